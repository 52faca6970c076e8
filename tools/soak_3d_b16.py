"""Soak: N training steps of PHiSeg3D 5/5 on a 4 x 128 x 128 x 64 volume in bf16 storage (graph replay, fresh noise per step): the loss
stays finite and falls, parameters and gradients stay finite.  usage: soak_3d_b16.py [steps=100]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("UZ_CONV_MATH", "bf16"); os.environ.setdefault("UZ_STORE_B16", "1")
import numpy as np, torch
import oracle
from oracle import refgraph3d as R3
from unet_zoo_amd.models.phiseg3D import PHISeg3D, phiseg3d_spec
from unet_zoo_amd.optim import FusedAdam
n = int(sys.argv[1]) if len(sys.argv) > 1 else 100
dev = torch.device("cuda", 0)
filters, K, Cin, dhw = [32, 64, 128, 192, 192], 3, 4, (128, 128, 64)
sd0 = oracle.deterministic_state_dict(phiseg3d_spec(Cin, K, filters, 5), seed=11)
shapes = R3.phiseg3d_eps_shapes(*dhw, 5, 5)
x, onehot, lab, eps = R3.synthetic_volume(Cin, K, dhw, 9, shapes + shapes)
xd, od, ld = (torch.from_numpy(a).to(dev) for a in (x, onehot, lab))
net = PHISeg3D(Cin, K, filters, latent_levels=5, image_size=(Cin, *dhw)); net.load_state_dict(sd0); net.train(); net.enable_graphs(True)
opt = FusedAdam(net, lr=1e-3, weight_decay=1e-5)
losses = []
for s in range(n):
    net.forward(xd, od, training=True)                 # device-side noise stream: fresh epsilon every step
    l = net.loss(ld); opt.zero_grad(); l.backward(); opt.step(); losses.append(float(l))
ok = all(np.isfinite(losses)) and all(bool(torch.isfinite(p).all()) for p in net.parameters())
print("b16 buffers", net._cur.b16_info, "| loss first", losses[0], "min", min(losses), "last", losses[-1], "| finite", ok, "| flags", net.check_bounds())
assert ok and losses[-1] < 0.5 * losses[0]
