"""bf16 STORAGE on the config-5 volume (4 x 128 x 128 x 64): loss / logits / gradients against the fp32-storage bf16-arithmetic run and
the fp32-accurate split run, step time of both storages.  usage: diag_b16_volume.py [D H W]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import oracle
from oracle import refgraph3d as R3
from unet_zoo_amd import _ffi
from unet_zoo_amd.models.phiseg3D import PHISeg3D, phiseg3d_spec
from unet_zoo_amd.optim import FusedAdam
L = _ffi.lib()
dev = torch.device("cuda", 0)
dhw = tuple(int(a) for a in sys.argv[1:4]) if len(sys.argv) > 3 else (128, 128, 64)
filters, K, Cin = [32, 64, 128, 192, 192], 3, 4
sd0 = oracle.deterministic_state_dict(phiseg3d_spec(Cin, K, filters, 5), seed=11)
shapes = R3.phiseg3d_eps_shapes(*dhw, 5, 5)
x, onehot, lab, eps = R3.synthetic_volume(Cin, K, dhw, 9, shapes + shapes)
xd, od, ld = (torch.from_numpy(a).to(dev) for a in (x, onehot, lab))
epsd = [torch.from_numpy(e).to(dev) for e in eps]
res = {}
for tag, mode, b16 in (("split", 1, "0"), ("bf16 math", 3, "0"), ("bf16 storage", 3, "1")):
    L.uz_set_conv_math(mode)
    os.environ["UZ_STORE_B16"] = b16
    net = PHISeg3D(Cin, K, filters, latent_levels=5, image_size=(Cin, *dhw))
    net.load_state_dict(sd0)
    net.train()
    s = net.forward(xd, od, training=True, eps=epsd)
    s = [t.float().clone() for t in s]
    loss = net.loss(ld)
    loss.backward()
    grads = {k: p.grad.detach().clone() for k, p in net.named_parameters() if p.grad is not None}
    info = getattr(net._cur, "b16_info", None)
    net.enable_graphs(True)
    opt = FusedAdam(net, lr=1e-3, weight_decay=1e-5)
    losses = []
    for _ in range(3):
        net.forward(xd, od, training=True, eps=epsd); l = net.loss(ld); opt.zero_grad(); l.backward(); opt.step(); losses.append(float(l))
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(8):
        net.forward(xd, od, training=True, eps=epsd); l = net.loss(ld); opt.zero_grad(); l.backward(); opt.step()
    torch.cuda.synchronize(); ms = (time.perf_counter() - t0) / 8 * 1e3
    res[tag] = (float(loss), s, grads)
    print(f"{tag}: loss {float(loss):.6g} losses {losses} step {ms:.2f} ms arena {net._cur.arena_floats * 4 / 1e9:.2f} GB b16 {info}", flush=True)
    del net, opt
    torch.cuda.empty_cache()
L.uz_set_conv_math(-1)
noise = set()
try:
    from tests import _golden as G
    noise = set(G.bn_shadowed_biases(res["split"][2].keys()))
except Exception as e:
    print("no noise filter", e)
def cmp(a, b):
    la, sa, ga = res[a]; lb, sb, gb = res[b]
    rl = abs(la - lb) / abs(lb)
    lg = max(float((p - q).abs().max()) / max(float(q.max() - q.min()), 1e-3) for p, q in zip(sa, sb))
    devs = sorted((float((ga[k] - gb[k]).norm() / (gb[k].norm() + 1e-12)), k) for k in gb if k in ga and k not in noise)
    print(f"{a} vs {b}: loss rel {rl:.2e}, logits / range {lg:.2e}, gradient rel-L2: median {devs[len(devs) // 2][0]:.2e}, 90 % {devs[int(len(devs) * .9)][0]:.2e}, worst {devs[-1][0]:.2e} {devs[-1][1]}")
cmp("bf16 math", "split"); cmp("bf16 storage", "split"); cmp("bf16 storage", "bf16 math")
