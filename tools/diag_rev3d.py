"""Diagnostic: reversible PHISeg3D with hipGraph replay at growing sizes."""
import os, sys, faulthandler
faulthandler.enable()
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from unet_zoo_amd.models.phiseg3D import PHISeg3D
from unet_zoo_amd.optim import FusedAdam
from unet_zoo_amd.synthetic import synthetic_volume
nf = [int(v) for v in os.environ.get("NF", "8,16,16").split(",")]
L = int(os.environ.get("LAT", "2"))
dhw = tuple(int(v) for v in os.environ.get("DHW", "16,32,32").split(","))
rev = os.environ.get("REV", "1") == "1"
net = PHISeg3D(4, 3, nf, latent_levels=L, reversible=rev)
net.train()
if os.environ.get("GRAPHS", "1") == "1":
    net.enable_graphs(True)
opt = FusedAdam(net, lr=1e-4, weight_decay=1e-5)
x, oh, lab = (torch.from_numpy(a).cuda() for a in synthetic_volume(4, 3, dhw))
for i in range(4):
    net.forward(x, oh, training=True)
    loss = net.loss(lab)
    opt.zero_grad()
    loss.backward()
    opt.step()
    torch.cuda.synchronize()
    print("step", i, float(loss), flush=True)
print("ok", nf, L, dhw, rev, net._cur.summary())
