"""Soak test: STEPS (default 240) training steps of the north-star configuration (graphs, split kernels) on synthetic data; the
loss must stay finite and fall by more than half, and no split-fp16 kernel may ever have been handed a magnitude bound that its
tensor exceeded (uz_device_flags, checked every 20 steps).  usage: soak_train.py [steps] [model: phiseg | unet | probunet]"""
import os, sys, time; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, numpy as np
from unet_zoo_amd.models.phiseg import PHISeg
from unet_zoo_amd.optim import FusedAdam
from unet_zoo_amd.synthetic import synthetic_batch
STEPS = int(sys.argv[1]) if len(sys.argv) > 1 else 240
MODEL = sys.argv[2] if len(sys.argv) > 2 else "phiseg"
torch.manual_seed(0)
import bench
net = bench.build(MODEL); net.train(); net.enable_graphs(True)
opt = FusedAdam(net, lr=1e-3, weight_decay=1e-5)
dev = torch.device('cuda',0)
batches = [synthetic_batch(32,128,128,seed=s)[:2] for s in range(8)]
batches = [(torch.from_numpy(x).to(dev), torch.from_numpy(m).to(dev)) for x,m in batches]
losses=[]; t0=time.time()
flags = 0
for it in range(STEPS):
    x,m = batches[it % 8]
    (net.forward(x) if MODEL == "unet" else net.forward(x,m,training=True)); loss = net.loss(m); opt.zero_grad(); loss.backward(); opt.step()
    if it % 20 == 0 or it == STEPS - 1:
        l=float(loss.detach()); losses.append(l); f = net.check_bounds(); flags |= f; print(it, round(l,1), "bound flags", f, flush=True)
torch.cuda.synchronize(); print('time/step ms', (time.time()-t0)/STEPS*1e3)
assert flags == 0, flags
assert all(np.isfinite(losses)) and losses[-1] < 0.5*losses[0], losses
# eval: sample + accumulate
if MODEL == "phiseg":
    net.eval()
    with torch.no_grad():
        s = net.forward(batches[0][0], batches[0][1], training=False)      # the reference's eval forward needs the mask too (posterior)
    print('eval ok', [tuple(t.shape) for t in s][:2])
