"""Soak test: 240 training steps of the north-star configuration (graphs, split kernels) on synthetic data; the loss
must stay finite and fall by more than half."""
import os, sys, time; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, numpy as np
from unet_zoo_amd.models.phiseg import PHISeg
from unet_zoo_amd.optim import FusedAdam
from unet_zoo_amd.synthetic import synthetic_batch
torch.manual_seed(0)
net = PHISeg(1, 2, [32,64,128,192,192,192,192], latent_levels=5, image_size=(1,128,128)); net.train(); net.enable_graphs(True)
opt = FusedAdam(net, lr=1e-3, weight_decay=1e-5)
dev = torch.device('cuda',0)
batches = [synthetic_batch(32,128,128,seed=s)[:2] for s in range(8)]
batches = [(torch.from_numpy(x).to(dev), torch.from_numpy(m).to(dev)) for x,m in batches]
losses=[]; t0=time.time()
for it in range(240):
    x,m = batches[it % 8]
    net.forward(x,m,training=True); loss = net.loss(m); opt.zero_grad(); loss.backward(); opt.step()
    if it % 20 == 0 or it == 239:
        l=float(loss.detach()); losses.append(l); print(it, round(l,1), flush=True)
torch.cuda.synchronize(); print('time/step ms', (time.time()-t0)/240*1e3)
assert all(np.isfinite(losses)) and losses[-1] < 0.5*losses[0], losses
# eval: sample + accumulate
net.eval()
with torch.no_grad():
    s = net.forward(batches[0][0], batches[0][1], training=False)      # the reference's eval forward needs the mask too (posterior)
print('eval ok', [tuple(t.shape) for t in s][:2])
