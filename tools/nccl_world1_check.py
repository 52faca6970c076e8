"""RCCL smoke on a one-GPU box: world_size 1 process group (backend nccl = RCCL), broadcast of the parameters, data-parallel
all-reduce between hipGraph replays of the backward tape and the fused Adam step, barrier - the code path of
`bench.py --gpus N` minus the peers."""
import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch, torch.distributed as dist
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29533")
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
from unet_zoo_amd.models.phiseg import PHISeg
from unet_zoo_amd.optim import FusedAdam
from unet_zoo_amd.synthetic import synthetic_batch
torch.manual_seed(1)
net = PHISeg(1, 2, [32,64,128,192,192,192,192], latent_levels=5, image_size=(1,128,128)); net.train()
dist.broadcast(net._ptab.pflat, src=0)
net.set_data_parallel(True); net.enable_graphs(True)
opt = FusedAdam(net, lr=1e-3, weight_decay=1e-5)
x, m, _ = synthetic_batch(32, 128, 128, seed=5)
x, m = torch.from_numpy(x).cuda(), torch.from_numpy(m).cuda()
for it in range(6):
    net.forward(x, m, training=True); loss = net.loss(m); opt.zero_grad(); loss.backward(); opt.step()
torch.cuda.synchronize(); dist.barrier(); torch.cuda.synchronize()
t0 = time.perf_counter()
for it in range(20):
    net.forward(x, m, training=True); loss = net.loss(m); opt.zero_grad(); loss.backward(); opt.step()
torch.cuda.synchronize(); dist.barrier(); torch.cuda.synchronize()
print("nccl world 1: ms/step", (time.perf_counter() - t0) / 20 * 1e3, "loss", float(loss.detach()))
dist.destroy_process_group()
