"""Which gradient differs between a non-DP step and a DP (world 1, overlapped exchange) step?  Both nets run in lockstep; after every
step the flat gradient and parameter buffers are compared per tensor (first differences printed, with a 'stale' check against the
DP net's gradient of the previous step)."""
import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import unet_zoo_amd  # noqa: F401
import torch, torch.distributed as dist
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29534")
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
torch.cuda.set_stream(torch.cuda.Stream())
from unet_zoo_amd.models.phiseg import PHISeg
from unet_zoo_amd.optim import FusedAdam
from unet_zoo_amd.synthetic import synthetic_batch
from unet_zoo_amd._plan import _numel

B = int(os.environ.get("UZ_DIAG_BATCH", "32"))
SEED = int(os.environ.get("UZ_DIAG_SEED", "5"))
x, m, _ = synthetic_batch(B, 128, 128, seed=SEED)
x, m = torch.from_numpy(x).cuda(), torch.from_numpy(m).cuda()
g = torch.Generator(device="cuda").manual_seed(7 + SEED)
noise = [torch.randn(s_, generator=g, device="cuda") for s_ in [(B, 2, 2 << k, 2 << k) for k in range(5)] * 2]


def make(dp, overlap=True):
    torch.manual_seed(1 + SEED)
    net = PHISeg(1, 2, [32, 64, 128, 192, 192, 192, 192], latent_levels=5, image_size=(1, 128, 128))
    net.train()
    if dp:
        net.set_data_parallel(True, overlap=overlap)
        net._dp.broadcast_params()
    net.enable_graphs(os.environ.get("UZ_DIAG_EAGER") is None)
    return net, FusedAdam(net, lr=1e-3, weight_decay=1e-5)


_kinds = os.environ.get("UZ_DIAG_NETS", "n,o").split(",")          # n = no DP, o = DP overlapped, s = DP serial
nets = [make(k != "n", k == "o") for k in _kinds]
prev = None
for step in range(int(os.environ.get("UZ_DIAG_STEPS", "10"))):
    losses = []
    for k_, (net, opt) in enumerate(nets):
        if os.environ.get("UZ_DEBUG_BIL"):
            torch.cuda.synchronize(); sys.stderr.write(f"== net {k_} step {step}\n"); sys.stderr.flush()
        net.forward(x, m, training=True, eps=noise); loss = net.loss(m); opt.zero_grad(); loss.backward(); opt.step()
        losses.append(float(loss.detach()))
    torch.cuda.synchronize()
    print(f"step {step}: losses {losses[0]!r} {losses[1]!r} equal {losses[0] == losses[1]}")
    t0, t1 = nets[0][0]._ptab, nets[1][0]._ptab
    bad = []
    for k, off in t0.poff.items():
        n = _numel(t0.shape[k])
        a, b = t0.gflat[off:off + n], t1.gflat[off:off + n]
        if not torch.equal(a, b):
            d = (a - b).abs()
            stale = float((prev[off:off + n] - b).abs().max()) if prev is not None else -1.0
            bad.append((k, n, int((d > 0).sum()), float(d.max()), float(a.abs().max()), stale))
    peq = torch.equal(t0.pflat, t1.pflat)
    print(f"step {step}: params equal {peq}; gradient tensors that differ: {len(bad)}")
    for r in bad[:12]:
        print("    %-70s n=%d differing=%d max|d|=%.3e max|g|=%.3e  max|prev step - dp|=%.3e" % r)
    if bad:
        print("    offsets of the differing tensors:", [t0.poff[r[0]] for r in bad[:12]])
        break
    prev = t1.gflat.clone()
dist.destroy_process_group()
