"""RCCL on a one-GPU box: world_size 1 (backend nccl = RCCL).  Exercises the product's data-parallel gradient exchange end to
end minus the peers: uz_comm_* over librccl.so (unique id, ncclCommInitRank, ncclBroadcast), the bucket events recorded inside
the backward hipGraph, the communication stream waiting for them and running ncclAllReduce(avg) per bucket
(GradSync.sync really calls the collective - with one rank it must return the gradients unchanged), and Adam starting behind
the last all-reduce.  Checks: parameters after 8 steps are BIT-IDENTICAL to a run without data parallelism (same seed, same
data), with overlap on and off; prints the step time of both and the exposed all-reduce time."""
import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import unet_zoo_amd  # noqa: F401  (first: it sets the HIP queue environment, which must precede HIP initialisation)
import torch, torch.distributed as dist
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29533")
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
if not os.environ.get("UZ_NULL_STREAM"):
    torch.cuda.set_stream(torch.cuda.Stream())          # as bench.py / train_model.py do for data-parallel runs
from unet_zoo_amd.models.phiseg import PHISeg
from unet_zoo_amd.optim import FusedAdam
from unet_zoo_amd.synthetic import synthetic_batch
from unet_zoo_amd import _ffi

B = int(os.environ.get("UZ_CHECK_BATCH", "32"))
x, m, _ = synthetic_batch(B, 128, 128, seed=5)
x, m = torch.from_numpy(x).cuda(), torch.from_numpy(m).cuda()
eps = None


def run(dp, overlap=True, steps=8, timed=0):
    # every leg starts from a clean process state: the previous leg's model (a 10.7 GB arena, its plans, lane streams' work) is collected
    # first - as the THIRD live model of the process the serial leg read 21.7 ms against 16.1 ms alone (round 6: not an exchange cost)
    import gc
    gc.collect(); torch.cuda.synchronize(); torch.cuda.empty_cache()
    torch.manual_seed(1)
    net = PHISeg(1, 2, [32, 64, 128, 192, 192, 192, 192], latent_levels=5, image_size=(1, 128, 128))
    net.train()
    if dp:
        net.set_data_parallel(True, overlap=overlap)
        assert net._dp.backend == "rccl" and net._dp.world == 1
        net._dp.broadcast_params()
    net.enable_graphs(True)
    opt = FusedAdam(net, lr=1e-3, weight_decay=1e-5)
    g = torch.Generator(device="cuda").manual_seed(7)
    shapes = [(B, 2, 2 << k, 2 << k) for k in range(5)] * 2
    noise = [torch.randn(s_, generator=g, device="cuda") for s_ in shapes]
    def step():
        net.forward(x, m, training=True, eps=noise); loss = net.loss(m); opt.zero_grad(); loss.backward(); opt.step(); return loss
    for _ in range(steps):
        loss = step()
    torch.cuda.synchronize()
    ms = None
    if timed:
        t0 = time.perf_counter()
        for _ in range(timed):
            loss = step()
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / timed * 1e3
    exposed = net._dp.exposed_ms() if dp else None
    nb = len(net._dp.buckets) if dp else 0
    nev = len(net._cur.events) if dp else 0
    out = (net._ptab.pflat.clone(), float(loss.detach()), ms, exposed, nb, nev)
    if dp:
        torch.cuda.synchronize()
        net._dp.close()                                  # communicator + communication stream: the next leg must not inherit a live stream
    return out


print("rccl version code", _ffi.lib().uz_comm_version())
if os.environ.get("UZ_WARM_NODP"):
    run(False, timed=3)
if os.environ.get("UZ_DP_ONLY"):
    if os.environ.get("UZ_SIDE_STREAM"):
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            p1, l1, ms1, ex1, nb, nev = run(True, overlap=True, timed=10)
    else:
        p1, l1, ms1, ex1, nb, nev = run(True, overlap=True, timed=10)
    print(f"side {os.environ.get('UZ_SIDE_STREAM')} prio {os.environ.get('UZ_DP_STREAM_PRIORITY')} lanes {os.environ.get('UZ_LANES')}: dp overlap {ms1:.3f} ms/step exposed {ex1}")
    dist.destroy_process_group()
    sys.exit(0)
if os.environ.get("UZ_SERIAL_ONLY"):      # the serial exchange alone in a fresh process (is its 32 ms the exchange, or being the third model of the process?)
    p2, l2, ms2, ex2, _, _ = run(True, overlap=False, timed=10)
    print(f"serial only (first and only model of the process): dp serial {ms2:.3f} ms/step, exposed {ex2:.3f} ms")
    dist.destroy_process_group()
    sys.exit(0)
p0, l0, ms0, _, _, _ = run(False, timed=10)
p1, l1, ms1, ex1, nb, nev = run(True, overlap=True, timed=10)
p2, l2, ms2, ex2, _, _ = run(True, overlap=False, timed=10)
same1, same2 = torch.equal(p0, p1), torch.equal(p0, p2)
print(f"buckets {nb}, bucket events in the backward graph {nev}")
print(f"no dp: {ms0:.3f} ms/step | dp overlap: {ms1:.3f} ms/step, exposed all-reduce {ex1:.3f} ms | dp serial: {ms2:.3f} ms/step, exposed {ex2:.3f} ms")
print(f"bit-identical to the non-DP run: overlap={same1} serial={same2}; losses {l0} {l1} {l2}")
dist.barrier()
assert same1 and same2 and nev == nb and nb >= 3
print("nccl world 1: ms/step", ms1, "loss", l1)
dist.destroy_process_group()
