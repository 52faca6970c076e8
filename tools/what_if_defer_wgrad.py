#!/usr/bin/env python3
"""What would the forward pass's head absorb?  (a measurement for the next round - timing only.)
The heaviest weight-gradient ops of the backward tape (the likelihood's: needed by nobody but the optimiser) are launched a SECOND time on a
side stream at the START of every forward pass, beside the encoders' 3-ms chain of small launches; their inputs are whatever the last
backward left in the buffers and their slabs are overwritten by the next backward, so the training state is not touched (the values they
compute are garbage: the next forward is re-zeroing the bound slots they read).  Reported: the step time without / with that extra work, and
the extra work's isolated time.  If the step grows by much less than the work takes alone, moving the real ops there would shorten the step by
about their realised time in the backward tape minus that growth.
usage: python tools/what_if_defer_wgrad.py [n_ops=6]"""
import ctypes as C, os, sys, time
os.environ.setdefault("UZ_REPLAY", "lanes")
os.environ.setdefault("GPU_MAX_HW_QUEUES", "4")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from unet_zoo_amd import _ffi
from unet_zoo_amd._plan import Plan
from unet_zoo_amd.models.phiseg import PHISeg
from unet_zoo_amd.optim import FusedAdam
n_ops = int(sys.argv[1]) if len(sys.argv) > 1 else 6
dev = torch.device("cuda", 0)
torch.manual_seed(0)
net = PHISeg(1, 2, [32, 64, 128, 192, 192, 192, 192], latent_levels=5, image_size=(1, 128, 128)).to(dev)
net.train(); net.enable_graphs(True)
opt = FusedAdam(net, lr=1e-3, weight_decay=1e-5)
x = torch.randn(32, 1, 128, 128, device=dev); m = torch.randint(0, 2, (32, 128, 128), device=dev)
L = _ffi.lib()
side = torch.cuda.Stream()
extra = None
def step(with_extra):
    if with_extra and extra is not None:
        side.wait_stream(torch.cuda.current_stream())
        _ffi.check(L.uz_run_tape(extra[0], extra[1], C.c_void_p(side.cuda_stream)), "extra")
    net.forward(x, m, training=True)
    loss = net.loss(m)
    opt.zero_grad(); loss.backward(); opt.step()
    if with_extra and extra is not None:
        torch.cuda.current_stream().wait_stream(side)
for _ in range(5): step(False)
net.tune_schedule(lambda: step(False), rounds=int(os.environ.get("UZ_TUNE_SCHEDULE", "6")))
plan = net._cur
ops = plan.bwd_ops
arr, n = plan.tapes["bwd"]
heavy = sorted((k for k, o in enumerate(ops) if o["code"] == "UZ_OP_CONV_BWD_WEIGHT"), key=lambda k: -Plan._op_cost(ops[k]))[:n_ops]
sub = (_ffi.uz_op * len(heavy))(*[arr[k] for k in heavy])
extra = (sub, len(heavy))
def shape(o):
    i = o["i"]; return f"{i[0]}->{i[2]}@{i[5]}x{i[6]}"
print("extra work:", [shape(ops[k]) for k in heavy])
def timed(fn, reps=30):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps): fn()
    torch.cuda.synchronize(); return 1e3 * (time.perf_counter() - t0) / reps
alone = timed(lambda: _ffi.check(L.uz_run_tape(sub, len(heavy), C.c_void_p(side.cuda_stream)), "extra"), 20)
for rep in range(3):
    a = timed(lambda: step(False)); b = timed(lambda: step(True))
    print(f"step {a:.3f} ms   with the extra weight gradients beside the forward's head {b:.3f} ms (+{b - a:.3f})   the extra work alone {alone:.3f} ms")
