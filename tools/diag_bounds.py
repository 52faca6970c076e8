"""How tight are the magnitude bounds the split-fp16 kernels scale with?  Runs one PHiSeg backward tape op by op and, behind every
BatchNorm-backward op, compares the bound slot it published for dY (analytic: |alpha| (max|dz| + |m1| + max|x_hat| |m2|)) with
the true max|dY|.  A bound within 2^10 of the maximum keeps the split at full accuracy (split_f16.h)."""
import os, sys, math, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from unet_zoo_amd import _ffi
from unet_zoo_amd.synthetic import synthetic_batch
B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
net = bench.build("phiseg"); net.train()
x, m, _ = synthetic_batch(B)
x, m = torch.from_numpy(x).cuda(), torch.from_numpy(m).cuda()
net.forward(x, m); loss = net.loss(m)
plan = net._cur
L = _ffi.lib(); st = C.c_void_p(net._stream())
plan.tensor(plan.loss_scale).fill_(1.0)
arr, n = plan.tapes["bwd"]
base = plan.arena.data_ptr()
ratios = []
for k in range(n):
    one = (type(arr[0]) * 1)(arr[k])
    _ffi.check(L.uz_run_tape(one, 1, st), "op")
    o = plan.bwd_ops[k]
    if o["code"] != "UZ_OP_BN_RELU_BWD":
        continue
    torch.cuda.synchronize()
    cout, N, H, W = o["i"][1], o["i"][4], o["i"][5], o["i"][6]
    dy_off = (arr[k].p[5] - base) // 4
    sl_off = (arr[k].p[10] - base) // 4
    dy = plan.arena[dy_off:dy_off + N * cout * H * W]
    slot = plan.arena[sl_off:sl_off + 256]
    true, bound = float(dy.abs().max()), float(slot.max())
    if true > 0:
        ratios.append((bound / true, H, cout))
ratios.sort(reverse=True)
import numpy as np
r = np.array([a for a, _, _ in ratios])
print(f"{len(r)} BN-backward ops: bound / true max|dY|: median {np.median(r):.2f}, p90 {np.percentile(r, 90):.2f}, max {r.max():.1f} (2^{math.log2(r.max()):.1f}); min {r.min():.3f}")
print("largest:", [(round(a, 1), h, c) for a, h, c in ratios[:6]])
assert r.min() >= 0.999, "a bound below the true maximum would overflow the fp16 pieces"
