#!/usr/bin/env python3
"""Time BatchNorm(train)+ReLU forward / backward through the C ABI.  usage: bench_bn.py C H W [N] [reps]"""
import os, sys, ctypes as Cc
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from unet_zoo_amd import _ffi
a = [int(v) for v in sys.argv[1:]]
C, H, W = a[:3]; N = a[3] if len(a) > 3 else 32; reps = a[4] if len(a) > 4 else 20
L = _ffi.lib(); dev = torch.device("cuda", 0)
y = torch.randn(N, C, H, W, device=dev); ab = torch.empty_like(y); da = torch.randn_like(y); dy = torch.empty_like(y)
gam, bet = torch.ones(C, device=dev), torch.zeros(C, device=dev); rm, rv = torch.zeros(C, device=dev), torch.ones(C, device=dev)
save = torch.empty(2 * C, device=dev); dg, db, dbi = (torch.empty(C, device=dev) for _ in range(3))
ws = torch.zeros(L.uz_bn_workspace(C, N, H, W) // 4 + 64, device=dev)
st = torch.cuda.current_stream().cuda_stream
def t(fn):
    fn(); e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); e1.synchronize(); return e0.elapsed_time(e1) / reps
f = lambda: _ffi.check(L.uz_bn_relu_fwd(y.data_ptr(), C, C, gam.data_ptr(), bet.data_ptr(), rm.data_ptr(), rv.data_ptr(), save.data_ptr(), ab.data_ptr(), C, N, H, W, Cc.c_float(1e-3), Cc.c_float(0.01), 1, 1, ws.data_ptr(), st), "f")
b = lambda: _ffi.check(L.uz_bn_relu_bwd(da.data_ptr(), C, y.data_ptr(), C, C, gam.data_ptr(), bet.data_ptr(), save.data_ptr(), dy.data_ptr(), C, dg.data_ptr(), db.data_ptr(), dbi.data_ptr(), N, H, W, 1, ws.data_ptr(), st), "b")
plane = y.numel() * 4
tf, tb = t(f), t(b)
print(f"bn fwd {tf*1e3:8.1f} us  {3*plane/tf/1e9:7.1f} GB/s (3 passes)   bwd {tb*1e3:8.1f} us  {5*plane/tb/1e9:7.1f} GB/s (5 passes)")
