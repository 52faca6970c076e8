#!/usr/bin/env python3
"""Time one conv layer (fwd / dgrad / wgrad) through the C ABI.  usage: bench_conv.py Cin Cout H W [N] [ks] [reps] [which]"""
import os, sys, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from unet_zoo_amd import _ffi
a = [int(v) for v in sys.argv[1:8] if v.lstrip('-').isdigit()]
Cin, Cout, H, W = a[:4]
N = a[4] if len(a) > 4 else 32
ks = a[5] if len(a) > 5 else 3
reps = a[6] if len(a) > 6 else 5
which = sys.argv[8] if len(sys.argv) > 8 else "all"
L = _ffi.lib(); dev = torch.device("cuda", 0)
x = torch.randn(N, Cin, H, W, device=dev); dy = torch.randn(N, Cout, H, W, device=dev)
w = torch.randn(Cout, Cin, ks, ks, device=dev) * 0.05; y = torch.empty(N, Cout, H, W, device=dev); dx = torch.empty_like(x); dw = torch.empty_like(w)
wsb = max(L.uz_conv_bwd_weight_workspace(Cin, Cout, N, H, W, ks), L.uz_conv_workspace(Cin, Cout, N, H, W, ks))
ws = torch.empty(wsb // 4 + 64, device=dev)
st = torch.cuda.current_stream().cuda_stream
# magnitude bounds as the model plans supply them (maintained by the producing kernels): no measuring pass inside the call
def slot(v):
    t = torch.zeros(256, device=dev); t[0] = v; return t
xa, wa, dya = slot(float(x.abs().max())), slot(float(w.abs().max())), slot(float(dy.abs().max()))
fl = 2.0 * N * H * W * Cin * Cout * ks * ks
def t(fn):
    fn(); torch.cuda.synchronize(); best = 1e9
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); e1.synchronize(); best = min(best, e0.elapsed_time(e1))
    return best
if which in ("all", "fwd"):
    ms = t(lambda: _ffi.check(L.uz_conv_fwd(x.data_ptr(), Cin, Cin, w.data_ptr(), None, y.data_ptr(), Cout, Cout, N, H, W, ks, 0, xa.data_ptr(), wa.data_ptr(), None, ws.data_ptr(), wsb, st), "fwd"))
    print(f"fwd   {ms*1e3:9.1f} us {fl/ms/1e9:7.1f} TF/s")
if which in ("all", "dgrad"):
    ms = t(lambda: _ffi.check(L.uz_conv_bwd_data(dy.data_ptr(), Cout, Cout, w.data_ptr(), dx.data_ptr(), Cin, Cin, N, H, W, ks, 0, dya.data_ptr(), wa.data_ptr(), ws.data_ptr(), wsb, st), "dgrad"))
    print(f"dgrad {ms*1e3:9.1f} us {fl/ms/1e9:7.1f} TF/s")
if which in ("all", "wgrad"):
    ms = t(lambda: _ffi.check(L.uz_conv_bwd_weight(x.data_ptr(), Cin, Cin, dy.data_ptr(), Cout, Cout, dw.data_ptr(), None, N, H, W, ks, xa.data_ptr(), dya.data_ptr(), ws.data_ptr(), wsb, st), "wgrad"))
    print(f"wgrad {ms*1e3:9.1f} us {fl/ms/1e9:7.1f} TF/s")
