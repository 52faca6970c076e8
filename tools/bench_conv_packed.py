#!/usr/bin/env python3
"""Time one conv layer (fwd / dgrad / wgrad) through the C ABI with fp32 operands and with the operands in split storage.
usage: bench_conv_packed.py Cin Cout H W [N] [reps] [relu_frac]   (relu_frac: fraction of exact zeros in x, like a post-ReLU activation)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from unet_zoo_amd import _ffi
a = sys.argv[1:]
Cin, Cout, H, W = (int(v) for v in a[:4])
N = int(a[4]) if len(a) > 4 else 32
reps = int(a[5]) if len(a) > 5 else 10
zf = float(a[6]) if len(a) > 6 else 0.5
L = _ffi.lib(); dev = torch.device("cuda", 0)
x = torch.randn(N, Cin, H, W, device=dev); x = torch.where(torch.rand_like(x) < zf, torch.zeros_like(x), x.abs())
dy = torch.randn(N, Cout, H, W, device=dev)
if os.environ.get("UZ_BENCH_ZERO"):          # same instruction stream on all-zero operands: what is left when no bit toggles (clock / power check)
    x.zero_(); dy.zero_(); x[0, 0, 0, 0] = 1.0; dy[0, 0, 0, 0] = 1.0
w = torch.randn(Cout, Cin, 3, 3, device=dev) * 0.05
y = torch.empty(N, Cout, H, W, device=dev); dx = torch.empty_like(x); dw = torch.empty_like(w)
wsb = max(L.uz_conv_bwd_weight_workspace(Cin, Cout, N, H, W, 3), L.uz_conv_workspace(Cin, Cout, N, H, W, 3))
ws = torch.empty(wsb // 4 + 64, device=dev)
st = torch.cuda.current_stream().cuda_stream
def slot(v):
    t = torch.zeros(256, device=dev); t[0] = v; return t
xa, wa, dya = slot(float(x.abs().max())), slot(float(w.abs().max())), slot(float(dy.abs().max()))
xp, dyp = torch.empty_like(x), torch.empty_like(dy)
_ffi.check(L.uz_pack_split(x.data_ptr(), xp.data_ptr(), x.numel(), xa.data_ptr(), st), "pack")
_ffi.check(L.uz_pack_split(dy.data_ptr(), dyp.data_ptr(), dy.numel(), dya.data_ptr(), st), "pack")
fl = 2.0 * N * H * W * Cin * Cout * 9
def t(fn):
    fn(); torch.cuda.synchronize(); best = 1e9
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); e1.synchronize(); best = min(best, e0.elapsed_time(e1))
    return best
P = lambda t_: t_.data_ptr()
for pk in (0, 1):
    X, DY = (xp, dyp) if pk else (x, dy)
    f = t(lambda: _ffi.check(L.uz_conv_fwd_ex(P(X), Cin, Cin, P(w), None, P(y), Cout, Cout, N, H, W, 3, 0, P(xa), P(wa), None, P(ws), wsb, None, None, pk, None, 0, st), "fwd"))
    d = t(lambda: _ffi.check(L.uz_conv_bwd_data_ex(P(DY), Cout, Cout, P(w), P(dx), Cin, Cin, N, H, W, 3, 0, P(dya), P(wa), P(ws), wsb, None, pk, None, 0, None, 0, None, st), "dgrad"))
    g = t(lambda: _ffi.check(L.uz_conv_bwd_weight_ex(P(x if not pk else xp), Cin, Cin, P(DY), Cout, Cout, P(dw), None, N, H, W, 3, P(xa), P(dya), P(ws), wsb, pk, None, 0, pk, None, st), "wgrad"))
    print(f"{'packed' if pk else 'fp32  '}  fwd {f*1e3:8.1f} us {fl/f/1e9:6.1f} TF/s | dgrad {d*1e3:8.1f} us {fl/d/1e9:6.1f} | wgrad {g*1e3:8.1f} us {fl/g/1e9:6.1f}")
