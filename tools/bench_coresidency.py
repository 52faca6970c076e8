#!/usr/bin/env python3
"""Does a chain of small launches keep its speed beside a device-filling convolution?  Stream A: a loop of the heaviest 3x3 layer
(224 -> 128 @ 32 x 128 x 128, split path); stream B: a dependent chain of 40 small launches of ONE kind.  Reported per kind: the
chain's time alone, beside the convolution, and the kernel's VGPRs (from the build) - the question is whether the slowdown follows
the register footprint (a workgroup of the chain needs a free slot on all four SIMDs of a CU that a convolution workgroup occupies)."""
import os, sys, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from unet_zoo_amd import _ffi
L = _ffi.lib(); dev = torch.device("cuda", 0)
P = lambda t: None if t is None else t.data_ptr()
def slot(v):
    t = torch.zeros(256, device=dev); t[0] = v; return t
# heavy op
N, Ci, Co, H, W = 32, 224, 128, 128, 128
x = torch.randn(N, Ci, H, W, device=dev).relu_(); w = torch.randn(Co, Ci, 3, 3, device=dev) * 0.05; y = torch.empty(N, Co, H, W, device=dev)
xa, wa = slot(float(x.abs().max())), slot(float(w.abs().max()))
xp = torch.empty_like(x); _ffi.check(L.uz_pack_split(P(x), P(xp), x.numel(), P(xa), torch.cuda.current_stream().cuda_stream), "pack")
wsb = L.uz_conv_workspace(Ci, Co, N, H, W, 3); ws = torch.empty(wsb // 4 + 64, device=dev)
sA, sB = torch.cuda.Stream(), torch.cuda.Stream(priority=int(os.environ.get("UZ_CHAIN_STREAM_PRIORITY", "0")))     # (-1 = high priority for the chain of small launches)
def heavy(st, reps):
    for _ in range(reps):
        _ffi.check(L.uz_conv_fwd_ex(P(xp), Ci, Ci, P(w), None, P(y), Co, Co, N, H, W, 3, 0, P(xa), P(wa), None, P(ws), wsb, None, None, 1, None, 0, st.cuda_stream), "heavy")
# small ops (192 channels on a 4 x 4 plane at batch 32, like the deep levels)
c, h = 192, 4
ys = torch.randn(N, c, h, h, device=dev); a = torch.empty_like(ys); da = torch.randn_like(ys); dy = torch.empty_like(ys)
gam, bet = torch.ones(c, device=dev), torch.zeros(c, device=dev); rm, rv = torch.zeros(c, device=dev), torch.ones(c, device=dev); save = torch.zeros(4 * c, device=dev)
dg, db, dbias = torch.zeros(c, device=dev), torch.zeros(c, device=dev), torch.zeros(c, device=dev)
wsm = torch.empty(1 << 20, device=dev)
ws2b = max(L.uz_conv_workspace(c, c, N, h, h, 3), 1 << 16); ws2 = torch.empty(ws2b // 4 + 64, device=dev); w2 = torch.randn(c, c, 3, 3, device=dev) * 0.05
pooled = torch.empty(N, c, h // 2, h // 2, device=dev)
big = torch.randn(N, 32, 32, 32, device=dev); bigp = torch.empty(N, 32, 16, 16, device=dev)
kinds = {
    "bn_fused_small_fwd<2> (34 VGPRs; 156 until round 5)": lambda st: L.uz_bn_relu_fwd(P(ys), c, c, P(gam), P(bet), P(rm), P(rv), P(save), P(a), c, N, h, h, 1e-3, 0.01, 1, 1, None, P(wsm), st),
    "bn_fused_small_bwd<2> (36 VGPRs; 260 until round 5)": lambda st: L.uz_bn_relu_bwd(P(da), c, P(ys), c, c, P(gam), P(bet), P(save), P(dy), c, P(dg), P(db), P(dbias), N, h, h, 1, None, P(wsm), st),
    "conv_mfma 192->192@4x4 (122-156 VGPRs)": lambda st: L.uz_conv_fwd(P(ys), c, c, P(w2), None, P(a), c, c, N, h, h, 3, 0, None, None, None, P(ws2), ws2b, st),
    "avgpool 32ch@32x32 (low VGPRs)": lambda st: L.uz_avgpool2_fwd_ex(P(big), 32, 32, P(bigp), 32, N, 32, 32, None, None, 0, st),
}
def chain(fn, st, n=40):
    for _ in range(n):
        _ffi.check(fn(st.cuda_stream), "small")
def timed(fn, with_heavy):
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    if with_heavy:
        heavy(sA, 6)                     # ~4 ms of device-filling work in flight on stream A
    with torch.cuda.stream(sB):
        e0.record(sB); chain(fn, sB); e1.record(sB)
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / 40
heavy(sA, 2); torch.cuda.synchronize()
for name, fn in kinds.items():
    chain(fn, sB, 3); torch.cuda.synchronize()
    alone = min(timed(fn, False) for _ in range(3)); beside = min(timed(fn, True) for _ in range(3))
    print(f"{name:54s} alone {alone:7.1f} us/launch   beside the convolution {beside:7.1f} us/launch   x{beside / alone:.2f}")
