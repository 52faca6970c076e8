#!/usr/bin/env python3
"""What does hipExtStreamCreateWithCUMask do on MI355X?  The heaviest convolution on a masked stream (its own duration tells how many
CUs it got) with a chain of small launches on an ordinary stream beside it (tools/bench_coresidency.py)."""
import os, sys, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from unet_zoo_amd import _ffi
L = _ffi.lib(); dev = torch.device("cuda", 0)
hip = C.CDLL("libamdhip64.so")
P = lambda t: None if t is None else t.data_ptr()
def slot(v):
    t = torch.zeros(256, device=dev); t[0] = v; return t
N, Ci, Co, H, W = 32, 224, 128, 128, 128
x = torch.randn(N, Ci, H, W, device=dev).relu_(); w = torch.randn(Co, Ci, 3, 3, device=dev) * 0.05; y = torch.empty(N, Co, H, W, device=dev)
xa, wa = slot(float(x.abs().max())), slot(float(w.abs().max()))
xp = torch.empty_like(x); _ffi.check(L.uz_pack_split(P(x), P(xp), x.numel(), P(xa), torch.cuda.current_stream().cuda_stream), "pack")
wsb = L.uz_conv_workspace(Ci, Co, N, H, W, 3); ws = torch.empty(wsb // 4 + 64, device=dev)
c, h = 192, 4
ys = torch.randn(N, c, h, h, device=dev); a = torch.empty_like(ys)
gam, bet = torch.ones(c, device=dev), torch.zeros(c, device=dev); rm, rv = torch.zeros(c, device=dev), torch.ones(c, device=dev); save = torch.zeros(4 * c, device=dev)
wsm = torch.empty(1 << 20, device=dev)
sB = torch.cuda.Stream()
torch.cuda.synchronize()
def heavy(st):
    _ffi.check(L.uz_conv_fwd_ex(P(xp), Ci, Ci, P(w), None, P(y), Co, Co, N, H, W, 3, 0, P(xa), P(wa), None, P(ws), wsb, None, None, 1, None, 0, st), "heavy")
def small(st):
    _ffi.check(L.uz_bn_relu_fwd(P(ys), c, c, P(gam), P(bet), P(rm), P(rv), P(save), P(a), c, N, h, h, 1e-3, 0.01, 1, 1, None, P(wsm), st), "small")
masks = {
    "all ones": [0xFFFFFFFF] * 8,
    "word0 = 0xFFFFFFFE (bit 0 off)": [0xFFFFFFFE] + [0xFFFFFFFF] * 7,
    "bits 0-7 off": [0xFFFFFF00] + [0xFFFFFFFF] * 7,
    "bits 0-31 off (word 0 zero)": [0] + [0xFFFFFFFF] * 7,
    "every word 0xFFFFFFF0": [0xFFFFFFF0] * 8,
    "every word 0x7FFFFFFF": [0x7FFFFFFF] * 8,
    "words 0-3 only": [0xFFFFFFFF] * 4 + [0] * 4,
}
ev = lambda: torch.cuda.Event(enable_timing=True)
for name, m in masks.items():
    arr = (C.c_uint32 * 8)(*m)
    st = C.c_void_p()
    rc = hip.hipExtStreamCreateWithCUMask(C.byref(st), 8, arr)
    if rc != 0:
        print(f"{name}: create failed rc={rc}"); continue
    for _ in range(2): heavy(st)
    hip.hipStreamSynchronize(st)
    # heavy alone on the masked stream: time via device sync + host timer
    import time
    t0 = time.perf_counter()
    for _ in range(5): heavy(st)
    hip.hipStreamSynchronize(st)
    th = (time.perf_counter() - t0) / 5 * 1e6
    # small chain beside it
    for _ in range(6): heavy(st)
    e0, e1 = ev(), ev()
    with torch.cuda.stream(sB):
        e0.record(sB)
        for _ in range(40): small(sB.cuda_stream)
        e1.record(sB)
    torch.cuda.synchronize(); hip.hipStreamSynchronize(st)
    print(f"{name:36s} heavy {th:7.1f} us   small chain beside it {e0.elapsed_time(e1) * 1e3 / 40:6.1f} us/launch")
    hip.hipStreamDestroy(st)
