#!/usr/bin/env python3
"""The likelihood's 1 x 1 heads (C -> 2 classes, phiseg.py:281-284) at the five levels of the headline plan, through the C ABI: forward, data gradient,
weight gradient - microseconds per call (best of 10) and the HBM rate of the algorithmic bytes.  They run at the tail of the forward tape and
at the head of the backward tape with nothing beside them, so their duration is step time."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from unet_zoo_amd import _ffi
L = _ffi.lib(); dev = torch.device("cuda", 0); N = 32
st = torch.cuda.current_stream().cuda_stream; P = lambda t: None if t is None else t.data_ptr()
def best(fn, reps=10):
    fn(); torch.cuda.synchronize(); b = 1e9
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); e1.synchronize(); b = min(b, e0.elapsed_time(e1))
    return b * 1e3
tot = [0.0, 0.0, 0.0]
for C, H in ((128, 128), (192, 64), (192, 32), (192, 16), (192, 8)):
    x = torch.randn(N, C, H, H, device=dev); w = torch.randn(2, C, 1, 1, device=dev) * 0.1; b = torch.zeros(2, device=dev)
    y = torch.empty(N, 2, H, H, device=dev); dy = torch.randn_like(y); dx = torch.empty_like(x); dw = torch.empty_like(w); db = torch.empty(2, device=dev)
    wsb = max(L.uz_conv_workspace(C, 2, N, H, H, 1), L.uz_conv_bwd_weight_workspace(C, 2, N, H, H, 1), 1 << 16); ws = torch.empty(wsb // 4 + 64, device=dev)
    f = best(lambda: _ffi.check(L.uz_conv_fwd(P(x), C, C, P(w), P(b), P(y), 2, 2, N, H, H, 1, 0, None, None, None, P(ws), wsb, st), "fwd"))
    d = best(lambda: _ffi.check(L.uz_conv_bwd_data(P(dy), 2, 2, P(w), P(dx), C, C, N, H, H, 1, 0, None, None, P(ws), wsb, st), "dgrad"))
    g = best(lambda: _ffi.check(L.uz_conv_bwd_weight(P(x), C, C, P(dy), 2, 2, P(dw), P(db), N, H, H, 1, None, None, P(ws), wsb, st), "wgrad"))
    mb = x.numel() * 4 / 1e6
    tot = [tot[0] + f, tot[1] + d, tot[2] + g]
    print(f"{C}->2 @ {N}x{H}x{H} ({mb:6.1f} MB of activations): fwd {f:7.1f} us {mb / f * 1e3:6.0f} GB/s | dgrad {d:7.1f} us {mb / d * 1e3:6.0f} GB/s | wgrad {g:7.1f} us {mb / g * 1e3:6.0f} GB/s")
print(f"five levels: fwd {tot[0]:.0f} us, dgrad {tot[1]:.0f} us, wgrad {tot[2]:.0f} us")
