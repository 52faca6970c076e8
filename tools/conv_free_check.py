#!/usr/bin/env python3
"""Forward and data gradient of the small-plane 3x3 convolutions through the C ABI against fp64 (the LDS-free one-wave kernel,
conv_mfma.hip conv_free_kernel, takes the split-K calls with N * H * W <= UZ_CONV_FREE_PX; 0 = the LDS-staged kernel).  Ragged channel
counts, channel-slice views, accumulate.  Prints one line per case and ALL OK; exit code 1 on a miss."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn.functional as F
from unet_zoo_amd import _ffi
L = _ffi.lib(); dev = torch.device("cuda", 0)
st = torch.cuda.current_stream().cuda_stream
P = lambda t: None if t is None else t.data_ptr()
ok = True
GATE = 6e-6 if os.environ.get("UZ_CONV_MATH") == "split" else 2e-6     # split forced onto the tiny planes: the split kernels' own gate (tests/test_ops_gpu.py)
for (Cin, Cout, N, H, W) in [(192, 192, 32, 4, 4), (256, 256, 32, 8, 8), (192, 192, 32, 2, 2), (100, 72, 7, 5, 6), (64, 2, 32, 8, 8), (576, 192, 8, 8, 8)]:
    g = torch.Generator(device="cpu").manual_seed(Cin + Cout)
    x = torch.randn(N, Cin + 3, H, W, generator=g).to(dev); w = (torch.randn(Cout, Cin, 3, 3, generator=g) * 0.05).to(dev); b = torch.randn(Cout, generator=g).to(dev)
    dy = torch.randn(N, Cout + 2, H, W, generator=g).to(dev)
    xv, dyv = x[:, 2:2 + Cin], dy[:, 1:1 + Cout]
    yr = F.conv2d(xv.double().cpu(), w.double().cpu(), b.double().cpu(), padding=1).to(dev)
    dxr = F.conv_transpose2d(dyv.double().cpu(), w.double().cpu(), padding=1).to(dev)
    wsb = max(L.uz_conv_workspace(Cin, Cout, N, H, W, 3), L.uz_conv_workspace(Cout, Cin, N, H, W, 3), 256); ws = torch.empty(wsb // 4 + 64, device=dev)
    y = torch.full((N, Cout, H, W), float("nan"), device=dev)
    xp = x.data_ptr() + 4 * 2 * H * W; dyp = dy.data_ptr() + 4 * 1 * H * W
    _ffi.check(L.uz_conv_fwd(xp, Cin, Cin + 3, P(w), P(b), P(y), Cout, Cout, N, H, W, 3, 0, None, None, None, P(ws), wsb, st), "fwd")
    dx0 = torch.randn(N, Cin, H, W, generator=g).to(dev); dx = dx0.clone()
    _ffi.check(L.uz_conv_bwd_data(dyp, Cout, Cout + 2, P(w), P(dx), Cin, Cin, N, H, W, 3, 1, None, None, P(ws), wsb, st), "dgrad")
    torch.cuda.synchronize()
    e1 = float((y.double() - yr).abs().max() / yr.abs().max()); e2 = float((dx.double() - dx0.double() - dxr).abs().max() / dxr.abs().max())
    parts = (L.uz_conv_splitk_parts(Cin, Cout, N, H, W, 3), L.uz_conv_bwd_splitk_parts(Cin, Cout, N, H, W, 3))
    good = e1 <= GATE and e2 <= GATE
    ok &= good
    print(f"{Cin}->{Cout} @ {N}x{H}x{W} split parts fwd/bwd {parts}: fwd {e1:.2e} dgrad(accumulate) {e2:.2e} {'ok' if good else 'MISS'}")
print("ALL OK" if ok else "FAILED"); sys.exit(0 if ok else 1)
