#!/usr/bin/env python3
"""Weight gradient through the C ABI (fp32 and split-storage operands) against an fp64 reference, on shapes that take the 64-channel
split kernels: ragged channel counts, a partial last tile row, a 16-wide plane.  Prints one line per case and ALL OK; exit code 1 on a
miss.  The form is chosen by the environment (UZ_WG9, UZ_WG_M16): tests/test_optional_forms_gpu.py."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from unet_zoo_amd import _ffi
L = _ffi.lib(); dev = torch.device("cuda", 0)
st = torch.cuda.current_stream().cuda_stream
P = lambda t: t.data_ptr()
def slot(v):
    t = torch.zeros(256, device=dev); t[0] = v; return t
ok = True
GATE = 2e-6
for (Cin, Cout, N, H, W) in [(64, 64, 4, 32, 32), (224, 128, 2, 64, 64), (100, 72, 3, 30, 32), (192, 192, 8, 16, 16), (96, 64, 2, 33, 64)]:
    g = torch.Generator(device="cpu").manual_seed(Cin * 7 + Cout)
    x = torch.randn(N, Cin, H, W, generator=g).clamp_min(0).to(dev); dy = torch.randn(N, Cout, H, W, generator=g).to(dev)
    ref = torch.nn.grad.conv2d_weight(x.double().cpu(), (Cout, Cin, 3, 3), dy.double().cpu(), padding=1).to(dev)
    wsb = L.uz_conv_bwd_weight_workspace(Cin, Cout, N, H, W, 3); ws = torch.empty(wsb // 4 + 64, device=dev)
    xa, dya = slot(float(x.abs().max())), slot(float(dy.abs().max()))
    xp, dyp = torch.empty_like(x), torch.empty_like(dy)
    _ffi.check(L.uz_pack_split(P(x), P(xp), x.numel(), P(xa), st), "pack"); _ffi.check(L.uz_pack_split(P(dy), P(dyp), dy.numel(), P(dya), st), "pack")
    route = L.uz_conv_route(2, Cin, Cout, N, H, W, 3)
    for pk in (0, 1):
        if pk and route != 1: continue
        dw = torch.full((Cout, Cin, 3, 3), float("nan"), device=dev)
        _ffi.check(L.uz_conv_bwd_weight_ex(P(xp if pk else x), Cin, Cin, P(dyp if pk else dy), Cout, Cout, P(dw), None, N, H, W, 3, P(xa), P(dya), P(ws), wsb, pk, None, 0, pk, None, st), "wgrad")
        torch.cuda.synchronize()
        err = float((dw.double() - ref).abs().max() / ref.abs().max())
        good = err <= GATE          # tensor-max-relative; these kernels measure 2e-7 .. 4e-7 on these shapes
        ok &= good
        print(f"{Cin}->{Cout} @ {N}x{H}x{W} route {route} {'split storage' if pk else 'fp32 operands'}: max rel err {err:.2e} {'ok' if good else 'MISS'}")
print("ALL OK" if ok else "FAILED"); sys.exit(0 if ok else 1)
