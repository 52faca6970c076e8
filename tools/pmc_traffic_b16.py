#!/usr/bin/env python3
"""Workload for the HBM-traffic PMC passes of the bf16-STORAGE volume path (BASELINE config 5), run under `rocprofv3 --pmc FETCH_SIZE`
and, separately, `--pmc WRITE_SIZE` (tools/prof_b16.sh): the heaviest layer of PHiSeg3D 5/5 on 4 x 128 x 128 x 64 - Conv3d 96 -> 96 as the
depth window 288 -> 96 over 128 slices of 128 x 64 - forward, data gradient and weight gradient with every tensor in bf16 storage, the
large-plane BatchNorm forward / backward on the same tensor, and the calibration kernel `channel_sum_partial` (one coalesced dword per
lane over a known byte count; MI355X_MICROARCH.md: calibrate FETCH_SIZE on the access pattern before trusting an absolute number)."""
import ctypes as Cc
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from unet_zoo_amd import _ffi

C, Cout, D, H, W = 96, 96, 128, 128, 64
L = _ffi.lib(); dev = torch.device("cuda", 0)
L.uz_set_conv_math(3)
st = torch.cuda.current_stream().cuda_stream
bf = torch.bfloat16
xv = torch.zeros(D + 2, C, H, W, device=dev, dtype=bf); xv[1:D + 1] = torch.randn(D, C, H, W, device=dev).to(bf)          # volume with zero border slices
dyv = torch.zeros(D + 2, Cout, H, W, device=dev, dtype=bf); dyv[1:D + 1] = torch.randn(D, Cout, H, W, device=dev).to(bf)
w = torch.randn(Cout, C, 3, 3, 3, device=dev) * 0.05
wf, wb = torch.empty(Cout * C * 27, device=dev), torch.empty(Cout * C * 27, device=dev)
_ffi.check(L.uz_w3d_permute(w.data_ptr(), wf.data_ptr(), Cout, C, 0, st), "permute")
_ffi.check(L.uz_w3d_permute(w.data_ptr(), wb.data_ptr(), Cout, C, 1, st), "permute")
y = torch.zeros(D + 2, Cout, H, W, device=dev, dtype=bf); dx = torch.zeros(D + 2, C, H, W, device=dev, dtype=bf); dw = torch.empty(Cout, C, 3, 3, 3, device=dev)
wsb = max(L.uz_conv_bwd_weight_workspace(3 * C, Cout, D, H, W, 3), L.uz_conv_workspace(3 * C, Cout, D, H, W, 3), L.uz_conv_workspace(C, 3 * Cout, D, H, W, 3))
ws = torch.zeros(wsb // 4 + 64, device=dev)
npart = L.uz_conv_bn_partials(3 * C, Cout, D, H, W, 3)
part = torch.empty(max(npart, 1) * Cout * 4, device=dev)
esz = 2
sl = C * H * W * esz                                   # bytes per slice
for _ in range(int(sys.argv[1]) if len(sys.argv) > 1 else 4):
    # window views: pointer = slice d - 1 (the zero border slice for d = 0), Cin = 3 C over a C-channel buffer
    _ffi.check(L.uz_conv_fwd_b16(xv.data_ptr(), 3 * C, C, wf.data_ptr(), None, y.data_ptr() + Cout * H * W * esz, Cout, Cout, D, H, W, 3, ws.data_ptr(), wsb, None,
                                 part.data_ptr() if npart else None, 1, 1, st), "fwd")
    _ffi.check(L.uz_conv_bwd_data_b16(dyv.data_ptr(), 3 * Cout, Cout, wb.data_ptr(), dx.data_ptr() + sl, C, C, D, H, W, 3, 0, ws.data_ptr(), wsb, None, 1, 1, st), "dgrad")
    _ffi.check(L.uz_conv_bwd_weight_b16(xv.data_ptr(), 3 * C, C, dyv.data_ptr() + Cout * H * W * esz, Cout, Cout, dw.data_ptr(), D, H, W, 3, ws.data_ptr(), wsb, 1, 1, None, st), "wgrad")
gam, bet = torch.ones(Cout, device=dev), torch.zeros(Cout, device=dev)
rm, rv = torch.zeros(Cout, device=dev), torch.ones(Cout, device=dev)
save = torch.empty(4 * Cout, device=dev); dg, dbt, dbi = (torch.empty(Cout, device=dev) for _ in range(3))
bws = torch.zeros(L.uz_bn_workspace(Cout, D, H, W) // 4 + 64, device=dev)
a = torch.empty(D, Cout, H, W, device=dev, dtype=bf); dA = torch.randn(D, Cout, H, W, device=dev).to(bf); dyo = torch.empty_like(a)
yreal = y[1:D + 1]
for _ in range(2):
    _ffi.check(L.uz_bn_relu_fwd_b16(yreal.data_ptr(), Cout, Cout, gam.data_ptr(), bet.data_ptr(), rm.data_ptr(), rv.data_ptr(), save.data_ptr(), a.data_ptr(), Cout,
                                    D, H, W, Cc.c_float(1e-3), Cc.c_float(0.01), 1, 1, bws.data_ptr(), part.data_ptr() if npart else None, npart, 1, 1, st), "bn fwd")
    _ffi.check(L.uz_bn_relu_bwd_b16(dA.data_ptr(), Cout, yreal.data_ptr(), Cout, Cout, gam.data_ptr(), bet.data_ptr(), save.data_ptr(), dyo.data_ptr(), Cout,
                                    dg.data_ptr(), dbt.data_ptr(), dbi.data_ptr(), D, H, W, 1, bws.data_ptr(), 1, 1, 1, st), "bn bwd")
# calibration: the fp32 bias-gradient kernel over a known byte count (a 2-D fp32 weight gradient with db on a tensor of its own)
cx = torch.randn(32, 32, 128, 128, device=dev); cdy = torch.randn(32, 32, 128, 128, device=dev); cdw = torch.empty(32, 32, 3, 3, device=dev); cdb = torch.empty(32, device=dev)
L.uz_set_conv_math(0)
cws_b = L.uz_conv_bwd_weight_workspace(32, 32, 32, 128, 128, 3)
cws = torch.zeros(cws_b // 4 + 64, device=dev)
_ffi.check(L.uz_conv_bwd_weight(cx.data_ptr(), 32, 32, cdy.data_ptr(), 32, 32, cdw.data_ptr(), cdb.data_ptr(), 32, 128, 128, 3, None, None, cws.data_ptr(), cws_b, st), "calibration")
torch.cuda.synchronize()
print("calibration read bytes (dy)", cdy.numel() * 4)
print("algorithmic bytes: x", D * C * H * W * esz, "y", D * Cout * H * W * esz, "w", w.numel() * 4)
