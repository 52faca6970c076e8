#!/usr/bin/env python3
"""Workload for the HBM-traffic PMC passes (run under `rocprofv3 --pmc FETCH_SIZE` and, separately,
`--pmc WRITE_SIZE`): the dominant PHiSeg layer (3x3, 224 -> 128 channels, 32 x 128 x 128) forward,
data gradient and weight gradient through the C ABI, plus the calibration kernel
`channel_sum_partial` (bias gradient: one coalesced dword per lane over a known byte count, the same
access width as the conv staging loads) - MI355X_MICROARCH.md asks to calibrate FETCH_SIZE on the
access pattern in use before trusting an absolute number."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from unet_zoo_amd import _ffi

Cin, Cout, N, H, W, ks = 224, 128, 32, 128, 128, 3
L = _ffi.lib(); dev = torch.device("cuda", 0)
x = torch.randn(N, Cin, H, W, device=dev); dy = torch.randn(N, Cout, H, W, device=dev)
w = torch.randn(Cout, Cin, ks, ks, device=dev) * 0.05
y = torch.empty(N, Cout, H, W, device=dev); dx = torch.empty_like(x); dw = torch.empty_like(w); db = torch.empty(Cout, device=dev)
wsb = max(L.uz_conv_bwd_weight_workspace(Cin, Cout, N, H, W, ks), L.uz_conv_workspace(Cin, Cout, N, H, W, ks))
ws = torch.zeros(wsb // 4 + 64, device=dev)
st = torch.cuda.current_stream().cuda_stream
# magnitude bounds as the model plans supply them (bound slots: 256 floats, value = max over 16 sub-slots)
def slot(v):
    t = torch.zeros(256, device=dev); t[0] = v; return t
xa, wa, dya = slot(float(x.abs().max())), slot(float(w.abs().max())), slot(float(dy.abs().max()))
for _ in range(int(sys.argv[1]) if len(sys.argv) > 1 else 4):
    _ffi.check(L.uz_conv_fwd(x.data_ptr(), Cin, Cin, w.data_ptr(), None, y.data_ptr(), Cout, Cout, N, H, W, ks, 0, xa.data_ptr(), wa.data_ptr(), None, ws.data_ptr(), wsb, st), "fwd")
    _ffi.check(L.uz_conv_bwd_data(dy.data_ptr(), Cout, Cout, w.data_ptr(), dx.data_ptr(), Cin, Cin, N, H, W, ks, 0, dya.data_ptr(), wa.data_ptr(), ws.data_ptr(), wsb, st), "dgrad")
    _ffi.check(L.uz_conv_bwd_weight(x.data_ptr(), Cin, Cin, dy.data_ptr(), Cout, Cout, dw.data_ptr(), db.data_ptr(), N, H, W, ks, xa.data_ptr(), dya.data_ptr(), ws.data_ptr(), wsb, st), "wgrad")
# memory-bound class: BatchNorm(train)+ReLU forward / backward on a 128-channel 128x128 plane set (large path)
C = 128
yb = torch.randn(N, C, H, W, device=dev); ab = torch.empty_like(yb); dab = torch.randn_like(yb); dyb = torch.empty_like(yb)
gam, bet = torch.ones(C, device=dev), torch.zeros(C, device=dev)
rm, rv = torch.zeros(C, device=dev), torch.ones(C, device=dev)
save = torch.empty(2 * C, device=dev); dg, dbt, dbi = torch.empty(C, device=dev), torch.empty(C, device=dev), torch.empty(C, device=dev)
bws = torch.zeros(L.uz_bn_workspace(C, N, H, W) // 4 + 64, device=dev)
import ctypes as Cc
for _ in range(2):
    _ffi.check(L.uz_bn_relu_fwd(yb.data_ptr(), C, C, gam.data_ptr(), bet.data_ptr(), rm.data_ptr(), rv.data_ptr(), save.data_ptr(), ab.data_ptr(), C,
                                N, H, W, Cc.c_float(1e-3), Cc.c_float(0.01), 1, 1, None, bws.data_ptr(), st), "bn fwd")
    _ffi.check(L.uz_bn_relu_bwd(dab.data_ptr(), C, yb.data_ptr(), C, C, gam.data_ptr(), bet.data_ptr(), save.data_ptr(), dyb.data_ptr(), C,
                                dg.data_ptr(), dbt.data_ptr(), dbi.data_ptr(), N, H, W, 1, None, bws.data_ptr(), st), "bn bwd")
torch.cuda.synchronize()
print("bn plane bytes", yb.numel() * 4)
print("algorithmic bytes: x", x.numel() * 4, "y", y.numel() * 4, "w", w.numel() * 4, "calibration read (dy)", dy.numel() * 4)
