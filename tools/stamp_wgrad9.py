#!/usr/bin/env python3
"""Phase breakdown of the nine-taps-per-wave weight gradient (wgrad9_kernel) from in-kernel cycle stamps.
usage: stamp_wgrad9.py Cin Cout H W [N] [packed]   -> per workgroup (median): cycles waiting for the tile's loads, staging, MFMA loop"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from unet_zoo_amd import _ffi
a = [int(v) for v in sys.argv[1:]]
Cin, Cout, H, W = a[:4]; N = a[4] if len(a) > 4 else 32; pk = a[5] if len(a) > 5 else 1
L = _ffi.lib(); dev = torch.device("cuda", 0)
x = torch.randn(N, Cin, H, W, device=dev).abs(); dy = torch.randn(N, Cout, H, W, device=dev); dw = torch.empty(Cout, Cin, 3, 3, device=dev)
wsb = L.uz_conv_bwd_weight_workspace(Cin, Cout, N, H, W, 3); ws = torch.empty(wsb // 4 + 64, device=dev)
st = torch.cuda.current_stream().cuda_stream
def slot(v):
    t = torch.zeros(256, device=dev); t[0] = v; return t
xa, dya = slot(float(x.abs().max())), slot(float(dy.abs().max()))
if pk:
    xp, dyp = torch.empty_like(x), torch.empty_like(dy)
    _ffi.check(L.uz_pack_split(x.data_ptr(), xp.data_ptr(), x.numel(), xa.data_ptr(), st), "pack")
    _ffi.check(L.uz_pack_split(dy.data_ptr(), dyp.data_ptr(), dy.numel(), dya.data_ptr(), st), "pack")
    x, dy = xp, dyp
P = lambda t: t.data_ptr()
fn = lambda: _ffi.check(L.uz_conv_bwd_weight_ex(P(x), Cin, Cin, P(dy), Cout, Cout, P(dw), None, N, H, W, 3, P(xa), P(dya), P(ws), wsb, pk, None, 0, pk, None, st), "wgrad")
for _ in range(3): fn()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record(); fn(); e1.record(); e1.synchronize(); ms = e0.elapsed_time(e1)
buf = torch.zeros(4096 * 8, dtype=torch.int64, device=dev)
L.uz_debug_stamps(buf.data_ptr()); fn(); torch.cuda.synchronize(); L.uz_debug_stamps(None)
s = buf.view(4096, 8).cpu().double(); mma = s[2048:, 0]; s = s[:2048]; keep = s[:, 4] > 0; s = s[keep]; mma = mma[keep]
med = lambda v: float(v.median())
clk = ((s[:, 4] - s[:, 0]) / ((s[:, 6] - s[:, 5]) * 10.0)).median()
tiles = med(s[:, 7])
print(f"wall {ms*1e3:.1f} us | workgroups {len(s)} tiles/wg {tiles:.0f} clock {float(clk):.2f} GHz | per tile (median wg): load wait {med(s[:,1])/tiles:.0f}  staging {med(s[:,2])/tiles:.0f}  MFMA loop {med(mma)/tiles:.0f} cycles (MFMA alone: 3456) | total {med(s[:,4]-s[:,0]):.0f} epilogue {med(s[:,4]-s[:,3]):.0f}")
