#!/usr/bin/env python3
"""Phase breakdown of the split-fp16 convolution kernels from in-kernel cycle stamps (uz_debug_stamps).

usage: stamp_conv.py Cin Cout H W [N]     -> per direction: wall time and the workgroups' median cycles per phase
(prologue = first tile staged, staging = barrier-to-barrier LDS fill phases, loop = main loop incl. staging, epilogue)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from unet_zoo_amd import _ffi

a = [int(v) for v in sys.argv[1:]]
Cin, Cout, H, W = a[:4]
N = a[4] if len(a) > 4 else 32
ks = 3
L = _ffi.lib(); dev = torch.device("cuda", 0)
x = torch.randn(N, Cin, H, W, device=dev); dy = torch.randn(N, Cout, H, W, device=dev)
w = torch.randn(Cout, Cin, ks, ks, device=dev) * 0.05; y = torch.empty(N, Cout, H, W, device=dev); dx = torch.empty_like(x); dw = torch.empty_like(w)
wsb = max(L.uz_conv_bwd_weight_workspace(Cin, Cout, N, H, W, ks), L.uz_conv_workspace(Cin, Cout, N, H, W, ks))
ws = torch.empty(wsb // 4 + 64, device=dev)
st = torch.cuda.current_stream().cuda_stream


def slot(v):
    t = torch.zeros(256, device=dev); t[0] = v; return t


xa, wa, dya = slot(float(x.abs().max())), slot(float(w.abs().max())), slot(float(dy.abs().max()))
fl = 2.0 * N * H * W * Cin * Cout * ks * ks
calls = {
    "fwd": lambda: _ffi.check(L.uz_conv_fwd(x.data_ptr(), Cin, Cin, w.data_ptr(), None, y.data_ptr(), Cout, Cout, N, H, W, ks, 0, xa.data_ptr(), wa.data_ptr(), None, ws.data_ptr(), wsb, st), "fwd"),
    "dgrad": lambda: _ffi.check(L.uz_conv_bwd_data(dy.data_ptr(), Cout, Cout, w.data_ptr(), dx.data_ptr(), Cin, Cin, N, H, W, ks, 0, dya.data_ptr(), wa.data_ptr(), ws.data_ptr(), wsb, st), "dgrad"),
    "wgrad": lambda: _ffi.check(L.uz_conv_bwd_weight(x.data_ptr(), Cin, Cin, dy.data_ptr(), Cout, Cout, dw.data_ptr(), None, N, H, W, ks, xa.data_ptr(), dya.data_ptr(), ws.data_ptr(), wsb, st), "wgrad"),
}
buf = torch.zeros(4096 * 8, dtype=torch.int64, device=dev)
for name, fn in calls.items():
    L.uz_debug_stamps(None)
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); fn(); e1.record(); e1.synchronize()
    ms = e0.elapsed_time(e1)
    buf.zero_(); L.uz_debug_stamps(buf.data_ptr()); fn(); torch.cuda.synchronize(); L.uz_debug_stamps(None)
    s = buf.view(4096, 8).cpu().double()
    s = s[s[:, 4] > 0]
    med = lambda v: float(v.median())
    tot = s[:, 4] - s[:, 0]
    clk = ((s[:, 4] - s[:, 0]) / ((s[:, 6] - s[:, 5]) * 10.0)).median()      # cycles per ns (real-time counter: 100 MHz)
    span = (s[:, 6].max() - s[:, 5].min()) * 10.0 / 1e3                       # us, first start -> last end
    print(f"{name:5s} wall {ms*1e3:8.1f} us  {fl/ms/1e9:6.1f} TF/s | workgroups {len(s)} tiles/wg {med(s[:,7]):.0f} | clock {float(clk):.2f} GHz span {float(span):.0f} us | "
          f"per-wg cycles: total {med(tot):.0f} prologue {med(s[:,1]-s[:,0]):.0f} loop {med(s[:,3]-s[:,1]):.0f} (staging phases {med(s[:,2]):.0f}) epilogue {med(s[:,4]-s[:,3]):.0f}")
