#!/usr/bin/env python3
"""Do two under-filled conv launches overlap when issued on two streams? (diagnostic)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from unet_zoo_amd import _ffi
L = _ffi.lib(); dev = torch.device("cuda", 0)
def mk(Cin, Cout, H, N=32):
    x = torch.randn(N, Cin, H, H, device=dev); w = torch.randn(Cout, Cin, 3, 3, device=dev) * 0.05
    y = torch.empty(N, Cout, H, H, device=dev)
    wsb = max(L.uz_conv_workspace(Cin, Cout, N, H, H, 3), L.uz_conv_bwd_weight_workspace(Cin, Cout, N, H, H, 3)); ws = torch.empty(wsb // 4 + 64, device=dev)
    dw = torch.empty_like(w)
    def fwd(st): _ffi.check(L.uz_conv_fwd(x.data_ptr(), Cin, Cin, w.data_ptr(), None, y.data_ptr(), Cout, Cout, N, H, H, 3, 0, ws.data_ptr(), wsb, st), "f")
    def wg(st): _ffi.check(L.uz_conv_bwd_weight(x.data_ptr(), Cin, Cin, y.data_ptr(), Cout, Cout, dw.data_ptr(), None, N, H, H, 3, ws.data_ptr(), wsb, st), "w")
    return fwd, wg
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
def timeit(fn, reps=20):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3
for (Cin, Cout, H) in [(192, 192, 32), (192, 192, 16), (192, 192, 8), (192, 192, 4), (128, 128, 32), (256, 256, 16)]:
    for kind in (0, 1):
        a = mk(Cin, Cout, H)[kind]; b = mk(Cin, Cout, H)[kind]
        cur = torch.cuda.current_stream()
        def seq():
            a(cur.cuda_stream); b(cur.cuda_stream)
        def par():
            s1.wait_stream(cur); s2.wait_stream(cur)
            a(s1.cuda_stream); b(s2.cuda_stream)
            cur.wait_stream(s1); cur.wait_stream(s2)
        print(f"{'fwd' if kind == 0 else 'wgrad'} {Cin}->{Cout}@{H}: sequential pair {timeit(seq):7.1f} us   two streams {timeit(par):7.1f} us")
