"""Times the bilinear x2 kernels on the layers that carry the traffic (HIP events, 20 launches each)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from unet_zoo_amd import _ffi
L = _ffi.lib()
st = torch.cuda.current_stream().cuda_stream
def timeit(fn, reps=20):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); e1.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3
for (N, C, H, W, ac) in [(32, 192, 64, 64, 1), (32, 192, 32, 32, 1), (32, 192, 16, 16, 1), (32, 128, 64, 64, 0), (128, 32, 64, 32, 1)]:
    x = torch.randn(N, C, H, W, device="cuda"); y = torch.empty(N, C, 2 * H, 2 * W, device="cuda")
    dy = torch.randn_like(y); dx = torch.empty_like(x)
    by = 4.0 * N * C * H * W * 5
    f = timeit(lambda: _ffi.check(L.uz_bilinear2x_fwd(x.data_ptr(), C, C, y.data_ptr(), C, N, H, W, ac, None, None, st), "f"))
    b = timeit(lambda: _ffi.check(L.uz_bilinear2x_bwd(dy.data_ptr(), C, C, dx.data_ptr(), C, N, H, W, ac, 0, st), "b"))
    print(f"bilinear {C}ch {H}x{W}->x2 N={N} ac={ac}: fwd {f:7.1f} us {by / f / 1e3:6.0f} GB/s | bwd {b:7.1f} us {by / b / 1e3:6.0f} GB/s")
