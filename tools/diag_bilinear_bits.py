"""Dumps a checksum of uz_bilinear2x_bwd outputs (run once with and once without UZ_BILINEAR_BWD_PAIR=1 and compare)."""
import os, sys, hashlib
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from unet_zoo_amd import _ffi
L = _ffi.lib(); st = torch.cuda.current_stream().cuda_stream
for (N, C, H, W, ac, acc) in [(32, 192, 64, 64, 1, 1), (32, 192, 32, 32, 1, 0), (8, 64, 64, 32, 0, 1), (4, 8, 40, 64, 1, 0)]:
    g = torch.Generator(device="cuda").manual_seed(3)
    dy = torch.randn(N, C, 2 * H, 2 * W, device="cuda", generator=g); dx = torch.randn(N, C, H, W, device="cuda", generator=g)
    _ffi.check(L.uz_bilinear2x_bwd(dy.data_ptr(), C, C, dx.data_ptr(), C, N, H, W, ac, acc, st), "b")
    torch.cuda.synchronize()
    print((N, C, H, W, ac, acc), hashlib.sha1(dx.cpu().numpy().tobytes()).hexdigest()[:16])
