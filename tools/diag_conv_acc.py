"""Error of the conv kernels against an fp64 convolution (CPU): fp32-MFMA path vs split-bf16 path (set UZ_CONV_MATH)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, torch.nn.functional as F
from unet_zoo_amd import _ffi
L = _ffi.lib(); dev = torch.device("cuda", 0)
torch.manual_seed(0)
for (N, Cin, Cout, H, W) in ((2, 224, 128, 64, 64), (2, 192, 192, 32, 32), (4, 192, 192, 8, 8)):
    x = torch.randn(N, Cin, H, W); w = torch.randn(Cout, Cin, 3, 3) * (2.0 / (Cin * 9)) ** 0.5; dy = torch.randn(N, Cout, H, W)
    x = F.relu(x) + 0.3          # post-ReLU-like, non-zero mean (harder for cancellation)
    ref = F.conv2d(x.double(), w.double(), None, padding=1)
    refd = torch.nn.grad.conv2d_input(x.shape, w.double(), dy.double(), padding=1)
    xd, wd, dyd = x.to(dev), w.to(dev), dy.to(dev)
    y = torch.empty(N, Cout, H, W, device=dev); dx = torch.empty(N, Cin, H, W, device=dev)
    wsb = L.uz_conv_workspace(Cin, Cout, N, H, W, 3); ws = torch.zeros(wsb // 4 + 64, device=dev)
    st = torch.cuda.current_stream().cuda_stream
    _ffi.check(L.uz_conv_fwd(xd.data_ptr(), Cin, Cin, wd.data_ptr(), None, y.data_ptr(), Cout, Cout, N, H, W, 3, 0, ws.data_ptr(), wsb, st), "f")
    _ffi.check(L.uz_conv_bwd_data(dyd.data_ptr(), Cout, Cout, wd.data_ptr(), dx.data_ptr(), Cin, Cin, N, H, W, 3, 0, ws.data_ptr(), wsb, st), "d")
    torch.cuda.synchronize()
    cpu32 = F.conv2d(x, w, None, padding=1)
    e = (y.cpu().double() - ref); ec = (cpu32.double() - ref); ed = (dx.cpu().double() - refd)
    print(f"{os.environ.get('UZ_CONV_MATH','default'):8s} {Cin}->{Cout}@{H}: fwd max {e.abs().max():.3e} rms {e.pow(2).mean().sqrt():.3e} | cpu fp32 max {ec.abs().max():.3e} rms {ec.pow(2).mean().sqrt():.3e} | dgrad max {ed.abs().max():.3e} rms {ed.pow(2).mean().sqrt():.3e}  (|y| rms {ref.pow(2).mean().sqrt():.2f})")
