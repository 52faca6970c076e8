"""Per-tap error of uz_conv_bwd_weight against torch (CPU) on a small 3x3 case."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from unet_zoo_amd import _ffi
L = _ffi.lib(); dev = torch.device("cuda", 0)
N, Cin, Cout, H, W = [int(v) for v in (sys.argv[1:6] if len(sys.argv) > 5 else (1, 32, 32, 4, 32))]
torch.manual_seed(1)
x = torch.randn(N, Cin, H, W); dy = torch.randn(N, Cout, H, W)
ref = torch.nn.grad.conv2d_weight(x, (Cout, Cin, 3, 3), dy, padding=1)
wsb = L.uz_conv_bwd_weight_workspace(Cin, Cout, N, H, W, 3)
ws = torch.zeros(wsb // 4 + 64, device=dev); dw = torch.zeros(Cout, Cin, 3, 3, device=dev)
st = torch.cuda.current_stream().cuda_stream
xd, dyd_ = x.to(dev), dy.to(dev)
_ffi.check(L.uz_conv_bwd_weight(xd.data_ptr(), Cin, Cin, dyd_.data_ptr(), Cout, Cout, dw.data_ptr(), None, N, H, W, 3, ws.data_ptr(), wsb, st), "wgrad")
torch.cuda.synchronize()
e = (dw.cpu() - ref).abs()
print("ref rms", float(ref.pow(2).mean().sqrt()))
for t in range(9):
    print("tap", t, "max err %.3e" % float(e[:, :, t // 3, t % 3].max()), " per-co-half", ["%.2e" % float(e[a:a + 16, :, t // 3, t % 3].max()) for a in range(0, Cout, 16)][:4],
          " per-ci-half", ["%.2e" % float(e[:, a:a + 16, t // 3, t % 3].max()) for a in range(0, Cin, 16)][:4])
