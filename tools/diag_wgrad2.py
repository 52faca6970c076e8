import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from unet_zoo_amd import _ffi
L = _ffi.lib(); dev = torch.device("cuda", 0)
N, Cin, Cout, H, W = 1, 32, 32, 4, 32
def run(x, dy):
    wsb = L.uz_conv_bwd_weight_workspace(Cin, Cout, N, H, W, 3)
    ws = torch.zeros(wsb // 4 + 64, device=dev); dw = torch.zeros(Cout, Cin, 3, 3, device=dev)
    st = torch.cuda.current_stream().cuda_stream
    xd, dyd_ = x.to(dev), dy.to(dev)
    _ffi.check(L.uz_conv_bwd_weight(xd.data_ptr(), Cin, Cin, dyd_.data_ptr(), Cout, Cout, dw.data_ptr(), None, N, H, W, 3, ws.data_ptr(), wsb, st), "wgrad")
    torch.cuda.synchronize()
    return dw.cpu()
one = torch.ones(N, Cin, H, W)
ref = torch.nn.grad.conv2d_weight(one, (Cout, Cin, 3, 3), one, padding=1)
dw = run(one, one)
print("ones: ref\n", ref[0, 0], "\n got\n", dw[0, 0], "\n got[5,7]\n", dw[5, 7])
# x = delta at (row 1, col 5) in channel 3; dy = ones -> dw[:,3,ky,kx] = 1 where y+ky-1=1, x+kx-1=5 has valid y,x (all) -> all 1
x = torch.zeros(N, Cin, H, W); x[0, 3, 1, 5] = 1.0
dw = run(x, one); ref = torch.nn.grad.conv2d_weight(x, (Cout, Cin, 3, 3), one, padding=1)
print("delta x: ref[0,3]\n", ref[0, 3], "\n got[0,3]\n", dw[0, 3], "\n got[0,4]\n", dw[0, 4], " total abs", float(dw.abs().sum()), float(ref.abs().sum()))
dyd = torch.zeros(N, Cout, H, W); dyd[0, 2, 1, 5] = 1.0
xr = torch.arange(H * W, dtype=torch.float32).reshape(1, 1, H, W).repeat(N, Cin, 1, 1) / 8.0
dw = run(xr, dyd); ref = torch.nn.grad.conv2d_weight(xr, (Cout, Cin, 3, 3), dyd, padding=1)
print("delta dy: ref[2,0]\n", ref[2, 0], "\n got[2,0]\n", dw[2, 0], "\n got[3,0]\n", dw[3, 0])
nz = (dw.abs() > 1e-6).nonzero()
print("nonzero entries:", nz.shape[0], "distinct co:", sorted(set(nz[:, 0].tolist()))[:10], "distinct ci count:", len(set(nz[:, 1].tolist())))
print("sample", dw[nz[0, 0], nz[0, 1]] if nz.shape[0] else None)
for row in range(4):
    for col in (0, 5, 17, 31):
        dyd = torch.zeros(N, Cout, H, W); dyd[0, 2, row, col] = 1.0
        dw = run(xr, dyd); ref = torch.nn.grad.conv2d_weight(xr, (Cout, Cin, 3, 3), dyd, padding=1)
        print(row, col, "ok" if torch.allclose(dw, ref, atol=1e-4) else "BAD", float(dw[2, 0, 1, 1]), float(ref[2, 0, 1, 1]))
