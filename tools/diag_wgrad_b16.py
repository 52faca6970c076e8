"""Debug: which pixels of a bf16-stored operand does the weight gradient see?"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tests import _gpu as g
from unet_zoo_amd import _ffi
L = _ffi.lib(); L.uz_set_conv_math(3)
N, Cin, Cout, H, W = 2, 64, 64, 64, 64
d = g.dev()
wsb = L.uz_conv_bwd_weight_workspace(Cin, Cout, N, H, W, 3)
ws = torch.empty(wsb // 4 + 64, device=d)
def run(x, dy, xb, db):
    dw = torch.full((Cout, Cin, 3, 3), float("nan"), device=d)
    xs = x.to(d).to(torch.bfloat16) if xb else x.to(d)
    ds = dy.to(d).to(torch.bfloat16) if db else dy.to(d)
    g.call("uz_conv_bwd_weight_b16", xs, Cin, Cin, ds, Cout, Cout, dw, N, H, W, 3, ws, wsb, xb, db, None)
    return dw.cpu()
# x: channel ci holds value (1 + x coordinate) on row y -> dw[co, ci, 1, 1] with a one-hot dy at (b, co, y0, x0) reads x[b, ci, y0, x0] = 1 + x0
x = torch.zeros(N, Cin, H, W)
x += (1 + torch.arange(W).float()).view(1, 1, 1, W)
x += 100 * torch.arange(H).float().view(1, 1, H, 1)
for (b, y0, x0) in [(0, 0, 0), (0, 0, 1), (0, 0, 2), (0, 0, 3), (0, 0, 4), (0, 0, 5), (0, 1, 0), (0, 5, 33), (1, 7, 63), (0, 3, 31), (0, 3, 32)]:
    dy = torch.zeros(N, Cout, H, W); dy[b, 3, y0, x0] = 1.0
    ref = run(x, dy, 0, 0)[3, 5]
    a = run(x, dy, 0, 1)[3, 5]
    c = run(x, dy, 1, 0)[3, 5]
    print((b, y0, x0), "ref centre", float(ref[1, 1]), "| dy16:", a.flatten().tolist(), "| x16:", c.flatten().tolist(), "| ref:", ref.flatten().tolist())
