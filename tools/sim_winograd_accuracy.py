"""CPU experiment (not product code): what would Winograd F(2x2, 3x3) cost in ACCURACY on the split-fp16 matrix pipe?  (VERDICT r5 item 4:
"price a fewer-products convolution"; gate = error against fp64 at most 2x the fp32-MFMA kernel's, the gate of
tests/test_full_configs_gpu.py::test_split_kernels_at_baseline_size_vs_fp64.)

The heaviest layer's forward (3x3, 224 -> 128, 128 x 128 planes, post-ReLU inputs, Kaiming weights) is evaluated five ways against an fp64
direct convolution:  (a) fp32 direct (the fp32-MFMA kernels' arithmetic);  (b) the product's split path: operands scaled by a power of two
from their bound, two fp16 pieces each, three piece products, fp32 accumulation;  (c) Winograd with fp32 transforms and fp32 products;
(d) Winograd with fp32 transforms and the TRANSFORMED operands split like (b) - 16 batched [Cout x Cin] x [Cin x tiles] products instead of
36 per 2 x 2 output tile;  (e) = (d) with the filter transform in fp64 (weights are transformed once per step: cheap to do better).
Printed: max and rms error relative to max|y| / rms(y), in absolute terms and in units of (a)."""
import sys
import torch
import torch.nn.functional as F

torch.manual_seed(0)
torch.set_num_threads(8)
N, Cin, Cout, H, W = int(sys.argv[1]) if len(sys.argv) > 1 else 2, 224, 128, 128, 128
x = torch.randn(N, Cin, H, W).clamp_min(0)
w = torch.randn(Cout, Cin, 3, 3) * (2.0 / (9 * Cin)) ** 0.5
ref = F.conv2d(x.double(), w.double(), padding=1)


def scale_of(t):
    """power of two s with max|t| * s < 2^14 (split_f16.h: split_scale)"""
    import math
    return 2.0 ** (13 - math.floor(math.log2(float(t.abs().max()))))


def split2(t, s):
    v = (t * s).clamp(-65504, 65504)
    h1 = v.half().float()
    h2 = (v - h1).half().float()
    return h1, h2


def report(name, y, base=None):
    e = (y.double() - ref)
    mx, rms = float(e.abs().max() / ref.abs().max()), float(e.pow(2).mean().sqrt() / ref.pow(2).mean().sqrt())
    extra = "" if base is None else f"   = {mx / base[0]:5.2f}x / {rms / base[1]:5.2f}x the fp32 direct error"
    print(f"{name:62s} max {mx:.3e}  rms {rms:.3e}{extra}")
    return mx, rms


base = report("(a) fp32 direct", F.conv2d(x, w, padding=1))
sx, sw = scale_of(x), scale_of(w)
x1, x2 = split2(x, sx)
w1, w2 = split2(w, sw)
y = (F.conv2d(x2, w1, padding=1) + F.conv2d(x1, w2, padding=1) + F.conv2d(x1, w1, padding=1)) / (sx * sw)
report("(b) split-fp16 direct (the product's arithmetic)", y, base)

Bt = torch.tensor([[1., 0, -1, 0], [0, 1, 1, 0], [0, -1, 1, 0], [0, 1, 0, -1]])
G = torch.tensor([[1., 0, 0], [.5, .5, .5], [.5, -.5, .5], [0, 0, 1]])
At = torch.tensor([[1., 1, 1, 0], [0, 1, -1, -1]])


def winograd(x, w, mode, g_dtype=torch.float32):
    xp = F.pad(x, (1, 1, 1, 1))
    d = xp.unfold(2, 4, 2).unfold(3, 4, 2)                          # N, C, H/2, W/2, 4, 4
    V = torch.einsum("ij,nchwjk,lk->nchwil", Bt, d, Bt)             # B^T d B
    U = torch.einsum("ij,ocjk,lk->ocil", G.to(g_dtype), w.to(g_dtype), G.to(g_dtype)).float()      # G g G^T
    n, c, th, tw = V.shape[:4]
    Vm = V.reshape(n, c, th * tw, 16).permute(3, 1, 0, 2).reshape(16, c, n * th * tw)      # [xi][ci][tiles]
    Um = U.reshape(Cout, Cin, 16).permute(2, 0, 1)                                          # [xi][co][ci]
    if mode == "fp32":
        M = torch.bmm(Um, Vm)
    else:
        # one scale per transformed tensor (per-frequency scales would be possible too: "perxi")
        if mode == "split":
            sv, su = scale_of(Vm), scale_of(Um)
            v1, v2 = split2(Vm, sv)
            u1, u2 = split2(Um, su)
            M = (torch.bmm(u1, v2) + torch.bmm(u2, v1) + torch.bmm(u1, v1)) / (sv * su)
        else:
            M = torch.empty(16, Cout, Vm.shape[2])
            for xi in range(16):
                sv, su = scale_of(Vm[xi]), scale_of(Um[xi])
                v1, v2 = split2(Vm[xi], sv)
                u1, u2 = split2(Um[xi], su)
                M[xi] = (u1 @ v2 + u2 @ v1 + u1 @ v1) / (sv * su)
    M = M.reshape(4, 4, Cout, n, th, tw)
    Y = torch.einsum("ij,jkonhw,lk->onhwil", At, M, At)            # A^T M A: 2 x 2 outputs per tile
    return Y.permute(1, 0, 2, 4, 3, 5).reshape(n, Cout, 2 * th, 2 * tw)


report("(c) Winograd F(2x2,3x3), fp32 transforms and products", winograd(x, w, "fp32"), base)
report("(d) Winograd, transformed operands split-fp16, one scale each", winograd(x, w, "split"), base)
report("(d') ... one scale per frequency plane", winograd(x, w, "perxi"), base)
report("(e) (d) with the filter transform in fp64", winograd(x, w, "split", torch.float64), base)
