"""Experiment (CPU, not product code): replace every 3x3 conv of the CPU oracle by a 3-way bf16 operand split (6 or 9 piece
products, fp32 accumulate) and compare PHiSeg loss / logits / gradients with the plain fp32 oracle and an fp64 run."""
import sys; import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, torch.nn.functional as F, numpy as np
import oracle
from oracle import refgraph as R
from tests import _golden as G

torch.set_num_threads(8)
def split3(t):
    a1 = t.to(torch.bfloat16).to(torch.float32); r = t - a1
    a2 = r.to(torch.bfloat16).to(torch.float32); r = r - a2
    a3 = r.to(torch.bfloat16).to(torch.float32)
    return a1, a2, a3
orig = F.conv2d
MODE = {'n': 0}
class SplitConv(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, w, nprod):
        ctx.save_for_backward(x, w); ctx.nprod = nprod
        return emu(x, w, nprod, 'fwd')
    @staticmethod
    def backward(ctx, gy):
        x, w = ctx.saved_tensors
        gx = emu_dgrad(gy, w, ctx.nprod)
        gw = emu_wgrad(x, gy, w.shape, ctx.nprod)
        return gx, gw, None
PAIRS6 = [(0,0),(0,1),(1,0),(0,2),(2,0),(1,1)]
PAIRS9 = [(i,j) for i in range(3) for j in range(3)]
def emu(x, w, nprod, _):
    xs, ws = split3(x), split3(w)
    out = None
    for i, j in (PAIRS6 if nprod == 6 else PAIRS9)[::-1]:      # small terms first
        t = orig(xs[i], ws[j], None, padding=1)
        out = t if out is None else out + t
    return out
def emu_dgrad(gy, w, nprod):
    gs, ws = split3(gy), split3(w)
    out = None
    for i, j in (PAIRS6 if nprod == 6 else PAIRS9)[::-1]:
        t = torch.nn.grad.conv2d_input(gy.shape[:1] + (w.shape[1],) + gy.shape[2:], ws[j], gs[i], padding=1)
        out = t if out is None else out + t
    return out
def emu_wgrad(x, gy, wshape, nprod):
    xs, gs = split3(x), split3(gy)
    out = None
    for i, j in (PAIRS6 if nprod == 6 else PAIRS9)[::-1]:
        t = torch.nn.grad.conv2d_weight(xs[i], wshape, gs[j], padding=1)
        out = t if out is None else out + t
    return out
def patched(x, w, b=None, stride=1, padding=0, *a, **k):
    if MODE['n'] and w.shape[-1] == 3 and x.dtype == torch.float32:
        y = SplitConv.apply(x, w, MODE['n'])
        return y if b is None else y + b.view(1, -1, 1, 1)
    return orig(x, w, b, stride, padding, *a, **k)
F.conv2d = patched; R.F.conv2d = patched

name = sys.argv[1] if len(sys.argv) > 1 else "phiseg_mid"
arrays, meta = G.load(name)
spec = G.spec_of(meta)
def run(mode, dtype):
    MODE['n'] = mode
    sd = oracle.deterministic_state_dict(spec, seed=meta["weight_seed"])
    leaves = {k: (v.to(dtype).clone().requires_grad_(True) if v.dtype.is_floating_point and "running_" not in k else (v.to(dtype) if v.dtype.is_floating_point else v.clone())) for k, v in sd.items()}
    shapes = oracle.phiseg_eps_shapes(meta["batch"], meta["hw"], meta["hw"])
    x, mask, eps = oracle.synthetic_batch(meta["batch"], meta["hw"], meta["hw"], seed=20201004, eps_shapes=shapes + shapes)
    e = [torch.from_numpy(a).to(dtype) for a in eps]
    out = oracle.phiseg_forward(leaves, torch.from_numpy(x).to(dtype), torch.from_numpy(mask), dict(posterior=e[:5], prior=e[5:]))
    total, _ = oracle.phiseg_loss(out, torch.from_numpy(mask))
    total.backward()
    s = out["s"] if isinstance(out, dict) and "s" in out else None
    return out, float(total), {k: v.grad.double() for k, v in leaves.items() if getattr(v, "grad", None) is not None}
o64, l64, g64 = run(0, torch.float64)
res = {}
for mode in (0, 6, 9):
    o, l, g = run(mode, torch.float32)
    # logits: find list of tensors in out
    def logits(o):
        for key in ("s_out_list", "s", "s_list"):
            if isinstance(o, dict) and key in o: return o[key]
        return [v for v in (o.values() if isinstance(o, dict) else o) if isinstance(v, (list, tuple))][0]
    ls, l6 = logits(o), logits(o64)
    err = max(float((a.double() - b).abs().max()) for a, b in zip(ls, l6))
    shadow = G.bn_shadowed_biases([k for k, _, _ in spec])
    rel = []
    for k in g:
        if k in shadow: continue
        d = float((g[k] - g64[k]).abs().max()); m = float(g64[k].abs().max())
        if m > 0: rel.append(d / m)
    print(f"mode {mode}: loss {l:.6f} (fp64 {l64:.6f}, rel {abs(l-l64)/abs(l64):.2e})  max logit err vs fp64 {err:.3e}  grad rel err median {np.median(rel):.2e} max {max(rel):.2e}")
