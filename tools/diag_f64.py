#!/usr/bin/env python3
"""Gradient accuracy against an fp64 ground truth: HIP fp32 path vs the fp32 CPU oracle (diagnostic)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, oracle
from tests import _golden as G
from tests.test_phiseg_gpu import _model, _inputs
arrays, meta = G.load(sys.argv[1] if len(sys.argv) > 1 else "phiseg_mid")
net, sd0 = _model(meta); net.train()
x, mask, eps = _inputs(meta, 0)
s = net.forward(x, mask, training=True, eps=eps); loss = net.loss(mask); loss.backward()
def cpu(dtype):
    lv = G.leaves({k: (v.to(dtype) if v.dtype.is_floating_point else v) for k, v in sd0.items()})
    e = [t.cpu().to(dtype) for t in eps]
    out = oracle.phiseg_forward(lv, x.cpu().to(dtype), mask.cpu().to(dtype), dict(posterior=e[:5], prior=e[5:]))
    total, _ = oracle.phiseg_loss(out, mask.cpu().to(dtype)); total.backward()
    return out, {k: v.grad for k, v in lv.items() if v.requires_grad and v.grad is not None}
o32, g32 = cpu(torch.float32); o64, g64 = cpu(torch.float64)
for l in range(5):
    print("logits lvl%d: |hip-f64| %.2e   |cpu32-f64| %.2e" % (l, float((s[l].cpu().double() - o64["s"][l]).abs().max()), float((o32["s"][l].double() - o64["s"][l]).abs().max())))
noise = G.bn_shadowed_biases(g64.keys())
rh, rc = [], []
for k, p in net.named_parameters():
    if k in noise or k not in g64: continue
    sc = float(g64[k].abs().max()) + 1e-12
    rh.append(float((p.grad.cpu().double() - g64[k]).abs().max()) / sc)
    rc.append(float((g32[k].double() - g64[k]).abs().max()) / sc)
rh, rc = np.array(rh), np.array(rc)
print("grad rel err vs f64:  hip median %.2e max %.2e | cpu32 median %.2e max %.2e" % (np.median(rh), rh.max(), np.median(rc), rc.max()))
print("tensors where hip err > 3x cpu32 err:", int((rh > 3 * rc + 1e-7).sum()), "of", len(rh), "; where hip err < cpu32 err:", int((rh < rc).sum()))
