#!/usr/bin/env python3
"""After one train step: which parameters deviate from the CPU oracle's (diagnostic, GPU box)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, oracle
from tests import _golden as G
from tests.test_phiseg_gpu import _model, _inputs
from unet_zoo_amd.optim import FusedAdam
arrays, meta = G.load(sys.argv[1] if len(sys.argv) > 1 else "phiseg_mid")
net, sd0 = _model(meta); net.train(); opt = FusedAdam(net, lr=1e-3, weight_decay=1e-5)
x, mask, eps = _inputs(meta, 0)
net.forward(x, mask, training=True, eps=eps); loss = net.loss(mask); opt.zero_grad(); loss.backward()
g_mine = {k: (None if p.grad is None else p.grad.detach().cpu().clone()) for k, p in net.named_parameters()}
opt.step()
lv = G.leaves(sd0); e = [t.cpu() for t in eps]
out = oracle.phiseg_forward(lv, x.cpu(), mask.cpu(), dict(posterior=e[:5], prior=e[5:]))
total, _ = oracle.phiseg_loss(out, mask.cpu()); total.backward()
params = {k: v for k, v in lv.items() if v.requires_grad}
new = oracle.adam_reference_step(params, {k: v.grad for k, v in params.items()}, {})
noise = G.bn_shadowed_biases(params.keys())
rows = []
for k, p in net.named_parameters():
    d = (p.detach().cpu() - new[k]).abs()
    moved = (new[k] - sd0[k]).abs().max()
    gr = params[k].grad
    gerr = 0.0 if gr is None else float((g_mine[k] - gr).abs().max() / (gr.abs().max() + 1e-12))
    # fraction of entries whose update sign differs
    flips = float(((p.detach().cpu() - sd0[k]).sign() != (new[k] - sd0[k]).sign()).float().mean())
    rows.append((float(d.max()), k, float(moved), gerr, flips, k in noise))
rows.sort(reverse=True)
for r in rows[:25]:
    print("dp %.2e  %-75s moved %.2e  grad relerr %.2e  signflips %.3f %s" % (r[0], r[1], r[2], r[3], r[4], "NOISE-BIAS" if r[5] else ""))
print("non-noise tensors with dp > 2e-4:", sum(1 for r in rows if r[0] > 2e-4 and not r[5]), "of", len(rows))
tot_flip = sum(r[4] for r in rows if not r[5]) / sum(1 for r in rows if not r[5])
print("mean sign-flip fraction (non-noise):", tot_flip)
