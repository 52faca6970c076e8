#!/usr/bin/env python3
"""Print per-tensor parity errors of the native PHISeg vs the golden fixtures (diagnostic, GPU box)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import oracle
from tests import _golden as G
from tests.test_phiseg_gpu import _model, _inputs

for name in (sys.argv[1:] or ["phiseg_small", "phiseg_full_digest"]):
    arrays, meta = G.load(name)
    net, _ = _model(meta); net.train()
    x, mask, eps = _inputs(meta, 0)
    s = net.forward(x, mask, training=True, eps=eps)
    loss = net.loss(mask); loss.backward()
    st = meta["steps"][0]
    print(name, "loss", float(loss), st["loss"], abs(float(loss)-st["loss"])/abs(st["loss"]))
    for k, v in st["loss_dict"].items():
        print("  ", k, float(net.loss_dict[k]), v)
    full = "s0" in arrays
    for l in range(5):
        if full:
            print("  s%d maxabs %.3e  mu %.3e sigma %.3e priormu %.3e" % (l, G.maxabs(s[l].cpu().numpy(), arrays[f"s{l}"]),
                  G.maxabs(net.posterior_mu[l].cpu().numpy(), arrays[f"post_mu{l}"]), G.maxabs(net.posterior_sigma[l].cpu().numpy(), arrays[f"post_sigma{l}"]),
                  G.maxabs(net.prior_mu[l].cpu().numpy(), arrays[f"prior_mu{l}"])))
        else:
            idx = arrays["s_idx"]
            print("  s%d maxabs %.3e  mu %.3e sigma %.3e priormu %.3e" % (l, G.maxabs(s[l].cpu().numpy().reshape(-1)[idx], arrays[f"s{l}_samp"]),
                  G.maxabs(net.posterior_mu[l].cpu().numpy(), arrays[f"post_mu{l}"]), G.maxabs(net.posterior_sigma[l].cpu().numpy(), arrays[f"post_sigma{l}"]),
                  G.maxabs(net.prior_mu[l].cpu().numpy(), arrays[f"prior_mu{l}"])))
    noise = G.bn_shadowed_biases(dict(net.named_parameters()).keys())
    rows = []
    for k, p in net.named_parameters():
        if p.grad is None or k in noise: continue
        if full:
            ref = arrays["grad:" + k]
            rows.append((G.maxabs(p.grad.cpu().numpy(), ref) / (1e-3 + float(np.abs(ref).max())), k, float(np.abs(ref).max())))
        else:
            n = st["grad_norms"][k]; mine = float(p.grad.double().norm())
            pick, vals = st["grad_samples"][k]
            got = p.grad.reshape(-1)[torch.tensor(pick)].cpu().numpy()
            rows.append((max(abs(mine-n), float(np.max(np.abs(got-np.array(vals))))) / max(n, 1e-3), k, n))
    rows.sort(reverse=True)
    for r in rows[:12]: print("   grad rel %.3e  %s  (scale %.3e)" % r)
    print("   median grad rel %.3e" % rows[len(rows)//2][0])
