"""Deviation of the full-size PHiSeg digests (B=2 / B=32) from the reference golden values in the current UZ_CONV_MATH mode."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, numpy as np
from tests import _golden as G
from tests.test_phiseg_gpu import _model, _inputs
for fixture in sys.argv[1:] or ["phiseg_full_digest", "phiseg_full_b32_digest"]:
    arrays, meta = G.load(fixture)
    net, _ = _model(meta); net.train()
    x, mask, eps = _inputs(meta, 0)
    s = net.forward(x, mask, training=True, eps=eps)
    loss = net.loss(mask); loss.backward()
    st = meta["steps"][0]
    idx = arrays["s_idx"]
    ml = max(G.maxabs(s[l].cpu().numpy().reshape(-1)[idx], arrays[f"s{l}_samp"]) for l in range(5))
    mm = max(G.maxabs(net.posterior_mu[l].cpu().numpy(), arrays[f"post_mu{l}"]) for l in range(5))
    ms = max(G.maxabs(net.prior_sigma[l].cpu().numpy(), arrays[f"prior_sigma{l}"]) for l in range(5))
    noise = G.bn_shadowed_biases(st["grad_norms"].keys())
    params = dict(net.named_parameters())
    devs = []
    for k, n in st["grad_norms"].items():
        if k in noise: continue
        mine = float(params[k].grad.double().norm())
        devs.append((abs(mine - n) / max(n, 1e-3), k))
    devs.sort(reverse=True)
    print(f"{os.environ.get('UZ_CONV_MATH','default'):8s} {fixture}: loss rel {abs(float(loss)-st['loss'])/abs(st['loss']):.2e} logits {ml:.2e} post_mu {mm:.2e} prior_sigma {ms:.2e}; grad-norm dev top: " + ", ".join(f"{d:.2e} {k.split('.')[0][:4]}..{'.'.join(k.split('.')[-3:])}" for d, k in devs[:4]))
