#!/usr/bin/env python3
"""Diagnostic: gradient / activation accuracy of the HIP path vs an fp64 oracle on the BASELINE architecture,
per conv-math mode (f32 MFMA / default / forced split), per sub-network, with intermediate activations."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, oracle
from tests import _golden as G
from unet_zoo_amd import _ffi
from unet_zoo_amd.models.phiseg import PHISeg

fixture = sys.argv[1] if len(sys.argv) > 1 else "phiseg_full_digest"
B = int(sys.argv[2]) if len(sys.argv) > 2 else None
arrays, meta = G.load(fixture)
if B:
    meta = dict(meta, batch=B)
sd0 = oracle.deterministic_state_dict(G.spec_of(meta), seed=meta["weight_seed"])
shapes = oracle.phiseg_eps_shapes(meta["batch"], meta["hw"], meta["hw"])
x, mask, eps = oracle.synthetic_batch(meta["batch"], meta["hw"], meta["hw"], seed=20201004, eps_shapes=shapes + shapes)
dev = torch.device("cuda", 0)


def cpu(dtype):
    lv = G.leaves({k: (v.to(dtype) if v.dtype.is_floating_point else v) for k, v in sd0.items()})
    e = [torch.from_numpy(t).to(dtype) for t in eps]
    out = oracle.phiseg_forward(lv, torch.from_numpy(x).to(dtype), torch.from_numpy(mask).to(dtype), dict(posterior=e[:5], prior=e[5:]))
    total, _ = oracle.phiseg_loss(out, torch.from_numpy(mask).to(dtype))
    total.backward()
    return out, {k: v.grad for k, v in lv.items() if v.requires_grad and v.grad is not None}


o32, g32 = cpu(torch.float32)
o64, g64 = cpu(torch.float64)
noise = G.bn_shadowed_biases(g64.keys())
L = _ffi.lib()
for mode, tag in ((0, "f32"), (1, "default"), (2, "split")):
    L.uz_set_conv_math(mode)
    net = PHISeg(1, 2, meta["filters"], latent_levels=5, image_size=(1, meta["hw"], meta["hw"]))
    net.load_state_dict(sd0)
    net.train()
    s = net.forward(torch.from_numpy(x).to(dev), torch.from_numpy(mask).to(dev), training=True, eps=[torch.from_numpy(e).to(dev) for e in eps])
    loss = net.loss(torch.from_numpy(mask).to(dev))
    loss.backward()
    print(f"== mode {tag}: loss {float(loss):.6f} (f64 {float(oracle.phiseg_loss(o64, torch.from_numpy(mask).double())[0]):.6f})")
    for l in range(5):
        print("  logits lvl%d |hip-f64| %.2e  |cpu32-f64| %.2e   post_mu |hip-f64| %.2e |cpu32-f64| %.2e" % (
            l, float((s[l].cpu().double() - o64["s"][l]).abs().max()), float((o32["s"][l].double() - o64["s"][l]).abs().max()),
            float((net.posterior_mu[l].cpu().double() - o64["posterior_mu"][l]).abs().max()),
            float((o32["posterior_mu"][l].double() - o64["posterior_mu"][l]).abs().max())))
    rows = []
    for k, p in net.named_parameters():
        if k in noise or k not in g64:
            continue
        sc = float(g64[k].abs().max()) + 1e-12
        rows.append((k, float((p.grad.cpu().double() - g64[k]).abs().max()) / sc, float((g32[k].double() - g64[k]).abs().max()) / sc))
    rh, rc = np.array([r[1] for r in rows]), np.array([r[2] for r in rows])
    print("  grad rel err vs f64: hip median %.2e max %.2e | cpu32 median %.2e max %.2e ; hip>3x cpu: %d of %d" % (
        np.median(rh), rh.max(), np.median(rc), rc.max(), int((rh > 3 * rc + 2e-5).sum()), len(rh)))
    for grp in ("posterior.contracting_path", "posterior.upsampling_path", "posterior.sample_z_path", "prior.contracting_path",
                "prior.sample_z_path", "likelihood.likelihood_ups_path", "likelihood.likelihood_post_ups_path", "likelihood.likelihood_post_c_path", "likelihood.s_layer"):
        sel = [r for r in rows if r[0].startswith(grp)]
        if sel:
            print("    %-40s hip median %.2e max %.2e | cpu32 median %.2e max %.2e" % (
                grp, np.median([r[1] for r in sel]), max(r[1] for r in sel), np.median([r[2] for r in sel]), max(r[2] for r in sel)))
    worst = sorted(rows, key=lambda r: -r[1] / (3 * r[2] + 2e-5))[:6]
    for k, a, b in worst:
        print("    worst: %-70s hip %.2e cpu32 %.2e" % (k, a, b))
L.uz_set_conv_math(-1)
