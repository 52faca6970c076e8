"""Functional fp32 CPU restatement of the reference's 3-D PHiSeg graph, models/phiseg3D.py (test oracle).

TEST INFRASTRUCTURE - see ``oracle/__init__.py``.  Same conventions as ``oracle/refgraph.py``: flat state_dict with the
reference's own key names, stock ATen ops, gradients from autograd.  Citations are relative to /root/reference.

What the reference can and cannot run (measured in the build container, tools/gen_golden.py `phiseg3d_case`):
  * ``Posterior.forward`` (:270-301, posterior and prior) and ``Likelihood.forward`` up to the level logits ``s_in`` (:357-397)
    run - PINNED by tests/golden/phiseg3d_*.npz, generated from those reference modules;
  * :398 hands a 2-element ``size`` to a 5-D ``interpolate`` and raises: the evident intent (nearest resize of each level to the
    full volume, as the 2-D model does at phiseg.py:341) is what `phiseg3d_forward` restates - that line, the loss built on its
    output, and the gradients are PARITY UNPINNED against the reference (they are pinned against autograd through this
    restatement);
  * ``forward(patch, mask)`` takes the mask ALREADY one-hot, (B, num_classes, D, H, W): utils.convert_to_onehot_torch returns a
    4-D label volume unchanged ("3D images from brats are already one hot encoded", utils.py:296-298), so :275-279 is
    cat(patch, mask - 0.5); ``loss(segm)`` takes the label map (:561);
  * the channel arithmetic of ``likelihood_post_c_path`` (:341-349) closes only when
    num_filters[latent_levels-1] == num_filters[latent_levels-1+lvl_diff]; other configurations raise in the reference and
    raise here.
"""
from __future__ import annotations

import numpy as np
import torch
import torch.nn.functional as F

from .refgraph import conv_unit, conv_bare, rev_sequence, _is_rev, kl_two_gauss_with_diag_cov, multinoulli_loss


def avgpool3d(x):
    """nn.AvgPool3d(2, 2, padding=0, ceil_mode=True) (phiseg3D.py:101)."""
    return F.avg_pool3d(x, kernel_size=2, stride=2, padding=0, ceil_mode=True)


def up2_trilinear(x):
    """interpolate(mode='trilinear', scale_factor=2, align_corners=True) (phiseg3D.py:146,306,376)."""
    return F.interpolate(x, mode="trilinear", scale_factor=2, align_corners=True)


def _stack(sd, prefixes, rev_prefix, x, bn_train):
    if _is_rev(sd, rev_prefix):
        return rev_sequence(sd, rev_prefix, x, bn_train)
    for p in prefixes:
        x = conv_unit(sd, p, x, bn_train)
    return x


def levels(sd, root):
    """(resolution levels, latent levels) of a Posterior as built at phiseg3D.py:222-248."""
    R = 0
    while any(k.startswith(f"{root}.contracting_path.{R}.") for k in sd):
        R += 1
    L = 0
    while any(k.startswith(f"{root}.sample_z_path.{L}.") for k in sd):
        L += 1
    return R, L


def phiseg3d_eps_shapes(depth, height, width, R, L, batch=1):
    """randn_like draws of one Posterior pass in draw order (coarsest latent level first, phiseg3D.py:289-293)."""
    return [(batch, 2, depth >> (R - 1 - k), height >> (R - 1 - k), width >> (R - 1 - k)) for k in range(L)]


def encoder3d(sd, root, x, eps, bn_train, z_override=None):
    """Posterior.forward (phiseg3D.py:270-301).  Returns (z, mu, sigma) lists indexed by latent level (0 = finest)."""
    R, L = levels(sd, root)
    blocks = []
    for i in range(R):
        base = 0
        if i != 0:
            x = avgpool3d(x)
            base = 1
        p = f"{root}.contracting_path.{i}.layers"
        x = _stack(sd, [f"{p}.{base + j}" for j in range(3)], f"{p}.{base}", x, bn_train)
        if i != R - 1:
            blocks.append(x)
    z, mu, sigma = [None] * L, [None] * L, [None] * L
    pre = x
    for k in range(L):
        if k != 0:
            u = up2_trilinear(z[-k])
            p = f"{root}.upsampling_path.{k - 1}.upconv_layer"
            u = _stack(sd, [p + ".0", p + ".1"], p, u, bn_train)
            pre = torch.cat([u, blocks[-k]], dim=1)
        p = f"{root}.sample_z_path.{k}"
        h = _stack(sd, [p + ".conv.0", p + ".conv.1"], p + ".conv.0", pre, bn_train)
        m = F.conv3d(h, sd[p + ".mu_conv.0.weight"], sd[p + ".mu_conv.0.bias"])
        s = F.softplus(F.conv3d(h, sd[p + ".sigma_conv.0.weight"], sd[p + ".sigma_conv.0.bias"]))
        mu[-k - 1], sigma[-k - 1] = m, s
        z[-k - 1] = m + s * eps[k]
        if z_override is not None:
            z[-k - 1] = z_override[-k - 1]
    return z, mu, sigma


def likelihood3d(sd, z, bn_train, full_size=None):
    """Likelihood.forward (phiseg3D.py:357-400).  Returns (s, s_in): the level logits resized to `full_size` (nearest; the
    reference's own resize call raises, see the module docstring) and un-resized."""
    root = "likelihood"
    L = len(z)
    diff = 0
    while f"{root}.likelihood_post_ups_path.0.{2 * diff + 1}.convolution.0.convolution.0.weight" in sd:
        diff += 1
    post_z = [None] * L
    for k in range(L):
        p = f"{root}.likelihood_ups_path.{k}"
        h = _stack(sd, [p + ".convolution.0", p + ".convolution.1"], p, z[-k - 1], bn_train)
        for t in range(diff):
            h = up2_trilinear(h)
            h = conv_unit(sd, f"{root}.likelihood_post_ups_path.{k}.{2 * t + 1}.convolution.0", h, bn_train)
        post_z[-k - 1] = h
    post_c = [None] * L
    post_c[L - 1] = post_z[L - 1]
    for i in reversed(range(L - 1)):
        cat = torch.cat([post_z[i], up2_trilinear(post_c[i + 1])], dim=1)
        p = f"{root}.likelihood_post_c_path.{i}"
        post_c[i] = _stack(sd, [p + ".convolution.0", p + ".convolution.1"], p, cat, bn_train)
    s, s_in = [None] * L, [None] * L
    for k in range(L):
        s_in[-k - 1] = conv_bare(sd, f"{root}.s_layer.{k}.convolution.0", post_c[-k - 1])
        s[-k - 1] = F.interpolate(s_in[-k - 1], size=list(full_size), mode="nearest") if full_size is not None else s_in[-k - 1]
    return s, s_in


def phiseg3d_forward(sd, patch, mask_onehot, eps, training=True, bn_train=True):
    """PHISeg3D.forward (phiseg3D.py:454-467) with the resize of :398 restated as intended.  mask_onehot: (B, K, D, H, W), see
    the module docstring; eps: 2 L noise volumes, the posterior's draws then the prior's."""
    R, L = levels(sd, "posterior")
    x = torch.cat([patch, mask_onehot.long().float() - 0.5], dim=1)          # utils.py:300 `.long()`, phiseg3D.py:278-279
    pz, pmu, psig = encoder3d(sd, "posterior", x, eps[:L], bn_train)
    if training:
        qz, qmu, qsig = encoder3d(sd, "prior", patch, eps[L:], bn_train, z_override=pz)
        s, s_in = likelihood3d(sd, pz, bn_train, patch.shape[-3:])
    else:
        qz, qmu, qsig = encoder3d(sd, "prior", patch, eps[L:], bn_train)
        s, s_in = likelihood3d(sd, qz, bn_train, patch.shape[-3:])
    return dict(s=s, s_in=s_in, post_z=pz, post_mu=pmu, post_sigma=psig, prior_z=qz, prior_mu=qmu, prior_sigma=qsig)


def phiseg3d_loss(out, mask, num_classes=2, exponential_weight=4.0):
    """PHISeg3D.loss (phiseg3D.py:529-611): hierarchical KL (4^level weights) + residual multinoulli loss.
    Returns (total, terms) with terms = [KL lvl 0..L-1, CE lvl 0..L-1]."""
    L = len(out["s"])
    kls = [exponential_weight ** i * kl_two_gauss_with_diag_cov(out["post_mu"][i], out["post_sigma"][i], out["prior_mu"][i], out["prior_sigma"][i])
           for i in range(L)]
    ces, acc = [None] * L, None
    for i in reversed(range(L)):
        acc = out["s"][i] if acc is None else acc + out["s"][i]
        ces[i] = multinoulli_loss(acc, mask, num_classes)
    total = 0
    for i in reversed(range(L)):
        total = total + kls[i]
    for i in reversed(range(L)):
        total = total + ces[i]
    return total, kls + ces


def synthetic_volume(in_ch, num_classes, dhw, seed, eps_shapes):
    """BraTS-like synthetic volume: channels N(0, 0.25^2) clipped to +-0.5; label map = nested random balls (labels
    0..num_classes-1), handed to forward() one-hot as the reference's BraTS pipeline does (utils.py:296-298) and to loss() as
    a label map (phiseg3D.py:561)."""
    rng = np.random.Generator(np.random.PCG64(seed))
    D, H, W = dhw
    x = np.clip(rng.standard_normal((1, in_ch, D, H, W)).astype(np.float32) * 0.25, -0.5, 0.5)
    zz, yy, xx = np.mgrid[0:D, 0:H, 0:W]
    lab = np.zeros((D, H, W), np.int64)
    c = [rng.uniform(0.35, 0.65) * n for n in dhw]
    for k in range(1, num_classes):
        r = min(dhw) * 0.45 / k
        lab[((zz - c[0]) ** 2 + (yy - c[1]) ** 2 + (xx - c[2]) ** 2) <= r * r] = k
    onehot = np.stack([(lab == k) for k in range(num_classes)]).astype(np.float32)[None]
    eps = [rng.standard_normal(s).astype(np.float32) for s in eps_shapes]
    return x, onehot, lab.astype(np.float32)[None, None], eps
