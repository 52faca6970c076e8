"""Helpers shared by the parity tests: load committed golden fixtures (data only)."""
import json
import os
from collections import OrderedDict

import numpy as np
import torch

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load(name):
    arrays = np.load(os.path.join(GOLDEN, name + ".npz"))
    with open(os.path.join(GOLDEN, name + ".json")) as f:
        meta = json.load(f)
    return arrays, meta


def spec_of(meta):
    return [(k, tuple(s), kd) for k, s, kd in meta["spec"]]


def leaves(sd):
    """Clone a state_dict into autograd leaves (float params) + plain buffers."""
    out = OrderedDict()
    for k, v in sd.items():
        t = v.clone()
        if t.dtype.is_floating_point and "running_" not in k:
            t.requires_grad_(True)
        out[k] = t
    return out


def maxabs(a, b):
    return float(np.max(np.abs(np.asarray(a, np.float64) - np.asarray(b, np.float64)))) if np.size(a) else 0.0


def bn_shadowed_biases(keys):
    """Conv biases that feed a training-mode BatchNorm (torchlayers.py:18-20): the BN mean
    subtraction cancels them, so their true gradient is exactly 0 and what autograd returns
    is floating-point cancellation noise (order 1e-7 x |dy| x pixels) that depends on the
    summation order of each backend.  Adam then turns that noise into +-lr updates.  These
    entries carry no signal in the reference either and are excluded from parity gates."""
    ks = set(keys)
    return {k for k in ks if k.endswith(".convolution.0.bias") and (k[:-len("0.bias")] + "1.weight") in ks}


def check_first_adam_step(theta0, theta1, grads_hip, grads_ref, skip=(), lr=1e-3, wd=1e-5, tol=1e-5, min_cover=0.5):
    """Non-vacuous check of the first optimiser step ("post-step parameters").

    torch.optim.Adam's first update is theta1 = theta0 - lr * g' / (|g'| + 1e-8) with g' = g + wd * theta0, i.e. a step of
    +-lr whose SIGN is the information.  For every entry whose reference gradient is larger than the measured
    gradient deviation of its tensor (3x max|g_hip - g_ref|, at least 1e-6 of the tensor's max) - so that both
    implementations agree on the sign by construction of the gradient gate - the native parameter after step 1
    must equal the reference update to `tol` (an optimiser with a wrong sign, a missing bias correction, a
    misapplied weight decay or a skipped tensor fails by ~lr = 100 x tol).  Returns the fraction of entries covered
    and asserts it is at least `min_cover`, so the gate cannot pass by excluding everything.
    theta0 / theta1 / grads_*: dicts of numpy arrays keyed like state_dict (grads may lack keys = grad None)."""
    covered = total = 0
    for k, g_ref in grads_ref.items():
        if g_ref is None or k in skip:
            continue
        g_ref = np.asarray(g_ref, np.float64)
        g_hip = np.asarray(grads_hip[k], np.float64)
        t0 = np.asarray(theta0[k], np.float64)
        gp = g_ref + wd * t0
        ref1 = t0 - lr * gp / (np.abs(gp) + 1e-8)
        thr = max(3.0 * float(np.max(np.abs(g_hip - g_ref))), 1e-6 * float(np.max(np.abs(g_ref))), 1e-7)
        sel = np.abs(gp) > thr
        total += g_ref.size
        covered += int(sel.sum())
        if sel.any():
            d = np.abs(np.asarray(theta1[k], np.float64) - ref1)[sel]
            assert float(d.max()) <= tol, (k, float(d.max()), int(sel.sum()), g_ref.size)
    assert total > 0 and covered / total >= min_cover, (covered, total)
    return covered / total
