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
