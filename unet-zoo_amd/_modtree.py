"""Parameter / buffer holders that reproduce the reference's ``state_dict`` key names.

The reference builds deep ``nn.Module`` trees (models/phiseg.py, torchlayers.py); here the trees
exist only so that ``state_dict()``, ``load_state_dict()``, ``parameters()``, ``train()`` /
``eval()`` behave identically (820 keys for PHISeg).  All tensors are views into the flat device
buffers of a ``ParamTable`` - the kernels read those buffers directly.
"""
import torch
import torch.nn as nn


class Holder(nn.Module):
    """Name-space node without behaviour."""

    def forward(self, *a, **k):  # pragma: no cover
        raise RuntimeError("Holder modules only carry parameters; call the model's forward()")


def attach(root, ptab):
    """Register every entry of the ParamTable under its dotted key on `root`."""
    for key, shape, kind in ptab.spec:
        parts = key.split(".")
        node = root
        for name in parts[:-1]:
            child = node._modules.get(name)
            if child is None:
                child = Holder()
                node.add_module(name, child)
            node = child
        leaf = parts[-1]
        if kind in ("conv_w", "conv_b", "bn_w", "bn_b"):
            node.register_parameter(leaf, nn.Parameter(ptab.pview(key), requires_grad=True))
        elif kind in ("bn_rm", "bn_rv"):
            node.register_buffer(leaf, ptab.bview(key))
        else:
            node.register_buffer(leaf, ptab.nbtview(key))


def conv_unit_spec(prefix, cin, cout, k=3, norm=True):
    """Entries of one reference Conv2D unit (torchlayers.py:7-29): convolution.0 = Conv2d,
    convolution.1 = BatchNorm2d (absent when norm=nn.Identity)."""
    out = [(f"{prefix}.convolution.0.weight", (cout, cin, k, k), "conv_w"),
           (f"{prefix}.convolution.0.bias", (cout,), "conv_b")]
    if norm:
        out += [(f"{prefix}.convolution.1.weight", (cout,), "bn_w"),
                (f"{prefix}.convolution.1.bias", (cout,), "bn_b"),
                (f"{prefix}.convolution.1.running_mean", (cout,), "bn_rm"),
                (f"{prefix}.convolution.1.running_var", (cout,), "bn_rv"),
                (f"{prefix}.convolution.1.num_batches_tracked", (), "bn_nbt")]
    return out


def rev_sequence_spec(prefix, cin, cout, depth):
    """Entries of one reference ReversibleSequence (torchlayers.py:55-82): `inital_conv` (sic) = 1x1 Conv2D unit when the
    channel counts differ, then `depth` revtorch ReversibleBlocks whose F / G are one 3x3 Conv2D unit on half the channels.
    Key names follow revtorch 0.2.0's module attributes (`reversible_blocks`, `f_block`, `g_block`)."""
    out = []
    if cin != cout:
        out += conv_unit_spec(f"{prefix}.inital_conv", cin, cout, k=1)
    for i in range(depth):
        out += conv_unit_spec(f"{prefix}.sequence.reversible_blocks.{i}.f_block.0", cout // 2, cout // 2)
        out += conv_unit_spec(f"{prefix}.sequence.reversible_blocks.{i}.g_block.0", cout // 2, cout // 2)
    return out


def plain_conv_spec(prefix, cin, cout, k):
    return [(f"{prefix}.weight", (cout, cin, k, k), "conv_w"), (f"{prefix}.bias", (cout,), "conv_b")]


def init_default(ptab):
    """PyTorch's default nn.Conv2d / nn.BatchNorm2d initialisation (what PHiSeg keeps, since
    phiseg.py:36 is commented out): kaiming_uniform(a=sqrt(5)) weights, U(+-1/sqrt(fan_in)) biases,
    BN gamma=1, beta=0, running stats (0, 1)."""
    import math
    last_fan_in = 1
    for key, shape, kind in ptab.spec:
        if kind == "conv_w":
            w = ptab.pview(key)
            nn.init.kaiming_uniform_(w, a=math.sqrt(5))
            last_fan_in = shape[1] * shape[2] * shape[3]
        elif kind == "conv_b":
            bound = 1.0 / math.sqrt(last_fan_in)
            nn.init.uniform_(ptab.pview(key), -bound, bound)
        elif kind == "bn_w":
            ptab.pview(key).fill_(1.0)
        elif kind == "bn_b":
            ptab.pview(key).zero_()
        elif kind == "bn_rm":
            ptab.bview(key).zero_()
        elif kind == "bn_rv":
            ptab.bview(key).fill_(1.0)
    ptab.nbt.zero_()
