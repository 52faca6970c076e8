"""Vanilla U-Net on the native HIP path - drop-in for the reference ``models/unet.py`` ``Unet``:
same constructor keywords (unet.py:88-90), ``forward(x, mask=None, training=True, val=False)``,
``loss(mask)`` (mean cross entropy over all pixels, unet.py:159-165), ``sample()``, and the same
``state_dict`` keys.  Blocks are [AvgPool2d(2,2,ceil_mode)] + 3 x (Conv3x3 + ReLU) without any
norm (unet.py:12-40); the decoder up-samples bilinearly with align_corners=False and concatenates
``[up, bridge]`` (unet.py:65-75) - here both halves are written in place into one buffer.
Also serves as the feature backbone of the Probabilistic U-Net (apply_last_layer=False)."""
import math

import torch
import torch.nn as nn

from .._engine import NativeModel, conv_unit
from .._modtree import plain_conv_spec, rev_sequence_spec


def unet_spec(input_channels, num_classes, num_filters, apply_last_layer=True, prefix="", reversible=False):
    """reversible=True: every DownConvBlock body is ReversibleSequence(in, out, reversible_depth=3) (unet.py:32-35)."""
    nf, out = list(num_filters), []
    for i in range(len(nf)):
        cin = input_channels if i == 0 else nf[i - 1]
        base = 0 if i == 0 else 1
        if reversible:
            out += rev_sequence_spec(f"{prefix}contracting_path.{i}.layers.{base}", cin, nf[i], 3)
            continue
        for j in range(3):
            out += plain_conv_spec(f"{prefix}contracting_path.{i}.layers.{base + 2 * j}", cin if j == 0 else nf[i], nf[i], 3)
    prev = nf[-1]
    for k, i in enumerate(range(len(nf) - 2, -1, -1)):
        cin = prev + nf[i]
        if reversible:
            out += rev_sequence_spec(f"{prefix}upsampling_path.{k}.conv_block.layers.0", cin, nf[i], 3)
        else:
            for j in range(3):
                out += plain_conv_spec(f"{prefix}upsampling_path.{k}.conv_block.layers.{2 * j}", cin if j == 0 else nf[i], nf[i], 3)
        prev = nf[i]
    if apply_last_layer:
        out += plain_conv_spec(f"{prefix}last_layer", prev, num_classes, 1)
    return out


def init_unet_weights(ptab, prefix="", skip=("last_layer",)):
    """utils.init_weights (utils.py:78-83): kaiming_normal_(fan_in, relu) weights, truncated-normal
    (std 1e-3) biases for every conv of the blocks; `last_layer` keeps PyTorch's default init."""
    for key, shape, kind in ptab.spec:
        if not key.startswith(prefix):
            continue
        tail = key[len(prefix):]
        if any(tail.startswith(s) for s in skip):
            if kind == "conv_w":
                nn.init.kaiming_uniform_(ptab.pview(key), a=math.sqrt(5))
                fan_in = shape[1] * shape[2] * shape[3]
            elif kind == "conv_b":
                nn.init.uniform_(ptab.pview(key), -1.0 / math.sqrt(fan_in), 1.0 / math.sqrt(fan_in))
            continue
        if kind == "conv_w":
            nn.init.kaiming_normal_(ptab.pview(key), mode="fan_in", nonlinearity="relu")
        elif kind == "conv_b":
            nn.init.trunc_normal_(ptab.pview(key), mean=0.0, std=1e-3, a=-2e-3, b=2e-3)
        elif kind == "bn_w":                       # BatchNorm2d of the reversible variant's Conv2D units: torch defaults
            ptab.pview(key).fill_(1.0)
        elif kind == "bn_rv":
            ptab.bview(key).fill_(1.0)


def build_unet_graph(plan, prefix, x, num_filters, apply_last_layer, final_out=None, reversible=False):
    """Emit the U-Net forward (unet.py:129-157) into `plan`; returns the output View
    (logits, or the last block's features when apply_last_layer is False)."""
    nf = list(num_filters)
    n = len(nf)
    cats = {}
    for i in range(n):
        if i != 0:
            x = plan.avgpool(x, f"{prefix}pool{i}")
        out = None
        if i != n - 1:      # this block is the `bridge` of torch.cat([up, bridge], 1): write it in place
            cats[i] = plan.buf(f"{prefix}cat{i}", nf[i + 1] + nf[i], x.H, x.W)
            out = cats[i].slice(nf[i + 1], nf[i])
        base = 0 if i == 0 else 1
        if reversible:
            x = plan.rev_sequence(x, f"{prefix}contracting_path.{i}.layers.{base}", nf[i], 3, conv_unit, out=out)
            continue
        for j in range(3):
            x = plan.conv_relu(x, f"{prefix}contracting_path.{i}.layers.{base + 2 * j}", out=out if j == 2 else None)
    for k, i in enumerate(range(n - 2, -1, -1)):
        cat = cats[i]
        plan.bilinear(x, False, out=cat.slice(0, nf[i + 1]))
        x = cat
        last_block = (i == 0)
        if reversible:
            o = final_out if (last_block and not apply_last_layer) else None
            x = plan.rev_sequence(x, f"{prefix}upsampling_path.{k}.conv_block.layers.0", nf[i], 3, conv_unit, out=o)
            continue
        for j in range(3):
            o = final_out if (last_block and j == 2 and not apply_last_layer) else None
            x = plan.conv_relu(x, f"{prefix}upsampling_path.{k}.conv_block.layers.{2 * j}", out=o)
    if apply_last_layer:
        x = plan.conv_bare(x, f"{prefix}last_layer", out=final_out)
    return x


class Unet(NativeModel):
    # one chain of device-filling kernels: a second dependency lane only adds cross-queue barriers (4.74 -> 4.63 ms per step)
    default_lanes = 1

    def __init__(self, input_channels, num_classes, num_filters, initializers=None, apply_last_layer=True, padding=True,
                 reversible=False, training=False, latent_dim=3, no_convs_fcomb=4, beta=1.0, device=None):
        super().__init__()
        self.reversible = bool(reversible)
        self.input_channels, self.num_classes, self.num_filters = input_channels, num_classes, list(num_filters)
        self.padding, self.activation_maps, self.apply_last_layer = padding, [], apply_last_layer
        self.prediction = None
        self._init_storage(unet_spec(input_channels, num_classes, num_filters, apply_last_layer, reversible=self.reversible), device)
        init_unet_weights(self._ptab)

    def _build(self, N, H, W):
        plan = self._new_plan(N, bool(self.training) if self.reversible else False)      # only the reversible variant has BatchNorms
        plan.bn_prefixes_nbt = []
        io = {"x": plan.buf("x", self.input_channels, H, W, requires_grad=False)}
        io["pred"] = build_unet_graph(plan, "", io["x"], self.num_filters, self.apply_last_layer, reversible=self.reversible)
        plan.loss_phase()
        plan.total = plan.vec("total", 1)
        io["mask"] = plan.buf("loss_mask", 1, H, W, requires_grad=False)
        if self.apply_last_layer:
            plan.residual_ce([io["pred"]], io["mask"], plan.total, post_scale=1.0 / (H * W))
        plan.finalize(want_backward=self.apply_last_layer and (plan.bn_training or not self.reversible))
        plan.io = io
        return plan

    def forward(self, x, mask=None, training=True, val=False):
        self._require_gpu()
        N, _, H, W = x.shape
        plan = self._plan((N, H, W, bool(self.training) and self.reversible), lambda: self._build(N, H, W))
        plan.tensor(plan.io["x"]).copy_(x)
        self._run(plan, "fwd")
        if self.training and self.reversible:
            self._bump_nbt(plan)
        self._cur = plan
        out = plan.tensor(plan.io["pred"])
        if val:
            self.activation_maps.append(out)
        self.prediction = out
        return out

    def sample(self, testing=True):
        return self.prediction

    def loss(self, mask):
        plan = self._cur
        if plan is None:
            raise RuntimeError("call forward() before loss()")
        N, _, H, W = plan.tensor(plan.io["mask"]).shape
        plan.tensor(plan.io["mask"]).copy_(mask.reshape(N, 1, H, W))
        if torch.is_grad_enabled():
            return self._loss_tensor(plan)
        plan.run("loss", self._stream())
        return plan.tensor(plan.total).reshape(()).clone()
