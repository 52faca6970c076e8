"""Functional fp32 CPU restatement of the reference's model graphs (test oracle).

TEST INFRASTRUCTURE - see ``oracle/__init__.py``.  Every function works on a
flat ``{state_dict key: tensor}`` mapping (the reference's own key names) and
uses stock ``torch.nn.functional`` ops, i.e. the same ATen ops the reference's
``nn.Module`` tree dispatches.  Gradients come from torch autograd exactly as in
the reference (``loss.backward()``, train_model.py:121).

Citations are relative to /root/reference.
"""
from __future__ import annotations

import math
from collections import OrderedDict

import numpy as np
import torch
import torch.nn.functional as F

BN_EPS = 1e-3        # torchlayers.py:20
BN_MOMENTUM = 0.01   # torchlayers.py:20
PHISEG_RES_LEVELS = 7     # phiseg.py:131-132 (hard-coded in Posterior)
PHISEG_LAT_LEVELS = 5


# --------------------------------------------------------------------------- #
# building blocks
# --------------------------------------------------------------------------- #
def _bn(sd, p, y, bn_train):
    """nn.BatchNorm2d(eps=1e-3, momentum=0.01) as built by torchlayers.py:20."""
    if bn_train:
        sd[p + ".num_batches_tracked"] += 1
    return F.batch_norm(y, sd[p + ".running_mean"], sd[p + ".running_var"],
                        sd[p + ".weight"], sd[p + ".bias"],
                        training=bn_train, momentum=BN_MOMENTUM, eps=BN_EPS)


# Test hook for the bf16 arithmetic mode of the native convolutions (UZ_CONV_MATH=bf16, BASELINE config 5): a callable
# (x, w) -> bool; where it returns True both operands are rounded to bf16 (round to nearest even) before the fp32 convolution -
# exactly what the single-piece kernels multiply (products of two bf16 values are exact in fp32, accumulation is fp32).
CONV_OPERAND_ROUNDING = None


def _maybe_round(x, w):
    if CONV_OPERAND_ROUNDING is not None and CONV_OPERAND_ROUNDING(x, w):
        return x.to(torch.bfloat16).to(torch.float32), w.to(torch.bfloat16).to(torch.float32)
    return x, w


def conv_unit(sd, p, x, bn_train):
    """Conv2D unit: Conv2d(k, pad = 1 if k == 3 else 0) -> BN -> ReLU (torchlayers.py:7-29); with a 5-D weight the Conv3D unit
    of models/phiseg3D.py:13-35 (Conv3d -> BatchNorm3d(eps=1e-3, momentum=0.01) -> ReLU)."""
    w = sd[p + ".convolution.0.weight"]
    conv = F.conv3d if w.dim() == 5 else F.conv2d
    x, w = _maybe_round(x, w)
    y = conv(x, w, sd[p + ".convolution.0.bias"], padding=1 if w.shape[-1] == 3 else 0)
    return F.relu(_bn(sd, p + ".convolution.1", y, bn_train))


def conv_bare(sd, p, x):
    """Conv2D with norm=activation=nn.Identity (phiseg.py:281-284, probabilistic_unet.py:244)."""
    w = sd[p + ".convolution.0.weight"]
    conv = F.conv3d if w.dim() == 5 else F.conv2d
    x, w = _maybe_round(x, w)
    return conv(x, w, sd[p + ".convolution.0.bias"], padding=1 if w.shape[-1] == 3 else 0)


def rev_sequence(sd, p, x, bn_train):
    """ReversibleSequence (torchlayers.py:55-82).  `inital_conv` (sic): a 1x1 Conv2D unit when the channel counts differ;
    then revtorch ReversibleBlocks.  revtorch==0.2.0 (requirements.txt:38) is NOT vendored under /root/reference and cannot be
    installed here, so its published algorithm is restated - PARITY UNPINNED for this function (no reference run, no golden
    vector): per block, with the channels split in halves (x1, x2),  y1 = x1 + F(x2),  y2 = x2 + G(y1),  output cat(y1, y2);
    F / G = the block's `f_block` / `g_block` (here one 3x3 Conv2D unit each).  revtorch's backward pass inverts the block
    (x2 = y2 - G(y1), x1 = y1 - F(x2)) and back-propagates through the recomputed F and G; plain autograd through this
    forward gives the same gradients up to rounding, which is what the oracle uses.  Side effect of the recomputation that
    the oracle does NOT reproduce by itself: BatchNorm running statistics of F and G are updated a second time per training
    step (see `revtorch_second_bn_update`)."""
    if f"{p}.inital_conv.convolution.0.weight" in sd:
        x = conv_unit(sd, p + ".inital_conv", x, bn_train)
    i = 0
    while f"{p}.sequence.reversible_blocks.{i}.f_block.0.convolution.0.weight" in sd:
        b = f"{p}.sequence.reversible_blocks.{i}"
        x1, x2 = torch.chunk(x, 2, dim=1)
        y1 = x1 + conv_unit(sd, b + ".f_block.0", x2, bn_train)
        y2 = x2 + conv_unit(sd, b + ".g_block.0", y1, bn_train)
        x = torch.cat([y1, y2], dim=1)
        i += 1
    return x


def revtorch_second_bn_update(sd_before, sd_after):
    """revtorch re-runs F and G in training mode while it recomputes activations in backward, so each of their BatchNorms
    applies its momentum update twice per step with (numerically) the same batch statistics and counts two batches.
    Given the buffers before the forward pass and after it, returns the state after the backward pass as well."""
    out = {k: v.clone() for k, v in sd_after.items()}
    for k, v in sd_after.items():
        if ".reversible_blocks." not in k:
            continue
        if k.endswith("running_mean") or k.endswith("running_var"):
            batch = (v - (1 - BN_MOMENTUM) * sd_before[k]) / BN_MOMENTUM
            out[k] = (1 - BN_MOMENTUM) * v + BN_MOMENTUM * batch
        elif k.endswith("num_batches_tracked"):
            out[k] = v + (v - sd_before[k])
    return out


def _is_rev(sd, p):
    return f"{p}.sequence.reversible_blocks.0.f_block.0.convolution.0.weight" in sd


def avgpool(x):
    """nn.AvgPool2d(2, 2, padding=0, ceil_mode=True) (phiseg.py:23, unet.py:22, probabilistic_unet.py:56)."""
    return F.avg_pool2d(x, kernel_size=2, stride=2, padding=0, ceil_mode=True)


def up2(x, align_corners):
    return F.interpolate(x, mode="bilinear", scale_factor=2, align_corners=align_corners)


def batch_to_onehot(mask, nlabels=2):
    """utils.convert_batch_to_onehot / convert_to_onehot_torch (utils.py:289-311):
    (B,1,H,W) float labels -> (B,nlabels,H,W) long one-hot."""
    lab = mask.reshape(mask.shape[0], 1, mask.shape[-2], mask.shape[-1])
    return torch.cat([(lab == ii) for ii in range(nlabels)], dim=1).long()


def kl_two_gauss_with_diag_cov(mu0, sigma0, mu1, sigma1):
    """phiseg.py:436-453 / probabilistic_unet.py:291-308, including the
    ``sigma1_fs = sigma1 * sigma0`` quirk that the reference ships."""
    s0 = torch.flatten(sigma0, start_dim=1)
    s1 = torch.flatten(sigma1, start_dim=1)
    sigma0_fs = s0 * s0
    sigma1_fs = s1 * s0
    logsigma0_fs = torch.log(sigma0_fs + 1e-10)
    logsigma1_fs = torch.log(sigma1_fs + 1e-10)
    d = torch.flatten(mu1, start_dim=1) - torch.flatten(mu0, start_dim=1)
    return torch.mean(0.5 * torch.sum((sigma0_fs + d * d) / (sigma1_fs + 1e-10)
                                      + logsigma1_fs - logsigma0_fs - 1, dim=1))


def multinoulli_loss(logits, target, num_classes):
    """phiseg.py:481-490: per-pixel CE summed over pixels, mean over batch."""
    b = logits.shape[0]
    ce = F.cross_entropy(logits.reshape(b, num_classes, -1), target.reshape(b, -1).long(), reduction="none")
    return torch.mean(torch.sum(ce, dim=1))


# --------------------------------------------------------------------------- #
# PHiSeg (models/phiseg.py)
# --------------------------------------------------------------------------- #
def phiseg_eps_shapes(batch, height, width):
    """Shapes of the randn_like draws of one Posterior/Prior pass in draw order
    (deepest first), phiseg.py:104,196-202."""
    out = []
    for i in range(PHISEG_LAT_LEVELS):
        s = 2 ** (PHISEG_RES_LEVELS - 1 - i)
        out.append((batch, 2, height // s, width // s))
    return out


def _phiseg_encoder(sd, root, x, eps, bn_train, z_override=None):
    """Posterior.forward (phiseg.py:175-206) without the one-hot concat."""
    blocks = []
    for i in range(PHISEG_RES_LEVELS):                      # phiseg.py:191-194
        base = 0
        if i != 0:
            x = avgpool(x)
            base = 1
        if _is_rev(sd, f"{root}.contracting_path.{i}.layers.{base}"):          # phiseg.py:25-26
            x = rev_sequence(sd, f"{root}.contracting_path.{i}.layers.{base}", x, bn_train)
        else:
            for j in range(3):
                x = conv_unit(sd, f"{root}.contracting_path.{i}.layers.{base + j}", x, bn_train)
        if i != PHISEG_RES_LEVELS - 1:
            blocks.append(x)
    L = PHISEG_LAT_LEVELS
    z, mu, sigma = [None] * L, [None] * L, [None] * L
    pre = x
    for i in range(L):                                       # phiseg.py:196-202
        if i != 0:
            u = up2(z[-i], True)                             # phiseg.py:66
            if _is_rev(sd, f"{root}.upsampling_path.{i - 1}.upconv_layer"):   # phiseg.py:53-54
                u = rev_sequence(sd, f"{root}.upsampling_path.{i - 1}.upconv_layer", u, bn_train)
            else:
                for j in range(2):
                    u = conv_unit(sd, f"{root}.upsampling_path.{i - 1}.upconv_layer.{j}", u, bn_train)
            pre = torch.cat([u, blocks[-i]], dim=1)          # phiseg.py:71
        h = pre
        if _is_rev(sd, f"{root}.sample_z_path.{i}.conv.0"):  # phiseg.py:87-88
            h = rev_sequence(sd, f"{root}.sample_z_path.{i}.conv.0", h, bn_train)
        else:
            for j in range(2):                               # SampleZBlock, phiseg.py:99-106
                h = conv_unit(sd, f"{root}.sample_z_path.{i}.conv.{j}", h, bn_train)
        p = f"{root}.sample_z_path.{i}"
        m = F.conv2d(h, sd[p + ".mu_conv.0.weight"], sd[p + ".mu_conv.0.bias"])
        s = F.softplus(F.conv2d(h, sd[p + ".sigma_conv.0.weight"], sd[p + ".sigma_conv.0.bias"]))
        mu[-i - 1], sigma[-i - 1] = m, s
        z[-i - 1] = m + s * eps[i]
        if z_override is not None:                           # training_prior, phiseg.py:201-202
            z[-i - 1] = z_override[-i - 1]
    return z, mu, sigma


def _phiseg_likelihood(sd, z, image_hw, bn_train):
    """Likelihood.forward (phiseg.py:286-323)."""
    L = PHISEG_LAT_LEVELS
    lvl_diff = PHISEG_RES_LEVELS - L
    post_z, post_c, s = [None] * L, [None] * L, [None] * L
    root = "likelihood"
    for i in range(L):                                       # phiseg.py:293-300
        h = z[-i - 1]
        if _is_rev(sd, f"{root}.likelihood_ups_path.{i}"):   # phiseg.py:261-262
            h = rev_sequence(sd, f"{root}.likelihood_ups_path.{i}", h, bn_train)
        else:
            for j in range(2):
                h = conv_unit(sd, f"{root}.likelihood_ups_path.{i}.convolution.{j}", h, bn_train)
        for t in range(lvl_diff):                            # increase_resolution, phiseg.py:209-221
            h = up2(h, True)
            h = conv_unit(sd, f"{root}.likelihood_post_ups_path.{i}.{2 * t + 1}.convolution.0", h, bn_train)
        post_z[-i - 1] = h
    post_c[L - 1] = post_z[L - 1]
    for i in reversed(range(L - 1)):                         # phiseg.py:304-317
        h = torch.cat([post_z[i], up2(post_c[i + 1], True)], dim=1)
        if _is_rev(sd, f"{root}.likelihood_post_c_path.{i}"):                  # phiseg.py:274-275
            h = rev_sequence(sd, f"{root}.likelihood_post_c_path.{i}", h, bn_train)
        else:
            for j in range(2):
                h = conv_unit(sd, f"{root}.likelihood_post_c_path.{i}.convolution.{j}", h, bn_train)
        post_c[i] = h
    for i in range(L):                                       # phiseg.py:319-321
        s_in = conv_bare(sd, f"{root}.s_layer.{i}.convolution.0", post_c[-i - 1])
        s[-i - 1] = F.interpolate(s_in, size=list(image_hw), mode="nearest")
    return s


def phiseg_forward(sd, patch, mask, eps, training=True, bn_train=True):
    """PHISeg.forward (phiseg.py:414-426).

    eps = {"posterior": [5 tensors in draw order, deepest first], "prior": [...]}.
    ``sd`` running statistics are updated in place when bn_train (net.train()).
    """
    with torch.no_grad():                                    # phiseg.py:178-183
        onehot = batch_to_onehot(mask, 2).float()
    xin = torch.cat([patch, onehot - 0.5], dim=1)
    pz, pmu, psig = _phiseg_encoder(sd, "posterior", xin, eps["posterior"], bn_train)
    if training:
        qz, qmu, qsig = _phiseg_encoder(sd, "prior", patch, eps["prior"], bn_train, z_override=pz)
        s = _phiseg_likelihood(sd, pz, patch.shape[-2:], bn_train)
    else:
        qz, qmu, qsig = _phiseg_encoder(sd, "prior", patch, eps["prior"], bn_train)
        s = _phiseg_likelihood(sd, qz, patch.shape[-2:], bn_train)
    return dict(s=s, posterior_z=pz, posterior_mu=pmu, posterior_sigma=psig,
                prior_z=qz, prior_mu=qmu, prior_sigma=qsig)


def phiseg_loss(out, mask, latent_levels=PHISEG_LAT_LEVELS, num_classes=2):
    """PHISeg.loss -> elbo (phiseg.py:519-537): sum_l 4^l KL_l + sum_l CE_l.
    Returns (total, loss_dict) with the reference's loss_dict key names."""
    terms = OrderedDict()
    total = 0
    for ii in reversed(range(latent_levels)):                # phiseg.py:455-479
        w = 4 ** ii
        terms["KL_divergence_loss_lvl%d" % ii] = w * kl_two_gauss_with_diag_cov(
            out["posterior_mu"][ii], out["posterior_sigma"][ii], out["prior_mu"][ii], out["prior_sigma"][ii])
        total = total + terms["KL_divergence_loss_lvl%d" % ii]
    s_acc = None
    for ii in reversed(range(latent_levels)):                # phiseg.py:492-513
        s_acc = out["s"][ii] if s_acc is None else s_acc + out["s"][ii]
        terms["residual_multinoulli_loss_lvl%d" % ii] = multinoulli_loss(s_acc, mask, num_classes)
        total = total + terms["residual_multinoulli_loss_lvl%d" % ii]
    return total, terms


def phiseg_accumulate_output(s_list, use_softmax=False):
    """PHISeg.accumulate_output (phiseg.py:428-434), without the in-place aliasing."""
    acc = s_list[-1].clone()
    for i in range(len(s_list) - 1):
        acc = acc + s_list[i]
    return F.softmax(acc, dim=1) if use_softmax else acc


# --------------------------------------------------------------------------- #
# vanilla U-Net (models/unet.py)
# --------------------------------------------------------------------------- #
def _unet_block(sd, p, x, first_idx):
    """DownConvBlock body: 3 x (Conv3x3 pad 1 + ReLU), no norm (unet.py:25-30)."""
    for j in range(3):
        q = f"{p}.{first_idx + 2 * j}"
        x = F.relu(F.conv2d(x, sd[q + ".weight"], sd[q + ".bias"], padding=1))
    return x


def unet_forward(sd, x, prefix="", apply_last_layer=True, bn_train=True):
    """Unet.forward (unet.py:129-157); reversible variant: every block body is a ReversibleSequence (unet.py:32-35)."""
    n = 0
    rev = _is_rev(sd, f"{prefix}contracting_path.0.layers.0")
    while (f"{prefix}contracting_path.{n}.layers.{0 if n == 0 else 1}.weight" in sd
           or _is_rev(sd, f"{prefix}contracting_path.{n}.layers.{0 if n == 0 else 1}")):
        n += 1
    blocks = []
    for i in range(n):
        if i != 0:
            x = avgpool(x)
        if rev:
            x = rev_sequence(sd, f"{prefix}contracting_path.{i}.layers.{0 if i == 0 else 1}", x, bn_train)
        else:
            x = _unet_block(sd, f"{prefix}contracting_path.{i}.layers", x, 0 if i == 0 else 1)
        if i != n - 1:
            blocks.append(x)
    for i in range(n - 1):                                   # UpConvBlock, unet.py:65-75
        up = up2(x, False)
        x = torch.cat([up, blocks[-i - 1]], dim=1)
        if rev:
            x = rev_sequence(sd, f"{prefix}upsampling_path.{i}.conv_block.layers.0", x, bn_train)
        else:
            x = _unet_block(sd, f"{prefix}upsampling_path.{i}.conv_block.layers", x, 0)
    if apply_last_layer:
        x = F.conv2d(x, sd[prefix + "last_layer.weight"], sd[prefix + "last_layer.bias"])
    return x


def unet_loss(pred, mask):
    """Unet.loss (unet.py:159-165): mean CE over all pixels."""
    return F.cross_entropy(pred, mask.reshape(-1, pred.shape[-2], pred.shape[-1]).long())


# --------------------------------------------------------------------------- #
# Probabilistic U-Net (models/probabilistic_unet.py)
# --------------------------------------------------------------------------- #
def _axis_aligned_gaussian(sd, root, x, bn_train):
    """AxisAlignedConvGaussian.forward (probabilistic_unet.py:102-130) -> (mu, sigma)."""
    i = 0
    while f"{root}.encoder.layers.{2 * i}.convolution.0.convolution.0.weight" in sd:
        if i != 0:
            x = avgpool(x)
        for j in range(3):
            x = conv_unit(sd, f"{root}.encoder.layers.{2 * i}.convolution.{j}", x, bn_train)
        i += 1
    enc = torch.mean(x, dim=2, keepdim=True)
    enc = torch.mean(enc, dim=3, keepdim=True)
    mls = F.conv2d(enc, sd[root + ".conv_layer.weight"], sd[root + ".conv_layer.bias"])[:, :, 0, 0]
    L = mls.shape[1] // 2
    return mls[:, :L], torch.exp(mls[:, L:])


def probunet_fcomb(sd, features, z, bn_train):
    """Fcomb.forward (probabilistic_unet.py:185-199)."""
    b, _, h, w = features.shape
    zt = z[:, :, None, None].expand(b, z.shape[1], h, w)
    x = torch.cat((features, zt), dim=1)
    k = 0
    while f"fcomb.layers.{k}.convolution.0.weight" in sd:
        x = conv_unit(sd, f"fcomb.layers.{k}", x, bn_train)
        k += 1
    return F.conv2d(x, sd["fcomb.last_layer.weight"], sd["fcomb.last_layer.bias"])


def probunet_forward(sd, patch, segm, bn_train=True):
    """ProbabilisticUnet.forward (probabilistic_unet.py:246-255)."""
    out = {}
    if segm is not None:
        with torch.no_grad():
            onehot = batch_to_onehot(segm, 2).float()
        out["posterior_mu"], out["posterior_sigma"] = _axis_aligned_gaussian(
            sd, "posterior", torch.cat([patch, onehot - 0.5], dim=1), bn_train)
    out["prior_mu"], out["prior_sigma"] = _axis_aligned_gaussian(sd, "prior", patch, bn_train)
    out["unet_features"] = unet_forward(sd, patch, prefix="unet.", apply_last_layer=False, bn_train=bn_train)
    out["last_conv"] = conv_bare(sd, "last_conv", out["unet_features"])
    return out


def _l2_regularisation(sd, prefix):
    """utils.l2_regularisation (utils.py:93-101): sum of (non-squared) 2-norms over m.parameters()."""
    tot = None
    for k, v in sd.items():
        if k.startswith(prefix) and v.dtype.is_floating_point and "running_" not in k:
            tot = v.norm(2) if tot is None else tot + v.norm(2)
    return tot


def probunet_loss(sd, out, segm, eps, bn_train=True, num_classes=2):
    """ProbabilisticUnet.loss / elbo (probabilistic_unet.py:343-370).
    eps: (B, latent_dim) standard-normal draw of posterior.rsample()."""
    z = out["posterior_mu"] + out["posterior_sigma"] * eps
    kl = torch.mean(kl_two_gauss_with_diag_cov(out["posterior_mu"], out["posterior_sigma"],
                                               out["prior_mu"], out["prior_sigma"]))
    recon = probunet_fcomb(sd, out["unet_features"], z, bn_train)
    rec_loss = torch.sum(multinoulli_loss(recon, segm, num_classes))
    elbo = -(rec_loss + 1.0 * kl)
    reg = _l2_regularisation(sd, "posterior.") + _l2_regularisation(sd, "prior.") + _l2_regularisation(sd, "fcomb.layers.")
    return -elbo + 1e-5 * reg, dict(kl=kl, reconstruction_loss=rec_loss, reg=reg, reconstruction=recon, z=z)


# --------------------------------------------------------------------------- #
# optimiser step of the harness (train_model.py:49,119-122)
# --------------------------------------------------------------------------- #
def adam_reference_step(params, grads, state, lr=1e-3, weight_decay=1e-5):
    """One torch.optim.Adam step exactly as the harness configures it
    (train_model.py:49).  ``params``/``grads`` are dicts; entries whose grad is
    None are skipped entirely, like torch.optim.Adam does.  ``state`` is a dict
    kept by the caller across steps."""
    if "opt" not in state:
        state["leaves"] = {k: torch.nn.Parameter(v.detach().clone()) for k, v in params.items()}
        state["opt"] = torch.optim.Adam(list(state["leaves"].values()), lr=lr, weight_decay=weight_decay)
    for k, leaf in state["leaves"].items():
        g = grads.get(k)
        leaf.grad = None if g is None else g.detach().clone()
    state["opt"].step()
    return {k: v.detach().clone() for k, v in state["leaves"].items()}


# --------------------------------------------------------------------------- #
# synthetic data + deterministic weights
# --------------------------------------------------------------------------- #
def synthetic_batch(batch, height=128, width=128, seed=20201004, eps_shapes=None):
    """LIDC-like synthetic inputs: images N(0, 0.25^2) clipped to +-0.5, random-disc
    binary masks, eps ~ N(0,1).  numpy PCG64, fixed seed."""
    rng = np.random.Generator(np.random.PCG64(seed))
    x = np.clip(rng.standard_normal((batch, 1, height, width)).astype(np.float32) * 0.25, -0.5, 0.5)
    yy, xx = np.mgrid[0:height, 0:width]
    mask = np.zeros((batch, 1, height, width), np.float32)
    for b in range(batch):
        r = rng.uniform(8, 24) * min(height, width) / 128.0
        cy, cx = rng.uniform(r, height - r), rng.uniform(r, width - r)
        mask[b, 0] = ((yy - cy) ** 2 + (xx - cx) ** 2 <= r * r).astype(np.float32)
    eps = None
    if eps_shapes is not None:
        eps = [rng.standard_normal(s).astype(np.float32) for s in eps_shapes]
    return x, mask, eps


def deterministic_state_dict(spec, seed=1234):
    """Deterministic, RNG-stream-independent weights for parity tests.

    ``spec`` is an ordered list of (key, shape, kind) with kind in
    {"conv_w", "conv_b", "bn_w", "bn_b", "bn_rm", "bn_rv", "bn_nbt"}.
    Conv weights/biases ~ U(-1/sqrt(fan_in), 1/sqrt(fan_in)) (the magnitude of
    torch's default Conv2d init that PHiSeg keeps, phiseg.py:36 commented out);
    BN affine/buffers get non-trivial values so every term is exercised."""
    rng = np.random.Generator(np.random.PCG64(seed))
    sd = OrderedDict()
    fan_in = 1
    for key, shape, kind in spec:
        shape = tuple(shape)
        if kind == "conv_w":
            fan_in = int(np.prod(shape[1:]))
            bound = 1.0 / math.sqrt(fan_in)
            a = rng.uniform(-bound, bound, shape)
        elif kind == "conv_b":
            bound = 1.0 / math.sqrt(fan_in)
            a = rng.uniform(-bound, bound, shape)
        elif kind == "bn_w":
            a = rng.uniform(0.5, 1.5, shape)
        elif kind == "bn_b":
            a = rng.uniform(-0.2, 0.2, shape)
        elif kind == "bn_rm":
            a = rng.uniform(-0.1, 0.1, shape)
        elif kind == "bn_rv":
            a = rng.uniform(0.5, 1.5, shape)
        elif kind == "bn_nbt":
            sd[key] = torch.zeros(shape, dtype=torch.int64)
            continue
        else:
            raise ValueError(kind)
        sd[key] = torch.from_numpy(np.asarray(a, dtype=np.float32).reshape(shape))
    return sd
