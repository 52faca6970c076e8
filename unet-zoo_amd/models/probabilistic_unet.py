"""Probabilistic U-Net on the native HIP path - drop-in for the reference
``models/probabilistic_unet.py`` ``ProbabilisticUnet`` (constructor keywords :212-221,
``forward(patch, segm, training)`` :246-255, ``loss`` / ``elbo`` :343-370, ``sample`` :257-270,
``reconstruct`` :272-283, same ``state_dict`` keys).

Structure restated for the op tape:
  * prior / posterior = AxisAlignedConvGaussian (:73-130): 7 x ([AvgPool] + 3 Conv-BN-ReLU units),
    spatial mean, 1x1 conv to (mu, log sigma); sigma = exp(log sigma).  The 1x1 conv is issued as
    two row-slices of the same parameter so that mu and log sigma land in contiguous tensors.
  * unet = vanilla U-Net features (apply_last_layer=False), written straight into the first 32
    channels of the Fcomb input buffer; z is tiled into the remaining channels (:185-197).
  * elbo (:343-363): z = mu_q + sigma_q * eps, KL with the sigma1*sigma0 quirk (:291-308), Fcomb
    decode, summed cross entropy; loss adds 1e-5 * sum of 2-norms of posterior, prior and
    fcomb.layers parameters (:365-370).
  * ``last_conv`` (:244,255) is computed in forward but never reaches the loss: its parameters keep
    ``grad is None`` exactly as in the reference.
"""
import math

import torch
import torch.nn as nn
from torch.distributions import Independent, Normal

from .._engine import NativeModel, conv_unit
from .._modtree import conv_unit_spec, plain_conv_spec
from .unet import unet_spec, init_unet_weights, build_unet_graph


def _gaussian_spec(root, in_ch, nf, latent_dim):
    out = []
    for i in range(len(nf)):
        cin = in_ch if i == 0 else nf[i - 1]
        for j in range(3):
            out += conv_unit_spec(f"{root}.encoder.layers.{2 * i}.convolution.{j}", cin if j == 0 else nf[i], nf[i])
    out += plain_conv_spec(f"{root}.conv_layer", nf[-1], 2 * latent_dim, 1)
    return out


def probunet_spec(input_channels, num_classes, num_filters, latent_dim, no_convs_fcomb, reversible=False):
    """reversible=True reaches only the U-Net backbone: the reference builds prior / posterior / fcomb without it
    (probabilistic_unet.py:232-243)."""
    nf = list(num_filters)
    out = unet_spec(input_channels, num_classes, nf, apply_last_layer=False, prefix="unet.", reversible=reversible)
    out += _gaussian_spec("prior", input_channels, nf, latent_dim)
    out += _gaussian_spec("posterior", input_channels + 2, nf, latent_dim)      # Encoder default num_classes=2 (:31,:44)
    out += conv_unit_spec("fcomb.layers.0", nf[0] + latent_dim, nf[0], k=1)
    for k in range(1, no_convs_fcomb - 1):
        out += conv_unit_spec(f"fcomb.layers.{k}", nf[0], nf[0], k=1)
    out += plain_conv_spec("fcomb.last_layer", nf[0], num_classes, 1)
    out += conv_unit_spec("last_conv", 32, num_classes, k=1, norm=False)          # hard-coded 32 (:244)
    return out


class ProbabilisticUnet(NativeModel):
    # U-Net, prior encoder and posterior encoder are three independent chains whose deep levels are latency-bound: three lanes
    # measure 12.0 ms per step against 12.4 with two (PHiSeg, whose chains are dominated by device-filling kernels, loses 8 %
    # with three)
    default_lanes = 3
    wgrad_workgroups = 192                     # NativeModel.wgrad_workgroups: 256 -> 3 230, 192 -> 3 255, 128 -> 3 208 images/s (one box)

    def __init__(self, input_channels=1, num_classes=1, num_filters=None, latent_levels=1, latent_dim=2, initializers=None,
                 no_convs_fcomb=4, image_size=(1, 128, 128), beta=10.0, reversible=False, device=None):
        super().__init__()
        self.reversible = bool(reversible)
        self.input_channels, self.num_classes, self.num_filters = input_channels, num_classes, list(num_filters)
        self.latent_dim, self.no_convs_per_block, self.no_convs_fcomb = latent_dim, 3, no_convs_fcomb
        self.initializers = {"w": "he_normal", "b": "normal"}
        self.z_prior_sample = 0
        if self.num_filters[0] != 32:
            raise ValueError("ProbabilisticUnet.last_conv is hard-wired to 32 input channels (probabilistic_unet.py:244)")
        self._init_storage(probunet_spec(input_channels, num_classes, self.num_filters, latent_dim, no_convs_fcomb, self.reversible), device)
        self._init_weights()

    def _init_weights(self):
        pt = self._ptab
        init_unet_weights(pt, prefix="unet.", skip=())
        fan_in = 1
        for key, shape, kind in pt.spec:
            if key.startswith("unet."):
                continue
            if kind == "conv_w":
                fan_in = shape[1] * shape[2] * shape[3]
                if key.startswith("fcomb."):
                    nn.init.orthogonal_(pt.pview(key))                            # init_weights_orthogonal_normal (utils.py:86-90)
                elif key.startswith("last_conv."):
                    nn.init.kaiming_uniform_(pt.pview(key), a=math.sqrt(5))       # PyTorch default
                else:
                    nn.init.kaiming_normal_(pt.pview(key), mode="fan_in", nonlinearity="relu")
            elif kind == "conv_b":
                if key.endswith("conv_layer.bias"):
                    nn.init.normal_(pt.pview(key))                                # :100
                elif key.startswith("last_conv."):
                    nn.init.uniform_(pt.pview(key), -1.0 / math.sqrt(fan_in), 1.0 / math.sqrt(fan_in))
                else:
                    nn.init.trunc_normal_(pt.pview(key), mean=0.0, std=1e-3, a=-2e-3, b=2e-3)
            elif kind == "bn_w":
                pt.pview(key).fill_(1.0)
            elif kind == "bn_b":
                pt.pview(key).zero_()
            elif kind == "bn_rv":
                pt.bview(key).fill_(1.0)

    # ------------------------------------------------------------------ plan
    def _gaussian(self, plan, root, x):
        for i in range(len(self.num_filters)):
            if i != 0:
                x = plan.avgpool(x, f"{root}.pool{i}")
            for j in range(3):
                x = conv_unit(plan, x, f"{root}.encoder.layers.{2 * i}.convolution.{j}")
        enc = plan.spatial_mean(x, f"{root}.mean")
        L = self.latent_dim
        mu = plan.conv_bare(enc, f"{root}.conv_layer", name=f"{root}.mu", rows=(0, L))
        ls = plan.conv_bare(enc, f"{root}.conv_layer", name=f"{root}.logsigma", rows=(L, L))
        return mu, ls

    def _fcomb(self, plan, fcat, tag):
        h = fcat
        for k in range(self.no_convs_fcomb - 1):
            h = conv_unit(plan, h, f"fcomb.layers.{k}")
        return plan.conv_bare(h, "fcomb.last_layer", name=f"fcomb.{tag}.out")

    def _build(self, N, H, W, with_posterior, bn_training):
        plan = self._new_plan(N, bn_training)
        nf0, L = self.num_filters[0], self.latent_dim
        io = {"patch": plan.buf("patch", self.input_channels, H, W, requires_grad=False)}
        if with_posterior:
            io["segm"] = plan.buf("segm", 1, H, W, requires_grad=False)
            xin = plan.posterior_input(io["patch"], io["segm"], 2, "posterior.input")
            io["q_mu"], io["q_ls"] = self._gaussian(plan, "posterior", xin)
            io["q_eps0"] = plan.buf("q_eps0", L, 1, 1, requires_grad=False)
            io["q"] = plan.latent(io["q_mu"], io["q_ls"], io["q_eps0"], "posterior.lat", want_z=False, act=1)
        io["p_mu"], io["p_ls"] = self._gaussian(plan, "prior", io["patch"])
        io["p_eps0"] = plan.buf("p_eps0", L, 1, 1, requires_grad=False)
        io["p"] = plan.latent(io["p_mu"], io["p_ls"], io["p_eps0"], "prior.lat", want_z=False, act=1)
        fcat = plan.buf("fcomb.in", nf0 + L, H, W)
        io["fcat"] = fcat
        io["features"] = build_unet_graph(plan, "unet.", io["patch"], self.num_filters, False, final_out=fcat.slice(0, nf0),
                                          reversible=self.reversible)
        io["last_conv"] = plan.conv_bare(io["features"], "last_conv.convolution.0", name="last_conv")
        plan.total = plan.vec("total", 1)
        if with_posterior:
            # ---- loss tape: elbo (:343-363) + regulariser (:365-370)
            plan.loss_phase()
            io["terms"] = plan.vec("loss_terms", 3)                     # [reconstruction, KL, 1e-5 * reg]
            io["loss_mask"] = plan.buf("loss_mask", 1, H, W, requires_grad=False)
            io["eps"] = plan.buf("eps", L, 1, 1, requires_grad=False)
            zlat = plan.latent(io["q_mu"], io["q_ls"], io["eps"], "posterior.rsample", want_z=True, act=1)
            # the rsample latent shares (mu, log sigma) with io["q"]; KL attaches to the rsample node so that
            # one backward op merges the KL and the z paths into d mu / d log sigma
            io["zq"] = zlat
            plan.bcast_channels(zlat.z, fcat.slice(nf0, L))
            io["recon"] = self._fcomb(plan, fcat, "train")
            plan.residual_ce([io["recon"]], io["loss_mask"], io["terms"].slice(0, 1))
            plan.kl(zlat, io["p"], 1.0, io["terms"].slice(1, 1))
            keys = [k for k, _, kd in self._ptab.spec if kd in ("conv_w", "conv_b", "bn_w", "bn_b")
                    and (k.startswith("posterior.") or k.startswith("prior.") or k.startswith("fcomb.layers."))]
            # reference order of the sum: posterior, prior, fcomb.layers (:367-368)
            keys = ([k for k in keys if k.startswith("posterior.")] + [k for k in keys if k.startswith("prior.")]
                    + [k for k in keys if k.startswith("fcomb.")])
            plan.l2_reg(keys, io["terms"].slice(2, 1), 1e-5)
            plan.sum_terms(io["terms"], 3, plan.total)
        # ---- decode tape: Fcomb on cached features with a caller-supplied z (sample / reconstruct)
        plan.extra_phase("decode")
        io["z_in"] = plan.buf("z_in", L, 1, 1, requires_grad=False)
        plan.bcast_channels(io["z_in"], fcat.slice(nf0, L))
        io["decoded"] = self._fcomb(plan, fcat, "decode")
        plan.finalize(want_backward=with_posterior and bn_training)
        plan.io = io
        return plan

    # ------------------------------------------------------------------ reference API
    def forward(self, patch, segm=None, training=True):
        self._require_gpu()
        N, _, H, W = patch.shape
        key = (N, H, W, segm is not None, bool(self.training))
        plan = self._plan(key, lambda: self._build(N, H, W, segm is not None, bool(self.training)))
        io, T = plan.io, plan.tensor
        T(io["patch"]).copy_(patch)
        if segm is not None:
            T(io["segm"]).copy_(segm.reshape(N, 1, H, W))
        self._run(plan, "fwd")
        if self.training:
            self._bump_nbt(plan)
        self._cur = plan
        L = self.latent_dim
        # validate_args=False: torch.distributions' argument check is a device -> host synchronisation per forward (and it
        # raises on sigma == 0, which exp(log sigma) reaches by underflow on un-normalised synthetic data)
        if segm is not None:
            self.posterior_latent_space = Independent(Normal(loc=T(io["q_mu"]).reshape(N, L), scale=T(io["q"].sigma).reshape(N, L), validate_args=False), 1)
        self.prior_latent_space = Independent(Normal(loc=T(io["p_mu"]).reshape(N, L), scale=T(io["p"].sigma).reshape(N, L), validate_args=False), 1)
        self.unet_features = T(io["features"])
        return T(io["last_conv"])

    def _decode(self, z):
        plan = self._cur
        plan.tensor(plan.io["z_in"]).copy_(z.reshape(-1, self.latent_dim, 1, 1))
        plan.run("decode", self._stream())
        if self.training:
            self._bump_nbt(plan, "bn_prefixes_nbt_extra")
        return plan.tensor(plan.io["decoded"]).clone()

    def sample(self, testing=False):
        """:257-270 - decode one prior draw on the cached U-Net features."""
        z_prior = self.prior_latent_space.rsample() if not testing else self.prior_latent_space.sample()
        self.z_prior_sample = z_prior
        return self._decode(z_prior)

    def reconstruct(self, use_posterior_mean=False, calculate_posterior=False, z_posterior=None):
        """:272-283"""
        if use_posterior_mean:
            z_posterior = self.posterior_latent_space.mean
        elif calculate_posterior:
            z_posterior = self.posterior_latent_space.rsample()
        return self._decode(z_posterior)

    def accumulate_output(self, output_list, use_softmax=False):
        return torch.nn.functional.softmax(output_list, dim=1) if use_softmax else output_list

    def elbo(self, segm, analytic_kl=False, reconstruct_posterior_mean=False, eps=None):
        return -self._loss_parts(segm, eps)[1]

    def loss(self, mask, eps=None):
        """:365-370.  `eps` (N, latent_dim) optionally injects the rsample noise."""
        return self._loss_parts(mask, eps)[0]

    def _loss_parts(self, mask, eps):
        plan = self._cur
        if plan is None or "terms" not in plan.io:
            raise RuntimeError("call forward(patch, segm) before loss()")
        T = plan.tensor
        N, _, H, W = T(plan.io["loss_mask"]).shape
        T(plan.io["loss_mask"]).copy_(mask.reshape(N, 1, H, W))
        if eps is None:
            self._fill_normal(T(plan.io["eps"]))
        else:
            T(plan.io["eps"]).copy_(eps.reshape(N, self.latent_dim, 1, 1))
        if torch.is_grad_enabled() and plan.tapes["bwd"][1]:
            total = self._loss_tensor(plan)
        else:
            plan.run("loss", self._stream())
            total = T(plan.total).reshape(()).clone()
        if self.training:
            self._bump_nbt(plan, "bn_prefixes_nbt_loss")
        terms = T(plan.io["terms"]).reshape(-1).clone()
        self.reconstruction_loss = self.mean_reconstruction_loss = terms[0]
        self.kl_divergence_loss = terms[1]
        self.reconstruction = T(plan.io["recon"])
        elbo_plus = terms[0] + terms[1]
        return total, elbo_plus

    def kl_divergence(self, analytic=True, calculate_posterior=False, z_posterior=None):
        return self.kl_divergence_loss
