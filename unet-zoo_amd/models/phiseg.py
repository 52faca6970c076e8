"""PHiSeg (7 resolution levels / 5 latent levels) on the native HIP path.

Drop-in for the reference ``models/phiseg.py`` ``PHISeg`` class: same constructor keywords
(phiseg.py:336-349), same ``forward(patch, mask, training)`` / ``loss(mask)`` /
``accumulate_output`` / ``sample`` / ``reconstruct`` surface, same ``state_dict`` keys (820 for
the standard filter list) and the same quirks:
  * the architecture is always 7/5 whatever ``latent_levels`` says (phiseg.py:131-132);
  * ``kl_divergence_loss`` and ``reconstruction_loss`` alias the total loss (phiseg.py:519-534);
  * KL uses ``sigma1*sigma0`` (phiseg.py:438-439);
  * ``upsampling_path.4`` of posterior and prior is never executed, so its parameters keep
    ``grad is None`` (phiseg.py:161-164,198-199);
  * ``accumulate_output`` sums the levels in place into ``output_list[-1]`` (phiseg.py:428-434).
The forward/backward arithmetic itself runs in libuz_hip.so (see _plan.py for the op tape).
"""
import ctypes as C

import torch

from .. import _ffi
from .._engine import NativeModel, conv_unit
from .._modtree import conv_unit_spec, plain_conv_spec, rev_sequence_spec, init_default
from .._plan import View

RES_LEVELS = 7
LAT_LEVELS = 5
LVL_DIFF = RES_LEVELS - LAT_LEVELS


def _encoder_spec(root, in_ch, nf, reversible=False):
    """Posterior / prior parameter entries in the reference's registration order (phiseg.py:144-173); with
    reversible=True every Conv2D stack is a ReversibleSequence (phiseg.py:25-26,53-54,87-88)."""
    out = []
    for i in range(RES_LEVELS):
        cin = in_ch if i == 0 else nf[i - 1]
        base = 0 if i == 0 else 1          # layers.0 is the AvgPool2d when pooling
        if reversible:
            out += rev_sequence_spec(f"{root}.contracting_path.{i}.layers.{base}", cin, nf[i], 3)
            continue
        for j in range(3):
            out += conv_unit_spec(f"{root}.contracting_path.{i}.layers.{base + j}", cin if j == 0 else nf[i], nf[i])
    for k in range(LAT_LEVELS):
        if reversible:
            out += rev_sequence_spec(f"{root}.upsampling_path.{k}.upconv_layer", 2, 2 * nf[0], 2)
            continue
        out += conv_unit_spec(f"{root}.upsampling_path.{k}.upconv_layer.0", 2, 2 * nf[0])
        out += conv_unit_spec(f"{root}.upsampling_path.{k}.upconv_layer.1", 2 * nf[0], 2 * nf[0])
    for k in range(LAT_LEVELS):
        i = LAT_LEVELS - 1 - k
        cin = nf[i + LVL_DIFF] if k == 0 else 2 * nf[0] + nf[i + LVL_DIFF]
        p = f"{root}.sample_z_path.{k}"
        if reversible:
            out += rev_sequence_spec(p + ".conv.0", cin, cin, 3)
        else:
            out += conv_unit_spec(p + ".conv.0", cin, cin) + conv_unit_spec(p + ".conv.1", cin, cin)
        out += plain_conv_spec(p + ".mu_conv.0", cin, 2, 1) + plain_conv_spec(p + ".sigma_conv.0", cin, 2, 1)
    return out


def _likelihood_spec(nf, num_classes, reversible=False):
    """Likelihood entries (phiseg.py:252-284): both ModuleLists of the first loop are registered
    before the loop, so all ups_path entries precede all post_ups_path entries."""
    root, out = "likelihood", []
    for k in range(LAT_LEVELS):
        c = nf[LAT_LEVELS - 1 - k]
        if reversible:
            out += rev_sequence_spec(f"{root}.likelihood_ups_path.{k}", 2, c, 2)
            continue
        out += conv_unit_spec(f"{root}.likelihood_ups_path.{k}.convolution.0", 2, c)
        out += conv_unit_spec(f"{root}.likelihood_ups_path.{k}.convolution.1", c, c)
    for k in range(LAT_LEVELS):
        c = nf[LAT_LEVELS - 1 - k]
        for t in range(LVL_DIFF):
            out += conv_unit_spec(f"{root}.likelihood_post_ups_path.{k}.{2 * t + 1}.convolution.0", c, c)
    for i in range(LAT_LEVELS - 1):
        cin, cout = nf[i] + nf[i + 1 + LVL_DIFF], nf[i + LVL_DIFF]
        if reversible:
            out += rev_sequence_spec(f"{root}.likelihood_post_c_path.{i}", cin, cout, 2)
            continue
        out += conv_unit_spec(f"{root}.likelihood_post_c_path.{i}.convolution.0", cin, cout)
        out += conv_unit_spec(f"{root}.likelihood_post_c_path.{i}.convolution.1", cout, cout)
    for k in range(LAT_LEVELS):
        cin = nf[LAT_LEVELS - 1 - k + LVL_DIFF]
        out += conv_unit_spec(f"{root}.s_layer.{k}.convolution.0", cin, num_classes, k=1, norm=False)
    return out


def phiseg_spec(input_channels, num_classes, num_filters, reversible=False):
    nf = list(num_filters)
    return (_encoder_spec("posterior", input_channels + 2, nf, reversible) + _likelihood_spec(nf, num_classes, reversible)
            + _encoder_spec("prior", input_channels, nf, reversible))


class PHISeg(NativeModel):
    wgrad_workgroups = 128                     # NativeModel.wgrad_workgroups: 256 -> 1 907, 192 -> 1 946, 128 -> 1 970, 96 -> 1 916, 64 -> 1 778 images/s (one box)
    default_lanes_by_mode = {"lanes": 3}     # host-issued lane replay: 2 lanes 17.7 ms, 3: 16.3, 4: 16.2 (one box, round 5)
    decouple_wgrad_px = 8192       # see NativeModel.decouple_wgrad_px: the 16 x 16 ... 2 x 2 levels at batch 32
    decouple_wgrad_prefixes = ("likelihood",)      # its backward is a single chain (posterior / prior pair up): A/B 1 726 -> 1 744 images/s

    def __init__(self, input_channels, num_classes, num_filters, latent_levels=5, latent_dim=2, initializers=None,
                 no_convs_fcomb=4, beta=10.0, image_size=(128, 128, 1), reversible=False, apply_last_layer=True,
                 exponential_weighting=True, padding=True, device=None):
        super().__init__()
        self.reversible = bool(reversible)
        if len(num_filters) < RES_LEVELS or num_filters[4] != num_filters[6]:
            raise ValueError("PHISeg needs >= 7 filters with num_filters[4] == num_filters[6] (phiseg.py:131-132,258-270)")
        self.input_channels, self.num_classes, self.num_filters = input_channels, num_classes, list(num_filters)
        self.latent_levels, self.image_size = latent_levels, image_size
        self.loss_tot, self.loss_dict = 0, {}
        self.kl_divergence_loss_weight, self.beta = 1.0, 1.0
        self.padding, self.activation_maps, self.apply_last_layer = padding, [], apply_last_layer
        self.exponential_weighting, self.exponential_weight = exponential_weighting, 4
        self.residual_multinoulli_loss_weight = 1.0
        self.kl_divergence_loss = self.reconstruction_loss = 0
        self.s_out_list = [None] * latent_levels
        self._init_storage(phiseg_spec(input_channels, num_classes, num_filters, self.reversible), device)
        init_default(self._ptab)

    # ------------------------------------------------------------------ plan construction
    def _encoder(self, plan, root, x, eps, z_override, want_z):
        nf = self.num_filters
        skips = {}
        for i in range(RES_LEVELS):
            base = 0
            if i != 0:
                x = plan.avgpool(x, f"{root}.pool{i}")
                base = 1
            out = None
            if 2 <= i <= 5:     # blocks[2..5] feed torch.cat([up, bridge]) (phiseg.py:71): write them in place
                cat = plan.buf(f"{root}.cat{i}", 2 * nf[0] + nf[i], x.H, x.W)
                out = cat.slice(2 * nf[0], nf[i])
                skips[i] = cat
            if self.reversible:
                x = plan.rev_sequence(x, f"{root}.contracting_path.{i}.layers.{base}", nf[i], 3, conv_unit, out=out)
                continue
            for j in range(3):
                x = conv_unit(plan, x, f"{root}.contracting_path.{i}.layers.{base + j}", out=out if j == 2 else None)
        lats, zs = [], []
        pre = x
        for k in range(LAT_LEVELS):
            if k != 0:
                cat = skips[RES_LEVELS - 1 - k]
                u = plan.bilinear(zs[k - 1], True, name=f"{root}.up{k}.bil")
                if self.reversible:
                    plan.rev_sequence(u, f"{root}.upsampling_path.{k - 1}.upconv_layer", 2 * nf[0], 2, conv_unit, out=cat.slice(0, 2 * nf[0]))
                else:
                    u = conv_unit(plan, u, f"{root}.upsampling_path.{k - 1}.upconv_layer.0")
                    conv_unit(plan, u, f"{root}.upsampling_path.{k - 1}.upconv_layer.1", out=cat.slice(0, 2 * nf[0]))
                pre = cat
            p = f"{root}.sample_z_path.{k}"
            if self.reversible:
                h = plan.rev_sequence(pre, p + ".conv.0", pre.C, 3, conv_unit)
            else:
                h = conv_unit(plan, pre, p + ".conv.0")
                h = conv_unit(plan, h, p + ".conv.1")
            lat = plan.latent_heads(h, p + ".mu_conv.0", p + ".sigma_conv.0", eps[k], f"{root}.lat{k}", want_z=want_z, act=0)
            lats.append(lat)
            zs.append(z_override[k] if z_override is not None else lat.z)
        return lats, zs

    def _likelihood(self, plan, zs):
        """zs in draw order (k = 0 deepest).  Returns s views by level (s[0] finest)."""
        nf, root = self.num_filters, "likelihood"
        L = LAT_LEVELS
        cats = {}
        post_c = [None] * L
        for k in range(L):
            lvl = L - 1 - k
            if self.reversible:
                h = plan.rev_sequence(zs[k], f"{root}.likelihood_ups_path.{k}", nf[lvl], 2, conv_unit)
            else:
                h = conv_unit(plan, zs[k], f"{root}.likelihood_ups_path.{k}.convolution.0")
                h = conv_unit(plan, h, f"{root}.likelihood_ups_path.{k}.convolution.1")
            for t in range(LVL_DIFF):
                h = plan.bilinear(h, True, name=f"{root}.ups{k}.bil{t}")
                out = None
                if t == LVL_DIFF - 1 and lvl < L - 1:
                    cats[lvl] = plan.buf(f"{root}.cat{lvl}", nf[lvl] + nf[lvl + 1 + LVL_DIFF], h.H, h.W)
                    out = cats[lvl].slice(0, nf[lvl])
                h = conv_unit(plan, h, f"{root}.likelihood_post_ups_path.{k}.{2 * t + 1}.convolution.0", out=out)
            if lvl == L - 1:
                post_c[lvl] = h
        for lvl in reversed(range(L - 1)):
            cat = cats[lvl]
            plan.bilinear(post_c[lvl + 1], True, out=cat.slice(nf[lvl], nf[lvl + 1 + LVL_DIFF]))
            if self.reversible:
                post_c[lvl] = plan.rev_sequence(cat, f"{root}.likelihood_post_c_path.{lvl}", nf[lvl + LVL_DIFF], 2, conv_unit)
            else:
                h = conv_unit(plan, cat, f"{root}.likelihood_post_c_path.{lvl}.convolution.0")
                post_c[lvl] = conv_unit(plan, h, f"{root}.likelihood_post_c_path.{lvl}.convolution.1")
        s = [None] * L
        for k in range(L):
            lvl = L - 1 - k
            s_in = plan.conv_bare(post_c[lvl], f"{root}.s_layer.{k}.convolution.0.convolution.0")
            factor = self._H // s_in.H
            s[lvl] = plan.nearest(s_in, factor, f"{root}.s{lvl}")
        return s

    def _build(self, N, H, W, training, bn_training, decode_only=False):
        if H % 64 or W % 64:
            raise ValueError("PHISeg needs H and W divisible by 64 (7 resolution levels)")
        self._H = H
        plan = self._new_plan(N, bn_training)
        plan.bn_prefixes_nbt = []
        io = {}
        shapes = [(2, H >> (RES_LEVELS - 1 - k), W >> (RES_LEVELS - 1 - k)) for k in range(LAT_LEVELS)]
        if decode_only:
            io["z_in"] = [plan.buf(f"z_in{k}", *shapes[k], requires_grad=False) for k in range(LAT_LEVELS)]
            io["s"] = self._likelihood(plan, io["z_in"])
            plan.total = plan.vec("total", 1)
            plan.finalize(want_backward=False)
            plan.io = io
            return plan
        io["patch"] = plan.buf("patch", self.input_channels, H, W, requires_grad=False)
        io["mask"] = plan.buf("mask", 1, H, W, requires_grad=False)
        eps = [plan.buf(f"eps{k}", *shapes[k % LAT_LEVELS], requires_grad=False) for k in range(2 * LAT_LEVELS)]
        io["eps"] = eps
        xin = plan.posterior_input(io["patch"], io["mask"], 2, "posterior.input")
        post, post_z = self._encoder(plan, "posterior", xin, eps[:LAT_LEVELS], None, True)
        if training:       # prior sees the posterior samples (phiseg.py:417-418, 201-202)
            prior, prior_z = self._encoder(plan, "prior", io["patch"], eps[LAT_LEVELS:], post_z, False)
            s = self._likelihood(plan, post_z)
        else:
            prior, prior_z = self._encoder(plan, "prior", io["patch"], eps[LAT_LEVELS:], None, True)
            s = self._likelihood(plan, prior_z)
        io.update(s=s, post=post, prior=prior, post_z=post_z, prior_z=prior_z)
        # ---- loss: sum_l 4^l KL_l + sum_l CE_l (phiseg.py:455-513)
        plan.loss_phase()
        io["terms"] = plan.vec("loss_terms", 2 * LAT_LEVELS)
        plan.total = plan.vec("total", 1)
        io["loss_mask"] = plan.buf("loss_mask", 1, H, W, requires_grad=False)
        for lvl in range(LAT_LEVELS):
            k = LAT_LEVELS - 1 - lvl
            w = float(self.exponential_weight ** lvl) if self.exponential_weighting else 1.0
            plan.kl(post[k], prior[k], w * self.kl_divergence_loss_weight, io["terms"].slice(lvl, 1))
        plan.residual_ce(s, io["loss_mask"], io["terms"].slice(LAT_LEVELS, LAT_LEVELS))
        plan.sum_terms(io["terms"], 2 * LAT_LEVELS, plan.total)
        plan.finalize(want_backward=bn_training)
        plan.io = io
        return plan

    # ------------------------------------------------------------------ reference API
    def forward(self, patch, mask, training=True, eps=None):
        """PHISeg.forward (phiseg.py:414-426).  `eps` (optional, list of 10 tensors: 5 posterior draws
        deepest first, then 5 prior draws) injects the latent noise; default draws it on the device."""
        self._require_gpu()
        N, _, H, W = patch.shape
        key = (N, H, W, bool(training), bool(self.training))
        plan = self._plan(key, lambda: self._build(N, H, W, bool(training), bool(self.training)))
        io = plan.io
        plan.tensor(io["patch"]).copy_(patch)
        plan.tensor(io["mask"]).copy_(mask.reshape(N, 1, H, W))
        if eps is None:
            flat = plan.__dict__.get("_eps_span", False)
            if flat is False:
                flat = plan._eps_span = plan.span(io["eps"])
            if flat is not None:
                self._fill_normal(flat)                       # the ten noise buffers are neighbours in the arena: one launch
            else:
                for e in io["eps"]:
                    self._fill_normal(plan.tensor(e))
        else:
            for e, src in zip(io["eps"], eps):
                plan.tensor(e).copy_(src)
        self._run(plan, "fwd")
        if self.training:
            self._bump_nbt(plan)
        self._cur = plan
        T = plan.tensor
        order = [LAT_LEVELS - 1 - lvl for lvl in range(LAT_LEVELS)]        # level -> draw index
        self.posterior_mu = [T(io["post"][k].mu) for k in order]
        self.posterior_sigma = [T(io["post"][k].sigma) for k in order]
        self.posterior_latent_space = [T(io["post_z"][k]) for k in order]
        self.prior_mu = [T(io["prior"][k].mu) for k in order]
        self.prior_sigma = [T(io["prior"][k].sigma) for k in order]
        self.prior_latent_space = [T(io["prior_z"][k]) for k in order]
        self.s_out_list = [T(v) for v in io["s"]]
        return self.s_out_list

    def loss(self, segm):
        return self.elbo(segm)

    def elbo(self, segm, reconstruct_posterior_mean=False):
        """phiseg.py:519-537.  Returns a 0-d tensor whose backward() runs the backward tape."""
        plan = self._cur
        if plan is None or "terms" not in plan.io:
            raise RuntimeError("call forward() before loss()")
        N, _, H, W = plan.tensor(plan.io["loss_mask"]).shape
        plan.tensor(plan.io["loss_mask"]).copy_(segm.reshape(N, 1, H, W))
        total = self._loss_tensor(plan) if (torch.is_grad_enabled() and plan.tapes["bwd"][1]) else self._loss_nograd(plan)
        terms = plan.tensor(plan.io["terms"]).reshape(-1).clone()
        self.loss_dict = {}
        for lvl in reversed(range(LAT_LEVELS)):
            self.loss_dict["KL_divergence_loss_lvl%d" % lvl] = terms[lvl]
        for lvl in reversed(range(LAT_LEVELS)):
            self.loss_dict["residual_multinoulli_loss_lvl%d" % lvl] = terms[LAT_LEVELS + lvl]
        # the reference accumulates everything into ONE tensor object (phiseg.py:523,477,512)
        self.loss_tot = self.kl_divergence_loss = self.reconstruction_loss = total
        return total

    def _loss_nograd(self, plan):
        plan.run("loss", self._stream())
        return plan.tensor(plan.total).reshape(()).clone()

    def kl_divergence(self):
        return self.kl_divergence_loss

    # ---- the reference's public loss helpers (phiseg.py:436-513) as stand-alone device evaluations.  They return
    # plain values (no autograd graph): gradients of the training loss come from the backward tape of loss().
    def KL_two_gauss_with_diag_cov(self, mu0, sigma0, mu1, sigma1):
        """phiseg.py:436-453 incl. the sigma1*sigma0 quirk; (N, ...) device tensors -> 0-d tensor."""
        self._require_gpu()
        t = [x.detach().to(self.device, torch.float32).contiguous() for x in (mu0, sigma0, mu1, sigma1)]
        N = t[0].shape[0]
        out = torch.empty(1, device=self.device)
        _ffi.check(_ffi.lib().uz_kl_fwd(t[0].data_ptr(), t[1].data_ptr(), t[2].data_ptr(), t[3].data_ptr(), N,
                                        t[0].numel() // N, 1.0, out.data_ptr(), self._stream()), "kl_fwd")
        return out.reshape(())

    def calculate_hierarchical_KL_div_loss(self):
        """phiseg.py:455-479 on the cached posterior / prior moments of the last forward()."""
        w = [self.exponential_weight ** i if self.exponential_weighting else 1 for i in range(self.latent_levels)]
        for ii in reversed(range(self.latent_levels)):
            self.loss_dict["KL_divergence_loss_lvl%d" % ii] = w[ii] * self.KL_two_gauss_with_diag_cov(
                self.posterior_mu[ii], self.posterior_sigma[ii], self.prior_mu[ii], self.prior_sigma[ii])
            self.loss_tot = self.loss_tot + self.kl_divergence_loss_weight * self.loss_dict["KL_divergence_loss_lvl%d" % ii]
        return self.loss_tot

    def _ce_levels(self, logits, target):
        self._require_gpu()
        logits = [x.detach().to(self.device, torch.float32).contiguous() for x in logits]
        N, K, H, W = logits[0].shape
        L = _ffi.lib()
        tgt = target.detach().to(self.device, torch.float32).reshape(N, 1, H, W).contiguous()
        tab = torch.tensor([x.data_ptr() for x in logits], dtype=torch.int64, device=self.device)
        ws = torch.empty(L.uz_ce_workspace(N, H, W, len(logits)) // 4 + 16, device=self.device)
        out = torch.empty(len(logits), device=self.device)
        _ffi.check(L.uz_residual_ce_fwd(tab.data_ptr(), len(logits), K, tgt.data_ptr(), N, H, W, out.data_ptr(), ws.data_ptr(),
                                        self._stream()), "residual_ce_fwd")
        return out

    def multinoulli_loss(self, reconstruction, target):
        """phiseg.py:481-490: per-pixel CE summed over pixels, mean over the batch."""
        return self._ce_levels([reconstruction], target)[0]

    def residual_multinoulli_loss(self, reconstruction, target):
        """phiseg.py:492-513 on a list of level logits (finest first)."""
        terms = self._ce_levels(list(reconstruction), target)
        for ii in reversed(range(len(reconstruction))):
            self.loss_dict["residual_multinoulli_loss_lvl%d" % ii] = terms[ii]
            self.loss_tot = self.loss_tot + self.residual_multinoulli_loss_weight * terms[ii]
        return self.loss_tot

    def accumulate_output(self, output_list, use_softmax=False):
        """phiseg.py:428-434 incl. the in-place accumulation into output_list[-1]."""
        self._require_gpu()
        last = output_list[-1]
        N, K, H, W = last.shape
        assert all(t.is_contiguous() for t in output_list)
        tab = torch.tensor([t.data_ptr() for t in output_list], dtype=torch.int64, device=self.device)
        soft = torch.empty_like(last) if use_softmax else None
        L = _ffi.lib()
        L.uz_accumulate_softmax_argmax.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                                                   C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        _ffi.check(L.uz_accumulate_softmax_argmax(tab.data_ptr(), len(output_list), K, N, H, W, last.data_ptr(),
                                                  soft.data_ptr() if use_softmax else None, None, self._stream()), "accumulate_output")
        return soft if use_softmax else last

    def sample_posterior(self):
        return [m + s * torch.randn_like(s) for m, s in zip(self.posterior_mu, self.posterior_sigma)]

    def sample_prior(self):
        return [m + s * torch.randn_like(s) for m, s in zip(self.prior_mu, self.prior_sigma)]

    def reconstruct(self, z_posterior, use_softmax=True):
        """Decode a list of latent samples (finest level first) through the likelihood (phiseg.py:403-405)."""
        self._require_gpu()
        N = z_posterior[0].shape[0]
        H, W = z_posterior[0].shape[2] * 4, z_posterior[0].shape[3] * 4
        key = ("decode", N, H, W, bool(self.training))
        plan = self._plan(key, lambda: self._build(N, H, W, False, bool(self.training), decode_only=True))
        for lvl, z in enumerate(z_posterior):
            plan.tensor(plan.io["z_in"][LAT_LEVELS - 1 - lvl]).copy_(z)
        self._run(plan, "fwd")
        if self.training:
            self._bump_nbt(plan)
        layer_recon = [plan.tensor(v).clone() for v in plan.io["s"]]
        return self.accumulate_output(layer_recon, use_softmax=use_softmax), layer_recon

    def sample(self, testing=True):
        if testing:
            sample, _ = self.reconstruct(self.sample_prior(), use_softmax=False)
            return sample
        raise NotImplementedError
