"""PHiSeg3D on the native HIP path - counterpart of the reference ``models/phiseg3D.py`` ``PHISeg3D`` (constructor keywords
:413-425; ``forward(patch, mask, training)`` :454-467; the same loss helpers as the 2-D model :469-611): Conv3d 3x3x3 +
BatchNorm3d + ReLU units (:13-35), AvgPool3d(2, ceil) (:101), trilinear x2 with align_corners=True (:146,306,376), 1x1x1 latent
heads (:184-186), optional ReversibleSequence with reversible_depth 1 (:61-86,104,142,176,341,352), resolution levels =
len(num_filters) and `latent_levels` free (:245-248 - unlike the 2-D model nothing is hard-wired to 7 / 5).

First slice, with these limits stated up front:
  * ONE sample per call (the reference's BraTS experiment runs batch_size 1, phiseg_brats.py:28): a volume is held as
    [D + 2][C][H][W], i.e. its depth slices are the batch of the 2-D kernels (csrc/vol.hip, _plan.Plan.vol);
  * fp32 storage; the 3x3x3 convolutions run on the split-fp16 matrix-pipe kernels of the 2-D path with K = 27 Cin (BASELINE
    config 5 asks for bf16 storage, which would make the model HBM-bound: next);
  * the reference's own forward raises at its last line (:398 hands a 2-element `size` to a 5-D `interpolate`); the evident
    intent - nearest resize of every level to the full volume - is what runs here, and parity is pinned on everything up to
    the un-resized level logits `s_in` (Posterior, prior, Likelihood), which the reference does execute.
"""
import torch

from .._engine import NativeModel, conv_unit
from .._modtree import conv_unit_spec, plain_conv_spec, rev_sequence_spec, init_default


def _unit3(prefix, cin, cout, k=3, norm=True):
    return [(key, (shape + (shape[-1],) if kind == "conv_w" else shape), kind) for key, shape, kind in conv_unit_spec(prefix, cin, cout, k=k, norm=norm)]


def _rev3(prefix, cin, cout, depth):
    return [(key, (shape + (shape[-1],) if kind == "conv_w" else shape), kind) for key, shape, kind in rev_sequence_spec(prefix, cin, cout, depth)]


def _plain3(prefix, cin, cout, k):
    return [(key, (shape + (shape[-1],) if kind == "conv_w" else shape), kind) for key, shape, kind in plain_conv_spec(prefix, cin, cout, k)]


def _encoder_spec(root, in_ch, nf, L, reversible):
    R = len(nf)
    out = []
    for i in range(R):
        cin = in_ch if i == 0 else nf[i - 1]
        base = 0 if i == 0 else 1
        if reversible:
            out += _rev3(f"{root}.contracting_path.{i}.layers.{base}", cin, nf[i], 1)
        else:
            for j in range(3):
                out += _unit3(f"{root}.contracting_path.{i}.layers.{base + j}", cin if j == 0 else nf[i], nf[i])
    for k in range(L):
        if reversible:
            out += _rev3(f"{root}.upsampling_path.{k}.upconv_layer", 2, 2 * nf[0], 1)
        else:
            out += _unit3(f"{root}.upsampling_path.{k}.upconv_layer.0", 2, 2 * nf[0]) + _unit3(f"{root}.upsampling_path.{k}.upconv_layer.1", 2 * nf[0], 2 * nf[0])
    for k in range(L):
        i = L - 1 - k
        cin = nf[i + R - L] if k == 0 else 2 * nf[0] + nf[i + R - L]
        p = f"{root}.sample_z_path.{k}"
        out += _rev3(p + ".conv.0", cin, cin, 1) if reversible else _unit3(p + ".conv.0", cin, cin) + _unit3(p + ".conv.1", cin, cin)
        out += _plain3(p + ".mu_conv.0", cin, 2, 1) + _plain3(p + ".sigma_conv.0", cin, 2, 1)
    return out


def _likelihood_spec(nf, L, num_classes, reversible):
    R, root, out = len(nf), "likelihood", []
    diff = R - L
    for k in range(L):
        c = nf[L - 1 - k]
        if reversible:
            out += _rev3(f"{root}.likelihood_ups_path.{k}", 2, c, 1)
        else:
            out += _unit3(f"{root}.likelihood_ups_path.{k}.convolution.0", 2, c) + _unit3(f"{root}.likelihood_ups_path.{k}.convolution.1", c, c)
    for k in range(L):
        c = nf[L - 1 - k]
        for t in range(diff):
            out += _unit3(f"{root}.likelihood_post_ups_path.{k}.{2 * t + 1}.convolution.0", c, c)
    for i in range(L - 1):
        cin, cout = nf[i] + nf[i + 1 + diff], nf[i + diff]
        if reversible:
            out += _rev3(f"{root}.likelihood_post_c_path.{i}", cin, cout, 1)
        else:
            out += _unit3(f"{root}.likelihood_post_c_path.{i}.convolution.0", cin, cout) + _unit3(f"{root}.likelihood_post_c_path.{i}.convolution.1", cout, cout)
    for k in range(L):
        out += _unit3(f"{root}.s_layer.{k}.convolution.0", nf[L - 1 - k + diff], num_classes, k=1, norm=False)
    return out


def phiseg3d_spec(input_channels, num_classes, num_filters, latent_levels, reversible=False):
    nf = list(num_filters)
    return (_encoder_spec("posterior", input_channels + num_classes, nf, latent_levels, reversible)
            + _likelihood_spec(nf, latent_levels, num_classes, reversible)
            + _encoder_spec("prior", input_channels, nf, latent_levels, reversible))


class PHISeg3D(NativeModel):
    def __init__(self, input_channels, num_classes, num_filters, latent_levels=5, initializers=None, no_convs_fcomb=4, beta=10.0,
                 image_size=(128, 128, 1), reversible=False, apply_last_layer=True, exponential_weighting=True, padding=True, device=None):
        super().__init__()
        self.reversible = bool(reversible)
        self.input_channels, self.num_classes, self.num_filters = input_channels, num_classes, list(num_filters)
        self.latent_levels, self.image_size = latent_levels, image_size
        if len(num_filters) < latent_levels:
            raise ValueError("PHISeg3D needs at least `latent_levels` filters")
        diff = len(num_filters) - latent_levels
        if latent_levels > 1 and num_filters[latent_levels - 1] != num_filters[latent_levels - 1 + diff]:
            # likelihood_post_c_path's first convolution is built for num_filters[i] + num_filters[i + 1 + lvl_diff] inputs
            # (:343) but is fed post_z[latent_levels - 1], which has num_filters[latent_levels - 1] channels (:372): the
            # reference raises on such a configuration at its first forward pass
            raise ValueError("PHISeg3D: num_filters[latent_levels-1] must equal num_filters[-1] (reference channel arithmetic, phiseg3D.py:343,372)")
        # the reference's Posterior concatenates a TWO-label one-hot whatever num_classes is (:275) while its first convolution
        # takes input_channels + num_classes (:218): only num_classes == 2 runs there; here the one-hot has num_classes labels
        self.loss_tot, self.loss_dict = 0, {}
        self.kl_divergence_loss_weight, self.beta = 1.0, 1.0
        self.exponential_weighting, self.exponential_weight = exponential_weighting, 4
        self.residual_multinoulli_loss_weight = 1.0
        self.kl_divergence_loss = self.reconstruction_loss = 0
        self.s_out_list = [None] * latent_levels
        self._init_storage(phiseg3d_spec(input_channels, num_classes, num_filters, latent_levels, self.reversible), device)
        init_default(self._ptab)

    # ------------------------------------------------------------------ plan construction
    def _stack(self, plan, x, prefix, cout, n_units, out=None, rev_prefix=None):
        """`n_units` Conv3D units - or, in the reversible variant, ReversibleSequence(cin, cout, reversible_depth=1)."""
        if self.reversible and rev_prefix is not None:
            return plan.rev_sequence(x, rev_prefix, cout, 1, conv_unit, out=out)
        for j, pfx in enumerate(prefix):
            x = conv_unit(plan, x, pfx, out=out if j == len(prefix) - 1 else None)
        return x

    def _encoder(self, plan, root, x, eps, z_override, want_z):
        nf, L = self.num_filters, self.latent_levels
        R = len(nf)
        skips = {}
        for i in range(R):
            base = 0
            if i != 0:
                x = plan.avgpool3d(x, f"{root}.pool{i}")
                base = 1
            out = None
            if R - L <= i <= R - 2:         # blocks[R-L .. R-2] are concatenated with an up-sampled z (phiseg3D.py:149)
                cat = plan.vol(f"{root}.cat{i}", 2 * nf[0] + nf[i], x.N, x.H, x.W)
                out = cat.slice(2 * nf[0], nf[i])
                skips[i] = cat
            pfx = [f"{root}.contracting_path.{i}.layers.{base + j}" for j in range(3)]
            x = self._stack(plan, x, pfx, nf[i], 3, out=out, rev_prefix=f"{root}.contracting_path.{i}.layers.{base}")
        lats, zs = [], []
        pre = x
        for k in range(L):
            if k != 0:
                cat = skips[R - 1 - k]
                u = plan.trilinear(zs[k - 1], name=f"{root}.up{k}.tri")
                self._stack(plan, u, [f"{root}.upsampling_path.{k - 1}.upconv_layer.{j}" for j in range(2)], 2 * nf[0], 2,
                            out=cat.slice(0, 2 * nf[0]), rev_prefix=f"{root}.upsampling_path.{k - 1}.upconv_layer")
                pre = cat
            p = f"{root}.sample_z_path.{k}"
            h = self._stack(plan, pre, [p + ".conv.0", p + ".conv.1"], pre.C, 2, rev_prefix=p + ".conv.0")
            mu = plan.conv_bare(h, p + ".mu_conv.0")
            ps = plan.conv_bare(h, p + ".sigma_conv.0")
            lat = plan.latent(mu, ps, eps[k], f"{root}.lat{k}", want_z=want_z, act=0)
            lats.append(lat)
            zs.append(z_override[k] if z_override is not None else lat.z)
        return lats, zs

    def _likelihood(self, plan, zs, full):
        nf, L, root = self.num_filters, self.latent_levels, "likelihood"
        diff = len(nf) - L
        cats, post_c, s_in = {}, [None] * L, [None] * L
        for k in range(L):
            lvl = L - 1 - k
            final = None
            if lvl < L - 1:                 # post_z[lvl] is concatenated with the up-sampled post_c[lvl + 1] (:381)
                cats[lvl] = plan.vol(f"{root}.cat{lvl}", nf[lvl] + nf[lvl + 1 + diff], zs[k].N << diff, zs[k].H << diff, zs[k].W << diff)
                final = cats[lvl].slice(0, nf[lvl])
            h = self._stack(plan, zs[k], [f"{root}.likelihood_ups_path.{k}.convolution.{j}" for j in range(2)], nf[lvl], 2,
                            out=final if diff == 0 else None, rev_prefix=f"{root}.likelihood_ups_path.{k}")
            for t in range(diff):
                h = plan.trilinear(h, name=f"{root}.ups{k}.tri{t}")
                h = conv_unit(plan, h, f"{root}.likelihood_post_ups_path.{k}.{2 * t + 1}.convolution.0", out=final if t == diff - 1 else None)
            if lvl == L - 1:
                post_c[lvl] = h
        for lvl in reversed(range(L - 1)):
            cat = cats[lvl]
            plan.trilinear(post_c[lvl + 1], name=f"{root}.cat{lvl}.tri", out=cat.slice(nf[lvl], nf[lvl + 1 + diff]))
            post_c[lvl] = self._stack(plan, cat, [f"{root}.likelihood_post_c_path.{lvl}.convolution.{j}" for j in range(2)], nf[lvl + diff], 2,
                                      rev_prefix=f"{root}.likelihood_post_c_path.{lvl}")
        s = [None] * L
        for k in range(L):
            lvl = L - 1 - k
            s_in[lvl] = plan.conv_bare(post_c[lvl], f"{root}.s_layer.{k}.convolution.0.convolution.0")
            f = full[1] // s_in[lvl].H
            s[lvl] = plan.nearest3d(s_in[lvl], f, full[0] // s_in[lvl].N, f"{root}.s{lvl}") if f > 1 else s_in[lvl]
        return s, s_in

    def _build(self, D, H, W, training, bn_training, decode_only=False):
        R, L, K = len(self.num_filters), self.latent_levels, self.num_classes
        if D % (1 << (R - 1)) or H % (1 << (R - 1)) or W % (1 << (R - 1)):
            raise ValueError("PHISeg3D needs every spatial size divisible by 2^(levels-1)")
        plan = self._new_plan(D, bn_training)
        plan.bn_prefixes_nbt = []
        io = {}
        shapes = [(2, D >> (R - 1 - k), H >> (R - 1 - k), W >> (R - 1 - k)) for k in range(L)]
        if decode_only:                                       # reconstruct() / sample(): the likelihood on given latent volumes
            io["z_in"] = [plan.vol(f"z_in{k}", *shapes[k], requires_grad=False) for k in range(L)]
            io["s"], io["s_in"] = self._likelihood(plan, io["z_in"], (D, H, W))
            plan.finalize(want_backward=False)
            plan.io = io
            return plan
        io["eps"] = [plan.vol(f"eps{k}", *shapes[k % L], requires_grad=False) for k in range(2 * L)]
        # cat(patch, one-hot mask - 0.5) (:275-279) is staged by forward(); the prior reads its first input_channels channels
        xin = io["input"] = plan.vol("posterior.input", self.input_channels + K, D, H, W, requires_grad=False)
        io["patch"] = xin.slice(0, self.input_channels)
        post, post_z = self._encoder(plan, "posterior", xin, io["eps"][:L], None, True)
        if training:
            prior, prior_z = self._encoder(plan, "prior", io["patch"], io["eps"][L:], post_z, False)
            s, s_in = self._likelihood(plan, post_z, (D, H, W))
        else:
            prior, prior_z = self._encoder(plan, "prior", io["patch"], io["eps"][L:], None, True)
            s, s_in = self._likelihood(plan, prior_z, (D, H, W))
        io.update(s=s, s_in=s_in, post=post, prior=prior, post_z=post_z, prior_z=prior_z)
        plan.loss_phase()
        io["terms"] = plan.vec("loss_terms", 2 * L)
        plan.total = plan.vec("total", 1)
        io["loss_mask"] = plan.vol("loss_mask", 1, D, H, W, requires_grad=False)
        for lvl in range(L):
            k = L - 1 - lvl
            w = float(self.exponential_weight ** lvl) if self.exponential_weighting else 1.0
            plan.kl(post[k], prior[k], w * self.kl_divergence_loss_weight, io["terms"].slice(lvl, 1))
        # the CE kernel averages over its batch axis, which here are the D slices of ONE sample: scale back to a sum over voxels
        plan.residual_ce(s, io["loss_mask"], io["terms"].slice(L, L), post_scale=float(D))
        plan.sum_terms(io["terms"], 2 * L, plan.total)
        plan.finalize(want_backward=bn_training)
        plan.io = io
        return plan

    # ------------------------------------------------------------------ reference API (volumes are NCDHW at the boundary)
    @staticmethod
    def _to_slices(t):
        return t[0].permute(1, 0, 2, 3)                       # (1, C, D, H, W) -> (D, C, H, W)

    @staticmethod
    def _to_volume(t):
        return t.permute(1, 0, 2, 3).unsqueeze(0)             # (D, C, H, W) -> (1, C, D, H, W)

    def forward(self, patch, mask, training=True, eps=None):
        self._require_gpu()
        if patch.shape[0] != 1:
            raise NotImplementedError("native PHISeg3D runs one volume per call (the reference's BraTS config uses batch_size 1)")
        _, _, D, H, W = patch.shape
        key = (D, H, W, bool(training), bool(self.training))
        plan = self._plan(key, lambda: self._build(D, H, W, bool(training), bool(self.training)))
        io, T = plan.io, plan.tensor
        K = self.num_classes
        T(io["patch"]).copy_(self._to_slices(patch))
        if mask.dim() == 5 and mask.shape[1] == K:            # already one-hot, as the reference's BraTS path hands it over (utils.py:296-298)
            onehot = self._to_slices(mask).to(torch.int64).to(torch.float32)
        else:                                                 # a label volume: one-hot encode it here
            lab = mask.reshape(D, 1, H, W)
            onehot = torch.cat([(lab == k) for k in range(K)], dim=1).to(torch.float32)
        T(io["input"].slice(self.input_channels, K)).copy_(onehot - 0.5)
        for k, e in enumerate(io["eps"]):
            if eps is None:
                self._fill_normal(T(e))
            else:
                T(e).copy_(self._to_slices(eps[k]))
        self._run(plan, "fwd")
        if self.training:
            self._bump_nbt(plan)
        self._cur = plan
        L = self.latent_levels
        order = [L - 1 - lvl for lvl in range(L)]
        V = lambda v: self._to_volume(T(v))                   # noqa: E731
        self.posterior_mu = [V(io["post"][k].mu) for k in order]
        self.posterior_sigma = [V(io["post"][k].sigma) for k in order]
        self.posterior_latent_space = [V(io["post_z"][k]) for k in order]
        self.prior_mu = [V(io["prior"][k].mu) for k in order]
        self.prior_sigma = [V(io["prior"][k].sigma) for k in order]
        self.s_in_list = [V(v) for v in io["s_in"]]
        self.s_out_list = [V(v) for v in io["s"]]
        return self.s_out_list

    def loss(self, segm):
        plan = self._cur
        if plan is None:
            raise RuntimeError("call forward() before loss()")
        D, _, H, W = plan.tensor(plan.io["loss_mask"]).shape
        plan.tensor(plan.io["loss_mask"]).copy_(self._to_slices(segm.reshape(1, 1, D, H, W)))
        if torch.is_grad_enabled() and plan.tapes["bwd"][1]:
            total = self._loss_tensor(plan)
        else:
            plan.run("loss", self._stream())
            total = plan.tensor(plan.total).reshape(()).clone()
        terms = plan.tensor(plan.io["terms"]).reshape(-1).clone()
        L = self.latent_levels
        self.loss_dict = {}
        for lvl in reversed(range(L)):
            self.loss_dict["KL_divergence_loss_lvl%d" % lvl] = terms[lvl]
        for lvl in reversed(range(L)):
            self.loss_dict["residual_multinoulli_loss_lvl%d" % lvl] = terms[L + lvl]
        # the reference accumulates both parts into loss_tot and returns the running total from each helper (:547-555,:582-583,
        # :597-604): kl_divergence_loss is the weighted KL sum, reconstruction_loss the grand total
        self.kl_divergence_loss = terms[:L].sum()
        self.loss_tot = self.reconstruction_loss = total
        return total

    def accumulate_output(self, output_list, use_softmax=False):
        """phiseg3D.py:469-475 (in place on the last level, as the reference)."""
        s_accum = output_list[-1]
        for i in range(len(output_list) - 1):
            s_accum += output_list[i]
        return torch.softmax(s_accum, dim=1) if use_softmax else s_accum

    def elbo(self, segm, reconstruct_posterior_mean=False):
        return self.loss(segm)

    def sample_posterior(self):
        """phiseg3D.py:426-433."""
        return [m + s * torch.randn_like(s) for m, s in zip(self.posterior_mu, self.posterior_sigma)]

    def sample_prior(self):
        """phiseg3D.py:435-441."""
        return [m + s * torch.randn_like(s) for m, s in zip(self.prior_mu, self.prior_sigma)]

    def reconstruct(self, z_posterior, use_softmax=True):
        """Decode latent volumes (finest level first) through the likelihood (phiseg3D.py:450-452)."""
        self._require_gpu()
        R, L = len(self.num_filters), self.latent_levels
        _, _, d, h, w = z_posterior[0].shape
        D, H, W = d << (R - L), h << (R - L), w << (R - L)
        key = ("decode", D, H, W, bool(self.training))
        plan = self._plan(key, lambda: self._build(D, H, W, False, bool(self.training), decode_only=True))
        for lvl, z in enumerate(z_posterior):
            plan.tensor(plan.io["z_in"][L - 1 - lvl]).copy_(self._to_slices(z))
        self._run(plan, "fwd")
        if self.training:
            self._bump_nbt(plan)
        layer_recon = [self._to_volume(plan.tensor(v)).clone() for v in plan.io["s"]]
        return self.accumulate_output(layer_recon, use_softmax=use_softmax), layer_recon

    def sample(self, testing=True):
        """phiseg3D.py:443-448."""
        if testing:
            sample, _ = self.reconstruct(self.sample_prior(), use_softmax=False)
            return sample
        raise NotImplementedError

    def kl_divergence(self):
        return self.kl_divergence_loss
