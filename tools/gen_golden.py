#!/usr/bin/env python3
"""Generate golden vectors from the REAL reference (build container only).

Imports /root/reference read-only (never copied, never shipped), runs its
models on CPU with fixed weights / inputs / noise, and writes plain-data
fixtures (npz arrays + json metadata) under tests/golden/.  The GPU box never
sees the reference; it only sees these vectors.

    python tools/gen_golden.py            # regenerates every fixture

Recipe for importing the reference: SURVEY.md Appendix B (three absent
third-party modules, never called on the hot path, are stubbed).
"""
import json
import os
import sys
import types

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(ROOT, "tests", "golden")
sys.path.insert(0, ROOT)

for _name in ("medpy", "medpy.metric", "nibabel", "revtorch"):
    sys.modules[_name] = types.ModuleType(_name)
# MedPy==0.4.0 (requirements.txt:17) is not vendored; its binary Jaccard / Dice coefficients are restated here so that the
# reference's own metric functions (utils.generalised_energy_distance, variance_ncc_dist) can produce golden values
def _jc(a, b):
    import numpy as _np
    a, b = _np.asarray(a).astype(bool), _np.asarray(b).astype(bool)
    return float(_np.count_nonzero(a & b)) / float(_np.count_nonzero(a | b))
def _dc(a, b):
    import numpy as _np
    a, b = _np.asarray(a).astype(bool), _np.asarray(b).astype(bool)
    return 2.0 * _np.count_nonzero(a & b) / float(_np.count_nonzero(a) + _np.count_nonzero(b))
sys.modules["medpy.metric"].jc, sys.modules["medpy.metric"].dc = _jc, _dc
sys.dont_write_bytecode = True
sys.path.insert(0, "/root/reference")

import torch  # noqa: E402

torch.set_num_threads(4)
from models.phiseg import PHISeg  # noqa: E402  (reference)
from models.unet import Unet  # noqa: E402  (reference)
from models.probabilistic_unet import ProbabilisticUnet  # noqa: E402  (reference)
import utils as ref_utils  # noqa: E402  (reference)

from oracle.refgraph import synthetic_batch, deterministic_state_dict, phiseg_eps_shapes  # noqa: E402


# --------------------------------------------------------------------------- #
def kinds_for(sd):
    """(key, shape, kind) spec for oracle.deterministic_state_dict from a real state_dict."""
    keys = list(sd.keys())
    spec = []
    for k in keys:
        v = sd[k]
        if k.endswith("num_batches_tracked"):
            kind = "bn_nbt"
        elif k.endswith("running_mean"):
            kind = "bn_rm"
        elif k.endswith("running_var"):
            kind = "bn_rv"
        elif v.dim() >= 4:
            kind = "conv_w"
        else:
            stem = k.rsplit(".", 1)[0]
            is_conv = sd[stem + ".weight"].dim() >= 4
            if is_conv:
                kind = "conv_b"
            else:
                kind = "bn_w" if k.endswith(".weight") else "bn_b"
        spec.append((k, tuple(v.shape), kind))
    return spec


class NoiseFeeder:
    """Replaces torch.randn_like / distributions' _standard_normal by a queue of given tensors."""

    def __init__(self, tensors):
        self.q = [torch.as_tensor(t) for t in tensors]
        self.used = 0

    def randn_like(self, ref, **kw):
        t = self.q[self.used]
        self.used += 1
        assert tuple(t.shape) == tuple(ref.shape), (t.shape, ref.shape)
        return t.clone()

    def standard_normal(self, shape, dtype, device):
        t = self.q[self.used]
        self.used += 1
        assert tuple(t.shape) == tuple(shape), (t.shape, shape)
        return t.clone()

    def __enter__(self):
        self._a = torch.randn_like
        self._b = torch.distributions.normal._standard_normal
        torch.randn_like = self.randn_like
        torch.distributions.normal._standard_normal = self.standard_normal
        return self

    def __exit__(self, *a):
        torch.randn_like = self._a
        torch.distributions.normal._standard_normal = self._b


def npf(t):
    return t.detach().cpu().numpy().astype(np.float32)


def save(name, arrays, meta):
    os.makedirs(OUT, exist_ok=True)
    np.savez_compressed(os.path.join(OUT, name + ".npz"), **arrays)
    with open(os.path.join(OUT, name + ".json"), "w") as f:
        json.dump(meta, f, indent=1)
    sz = os.path.getsize(os.path.join(OUT, name + ".npz"))
    print(f"wrote {name}: {len(arrays)} arrays, {sz / 1024:.0f} KiB")


def grads_of(net):
    return {n: (None if p.grad is None else npf(p.grad)) for n, p in net.named_parameters()}


# --------------------------------------------------------------------------- #
def phiseg_case(name, filters, hw, batch, n_steps, store_full, seed, store_inputs=True):
    net = PHISeg(input_channels=1, num_classes=2, num_filters=filters, latent_levels=5,
                 image_size=(1, hw, hw))
    spec = kinds_for(net.state_dict())
    sd0 = deterministic_state_dict(spec, seed=seed)
    net.load_state_dict(sd0)
    net.train()
    opt = torch.optim.Adam(net.parameters(), lr=1e-3, weight_decay=1e-5)   # train_model.py:49

    shapes = phiseg_eps_shapes(batch, hw, hw)
    arrays, meta = {}, dict(model="PHISeg", filters=filters, hw=hw, batch=batch, weight_seed=seed,
                            spec=[[k, list(s), kd] for k, s, kd in spec], steps=[])
    for step in range(n_steps):
        x, mask, eps = synthetic_batch(batch, hw, hw, seed=20201004 + step, eps_shapes=shapes + shapes)
        xt, mt = torch.from_numpy(x), torch.from_numpy(mask)
        with NoiseFeeder(eps) as nf:
            s_list = net.forward(xt, mt, training=True)
            assert nf.used == 10
        loss = net.loss(mt)
        opt.zero_grad()
        loss.backward()
        g = grads_of(net)
        st = dict(loss=float(loss), loss_dict={k: float(v) for k, v in net.loss_dict.items()},
                  none_grads=[k for k, v in g.items() if v is None],
                  kl_alias=float(net.kl_divergence_loss), recon_alias=float(net.reconstruction_loss))
        if step == 0:
            if store_inputs:
                arrays["x"], arrays["mask"] = x, mask
                for i, e in enumerate(eps):
                    arrays[f"eps{i}"] = e
            if store_full:
                for l in range(5):
                    arrays[f"s{l}"] = npf(s_list[l])
                    for nm, lst in (("post_mu", net.posterior_mu), ("post_sigma", net.posterior_sigma),
                                    ("post_z", net.posterior_latent_space), ("prior_mu", net.prior_mu),
                                    ("prior_sigma", net.prior_sigma)):
                        arrays[f"{nm}{l}"] = npf(lst[l])
                for k, v in g.items():
                    if v is not None:
                        arrays["grad:" + k] = v
                for k, v in net.state_dict().items():
                    if "running_" in k:
                        arrays["buf1:" + k] = npf(v)
            else:
                # digests only (full-size case): sampled logits, grad norms + 8 sampled entries
                rs = np.random.Generator(np.random.PCG64(7))
                idx = rs.integers(0, batch * 2 * hw * hw, size=256)
                arrays["s_idx"] = idx
                for l in range(5):
                    arrays[f"s{l}_samp"] = npf(s_list[l]).reshape(-1)[idx]
                    arrays[f"post_mu{l}"] = npf(net.posterior_mu[l])
                    arrays[f"post_sigma{l}"] = npf(net.posterior_sigma[l])
                    arrays[f"prior_mu{l}"] = npf(net.prior_mu[l])
                    arrays[f"prior_sigma{l}"] = npf(net.prior_sigma[l])
                gn, gs = {}, {}
                for k, v in g.items():
                    if v is not None:
                        gn[k] = float(np.sqrt((v.astype(np.float64) ** 2).sum()))
                        flat = v.reshape(-1)
                        pick = rs.integers(0, flat.size, size=min(8, flat.size))
                        gs[k] = [pick.tolist(), flat[pick].astype(float).tolist()]
                st["grad_norms"], st["grad_samples"] = gn, gs
        opt.step()
        meta["steps"].append(st)
    if store_full:
        for k, v in net.state_dict().items():
            if v.dtype.is_floating_point:
                arrays["final:" + k] = npf(v)
        meta["final_nbt"] = int(net.state_dict()[spec[6][0]]) if spec[6][2] == "bn_nbt" else None

    # eval-mode pass (validate(): net.eval(), forward(training=False), accumulate_output(softmax), argmax)
    net.load_state_dict(sd0)
    net.eval()
    x, mask, eps = synthetic_batch(batch, hw, hw, seed=20201004, eps_shapes=shapes + shapes)
    with torch.no_grad(), NoiseFeeder(eps):
        s_list = net.forward(torch.from_numpy(x), torch.from_numpy(mask), training=False)
        s_copy = [t.clone() for t in s_list]
        soft = net.accumulate_output(s_list, use_softmax=True)
    arg = torch.argmax(soft, dim=1).numpy().astype(np.uint8)
    acc = sum(s_copy)
    margin = float((acc[:, 1] - acc[:, 0]).abs().min())
    arrays["eval_argmax_bits"] = np.packbits(arg.reshape(-1))
    meta["eval_margin_min"] = margin
    meta["eval_alias_inplace"] = bool(not torch.equal(s_list[-1], s_copy[-1]))   # in-place accumulate quirk (phiseg.py:428-434)
    if store_full:
        arrays["eval_softmax"] = npf(soft)
        for l in range(5):
            arrays[f"eval_s{l}"] = npf(s_copy[l])
    else:
        arrays["eval_acc_samp"] = npf(acc).reshape(-1)[arrays["s_idx"]]
    save(name, arrays, meta)


def phiseg_traj_case(name, filters, hw, batch, n_steps, seed, n_samp=8):
    """Train-step contract AT THE HEADLINE SIZE: n_steps x [forward, loss, zero_grad, backward,
    Adam(lr 1e-3, wd 1e-5)] of the real reference from the deterministic state_dict; per step the loss terms and - behind the
    optimiser step - digests of every parameter and BatchNorm buffer (L2 norm + n_samp sampled entries).  Inputs and noise are
    regenerated from the seeds (synthetic_batch(seed = 20201004 + step))."""
    net = PHISeg(input_channels=1, num_classes=2, num_filters=filters, latent_levels=5, image_size=(1, hw, hw))
    spec = kinds_for(net.state_dict())
    sd0 = deterministic_state_dict(spec, seed=seed)
    net.load_state_dict(sd0)
    net.train()
    opt = torch.optim.Adam(net.parameters(), lr=1e-3, weight_decay=1e-5)   # train_model.py:49
    shapes = phiseg_eps_shapes(batch, hw, hw)
    rs = np.random.Generator(np.random.PCG64(11))
    picks = {k: rs.integers(0, int(np.prod(sh)) if len(sh) else 1, size=min(n_samp, max(1, int(np.prod(sh))))) for k, sh, kd in spec if kd != "bn_nbt"}
    arrays = {"pick:" + k: v for k, v in picks.items()}
    meta = dict(model="PHISeg", filters=filters, hw=hw, batch=batch, weight_seed=seed, lr=1e-3, weight_decay=1e-5,
                spec=[[k, list(sh), kd] for k, sh, kd in spec], steps=[])
    for step in range(n_steps):
        x, mask, eps = synthetic_batch(batch, hw, hw, seed=20201004 + step, eps_shapes=shapes + shapes)
        xt, mt = torch.from_numpy(x), torch.from_numpy(mask)
        with NoiseFeeder(eps) as nf:
            net.forward(xt, mt, training=True)
            assert nf.used == 10
        loss = net.loss(mt)
        opt.zero_grad()
        loss.backward()
        st = dict(loss=float(loss), loss_dict={k: float(v) for k, v in net.loss_dict.items()},
                  none_grads=[k for k, p in net.named_parameters() if p.grad is None])
        opt.step()
        sd = net.state_dict()
        norms = {}
        for k, sh, kd in spec:
            if kd == "bn_nbt":
                continue
            v = npf(sd[k]).reshape(-1)
            norms[k] = float(np.sqrt((v.astype(np.float64) ** 2).sum()))
            arrays[f"step{step}:" + k] = v[picks[k]]
        st["norms"] = norms
        st["nbt"] = int(sd[[k for k, _, kd in spec if kd == "bn_nbt"][0]])
        meta["steps"].append(st)
        print(f"step {step}: loss {float(loss):.6f}", flush=True)
    save(name, arrays, meta)


def unet_case(name, filters, batch, n_steps, seed):
    hw = 128                                                           # Unet.loss hard-codes 128 (unet.py:163)
    net = Unet(1, 2, filters)
    spec = kinds_for(net.state_dict())
    sd0 = deterministic_state_dict(spec, seed=seed)
    net.load_state_dict(sd0)
    net.train()
    opt = torch.optim.Adam(net.parameters(), lr=1e-3, weight_decay=1e-5)
    arrays, meta = {}, dict(model="Unet", filters=filters, hw=hw, batch=batch, weight_seed=seed,
                            spec=[[k, list(s), kd] for k, s, kd in spec], steps=[])
    for step in range(n_steps):
        x, mask, _ = synthetic_batch(batch, hw, hw, seed=20201004 + step)
        pred = net.forward(torch.from_numpy(x))
        loss = net.loss(torch.from_numpy(mask))
        opt.zero_grad()
        loss.backward()
        if step == 0:
            arrays["x"], arrays["mask"], arrays["pred"] = x, mask, npf(pred)
            for k, v in grads_of(net).items():
                arrays["grad:" + k] = v
        opt.step()
        meta["steps"].append(dict(loss=float(loss)))
    for k, v in net.state_dict().items():
        arrays["final:" + k] = npf(v)
    save(name, arrays, meta)


def probunet_case(name, filters, latent_dim, batch, n_steps, seed):
    hw = 128
    net = ProbabilisticUnet(input_channels=1, num_classes=2, num_filters=filters, latent_dim=latent_dim,
                            no_convs_fcomb=3, image_size=(1, hw, hw))
    spec = kinds_for(net.state_dict())
    sd0 = deterministic_state_dict(spec, seed=seed)
    net.load_state_dict(sd0)
    net.train()
    opt = torch.optim.Adam(net.parameters(), lr=1e-3, weight_decay=1e-5)
    arrays, meta = {}, dict(model="ProbabilisticUnet", filters=filters, latent_dim=latent_dim, hw=hw, batch=batch,
                            weight_seed=seed, spec=[[k, list(s), kd] for k, s, kd in spec], steps=[])
    for step in range(n_steps):
        x, mask, eps = synthetic_batch(batch, hw, hw, seed=20201004 + step, eps_shapes=[(batch, latent_dim)])
        xt, mt = torch.from_numpy(x), torch.from_numpy(mask)
        last = net.forward(xt, mt, training=True)
        with NoiseFeeder(eps) as nf:
            loss = net.loss(mt)
            assert nf.used == 1
        opt.zero_grad()
        loss.backward()
        g = grads_of(net)
        if step == 0:
            arrays["x"], arrays["mask"], arrays["eps0"] = x, mask, eps[0]
            arrays["last_conv"] = npf(last)
            arrays["unet_features"] = npf(net.unet_features)
            arrays["reconstruction"] = npf(net.reconstruction)
            arrays["post_mu"] = npf(net.posterior_latent_space.mean)
            arrays["post_sigma"] = npf(net.posterior_latent_space.stddev)
            arrays["prior_mu"] = npf(net.prior_latent_space.mean)
            arrays["prior_sigma"] = npf(net.prior_latent_space.stddev)
            for k, v in g.items():
                if v is not None:
                    arrays["grad:" + k] = v
        opt.step()
        meta["steps"].append(dict(loss=float(loss), kl=float(net.kl_divergence_loss),
                                  recon=float(net.reconstruction_loss),
                                  none_grads=[k for k, v in g.items() if v is None]))
    for k, v in net.state_dict().items():
        if v.dtype.is_floating_point:
            arrays["final:" + k] = npf(v)
    save(name, arrays, meta)



def _grad_digest(g, rs):
    """Per-tensor gradient L2 norms + 8 sampled entries each (digest form for the full-size cases)."""
    gn, gs = {}, {}
    for k, v in g.items():
        if v is not None:
            gn[k] = float(np.sqrt((v.astype(np.float64) ** 2).sum()))
            flat = v.reshape(-1)
            pick = rs.integers(0, flat.size, size=min(8, flat.size))
            gs[k] = [pick.tolist(), flat[pick].astype(float).tolist()]
    return gn, gs


def unet_digest_case(name, filters, batch, seed):
    """BASELINE config 2 (Unet(1,2,[32,64,128,192]), batch 32) in digest form: sampled logits, loss, per-tensor
    gradient norms + sampled entries, packed argmax bits and the minimum logit margin (unet.py:78-165)."""
    hw = 128
    net = Unet(1, 2, filters)
    spec = kinds_for(net.state_dict())
    net.load_state_dict(deterministic_state_dict(spec, seed=seed))
    net.train()
    rs = np.random.Generator(np.random.PCG64(7))
    x, mask, _ = synthetic_batch(batch, hw, hw, seed=20201004)
    pred = net.forward(torch.from_numpy(x))
    loss = net.loss(torch.from_numpy(mask))
    loss.backward()
    g = grads_of(net)
    idx = rs.integers(0, batch * 2 * hw * hw, size=1024)
    p = npf(pred)
    # near-ties are unavoidable over 524 288 pixels of a randomly initialised net: the bit-exact argmax gate applies to
    # the pixels whose logit margin exceeds 2x the 1e-4 logit tolerance; the rest are counted and reported
    conf = np.abs(p[:, 1] - p[:, 0]) > 2e-4
    arrays = dict(s_idx=idx, pred_samp=p.reshape(-1)[idx],
                  argmax_bits=np.packbits(np.argmax(p, axis=1).astype(np.uint8).reshape(-1)),
                  argmax_conf_bits=np.packbits(conf.astype(np.uint8).reshape(-1)))
    gn, gs = _grad_digest(g, rs)
    meta = dict(model="Unet", filters=filters, hw=hw, batch=batch, weight_seed=seed,
                spec=[[k, list(s), kd] for k, s, kd in spec],
                steps=[dict(loss=float(loss), grad_norms=gn, grad_samples=gs,
                            none_grads=[k for k, v in g.items() if v is None])],
                margin_min=float(np.abs(p[:, 1] - p[:, 0]).min()), n_near_ties=int((~conf).sum()))
    save(name, arrays, meta)


def probunet_digest_case(name, filters, latent_dim, batch, seed, n_decode=8):
    """BASELINE config 3 (ProbabilisticUnet(1,2,[32,64,128,192,192,192,192], latent_dim=6, no_convs_fcomb=3), batch 32)
    in digest form, plus `n_decode` posterior-sample decodes `reconstruct(calculate_posterior=True)`
    (probabilistic_unet.py:272-283) in eval mode with recorded eps."""
    hw = 128
    net = ProbabilisticUnet(input_channels=1, num_classes=2, num_filters=filters, latent_dim=latent_dim,
                            no_convs_fcomb=3, image_size=(1, hw, hw))
    spec = kinds_for(net.state_dict())
    sd0 = deterministic_state_dict(spec, seed=seed)
    net.load_state_dict(sd0)
    net.train()
    rs = np.random.Generator(np.random.PCG64(7))
    x, mask, eps = synthetic_batch(batch, hw, hw, seed=20201004, eps_shapes=[(batch, latent_dim)] * (1 + n_decode))
    xt, mt = torch.from_numpy(x), torch.from_numpy(mask)
    last = net.forward(xt, mt, training=True)
    with NoiseFeeder(eps[:1]) as nf:
        loss = net.loss(mt)
        assert nf.used == 1
    loss.backward()
    g = grads_of(net)
    idx = rs.integers(0, batch * 2 * hw * hw, size=1024)
    fidx = rs.integers(0, batch * 32 * hw * hw, size=1024)
    arrays = dict(s_idx=idx, f_idx=fidx,
                  last_conv_samp=npf(last).reshape(-1)[idx], features_samp=npf(net.unet_features).reshape(-1)[fidx],
                  reconstruction_samp=npf(net.reconstruction).reshape(-1)[idx],
                  post_mu=npf(net.posterior_latent_space.mean), post_sigma=npf(net.posterior_latent_space.stddev),
                  prior_mu=npf(net.prior_latent_space.mean), prior_sigma=npf(net.prior_latent_space.stddev))
    gn, gs = _grad_digest(g, rs)
    meta = dict(model="ProbabilisticUnet", filters=filters, latent_dim=latent_dim, hw=hw, batch=batch, weight_seed=seed,
                n_decode=n_decode, spec=[[k, list(s), kd] for k, s, kd in spec],
                steps=[dict(loss=float(loss), kl=float(net.kl_divergence_loss), recon=float(net.reconstruction_loss),
                            grad_norms=gn, grad_samples=gs, none_grads=[k for k, v in g.items() if v is None])])
    # eval-mode decode of n_decode posterior samples on the cached U-Net features
    net.load_state_dict(sd0)
    net.eval()
    margins = []
    with torch.no_grad():
        net.forward(xt, mt, training=False)
        arrays["eval_post_mu"] = npf(net.posterior_latent_space.mean)
        arrays["eval_post_sigma"] = npf(net.posterior_latent_space.stddev)
        for j in range(n_decode):
            with NoiseFeeder([eps[1 + j]]) as nf:
                rec = net.reconstruct(use_posterior_mean=False, calculate_posterior=True)
                assert nf.used == 1
            r = npf(rec)
            arrays[f"dec{j}_samp"] = r.reshape(-1)[idx]
            arrays[f"dec{j}_argmax_bits"] = np.packbits(np.argmax(r, axis=1).astype(np.uint8).reshape(-1))
            margins.append(float(np.abs(r[:, 1] - r[:, 0]).min()))
    meta["decode_margin_min"] = margins
    save(name, arrays, meta)


def init_moment_cases():
    """Per-tensor moments of the three initialisers as the REAL reference applies them at construction
    (utils.init_weights utils.py:78-83 via unet.py:37 / probabilistic_unet.py:66; init_weights_orthogonal_normal
    utils.py:86-90 via probabilistic_unet.py:165-170; kaiming-normal + normal bias probabilistic_unet.py:99-100;
    PHiSeg keeps torch's default Conv2d init since phiseg.py:36 is commented out)."""
    meta = {}
    torch.manual_seed(0)
    nets = dict(unet=Unet(1, 2, [32, 64, 128, 192]),
                phiseg=PHISeg(1, 2, [32, 64, 128, 192, 192, 192, 192], latent_levels=5, image_size=(1, 128, 128)),
                probunet=ProbabilisticUnet(1, 2, [32, 64, 128, 192, 192, 192, 192], latent_dim=6, no_convs_fcomb=3,
                                           image_size=(1, 128, 128)))
    for name, net in nets.items():
        m = {}
        for k, v in net.state_dict().items():
            if not v.dtype.is_floating_point:
                continue
            a = v.detach().double()
            e = dict(n=int(a.numel()), mean=float(a.mean()), std=float(a.std(unbiased=False)), absmax=float(a.abs().max()))
            if k.startswith("fcomb.") and a.dim() == 4:
                w = a.reshape(a.shape[0], -1)
                g = w @ w.t() if w.shape[0] <= w.shape[1] else w.t() @ w
                e["orth_err"] = float((g - torch.eye(g.shape[0], dtype=torch.float64)).abs().max())
            m[k] = e
        meta[name] = m
    os.makedirs(OUT, exist_ok=True)
    with open(os.path.join(OUT, "init_moments.json"), "w") as f:
        json.dump(meta, f, indent=0)
    print("wrote init_moments:", {k: len(v) for k, v in meta.items()})


def phiseg_f64_case(name, filters, hw, batch, seed, n_samp=256):
    """fp64 ground truth of the REAL reference on the headline configuration: the reference model run in double precision
    (net.double()) next to its own fp32 run, same weights / inputs / noise.  Stored per parameter tensor: up to `n_samp`
    sampled gradient entries in fp64 and in the reference's fp32, so that a test can measure any implementation's error
    against the real-valued graph and compare it with the reference's OWN fp32 error on the same entries."""
    shapes = phiseg_eps_shapes(batch, hw, hw)
    x, mask, eps = synthetic_batch(batch, hw, hw, seed=20201004, eps_shapes=shapes + shapes)
    rs = np.random.Generator(np.random.PCG64(11))
    res = {}
    for dt in (torch.float32, torch.float64):
        net = PHISeg(input_channels=1, num_classes=2, num_filters=filters, latent_levels=5, image_size=(1, hw, hw))
        spec = kinds_for(net.state_dict())
        net.load_state_dict(deterministic_state_dict(spec, seed=seed))
        net = net.to(dt)
        net.train()
        with NoiseFeeder([torch.from_numpy(e).to(dt) for e in eps]):
            s_list = net.forward(torch.from_numpy(x).to(dt), torch.from_numpy(mask).to(dt), training=True)
        loss = net.loss(torch.from_numpy(mask).to(dt))
        loss.backward()
        res[dt] = (float(loss), [t.detach() for t in s_list], {n: (None if p.grad is None else p.grad.detach()) for n, p in net.named_parameters()})
        del net
    arrays, meta = {}, dict(model="PHISeg", filters=filters, hw=hw, batch=batch, weight_seed=seed,
                            spec=[[k, list(s), kd] for k, s, kd in spec], loss32=res[torch.float32][0], loss64=res[torch.float64][0])
    idx = rs.integers(0, batch * 2 * hw * hw, size=1024)
    arrays["s_idx"] = idx
    for l in range(5):
        arrays[f"s{l}_f64"] = res[torch.float64][1][l].numpy().reshape(-1)[idx]
        arrays[f"s{l}_f32"] = res[torch.float32][1][l].numpy().reshape(-1)[idx]
    for k, g64 in res[torch.float64][2].items():
        if g64 is None:
            continue
        g32 = res[torch.float32][2][k]
        flat = g64.reshape(-1)
        pick = np.arange(flat.numel()) if flat.numel() <= n_samp else np.sort(rs.choice(flat.numel(), size=n_samp, replace=False))
        arrays["i:" + k] = pick.astype(np.int64)
        arrays["g64:" + k] = flat.numpy()[pick]
        arrays["g32:" + k] = g32.reshape(-1).numpy()[pick]
        arrays["m64:" + k] = np.array([float(g64.abs().max()), float(g64.double().norm())])
    save(name, arrays, meta)


def phiseg3d_case(name, in_ch, num_classes, filters, latent_levels, dhw, seed):
    """models/phiseg3D.py, the parts the reference can execute (see oracle/refgraph3d.py): Posterior, prior (training_prior)
    and Likelihood modules of a real PHISeg3D instance.  Likelihood.forward's last statement (:398) raises on 5-D input; for
    the duration of that call `interpolate(x, size=<2 elements>)` is answered with the nearest resize to the full volume (the
    evident intent), everything else is the reference's own code - including its loss functions, fed through the attributes
    PHISeg3D.forward would have set (:456-463), and autograd through its modules."""
    from models.phiseg3D import PHISeg3D as Ref3D
    import torch.nn.functional as TF
    from oracle.refgraph3d import phiseg3d_eps_shapes, synthetic_volume
    D, H, W = dhw
    net = Ref3D(input_channels=in_ch, num_classes=num_classes, num_filters=filters, latent_levels=latent_levels,
                image_size=(in_ch, D, H, W))
    spec = kinds_for(net.state_dict())
    net.load_state_dict(deterministic_state_dict(spec, seed=seed))
    net.train()
    R, L = len(filters), latent_levels
    shapes = phiseg3d_eps_shapes(D, H, W, R, L)
    x, onehot, lab, eps = synthetic_volume(in_ch, num_classes, dhw, 20201005, shapes + shapes)
    xt, ot, lt = torch.from_numpy(x), torch.from_numpy(onehot), torch.from_numpy(lab)
    s_in = {}
    hooks = [blk.register_forward_hook(lambda m, i, o, k=k: s_in.__setitem__(k, o)) for k, blk in enumerate(net.likelihood.s_layer)]
    orig = TF.interpolate

    def resize(inp, size=None, *a, **kw):
        if size is not None and inp.dim() == 5 and len(size) == 2:
            return orig(inp, size=[D, H, W], mode="nearest")
        return orig(inp, size, *a, **kw)
    with NoiseFeeder([torch.from_numpy(e) for e in eps]):
        pz, pmu, psig = net.posterior(xt, ot)
        qz, qmu, qsig = net.prior(xt, training_prior=True, z_list=pz)
    TF.interpolate = resize
    try:
        s = net.likelihood(pz)
    finally:
        TF.interpolate = orig
    for h in hooks:
        h.remove()
    net.posterior_latent_space, net.posterior_mu, net.posterior_sigma = pz, pmu, psig
    net.prior_latent_space, net.prior_mu, net.prior_sigma = qz, qmu, qsig
    net.s_out_list = s
    loss = net.loss(lt)
    loss.backward()
    arrays = {"patch": x, "mask_onehot": onehot, "labels": lab}
    for k, e in enumerate(eps):
        arrays[f"eps{k}"] = e
    for l in range(L):
        arrays[f"post_mu{l}"], arrays[f"post_sigma{l}"], arrays[f"post_z{l}"] = npf(pmu[l]), npf(psig[l]), npf(pz[l])
        arrays[f"prior_mu{l}"], arrays[f"prior_sigma{l}"] = npf(qmu[l]), npf(qsig[l])
        arrays[f"s_in{l}"] = npf(s_in[L - 1 - l])            # s_layer[k] serves level L-1-k (:395-397)
    for k, v in net.loss_dict.items():
        arrays["loss:" + k] = np.float32(float(v))
    arrays["loss"] = np.float32(float(loss))
    for n_, g in grads_of(net).items():
        if g is not None:
            arrays["g:" + n_] = g
    sd1 = net.state_dict()
    for k, v in sd1.items():
        if "running_" in k:
            arrays["sd1:" + k] = npf(v)
    save(name, arrays, dict(model="PHISeg3D", input_channels=in_ch, num_classes=num_classes, filters=filters, latent_levels=L,
                            dhw=list(dhw), weight_seed=seed, spec=[[k, list(s_), kd] for k, s_, kd in spec],
                            no_grad=[n_ for n_, g in grads_of(net).items() if g is None]))


def phiseg3d_digest_case(name, in_ch, num_classes, filters, latent_levels, dhw, seed, vol_seed, n_samp=2048, backward=True):
    """BASELINE config 5 at its LITERAL size (4 x 128 x 128 x 64, filters 32 / 64 / 128 / 192 / 192) through the reference's own
    modules in fp32 on the CPU: digests only - losses, sampled level logits / s_in / posterior and prior moments,
    gradient norms and sampled gradient entries.  Inputs and weights are regenerated by the test from the same seeds
    (oracle.deterministic_state_dict(seed), oracle.refgraph3d.synthetic_volume(vol_seed)).  Same call protocol as phiseg3d_case."""
    from models.phiseg3D import PHISeg3D as Ref3D
    import torch.nn.functional as TF
    from oracle.refgraph3d import phiseg3d_eps_shapes, synthetic_volume
    D, H, W = dhw
    net = Ref3D(input_channels=in_ch, num_classes=num_classes, num_filters=filters, latent_levels=latent_levels,
                image_size=(in_ch, D, H, W))
    spec = kinds_for(net.state_dict())
    net.load_state_dict(deterministic_state_dict(spec, seed=seed))
    net.train()
    R, L = len(filters), latent_levels
    shapes = phiseg3d_eps_shapes(D, H, W, R, L)
    x, onehot, lab, eps = synthetic_volume(in_ch, num_classes, dhw, vol_seed, shapes + shapes)
    xt, ot, lt = torch.from_numpy(x), torch.from_numpy(onehot), torch.from_numpy(lab)
    s_in = {}
    hooks = [blk.register_forward_hook(lambda m, i, o, k=k: s_in.__setitem__(k, o)) for k, blk in enumerate(net.likelihood.s_layer)]
    orig = TF.interpolate

    def resize(inp, size=None, *a, **kw):
        if size is not None and inp.dim() == 5 and len(size) == 2:
            return orig(inp, size=[D, H, W], mode="nearest")
        return orig(inp, size, *a, **kw)
    ctx = torch.enable_grad() if backward else torch.no_grad()
    with ctx:
        with NoiseFeeder([torch.from_numpy(e) for e in eps]):
            pz, pmu, psig = net.posterior(xt, ot)
            qz, qmu, qsig = net.prior(xt, training_prior=True, z_list=pz)
        TF.interpolate = resize
        try:
            s = net.likelihood(pz)
        finally:
            TF.interpolate = orig
        for h in hooks:
            h.remove()
        net.posterior_latent_space, net.posterior_mu, net.posterior_sigma = pz, pmu, psig
        net.prior_latent_space, net.prior_mu, net.prior_sigma = qz, qmu, qsig
        net.s_out_list = s
        loss = net.loss(lt)
        if backward:
            loss.backward()
    rs = np.random.default_rng(seed + 7)
    arrays = {}

    def sample(key, t):
        flat = t.detach().reshape(-1)
        pick = np.arange(flat.numel()) if flat.numel() <= n_samp else np.sort(rs.choice(flat.numel(), size=n_samp, replace=False))
        arrays["i:" + key] = pick.astype(np.int64)
        arrays["v:" + key] = flat.numpy()[pick].astype(np.float32)
        arrays["m:" + key] = np.array([float(flat.abs().max()), float(flat.double().norm()), float(flat.double().mean())])
    for l in range(L):
        sample(f"s{l}", s[l]); sample(f"s_in{l}", s_in[L - 1 - l])
        sample(f"post_mu{l}", pmu[l]); sample(f"post_sigma{l}", psig[l]); sample(f"prior_mu{l}", qmu[l]); sample(f"prior_sigma{l}", qsig[l])
    for k, v in net.loss_dict.items():
        arrays["loss:" + k] = np.float32(float(v))
    arrays["loss"] = np.float32(float(loss))
    no_grad = []
    if backward:
        for n_, p_ in net.named_parameters():
            if p_.grad is None:
                no_grad.append(n_)
            else:
                sample("g:" + n_, p_.grad)
    save(name, arrays, dict(model="PHISeg3D", input_channels=in_ch, num_classes=num_classes, filters=filters, latent_levels=L,
                            dhw=list(dhw), weight_seed=seed, volume_seed=vol_seed, backward=bool(backward), no_grad=no_grad,
                            spec=[[k, list(s_), kd] for k, s_, kd in spec]))


def phiseg3d_bf16_case(name, in_ch, num_classes, filters, latent_levels, dhw, seed):
    """The reference's 3-D modules run IN BF16 the way PyTorch runs a model in bf16 with fp32 master weights:
    torch.autocast('cpu', dtype=torch.bfloat16) around Posterior / prior / Likelihood of a real PHISeg3D instance (Conv3d in bf16
    with bf16 outputs; whatever autocast keeps in fp32 stays fp32).  Fixture for the native bf16 ARITHMETIC mode
    (UZ_CONV_MATH=bf16: bf16 products, fp32 accumulation, fp32 storage): the two differ by the rounding of every conv OUTPUT to bf16
    on the reference side, so the gate is loose and stated in the test; the tight pin of that mode is the operand-rounding oracle.
    Digests only: every 7th element of the latent statistics and of the level logits."""
    from models.phiseg3D import PHISeg3D as Ref3D
    import torch.nn.functional as TF
    from oracle.refgraph3d import phiseg3d_eps_shapes, synthetic_volume
    D, H, W = dhw
    net = Ref3D(input_channels=in_ch, num_classes=num_classes, num_filters=filters, latent_levels=latent_levels, image_size=(in_ch, D, H, W))
    spec = kinds_for(net.state_dict())
    net.load_state_dict(deterministic_state_dict(spec, seed=seed))
    net.train()
    R, L = len(filters), latent_levels
    shapes = phiseg3d_eps_shapes(D, H, W, R, L)
    x, onehot, lab, eps = synthetic_volume(in_ch, num_classes, dhw, 20201006, shapes + shapes)
    xt, ot, lt = torch.from_numpy(x), torch.from_numpy(onehot), torch.from_numpy(lab)
    s_in = {}
    hooks = [blk.register_forward_hook(lambda m, i, o, k=k: s_in.__setitem__(k, o)) for k, blk in enumerate(net.likelihood.s_layer)]
    orig = TF.interpolate

    def resize(inp, size=None, *a, **kw):
        if size is not None and inp.dim() == 5 and len(size) == 2:
            return orig(inp, size=[D, H, W], mode="nearest")
        return orig(inp, size, *a, **kw)
    out = {}
    for mode in ("fp32", "bf16"):
        # fresh running statistics for each run (train-mode BatchNorm updates them)
        net.load_state_dict(deterministic_state_dict(spec, seed=seed))
        ctx = torch.autocast("cpu", dtype=torch.bfloat16) if mode == "bf16" else torch.autocast("cpu", enabled=False)
        with torch.no_grad(), ctx:
            with NoiseFeeder([torch.from_numpy(e) for e in eps]):
                pz, pmu, psig = net.posterior(xt, ot)
                qz, qmu, qsig = net.prior(xt, training_prior=True, z_list=pz)
            TF.interpolate = resize
            try:
                s = net.likelihood(pz)
            finally:
                TF.interpolate = orig
        net.posterior_latent_space, net.posterior_mu, net.posterior_sigma = [t.float() for t in pz], [t.float() for t in pmu], [t.float() for t in psig]
        net.prior_latent_space, net.prior_mu, net.prior_sigma = [t.float() for t in qz], [t.float() for t in qmu], [t.float() for t in qsig]
        net.s_out_list = [t.float() for t in s]
        with torch.no_grad():
            loss = net.loss(lt)
        out[mode] = dict(pmu=pmu, psig=psig, qmu=qmu, qsig=qsig, s_in=dict(s_in), loss=float(loss))
    for h in hooks:
        h.remove()
    arrays = {}
    for mode, o in out.items():
        for l in range(L):
            sub = lambda t: npf(t.float()).reshape(-1)[::7].copy()          # every 7th element, flat
            arrays[f"{mode}:post_mu{l}"], arrays[f"{mode}:post_sigma{l}"] = sub(o["pmu"][l]), sub(o["psig"][l])
            arrays[f"{mode}:prior_mu{l}"], arrays[f"{mode}:prior_sigma{l}"] = sub(o["qmu"][l]), sub(o["qsig"][l])
            arrays[f"{mode}:s_in{l}"] = sub(o["s_in"][L - 1 - l])
        arrays[f"{mode}:loss"] = np.float32(o["loss"])
    save(name, arrays, dict(model="PHISeg3D", input_channels=in_ch, num_classes=num_classes, filters=filters, latent_levels=L, dhw=list(dhw),
                            weight_seed=seed, input_seed=20201006, logit_stride=7, spec=[[k, list(s_), kd] for k, s_, kd in spec],
                            how="reference Posterior / prior / Likelihood under torch.autocast('cpu', torch.bfloat16) and in fp32, no_grad"))


def batch_provider_stream():
    """Index / annotator stream of the REAL reference BatchProvider.next_batch (data/batch_provider.py:43-67,131-137) under a
    fixed numpy seed: pins the native provider's sampling logic and its order of RNG draws.  (The augmentation draws cannot be
    generated: the reference's _augmentation_function returns False without cv2, which this image lacks.)"""
    from data.batch_provider import BatchProvider
    N, A = 23, 4
    X = np.arange(N, dtype=np.float32)[:, None, None] * np.ones((1, 4, 4), np.float32)
    y = np.zeros((N, 4, 4, A), np.uint8)
    for a in range(A):
        y[..., a] = a
    np.random.seed(1234)
    bp = BatchProvider(X, y, np.arange(N), add_dummy_dimension=True, num_labels_per_subject=A, annotator_range=range(A))
    rec = []
    for _ in range(9):
        xb, yb = bp.next_batch(5)
        rec.append(dict(idx=[int(v) for v in xb[:, 0, 0, 0]], ann=[int(v) for v in yb[:, 0, 0]]))
    with open(os.path.join(OUT, "batch_provider_stream.json"), "w") as f:
        json.dump(dict(N=N, A=A, seed=1234, batch=5, batches=rec), f)
    print("wrote batch_provider_stream")


def op_cases():
    """G1: reference-authored arithmetic that is not a stock torch op."""
    rs = np.random.Generator(np.random.PCG64(99))
    arrays, meta = {}, {}
    net = PHISeg(1, 2, [4, 8, 8, 8, 8, 8, 8], image_size=(1, 64, 64))
    # KL with the sigma1*sigma0 quirk, incl. tiny sigmas (log(... + 1e-10) edge)
    for i, scale in enumerate([1.0, 1e-3, 1e-6]):
        mu0, mu1 = (rs.standard_normal((3, 2, 4, 4)).astype(np.float32) for _ in range(2))
        s0, s1 = (np.abs(rs.standard_normal((3, 2, 4, 4))).astype(np.float32) * scale + 1e-7 for _ in range(2))
        t = [torch.tensor(a, requires_grad=True) for a in (mu0, s0, mu1, s1)]
        kl = net.KL_two_gauss_with_diag_cov(*t)
        kl.backward()
        for nm, a, tt in zip(("mu0", "s0", "mu1", "s1"), (mu0, s0, mu1, s1), t):
            arrays[f"kl{i}_{nm}"] = a
            arrays[f"kl{i}_d{nm}"] = npf(tt.grad)
        meta[f"kl{i}"] = float(kl)
    # one-hot of a batch of label maps (utils.py:289-311)
    lab = (rs.uniform(size=(3, 1, 8, 8)) > 0.5).astype(np.float32)
    arrays["onehot_in"] = lab
    arrays["onehot_out"] = ref_utils.convert_batch_to_onehot(torch.from_numpy(lab), nlabels=2).numpy().astype(np.int64)
    # l2_regularisation (utils.py:93-101)
    lin = torch.nn.Sequential(torch.nn.Conv2d(3, 4, 3), torch.nn.BatchNorm2d(4))
    arrays["l2_w0"], arrays["l2_b0"] = npf(lin[0].weight), npf(lin[0].bias)
    arrays["l2_w1"], arrays["l2_b1"] = npf(lin[1].weight), npf(lin[1].bias)
    meta["l2"] = float(ref_utils.l2_regularisation(lin))
    # residual multinoulli loss on cumulative logits (phiseg.py:481-513)
    s = [rs.standard_normal((2, 2, 8, 8)).astype(np.float32) for _ in range(5)]
    tgt = (rs.uniform(size=(2, 1, 8, 8)) > 0.5).astype(np.float32)
    net.loss_tot = 0
    st = [torch.tensor(a, requires_grad=True) for a in s]
    tot = net.residual_multinoulli_loss(st, torch.from_numpy(tgt))
    tot.backward()
    for i in range(5):
        arrays[f"rm_s{i}"], arrays[f"rm_ds{i}"] = s[i], npf(st[i].grad)
        meta[f"rm_lvl{i}"] = float(net.loss_dict["residual_multinoulli_loss_lvl%d" % i])
    arrays["rm_target"] = tgt
    meta["rm_total"] = float(tot)
    save("ops", arrays, meta)


def metric_cases():
    """Golden values of the reference's validation metrics (utils.py:148-247) on synthetic samples."""
    rs = np.random.Generator(np.random.PCG64(5))
    arrays, meta = {}, {}
    H = W = 32
    yy, xx = np.mgrid[0:H, 0:W]
    def disc(cy, cx, r):
        return ((yy - cy) ** 2 + (xx - cx) ** 2 <= r * r).astype(np.int64)
    for case in range(3):
        N, M = (6, 4) if case < 2 else (4, 3)
        samples = np.stack([disc(16 + rs.normal(0, 2), 16 + rs.normal(0, 2), 6 + rs.normal(0, 1.5)) for _ in range(N)])
        gts = np.stack([disc(16 + rs.normal(0, 2), 15 + rs.normal(0, 2), 6 + rs.normal(0, 1.5)) for _ in range(M)])
        if case == 1:
            samples[0] = 0          # empty prediction vs non-empty gt, and
            gts[1] = 0              # empty gt: exercises the 0 / 1 conventions
        logits = rs.standard_normal((N, 2, H, W)).astype(np.float32) + 3.0 * np.stack([1 - samples, samples], 1).astype(np.float32)
        soft = torch.softmax(torch.from_numpy(logits), dim=1)
        ged = ref_utils.generalised_energy_distance(torch.from_numpy(samples), torch.from_numpy(gts), nlabels=1, label_range=range(1, 2))
        onehot = ref_utils.convert_batch_to_onehot(torch.from_numpy(gts).unsqueeze(1), nlabels=2)
        ncc = ref_utils.variance_ncc_dist(soft, onehot)
        arrays[f"m{case}_samples"], arrays[f"m{case}_gts"], arrays[f"m{case}_soft"] = samples.astype(np.uint8), gts.astype(np.uint8), soft.numpy()
        meta[f"m{case}_ged"], meta[f"m{case}_ncc"] = float(ged), float(np.asarray(ncc).reshape(-1)[0])
    save("metrics", arrays, meta)


if __name__ == "__main__":
    torch.manual_seed(0)
    if len(sys.argv) > 1 and sys.argv[1] == "metrics":
        metric_cases()
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "b32":
        # BASELINE config 4 exactly (batch 32): digests only, inputs are regenerated from the seed
        phiseg_case("phiseg_full_b32_digest", [32, 64, 128, 192, 192, 192, 192], 128, 32, 1, False, 1238, store_inputs=False)
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "b32traj":
        # BASELINE config 4, three training steps: losses, post-step parameter / buffer digests
        phiseg_traj_case("phiseg_full_b32_traj", [32, 64, 128, 192, 192, 192, 192], 128, 32, 3, 1238)
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "3d":
        phiseg3d_case("phiseg3d_small", 2, 3, [4, 8, 8], 2, (16, 16, 8), 1242)       # lvl_diff 1
        phiseg3d_case("phiseg3d_l3", 4, 3, [8, 8, 16], 3, (8, 16, 16), 1243)         # one latent level per resolution level
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "3d_full":
        # BASELINE config 5 at its literal size: digests of the reference's fp32 CPU run (minutes, ~40 GB of host memory with the backward)
        phiseg3d_digest_case("phiseg3d_full_digest", 4, 3, [32, 64, 128, 192, 192], 5, (128, 128, 64), 11, 9, backward=(len(sys.argv) < 3 or sys.argv[2] != "fwd"))
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "3d_bf16":
        # large enough for the library to route its convolutions to the matrix-pipe kernels (32+ channels on 64-wide planes)
        phiseg3d_bf16_case("phiseg3d_bf16", 4, 3, [32, 32, 64], 3, (32, 64, 64), 1244)
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "batches":
        batch_provider_stream()
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "f64":
        b = int(sys.argv[2]) if len(sys.argv) > 2 else 32
        phiseg_f64_case("phiseg_full_b%d_f64" % b, [32, 64, 128, 192, 192, 192, 192], 128, b, 1238)
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "init":
        init_moment_cases()
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "unet_b32":
        # BASELINE config 2 exactly
        unet_digest_case("unet_full_b32_digest", [32, 64, 128, 192], 32, 1240)
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "probunet_b32":
        # BASELINE config 3 exactly (+ 8 posterior-sample decodes)
        probunet_digest_case("probunet_full_b32_digest", [32, 64, 128, 192, 192, 192, 192], 6, 32, 1241)
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "mid":
        # better-conditioned multi-step trajectory (16+ samples per BN channel at the deepest level)
        phiseg_case("phiseg_mid", [8, 16, 16, 16, 16, 16, 16], 128, 4, 3, True, 1239)
        sys.exit(0)
    op_cases()
    phiseg_case("phiseg_small", [4, 8, 8, 8, 8, 8, 8], 64, 2, 3, True, 1234)
    unet_case("unet_small", [4, 8, 8, 8], 2, 3, 1235)
    probunet_case("probunet_small", [32, 8, 8, 8, 8, 8, 8], 6, 2, 3, 1236)
    phiseg_case("phiseg_full_digest", [32, 64, 128, 192, 192, 192, 192], 128, 2, 1, False, 1237)
