"""Pin the CPU oracle (oracle/refgraph.py) against golden vectors produced by the real
reference (tools/gen_golden.py).  CPU only.  Tolerances: the oracle and the reference run
the same ATen ops in the same order, so agreement is expected to ~1e-6; gates are stated
per check."""
import numpy as np
import pytest
import torch

import oracle
from oracle import refgraph as R
from tests import _golden as G

torch.set_num_threads(4)


def _t(a):
    return torch.from_numpy(np.asarray(a))


# ----------------------------------------------------------------------------- ops (G1)
def test_kl_quirk_matches_reference():
    arrays, meta = G.load("ops")
    for i in range(3):
        t = [_t(arrays[f"kl{i}_{n}"]).clone().requires_grad_(True) for n in ("mu0", "s0", "mu1", "s1")]
        kl = oracle.kl_two_gauss_with_diag_cov(*t)
        kl.backward()
        assert abs(float(kl) - meta[f"kl{i}"]) <= 1e-6 * max(1.0, abs(meta[f"kl{i}"]))
        for n, tt in zip(("mu0", "s0", "mu1", "s1"), t):
            ref = arrays[f"kl{i}_d{n}"]
            assert G.maxabs(tt.grad.numpy(), ref) <= 1e-6 * max(1.0, float(np.abs(ref).max()))


def test_onehot_matches_reference():
    arrays, _ = G.load("ops")
    out = oracle.batch_to_onehot(_t(arrays["onehot_in"]), 2).numpy()
    assert out.dtype == np.int64
    assert np.array_equal(out, arrays["onehot_out"])          # integer path: bit exact


def test_l2_regularisation_matches_reference():
    arrays, meta = G.load("ops")
    sd = {"m.0.weight": _t(arrays["l2_w0"]), "m.0.bias": _t(arrays["l2_b0"]),
          "m.1.weight": _t(arrays["l2_w1"]), "m.1.bias": _t(arrays["l2_b1"])}
    assert abs(float(R._l2_regularisation(sd, "m.")) - meta["l2"]) <= 1e-6 * meta["l2"]


def test_residual_multinoulli_matches_reference():
    arrays, meta = G.load("ops")
    s = [_t(arrays[f"rm_s{i}"]).clone().requires_grad_(True) for i in range(5)]
    tgt = _t(arrays["rm_target"])
    dummy = [torch.zeros(1, 1)] * 5
    out = dict(s=s, posterior_mu=dummy, posterior_sigma=[torch.ones(1, 1)] * 5, prior_mu=dummy,
               prior_sigma=[torch.ones(1, 1)] * 5)
    total, terms = oracle.phiseg_loss(out, tgt)
    total.backward()
    for i in range(5):
        assert abs(float(terms["residual_multinoulli_loss_lvl%d" % i]) - meta[f"rm_lvl{i}"]) <= 1e-5 * abs(meta[f"rm_lvl{i}"])
        assert G.maxabs(s[i].grad.numpy(), arrays[f"rm_ds{i}"]) <= 1e-6


# ----------------------------------------------------------------------------- PHiSeg small (G2)
def _phiseg_inputs(arrays, batch, hw, step):
    shapes = oracle.phiseg_eps_shapes(batch, hw, hw)
    x, mask, eps = oracle.synthetic_batch(batch, hw, hw, seed=20201004 + step, eps_shapes=shapes + shapes)
    if step == 0 and "x" in arrays:
        assert np.array_equal(x, arrays["x"]) and np.array_equal(mask, arrays["mask"])
        for i in range(10):
            assert np.array_equal(eps[i], arrays[f"eps{i}"])
    e = [_t(a) for a in eps]
    return _t(x), _t(mask), dict(posterior=e[:5], prior=e[5:])


def test_phiseg_small_forward_loss_grads_adam():
    arrays, meta = G.load("phiseg_small")
    spec = G.spec_of(meta)
    sd = G.leaves(oracle.deterministic_state_dict(spec, seed=meta["weight_seed"]))
    state = {}
    for step, st in enumerate(meta["steps"]):
        x, mask, eps = _phiseg_inputs(arrays, meta["batch"], meta["hw"], step)
        out = oracle.phiseg_forward(sd, x, mask, eps, training=True, bn_train=True)
        total, terms = oracle.phiseg_loss(out, mask)
        for k in [k for k, v in sd.items() if v.requires_grad]:
            sd[k].grad = None
        total.backward()
        assert abs(float(total) - st["loss"]) <= 2e-6 * abs(st["loss"])
        for k, v in st["loss_dict"].items():
            assert abs(float(terms[k]) - v) <= 1e-5 * max(1.0, abs(v)), k
        # kl_divergence_loss / reconstruction_loss alias the total
        assert st["kl_alias"] == st["loss"] and st["recon_alias"] == st["loss"]
        none = sorted(k for k, v in sd.items() if v.requires_grad and v.grad is None)
        assert none == sorted(st["none_grads"]) and len(none) == 16
        if step == 0:
            for l in range(5):
                assert G.maxabs(out["s"][l].detach().numpy(), arrays[f"s{l}"]) <= 1e-5
                assert G.maxabs(out["posterior_mu"][l].detach().numpy(), arrays[f"post_mu{l}"]) <= 1e-5
                assert G.maxabs(out["posterior_sigma"][l].detach().numpy(), arrays[f"post_sigma{l}"]) <= 1e-5
                assert G.maxabs(out["posterior_z"][l].detach().numpy(), arrays[f"post_z{l}"]) <= 1e-5
                assert G.maxabs(out["prior_mu"][l].detach().numpy(), arrays[f"prior_mu{l}"]) <= 1e-5
            worst = 0.0
            noise = G.bn_shadowed_biases(sd.keys())
            for k, v in sd.items():
                if v.requires_grad and v.grad is not None and k not in noise:
                    ref = arrays["grad:" + k]
                    worst = max(worst, G.maxabs(v.grad.numpy(), ref) / (1e-3 + float(np.abs(ref).max())))
            assert worst <= 2e-3, worst
            for k, v in sd.items():
                if "running_" in k:
                    assert G.maxabs(v.numpy(), arrays["buf1:" + k]) <= 1e-6
        params = {k: v for k, v in sd.items() if v.requires_grad}
        grads = {k: v.grad for k, v in params.items()}
        new = oracle.adam_reference_step(params, grads, state)
        for k, v in new.items():
            sd[k] = v.requires_grad_(True)


def test_phiseg_small_eval_argmax_bit_exact():
    arrays, meta = G.load("phiseg_small")
    sd = oracle.deterministic_state_dict(G.spec_of(meta), seed=meta["weight_seed"])
    x, mask, eps = _phiseg_inputs(arrays, meta["batch"], meta["hw"], 0)
    with torch.no_grad():
        out = oracle.phiseg_forward(sd, x, mask, eps, training=False, bn_train=False)
        soft = oracle.phiseg_accumulate_output(out["s"], use_softmax=True)
    for l in range(5):
        assert G.maxabs(out["s"][l].numpy(), arrays[f"eval_s{l}"]) <= 1e-5
    assert G.maxabs(soft.numpy(), arrays["eval_softmax"]) <= 1e-6
    bits = np.packbits(torch.argmax(soft, dim=1).numpy().astype(np.uint8).reshape(-1))
    assert meta["eval_margin_min"] > 1e-4, "fixture has a near-tie; argmax gate would be meaningless"
    assert np.array_equal(bits, arrays["eval_argmax_bits"])
    assert meta["eval_alias_inplace"] is True


# ----------------------------------------------------------------------------- PHiSeg full-size digest (G3)
@pytest.mark.parametrize("fixture", ["phiseg_full_digest", "phiseg_full_b32_digest"])
def test_phiseg_full_digest(fixture):
    arrays, meta = G.load(fixture)
    sd = G.leaves(oracle.deterministic_state_dict(G.spec_of(meta), seed=meta["weight_seed"]))
    x, mask, eps = _phiseg_inputs(arrays, meta["batch"], meta["hw"], 0)
    out = oracle.phiseg_forward(sd, x, mask, eps, training=True, bn_train=True)
    total, terms = oracle.phiseg_loss(out, mask)
    total.backward()
    st = meta["steps"][0]
    assert abs(float(total) - st["loss"]) <= 1e-5 * abs(st["loss"])
    idx = arrays["s_idx"]
    for l in range(5):
        assert G.maxabs(out["s"][l].detach().numpy().reshape(-1)[idx], arrays[f"s{l}_samp"]) <= 1e-4
        assert G.maxabs(out["posterior_mu"][l].detach().numpy(), arrays[f"post_mu{l}"]) <= 1e-4
    noise = G.bn_shadowed_biases(sd.keys())
    for k, n in st["grad_norms"].items():
        if k in noise:
            continue
        mine = float(sd[k].grad.double().norm())
        assert abs(mine - n) <= 2e-3 * max(n, 1e-3), (k, mine, n)


# ----------------------------------------------------------------------------- U-Net small
def test_unet_small():
    arrays, meta = G.load("unet_small")
    sd = G.leaves(oracle.deterministic_state_dict(G.spec_of(meta), seed=meta["weight_seed"]))
    state = {}
    for step, st in enumerate(meta["steps"]):
        x, mask, _ = oracle.synthetic_batch(meta["batch"], 128, 128, seed=20201004 + step)
        pred = oracle.unet_forward(sd, _t(x))
        loss = oracle.unet_loss(pred, _t(mask))
        for v in sd.values():
            v.grad = None
        loss.backward()
        assert abs(float(loss) - st["loss"]) <= 1e-5 * abs(st["loss"])
        if step == 0:
            assert G.maxabs(pred.detach().numpy(), arrays["pred"]) <= 1e-5
            for k, v in sd.items():
                ref = arrays["grad:" + k]
                assert G.maxabs(v.grad.numpy(), ref) <= 1e-3 * (1e-4 + float(np.abs(ref).max())), k
        new = oracle.adam_reference_step(dict(sd), {k: v.grad for k, v in sd.items()}, state)
        sd = {k: v.requires_grad_(True) for k, v in new.items()}
    for k, v in sd.items():
        assert G.maxabs(v.detach().numpy(), arrays["final:" + k]) <= 2e-4, k


# ----------------------------------------------------------------------------- Probabilistic U-Net small
def test_probunet_small():
    arrays, meta = G.load("probunet_small")
    sd = G.leaves(oracle.deterministic_state_dict(G.spec_of(meta), seed=meta["weight_seed"]))
    st = meta["steps"][0]
    x, mask, eps = oracle.synthetic_batch(meta["batch"], 128, 128, seed=20201004,
                                          eps_shapes=[(meta["batch"], meta["latent_dim"])])
    assert np.array_equal(eps[0], arrays["eps0"])
    out = oracle.probunet_forward(sd, _t(x), _t(mask), bn_train=True)
    loss, aux = oracle.probunet_loss(sd, out, _t(mask), _t(eps[0]), bn_train=True)
    loss.backward()
    assert abs(float(loss) - st["loss"]) <= 1e-5 * abs(st["loss"])
    assert abs(float(aux["kl"]) - st["kl"]) <= 1e-4 * max(1.0, abs(st["kl"]))
    assert G.maxabs(out["unet_features"].detach().numpy(), arrays["unet_features"]) <= 1e-4
    assert G.maxabs(out["last_conv"].detach().numpy(), arrays["last_conv"]) <= 1e-4
    assert G.maxabs(aux["reconstruction"].detach().numpy(), arrays["reconstruction"]) <= 1e-4
    assert G.maxabs(out["posterior_mu"].detach().numpy(), arrays["post_mu"]) <= 1e-5
    assert G.maxabs(out["prior_sigma"].detach().numpy(), arrays["prior_sigma"]) <= 1e-5
    none = sorted(k for k, v in sd.items() if v.requires_grad and v.grad is None)
    assert none == sorted(st["none_grads"])
    noise = G.bn_shadowed_biases(sd.keys())
    for k, v in sd.items():
        if v.requires_grad and v.grad is not None and k not in noise:
            ref = arrays["grad:" + k]
            assert G.maxabs(v.grad.numpy(), ref) <= 2e-3 * (1e-3 + float(np.abs(ref).max())), k


# ----------------------------------------------------------------------------- BASELINE configs 2 and 3 at full size (digests)
def _check_grad_digest(sd, st, rel=2e-3):
    noise = G.bn_shadowed_biases(sd.keys())
    for k, n in st["grad_norms"].items():
        if k in noise:
            continue
        mine = float(sd[k].grad.double().norm())
        assert abs(mine - n) <= rel * max(n, 1e-3), (k, mine, n)


def test_unet_full_b32_digest():
    """BASELINE config 2: Unet(1,2,[32,64,128,192]), batch 32 (unet.py:78-165)."""
    arrays, meta = G.load("unet_full_b32_digest")
    sd = G.leaves(oracle.deterministic_state_dict(G.spec_of(meta), seed=meta["weight_seed"]))
    x, mask, _ = oracle.synthetic_batch(meta["batch"], 128, 128, seed=20201004)
    pred = oracle.unet_forward(sd, _t(x))
    loss = oracle.unet_loss(pred, _t(mask))
    loss.backward()
    st = meta["steps"][0]
    assert abs(float(loss) - st["loss"]) <= 1e-5 * abs(st["loss"])
    p = pred.detach().numpy()
    assert G.maxabs(p.reshape(-1)[arrays["s_idx"]], arrays["pred_samp"]) <= 1e-5
    _check_grad_digest(sd, st)
    conf = np.unpackbits(arrays["argmax_conf_bits"]).astype(bool)
    got = np.argmax(p, axis=1).astype(np.uint8).reshape(-1)
    ref = np.unpackbits(arrays["argmax_bits"])
    assert np.array_equal(got[conf], ref[conf]) and int((~conf).sum()) == meta["n_near_ties"]


def test_probunet_full_b32_digest():
    """BASELINE config 3: ProbabilisticUnet(1,2,[32,64,128,192,192,192,192], latent_dim=6, no_convs_fcomb=3), batch 32,
    plus the 8 posterior-sample decodes (probabilistic_unet.py:246-283,343-370)."""
    arrays, meta = G.load("probunet_full_b32_digest")
    B, L, nd = meta["batch"], meta["latent_dim"], meta["n_decode"]
    sd0 = oracle.deterministic_state_dict(G.spec_of(meta), seed=meta["weight_seed"])
    sd = G.leaves(sd0)
    x, mask, eps = oracle.synthetic_batch(B, 128, 128, seed=20201004, eps_shapes=[(B, L)] * (1 + nd))
    out = oracle.probunet_forward(sd, _t(x), _t(mask), bn_train=True)
    loss, aux = oracle.probunet_loss(sd, out, _t(mask), _t(eps[0]), bn_train=True)
    loss.backward()
    st = meta["steps"][0]
    assert abs(float(loss) - st["loss"]) <= 1e-5 * abs(st["loss"])
    assert abs(float(aux["kl"]) - st["kl"]) <= 1e-4 * max(1.0, abs(st["kl"]))
    idx, fidx = arrays["s_idx"], arrays["f_idx"]
    assert G.maxabs(out["last_conv"].detach().numpy().reshape(-1)[idx], arrays["last_conv_samp"]) <= 1e-4
    assert G.maxabs(out["unet_features"].detach().numpy().reshape(-1)[fidx], arrays["features_samp"]) <= 1e-4
    assert G.maxabs(aux["reconstruction"].detach().numpy().reshape(-1)[idx], arrays["reconstruction_samp"]) <= 1e-4
    assert G.maxabs(out["posterior_mu"].detach().numpy(), arrays["post_mu"]) <= 1e-5
    assert G.maxabs(out["prior_sigma"].detach().numpy(), arrays["prior_sigma"]) <= 1e-5
    none = sorted(k for k, v in sd.items() if v.requires_grad and v.grad is None)
    assert none == sorted(st["none_grads"])
    _check_grad_digest(sd, st)
    # eval-mode decodes: z = mu_q + sigma_q * eps_j, Fcomb on the cached features
    with torch.no_grad():
        ev = {k: v.clone() for k, v in sd0.items()}
        o = oracle.probunet_forward(ev, _t(x), _t(mask), bn_train=False)
        assert G.maxabs(o["posterior_mu"].numpy(), arrays["eval_post_mu"]) <= 1e-5
        for j in range(nd):
            z = o["posterior_mu"] + o["posterior_sigma"] * _t(eps[1 + j])
            rec = oracle.probunet_fcomb(ev, o["unet_features"], z, bn_train=False).numpy()
            assert G.maxabs(rec.reshape(-1)[idx], arrays[f"dec{j}_samp"]) <= 1e-4
            assert meta["decode_margin_min"][j] > 1e-3
            assert np.array_equal(np.packbits(np.argmax(rec, axis=1).astype(np.uint8).reshape(-1)), arrays[f"dec{j}_argmax_bits"])


@pytest.mark.parametrize("name", ["phiseg3d_small", "phiseg3d_l3"])
def test_phiseg3d_oracle_vs_reference_modules(name):
    """oracle/refgraph3d.py against the reference's own 3-D Posterior / prior / Likelihood modules, its loss functions and
    autograd (fixtures from tools/gen_golden.py `3d`; the one statement the reference cannot execute is documented there)."""
    from oracle import refgraph3d as R3
    arrays, meta = G.load(name)
    sd = G.leaves(oracle.deterministic_state_dict(G.spec_of(meta), seed=meta["weight_seed"]))
    L = meta["latent_levels"]
    T = lambda k: torch.from_numpy(arrays[k])       # noqa: E731
    eps = [T(f"eps{k}") for k in range(2 * L)]
    out = R3.phiseg3d_forward(sd, T("patch"), T("mask_onehot"), eps, training=True, bn_train=True)
    for l in range(L):
        for a, b in (("post_mu", "post_mu"), ("post_sigma", "post_sigma"), ("post_z", "post_z"), ("prior_mu", "prior_mu"),
                     ("prior_sigma", "prior_sigma"), ("s_in", "s_in")):
            got, want = out[a][l].detach().numpy(), arrays[f"{b}{l}"]
            assert got.shape == want.shape
            np.testing.assert_allclose(got, want, rtol=1e-5, atol=1e-5, err_msg=f"{a}{l}")
    total, terms = R3.phiseg3d_loss(out, T("labels"), num_classes=meta["num_classes"])
    assert abs(float(total) - float(arrays["loss"])) <= 1e-5 * abs(float(arrays["loss"]))
    for l in range(L):
        assert abs(float(terms[l]) - float(arrays[f"loss:KL_divergence_loss_lvl{l}"])) <= 1e-5 * abs(float(arrays[f"loss:KL_divergence_loss_lvl{l}"])) + 1e-6
        assert abs(float(terms[L + l]) - float(arrays[f"loss:residual_multinoulli_loss_lvl{l}"])) <= 1e-5 * float(arrays[f"loss:residual_multinoulli_loss_lvl{l}"])
    total.backward()
    n = 0
    for k, v in sd.items():
        if ("g:" + k) in arrays:
            g = arrays["g:" + k]
            scale = max(float(np.abs(g).max()), 1e-6)
            assert v.grad is not None, k
            assert float(np.abs(v.grad.numpy() - g).max()) <= 2e-4 * scale + 1e-6, k
            n += 1
        elif v.dtype.is_floating_point and "running_" not in k:
            assert v.grad is None or float(v.grad.abs().max()) == 0.0, k
    assert n > 50
    for k in sd:
        if "running_" in k:
            np.testing.assert_allclose(sd[k].detach().numpy(), arrays["sd1:" + k], rtol=1e-5, atol=1e-6)


def test_oracle3d_matches_the_reference_fp32_leg_of_the_bf16_fixture():
    """The fp32 leg of tests/golden/phiseg3d_bf16 (reference modules on a 32 x 64 x 64 volume, filters 32-32-64) pins the 3-D
    oracle at a size where the native library uses its matrix-pipe kernels; with operand rounding switched on for every 3x3x3
    layer the oracle lands between the reference's fp32 and bf16 runs (closer to fp32 than the reference's own bf16 run)."""
    from oracle import refgraph as RG
    from oracle import refgraph3d as R3
    from oracle.refgraph3d import phiseg3d_eps_shapes, synthetic_volume
    arrays, meta = G.load("phiseg3d_bf16")
    D, H, W = meta["dhw"]
    Lv = meta["latent_levels"]
    shapes = phiseg3d_eps_shapes(D, H, W, len(meta["filters"]), Lv)
    x, onehot, lab, eps = synthetic_volume(meta["input_channels"], meta["num_classes"], (D, H, W), meta["input_seed"], shapes + shapes)
    sd = oracle.deterministic_state_dict(G.spec_of(meta), seed=meta["weight_seed"])
    args = (torch.from_numpy(x), torch.from_numpy(onehot), [torch.from_numpy(e) for e in eps])
    st = meta["logit_stride"]

    def gaps(out):
        g32 = gbf = 0.0
        for l in range(Lv):
            for name, t in (("post_mu", out["post_mu"][l]), ("post_sigma", out["post_sigma"][l]), ("prior_mu", out["prior_mu"][l]),
                            ("prior_sigma", out["prior_sigma"][l]), ("s_in", out["s_in"][l])):
                got = t.detach().numpy().reshape(-1)[::st]
                rf, rb = arrays[f"fp32:{name}{l}"], arrays[f"bf16:{name}{l}"]
                rng = float(np.abs(rf).max())
                g32, gbf = max(g32, G.maxabs(got, rf) / rng), max(gbf, G.maxabs(got, rb) / rng)
        return g32, gbf
    with torch.no_grad():
        out = R3.phiseg3d_forward(G.leaves(sd), *args)
        total, _ = R3.phiseg3d_loss(out, torch.from_numpy(lab), num_classes=meta["num_classes"])
    g32, _ = gaps(out)
    assert g32 <= 1e-4, g32
    assert abs(float(total) - float(arrays["fp32:loss"])) <= 2e-5 * abs(float(arrays["fp32:loss"]))
    RG.CONV_OPERAND_ROUNDING = lambda xx, ww: ww.dim() == 5 and ww.shape[-1] == 3
    try:
        with torch.no_grad():
            out_r = R3.phiseg3d_forward(G.leaves(oracle.deterministic_state_dict(G.spec_of(meta), seed=meta["weight_seed"])), *args)
    finally:
        RG.CONV_OPERAND_ROUNDING = None
    r32, rbf = gaps(out_r)
    ref_gap = max(G.maxabs(arrays[f"bf16:s_in{l}"], arrays[f"fp32:s_in{l}"]) / float(np.abs(arrays[f"fp32:s_in{l}"]).max()) for l in range(Lv))
    assert 1e-5 < r32 <= 1.2 * ref_gap and rbf <= 4e-2, (r32, rbf, ref_gap)
