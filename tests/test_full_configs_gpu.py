"""BASELINE.json configs 2 and 3 at their stated size on the HIP path, against digests of the REAL reference
(tools/gen_golden.py `unet_b32` / `probunet_b32`), plus the inference entry points of PHISeg and a
direct parity test of the split-fp16 kernels on the heaviest layer at batch 32.

Gates (BASELINE.json north_star): logits within 1e-4 of the reference, bit-exact argmax label maps (on every pixel
whose reference logit margin exceeds 2e-4 = twice the logit tolerance; the handful of nearer ties is counted in the
fixture and cannot be decided by any fp32 implementation), loss to 2e-5 relative; gradients per tensor."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

import oracle
from tests import _golden as G

pytestmark = pytest.mark.gpu
DEV = lambda: torch.device("cuda", 0)  # noqa: E731


def _grad_digest_check(net, st, noise, rel=1e-2):
    params = dict(net.named_parameters())
    worst = 0.0
    for k, n in st["grad_norms"].items():
        if k in noise:
            continue
        mine = float(params[k].grad.double().norm())
        e = abs(mine - n) / max(n, 1e-3)
        worst = max(worst, e)
        assert e <= rel, (k, mine, n)
        pick, vals = st["grad_samples"][k]
        got = params[k].grad.reshape(-1)[torch.tensor(pick)].cpu().numpy()
        # single entries of a deep layer move by up to ~2 % of the tensor's norm between two fp32 implementations (chaotic
        # amplification of rounding noise, see test_phiseg_b32_gradients_vs_fp64_reference): twice the norm gate
        assert np.max(np.abs(got - np.array(vals))) <= 2 * rel * max(n, 1e-3), k
    return worst


def test_unet_full_b32_vs_reference_digest():
    """Config 2: Unet(1,2,[32,64,128,192]), batch 32 (unet.py:78-165)."""
    from unet_zoo_amd.models.unet import Unet
    arrays, meta = G.load("unet_full_b32_digest")
    net = Unet(1, 2, meta["filters"])
    net.load_state_dict(oracle.deterministic_state_dict(G.spec_of(meta), seed=meta["weight_seed"]))
    net.train()
    x, mask, _ = oracle.synthetic_batch(meta["batch"], 128, 128, seed=20201004)
    pred = net.forward(torch.from_numpy(x).to(DEV()))
    loss = net.loss(torch.from_numpy(mask).to(DEV()))
    loss.backward()
    st = meta["steps"][0]
    assert abs(float(loss) - st["loss"]) <= 2e-5 * abs(st["loss"]), (float(loss), st["loss"])
    p = pred.cpu().numpy()
    assert G.maxabs(p.reshape(-1)[arrays["s_idx"]], arrays["pred_samp"]) <= 1e-4
    assert all(q.grad is not None for q in net.parameters()) and st["none_grads"] == []
    worst = _grad_digest_check(net, st, noise=set(), rel=5e-3)
    conf = np.unpackbits(arrays["argmax_conf_bits"]).astype(bool)
    got = np.argmax(p, axis=1).astype(np.uint8).reshape(-1)
    ref = np.unpackbits(arrays["argmax_bits"])
    # the pixels the fixture leaves out of the bit-exact comparison are exactly the ones it recorded as near ties (reference margin
    # below 2e-4) - a count the generator stored, not an allowance
    assert int(conf.size - conf.sum()) == meta["n_near_ties"]
    assert np.array_equal(got[conf], ref[conf])                   # bit-exact label map wherever the margin decides it
    print(f"unet b32: loss rel {abs(float(loss) - st['loss']) / st['loss']:.1e}, worst grad-norm dev {worst:.1e}, "
          f"{int((got != ref).sum())} of {meta['n_near_ties']} near-tie pixels differ")


def test_probunet_full_b32_vs_reference_digest_and_8_decodes():
    """Config 3: ProbabilisticUnet(1,2,[32,64,128,192,192,192,192], latent_dim=6, no_convs_fcomb=3), batch 32, then the
    8 posterior-sample decodes reconstruct(calculate_posterior=True) in eval mode (probabilistic_unet.py:246-283,343-370)."""
    from unet_zoo_amd.models.probabilistic_unet import ProbabilisticUnet
    arrays, meta = G.load("probunet_full_b32_digest")
    B, L, nd = meta["batch"], meta["latent_dim"], meta["n_decode"]
    sd0 = oracle.deterministic_state_dict(G.spec_of(meta), seed=meta["weight_seed"])
    net = ProbabilisticUnet(1, 2, meta["filters"], latent_dim=L, no_convs_fcomb=3, image_size=(1, 128, 128))
    net.load_state_dict(sd0)
    net.train()
    x, mask, eps = oracle.synthetic_batch(B, 128, 128, seed=20201004, eps_shapes=[(B, L)] * (1 + nd))
    xd, md = torch.from_numpy(x).to(DEV()), torch.from_numpy(mask).to(DEV())
    last = net.forward(xd, md, training=True)
    loss = net.loss(md, eps=torch.from_numpy(eps[0]).to(DEV()))
    loss.backward()
    st = meta["steps"][0]
    assert abs(float(loss) - st["loss"]) <= 2e-5 * abs(st["loss"]), (float(loss), st["loss"])
    assert abs(float(net.kl_divergence_loss) - st["kl"]) <= 1e-3 * max(1.0, abs(st["kl"]))
    assert abs(float(net.reconstruction_loss) - st["recon"]) <= 2e-5 * abs(st["recon"])
    idx, fidx = arrays["s_idx"], arrays["f_idx"]
    assert G.maxabs(last.cpu().numpy().reshape(-1)[idx], arrays["last_conv_samp"]) <= 1e-4
    assert G.maxabs(net.unet_features.cpu().numpy().reshape(-1)[fidx], arrays["features_samp"]) <= 1e-4
    assert G.maxabs(net.reconstruction.cpu().numpy().reshape(-1)[idx], arrays["reconstruction_samp"]) <= 1e-4
    assert G.maxabs(net.posterior_latent_space.mean.cpu().numpy(), arrays["post_mu"]) <= 1e-4
    assert G.maxabs(net.posterior_latent_space.stddev.cpu().numpy(), arrays["post_sigma"]) <= 1e-4
    assert G.maxabs(net.prior_latent_space.mean.cpu().numpy(), arrays["prior_mu"]) <= 1e-4
    assert G.maxabs(net.prior_latent_space.stddev.cpu().numpy(), arrays["prior_sigma"]) <= 1e-4
    none = sorted(k for k, p in net.named_parameters() if p.grad is None)
    assert none == sorted(st["none_grads"])
    noise = G.bn_shadowed_biases(dict(net.named_parameters()).keys())
    worst = _grad_digest_check(net, st, noise, rel=1e-2)
    # ---- 8 posterior-sample decodes on the cached U-Net features, eval mode
    net.load_state_dict(sd0)
    net.eval()
    with torch.no_grad():
        net.forward(xd, md, training=False)
        mu, sig = net.posterior_latent_space.mean, net.posterior_latent_space.stddev
        assert G.maxabs(mu.cpu().numpy(), arrays["eval_post_mu"]) <= 1e-4
        assert G.maxabs(sig.cpu().numpy(), arrays["eval_post_sigma"]) <= 1e-4
        for j in range(nd):
            z = mu + sig * torch.from_numpy(eps[1 + j]).to(DEV())           # posterior.rsample() with the recorded eps
            rec = net.reconstruct(z_posterior=z).cpu().numpy()
            assert G.maxabs(rec.reshape(-1)[idx], arrays[f"dec{j}_samp"]) <= 1e-4, j
            assert meta["decode_margin_min"][j] > 1e-3
            bits = np.packbits(np.argmax(rec, axis=1).astype(np.uint8).reshape(-1))
            assert np.array_equal(bits, arrays[f"dec{j}_argmax_bits"]), j    # bit-exact label maps
    print(f"probunet b32: loss rel {abs(float(loss) - st['loss']) / st['loss']:.1e}, worst grad-norm dev {worst:.1e}")


def test_probunet_decode_between_loss_and_backward_keeps_gradients():
    """sample() / reconstruct() between loss() and backward() is legal in the reference; the decode tape shares the
    Fcomb input buffer's z channels with the loss tape, so the backward tape re-tiles z before it reads them."""
    from unet_zoo_amd.models.probabilistic_unet import ProbabilisticUnet
    arrays, meta = G.load("probunet_small")
    B, L = meta["batch"], meta["latent_dim"]
    x, mask, eps = oracle.synthetic_batch(B, 128, 128, seed=20201004, eps_shapes=[(B, L)])
    xd, md, ed = torch.from_numpy(x).to(DEV()), torch.from_numpy(mask).to(DEV()), torch.from_numpy(eps[0]).to(DEV())
    grads = []
    for interleave in (False, True):
        net = ProbabilisticUnet(1, 2, meta["filters"], latent_dim=L, no_convs_fcomb=3, image_size=(1, 128, 128))
        net.load_state_dict(oracle.deterministic_state_dict(G.spec_of(meta), seed=meta["weight_seed"]))
        net.eval()                       # eval-mode BN would make decode side-effect free; train mode updates running stats only
        net.train()
        net.forward(xd, md, training=True)
        loss = net.loss(md, eps=ed)
        if interleave:
            net.sample(testing=True)
            net.reconstruct(use_posterior_mean=True)
        loss.backward()
        grads.append({k: p.grad.clone() for k, p in net.named_parameters() if p.grad is not None})
    for k in grads[0]:
        assert torch.equal(grads[0][k], grads[1][k]), k


def test_backward_twice_accumulates_like_torch():
    from unet_zoo_amd.models.unet import Unet
    arrays, meta = G.load("unet_small")
    net = Unet(1, 2, meta["filters"])
    net.load_state_dict(oracle.deterministic_state_dict(G.spec_of(meta), seed=meta["weight_seed"]))
    x, mask, _ = oracle.synthetic_batch(meta["batch"], 128, 128, seed=20201004)
    xd, md = torch.from_numpy(x).to(DEV()), torch.from_numpy(mask).to(DEV())
    net.forward(xd)
    net.loss(md).backward()
    g1 = {k: p.grad.clone() for k, p in net.named_parameters()}
    net.forward(xd)
    net.loss(md).backward()                                # no zero_grad in between
    for k, p in net.named_parameters():
        assert torch.allclose(p.grad, 2 * g1[k], rtol=1e-6, atol=0), k


# ----------------------------------------------------------------------------- A13: PHISeg inference entry points
def test_phiseg_reconstruct_sample_vs_oracle():
    """PHISeg.reconstruct / sample / sample_prior / sample_posterior (phiseg.py:386-412): the decode-only plan against
    the oracle's likelihood + accumulate_output on the same latent samples; argmax bit-exact."""
    arrays, meta = G.load("phiseg_mid")
    from unet_zoo_amd.models.phiseg import PHISeg
    sd0 = oracle.deterministic_state_dict(G.spec_of(meta), seed=meta["weight_seed"])
    net = PHISeg(1, 2, meta["filters"], image_size=(1, meta["hw"], meta["hw"]))
    net.load_state_dict(sd0)
    net.eval()
    B, hw = meta["batch"], meta["hw"]
    shapes = oracle.phiseg_eps_shapes(B, hw, hw)
    x, mask, eps = oracle.synthetic_batch(B, hw, hw, seed=20201004, eps_shapes=shapes + shapes)
    e_dev = [torch.from_numpy(e).to(DEV()) for e in eps]
    from oracle import refgraph as R
    with torch.no_grad():
        net.forward(torch.from_numpy(x).to(DEV()), torch.from_numpy(mask).to(DEV()), training=False, eps=e_dev)
        # sample_prior / sample_posterior draw mu + sigma * randn on the device with the reference's shapes
        torch.manual_seed(3)
        zp = net.sample_prior()
        zq = net.sample_posterior()
        assert [tuple(z.shape) for z in zp] == [tuple(s) for s in reversed(shapes)] == [tuple(z.shape) for z in zq]
        for z, m, s_ in zip(zp, net.prior_mu, net.prior_sigma):
            assert torch.isfinite(z).all() and float(((z - m) / s_).std()) > 0.5
        # reconstruct on given latents (finest level first) vs the oracle's likelihood
        z_list = [m.clone() for m in net.posterior_mu]
        for use_softmax in (True, False):
            rec, layers = net.reconstruct(z_list, use_softmax=use_softmax)
            sd = {k: v.clone() for k, v in sd0.items()}
            s_ref = R._phiseg_likelihood(sd, [z.cpu() for z in z_list], (hw, hw), bn_train=False)
            for l in range(4):                                   # layers[4] is accumulated in place (phiseg.py:428-434)
                assert G.maxabs(layers[l].cpu().numpy(), s_ref[l].numpy()) <= 1e-4
            ref = oracle.phiseg_accumulate_output([t.clone() for t in s_ref], use_softmax=use_softmax)
            assert G.maxabs(rec.cpu().numpy(), ref.numpy()) <= (1e-5 if use_softmax else 3e-4)
            assert torch.equal(layers[-1], rec) or use_softmax            # in-place accumulation into the last level (phiseg.py:428-434)
            acc = oracle.phiseg_accumulate_output([t.clone() for t in s_ref], use_softmax=False)
            margin = (acc[:, 1] - acc[:, 0]).abs()
            ok = (margin > 6e-4).numpy()
            got = torch.argmax(rec, dim=1).cpu().numpy()
            assert np.array_equal(got[ok], torch.argmax(acc, dim=1).numpy()[ok]) and ok.mean() > 0.99
        # sample(testing=True) = reconstruct(sample_prior(), use_softmax=False)[0]
        torch.manual_seed(5)
        a = net.sample(testing=True)
        torch.manual_seed(5)
        b, _ = net.reconstruct(net.sample_prior(), use_softmax=False)
        assert torch.equal(a, b) and a.shape == (B, 2, hw, hw)
        with pytest.raises(NotImplementedError):
            net.sample(testing=False)


def test_phiseg_public_loss_helpers_match_reference_golden():
    """KL_two_gauss_with_diag_cov / multinoulli_loss / residual_multinoulli_loss / calculate_hierarchical_KL_div_loss
    are public on the reference class (phiseg.py:436-513); the native class exposes them as device evaluations."""
    from unet_zoo_amd.models.phiseg import PHISeg
    arrays, meta = G.load("ops")
    net = PHISeg(1, 2, [4, 8, 8, 8, 8, 8, 8], image_size=(1, 64, 64))
    d = DEV()
    for i in range(3):
        t = [torch.from_numpy(arrays[f"kl{i}_{n}"]).to(d) for n in ("mu0", "s0", "mu1", "s1")]
        kl = float(net.KL_two_gauss_with_diag_cov(*t))
        assert abs(kl - meta[f"kl{i}"]) <= 2e-5 * max(1.0, abs(meta[f"kl{i}"])), (i, kl, meta[f"kl{i}"])
    s = [torch.from_numpy(arrays[f"rm_s{i}"]).to(d) for i in range(5)]
    tgt = torch.from_numpy(arrays["rm_target"]).to(d)
    net.loss_tot, net.loss_dict = 0, {}
    tot = net.residual_multinoulli_loss(s, tgt)
    assert abs(float(tot) - meta["rm_total"]) <= 2e-5 * abs(meta["rm_total"])
    for i in range(5):
        assert abs(float(net.loss_dict["residual_multinoulli_loss_lvl%d" % i]) - meta[f"rm_lvl{i}"]) <= 2e-5 * abs(meta[f"rm_lvl{i}"])
    ref = oracle.refgraph.multinoulli_loss(s[4].cpu(), tgt.cpu(), 2)
    assert abs(float(net.multinoulli_loss(s[4], tgt)) - float(ref)) <= 2e-5 * abs(float(ref))


# ----------------------------------------------------------------------------- dominant kernels at BASELINE size, directly
def _err64(got, ref64):
    return float((got.double().cpu() - ref64).abs().max() / ref64.abs().max())


@pytest.mark.parametrize("Cin,Cout,H", [(224, 128, 128), (256, 192, 64)])
def test_split_kernels_at_baseline_size_vs_fp64(Cin, Cout, H):
    """conv_split_kernel / wgrad_split_kernel on the two heaviest PHiSeg layers at batch 32, reached through the C ABI
    WITH a workspace (the product path), against an fp64 CPU convolution: forward and data gradient on two images of
    the batch, weight gradient over all 32 images on 8 of the output channels (a few seconds of fp64 CPU work).  The split-fp16
    kernels must be as accurate as the fp32-MFMA kernels of the same library: error vs fp64 <= 2x theirs."""
    from tests import _gpu as g
    from unet_zoo_amd import _ffi
    L = _ffi.lib()
    N, W = 32, H
    gen = torch.Generator().manual_seed(11)
    x = torch.randn(N, Cin, H, W, generator=gen)
    w = torch.randn(Cout, Cin, 3, 3, generator=gen) * 0.05
    dy = torch.randn(N, Cout, H, W, generator=gen)
    xd, wd, dyd = x.to(g.dev()), w.to(g.dev()), dy.to(g.dev())
    pick = [5, 29]
    y64 = F.conv2d(x[pick].double(), w.double(), None, padding=1)
    dx64 = F.conv_transpose2d(dy[pick].double(), w.double(), None, padding=1)
    sel = [0, 1, 37, 38, 63, 64, Cout - 2, Cout - 1]            # 8 output channels across the 64-channel tiles: 1/16 of the fp64 CPU work
    dw64 = torch.nn.grad.conv2d_weight(x.double(), (len(sel), Cin, 3, 3), dy[:, sel].double(), padding=1)
    errs = {}
    try:
        for mode, tag in ((0, "f32"), (1, "default")):
            assert L.uz_set_conv_math(mode) == 0
            if mode == 1:
                assert L.uz_get_conv_math() == 1
            cws_b = L.uz_conv_workspace(Cin, Cout, N, H, W, 3)
            cws = torch.zeros(cws_b // 4 + 16, device=g.dev())
            wws_b = L.uz_conv_bwd_weight_workspace(Cin, Cout, N, H, W, 3)
            wws = torch.zeros(wws_b // 4 + 16, device=g.dev())
            y = torch.empty(N, Cout, H, W, device=g.dev())
            dx = torch.empty(N, Cin, H, W, device=g.dev())
            dw = torch.empty_like(wd)
            g.call("uz_conv_fwd", xd, Cin, Cin, wd, None, y, Cout, Cout, N, H, W, 3, 0, None, None, None, cws, cws_b)
            g.call("uz_conv_bwd_data", dyd, Cout, Cout, wd, dx, Cin, Cin, N, H, W, 3, 0, None, None, cws, cws_b)
            g.call("uz_conv_bwd_weight", xd, Cin, Cin, dyd, Cout, Cout, dw, None, N, H, W, 3, None, None, wws, wws_b)
            errs[tag] = (_err64(y[pick], y64), _err64(dx[pick], dx64), _err64(dw[sel], dw64))
            if mode == 1:       # the default policy must actually have selected the split kernels for this layer
                assert cws_b > 0, "split forward kernel not selected"
    finally:
        L.uz_set_conv_math(-1)
    print(f"{Cin}->{Cout}@{H}: err vs fp64 (fwd, dgrad, wgrad)  f32-MFMA {errs['f32']}  split {errs['default']}")
    for a, b, what in zip(errs["default"], errs["f32"], ("fwd", "dgrad", "wgrad")):
        assert a <= 2.0 * b + 1e-7, (what, a, b)
        assert a <= 5e-6, (what, a)
