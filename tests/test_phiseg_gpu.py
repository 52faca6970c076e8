"""End-to-end parity of the native PHISeg against (a) golden vectors generated from the real
reference and (b) the CPU oracle run live on the same seeded inputs.  Gates follow BASELINE.json:
segmentation logits within 1e-4 (fp32), bit-exact argmax label maps."""
import os

import numpy as np
import pytest
import torch

import oracle
from tests import _golden as G

pytestmark = pytest.mark.gpu


def _model(meta, **kw):
    from unet_zoo_amd.models.phiseg import PHISeg
    net = PHISeg(1, 2, meta["filters"], latent_levels=5, image_size=(1, meta["hw"], meta["hw"]), **kw)
    sd = oracle.deterministic_state_dict(G.spec_of(meta), seed=meta["weight_seed"])
    missing = net.load_state_dict(sd)
    assert not missing.missing_keys and not missing.unexpected_keys
    return net, sd


def _inputs(meta, step):
    shapes = oracle.phiseg_eps_shapes(meta["batch"], meta["hw"], meta["hw"])
    x, mask, eps = oracle.synthetic_batch(meta["batch"], meta["hw"], meta["hw"], seed=20201004 + step, eps_shapes=shapes + shapes)
    dev = torch.device("cuda", 0)
    return (torch.from_numpy(x).to(dev), torch.from_numpy(mask).to(dev), [torch.from_numpy(e).to(dev) for e in eps])


def test_state_dict_surface():
    arrays, meta = G.load("phiseg_small")
    net, sd = _model(meta)
    mine = net.state_dict()
    assert list(mine.keys()) == [k for k, _, _ in G.spec_of(meta)]
    for k, v in mine.items():
        assert tuple(v.shape) == tuple(sd[k].shape) and torch.equal(v.cpu(), sd[k])


# The small fixture (filters 4/8, 64x64, batch 2) normalises over as few as TWO values per channel at
# its 1x1 deepest level: BatchNorm(train) is then ill-conditioned (d out/d in ~ eps/(d^2+eps) by
# cancellation), so two correct fp32 implementations differ by ~1e-4 on logits and ~1e-2 on the
# deepest-layer gradients.  Its gates are therefore looser than BASELINE's 1e-4; the 1e-4 logit gate
# is enforced on the real architecture (full-size digests, batch 2 and batch 32) below.
SMALL_LOGIT_TOL = 5e-4
SMALL_GRAD_TOL = 5e-2
# fixture -> (logit tol, grad tol rel. to the tensor's max, loss rel tol per step)
# Loss tolerances grow per step: Adam's first update is lr*sign(g), so every near-zero gradient entry whose
# sign differs between two fp32 implementations moves that weight by 2*lr (measured: ~6e-5 of all entries);
# the CPU oracle run in fp64 instead of fp32 drifts from the fp32 reference by 7e-6 / 1e-3 at steps 1 / 2.
TRAIN_GATES = {"phiseg_small": (SMALL_LOGIT_TOL, SMALL_GRAD_TOL, (1e-4, 3e-4, 3e-3)),
               "phiseg_mid": (1e-4, 1e-1, (2e-5, 3e-4, 6e-3))}
# Gradient gate of the mid fixture = a distribution (median <= 1e-3, 90th percentile <= 1e-2 of the tensor's max) plus a cap on
# the worst tensor.  The worst tensors are always the prior's finest-level sample_z / upsampling units: their gradient is the KL
# term's d/d sigma1, a difference of near-equal numbers that amplifies ANY rounding difference upstream by ~1e5 - two correct
# fp32 first-layer kernels (MFMA tile vs the streaming thin-input kernel) realise 1.1e-2 and 6.0e-2 there while the median over
# tensors moves from 1.5e-4 to 1.0e-4.  Accuracy against the real-valued graph is gated by the fp64 tests below.
# (third-step loss of the mid fixture, measured: 1.8e-3 with the fp32-MFMA convolutions, 3.9e-3 with the split
#  convolutions forced onto every layer, 1e-3 for the CPU oracle in fp64 vs fp32 - all of them sign-flip noise of Adam's
#  first updates, not arithmetic error: the first-step loss agrees to 1e-7 and the logits to < 1e-4 in every mode)
# (the largest gradient deviations sit in the KL path: d/d sigma1 = s0/B - A*s0/B^2 is a difference of
#  near-equal terms whenever posterior ~ prior, in the reference's autograd as much as here)


@pytest.mark.parametrize("fixture", ["phiseg_small", "phiseg_mid"])
def test_phiseg_train_steps_vs_reference_golden(fixture):
    """Train-step contract: 3 x [forward, loss, zero_grad, backward, Adam(lr 1e-3, wd 1e-5)]
    from the reference's captured inputs / noise / initial state_dict."""
    from unet_zoo_amd.optim import FusedAdam
    logit_tol, grad_tol, loss_tols = TRAIN_GATES[fixture]
    arrays, meta = G.load(fixture)
    net, _ = _model(meta)
    net.train()
    opt = FusedAdam(net, lr=1e-3, weight_decay=1e-5)
    noise = G.bn_shadowed_biases(dict(net.named_parameters()).keys())
    theta0 = {k: v.detach().cpu().numpy().copy() for k, v in net.named_parameters()}
    for step, st in enumerate(meta["steps"]):
        x, mask, eps = _inputs(meta, step)
        s = net.forward(x, mask, training=True, eps=eps)
        loss = net.loss(mask)
        opt.zero_grad()
        loss.backward()
        assert abs(float(loss) - st["loss"]) <= loss_tols[step] * abs(st["loss"]), (step, float(loss), st["loss"])
        for k, v in st["loss_dict"].items():
            assert abs(float(net.loss_dict[k]) - v) <= 50 * loss_tols[step] * max(1.0, abs(v)), (step, k)
        assert float(net.kl_divergence_loss) == float(loss) == float(net.reconstruction_loss)   # alias quirk
        none = sorted(k for k, p in net.named_parameters() if p.grad is None)
        assert none == sorted(st["none_grads"])
        if step == 0:
            for l in range(5):
                assert G.maxabs(s[l].cpu().numpy(), arrays[f"s{l}"]) <= logit_tol, l
                assert G.maxabs(net.posterior_mu[l].cpu().numpy(), arrays[f"post_mu{l}"]) <= 1e-4
                assert G.maxabs(net.posterior_sigma[l].cpu().numpy(), arrays[f"post_sigma{l}"]) <= 1e-4
                assert G.maxabs(net.posterior_latent_space[l].cpu().numpy(), arrays[f"post_z{l}"]) <= 1e-4
                assert G.maxabs(net.prior_mu[l].cpu().numpy(), arrays[f"prior_mu{l}"]) <= 1e-4
                assert G.maxabs(net.prior_sigma[l].cpu().numpy(), arrays[f"prior_sigma{l}"]) <= 1e-4
            worst, wk, devs = 0.0, None, []
            for k, p in net.named_parameters():
                if p.grad is not None and k not in noise:
                    ref = arrays["grad:" + k]
                    e = G.maxabs(p.grad.cpu().numpy(), ref) / (1e-3 + float(np.abs(ref).max()))
                    devs.append(e)
                    if e > worst:
                        worst, wk = e, k
            assert worst <= grad_tol, (worst, wk)
            if fixture == "phiseg_mid":
                assert np.median(devs) <= 1e-3 and np.percentile(devs, 90) <= 1e-2, (float(np.median(devs)), float(np.percentile(devs, 90)))
            for k, v in net.state_dict().items():
                if "running_" in k:
                    assert G.maxabs(v.cpu().numpy(), arrays["buf1:" + k]) <= 1e-5, k
            g_hip = {k: p.grad.cpu().numpy().copy() for k, p in net.named_parameters() if p.grad is not None}
        opt.step()
        if step == 0:
            # post-step parameters (row H): the first Adam update must equal the reference's on every entry whose
            # gradient sign is determined (see _golden.check_first_adam_step) - a wrong-sign / wrong-scale optimiser fails
            theta1 = {k: v.detach().cpu().numpy() for k, v in net.named_parameters()}
            g_ref = {k: arrays["grad:" + k] for k in g_hip}
            cover = G.check_first_adam_step(theta0, theta1, g_hip, g_ref, skip=noise)
            assert cover >= 0.5
            for k in st["none_grads"]:                       # skipped, not zero-stepped (no weight decay either)
                assert np.array_equal(theta1[k], theta0[k]), k
    sd = net.state_dict()
    for k, v in sd.items():
        if v.dtype.is_floating_point and k not in noise:
            # Adam turns every gradient into a step of magnitude ~lr whatever its size: an entry whose
            # near-zero gradient has the opposite sign in the two implementations drifts by 2*lr per step, so after
            # 3 steps only a statistical statement is possible: the bulk of the entries agrees closely (a systematic
            # optimiser error would move EVERY entry by ~3*lr); the sharp per-entry gate is the step-1 check above
            d = np.abs(v.cpu().numpy().astype(np.float64) - arrays["final:" + k]).reshape(-1)
            assert d.max() <= 6.5e-3, k
            if d.size >= 64:
                assert np.median(d) <= 5e-4, (k, float(np.median(d)))
    nbt = [int(v) for k, v in sd.items() if k.endswith("num_batches_tracked") and "upsampling_path.4" not in k]
    assert set(nbt) == {len(meta["steps"])}
    assert all(int(v) == 0 for k, v in sd.items() if k.endswith("num_batches_tracked") and "upsampling_path.4" in k)


def test_phiseg_small_eval_argmax_bit_exact():
    arrays, meta = G.load("phiseg_small")
    net, _ = _model(meta)
    net.eval()
    x, mask, eps = _inputs(meta, 0)
    with torch.no_grad():
        s = net.forward(x, mask, training=False, eps=eps)
        for l in range(5):
            assert G.maxabs(s[l].cpu().numpy(), arrays[f"eval_s{l}"]) <= SMALL_LOGIT_TOL
        last_before = s[-1].clone()
        soft = net.accumulate_output(s, use_softmax=True)
    assert not torch.equal(s[-1], last_before)                 # accumulated in place, like the reference
    assert G.maxabs(soft.cpu().numpy(), arrays["eval_softmax"]) <= 1e-5
    assert meta["eval_margin_min"] > 1e-3
    bits = np.packbits(torch.argmax(soft, dim=1).cpu().numpy().astype(np.uint8).reshape(-1))
    assert np.array_equal(bits, arrays["eval_argmax_bits"])


@pytest.mark.parametrize("fixture", ["phiseg_full_digest", "phiseg_full_b32_digest"])
def test_phiseg_full_size_digest_vs_reference_golden(fixture):
    """BASELINE config 4 architecture (filters 32..192, 128x128) at batch 2 and at the headline batch 32
    against digests of the real reference: logits within 1e-4, loss terms, per-tensor gradient norms and
    sampled gradient entries, bit-exact argmax label map."""
    arrays, meta = G.load(fixture)
    net, _ = _model(meta)
    net.train()
    x, mask, eps = _inputs(meta, 0)
    s = net.forward(x, mask, training=True, eps=eps)
    loss = net.loss(mask)
    loss.backward()
    assert net.check_bounds() == 0          # every magnitude bound a split-fp16 kernel was handed covered its tensor (uz_device_flags)
    st = meta["steps"][0]
    assert abs(float(loss) - st["loss"]) <= 2e-5 * abs(st["loss"])
    idx = arrays["s_idx"]
    for l in range(5):
        assert G.maxabs(s[l].cpu().numpy().reshape(-1)[idx], arrays[f"s{l}_samp"]) <= 1e-4, l
        assert G.maxabs(net.posterior_mu[l].cpu().numpy(), arrays[f"post_mu{l}"]) <= 1e-4
        assert G.maxabs(net.prior_sigma[l].cpu().numpy(), arrays[f"prior_sigma{l}"]) <= 1e-4
    noise = G.bn_shadowed_biases(st["grad_norms"].keys())
    params = dict(net.named_parameters())
    for k, n in st["grad_norms"].items():
        if k in noise:
            continue
        mine = float(params[k].grad.double().norm())
        # Calibration (tests/golden/phiseg_full_b32_f64, the REAL reference run in double precision): the reference's own
        # fp32 gradients deviate from fp64 by 1.2e-3 of a tensor's norm in the median and by > 1e-2 for the worst tensors at
        # this batch size - rounding noise amplified through 30+ stacked batch normalisations.  Two fp32 implementations can
        # therefore differ by ~1e-2 of a tensor's norm; the sharp gate (error against fp64 no larger than the reference's own)
        # is test_phiseg_b32_gradients_vs_fp64_reference below.
        tol = float(os.environ.get("UZ_TEST_GRAD_NORM_TOL", "1e-2"))      # (the forced-split run of test_ops_gpu.py widens it, see there)
        if os.environ.get("UZ_CONV_MATH") == "split":
            # The whole tier under UZ_CONV_MATH=split (NOT the default routing: the split path forced onto the 16 x 16 ... 2 x 2 planes the
            # default keeps on fp32, test_default_routing_keeps_small_planes_off_the_split_path): its error is relative to a TENSOR's
            # maximum, and the KL gradients of the latent heads are differences of nearly equal terms - at batch 2 the posterior's
            # sigma_conv biases move by up to 11 % (the fp32 path sits at 0.5 % there); everything else stays within 1.5 %.
            tol = max(tol, 0.15 if ("sigma_conv" in k and fixture == "phiseg_full_digest") else 1.5e-2)
        assert abs(mine - n) <= tol * max(n, 1e-3), (k, mine, n)
        pick, vals = st["grad_samples"][k]
        got = params[k].grad.reshape(-1)[torch.tensor(pick)].cpu().numpy()
        assert np.max(np.abs(got - np.array(vals))) <= tol * max(n, 1e-3), k
    # eval pass: packed argmax bits must be identical
    net.load_state_dict(oracle.deterministic_state_dict(G.spec_of(meta), seed=meta["weight_seed"]))
    net.eval()
    with torch.no_grad():
        s = net.forward(x, mask, training=False, eps=eps)
        soft = net.accumulate_output(s, use_softmax=True)
    assert G.maxabs(s[-1].cpu().numpy().reshape(-1)[idx], arrays["eval_acc_samp"]) <= 2e-4
    assert meta["eval_margin_min"] > 1e-3, meta["eval_margin_min"]
    bits = np.packbits(torch.argmax(soft, dim=1).cpu().numpy().astype(np.uint8).reshape(-1))
    assert np.array_equal(bits, arrays["eval_argmax_bits"])


def test_phiseg_b32_three_training_steps_vs_reference_trajectory():
    """Train-step contract AT THE HEADLINE CONFIG (filters 32..192, 128 x 128, batch 32, default math): three
    steps of [forward, loss, zero_grad, backward, Adam(lr 1e-3, wd 1e-5)] against the real reference's trajectory
    (tests/golden/phiseg_full_b32_traj, tools/gen_golden.py b32traj): per-step loss terms, and behind every optimiser step the
    L2 norm and sampled entries of every parameter and BatchNorm buffer.
    Adam turns each gradient entry into a step of ~lr whatever its size, so an entry whose near-zero gradient has the other
    sign in two fp32 implementations moves by 2 lr: per entry the gate is that bound, the sharp statement is statistical -
    after the first step all but a small fraction of the sampled entries agree to 1e-6."""
    from unet_zoo_amd.optim import FusedAdam
    arrays, meta = G.load("phiseg_full_b32_traj")
    net, _ = _model(meta)
    net.train()
    net.enable_graphs(True)
    lr = meta["lr"]
    opt = FusedAdam(net, lr=lr, weight_decay=meta["weight_decay"])
    noise = G.bn_shadowed_biases([k for k, _, _ in G.spec_of(meta)])
    loss_tols = (2e-5, 1e-3, 2e-2)          # (step 2 measured: 4e-3 default math, 1.1e-2 fp32-MFMA only - two fp32 runs two Adam steps apart)
    for step, st in enumerate(meta["steps"]):
        x, mask, eps = _inputs(meta, step)
        net.forward(x, mask, training=True, eps=eps)
        loss = net.loss(mask)
        opt.zero_grad()
        loss.backward()
        rel = abs(float(loss.detach()) - st["loss"]) / abs(st["loss"])
        assert rel <= loss_tols[step], (step, float(loss.detach()), st["loss"])
        for k, v in st["loss_dict"].items():
            assert abs(float(net.loss_dict[k]) - v) <= 50 * loss_tols[step] * max(1.0, abs(v)), (step, k)
        assert sorted(k for k, p in net.named_parameters() if p.grad is None) == sorted(st["none_grads"])
        opt.step()
        assert net.check_bounds() == 0
        sd = net.state_dict()
        dev_all, flips = [], 0
        for k, sh, kd in G.spec_of(meta):
            if kd == "bn_nbt":
                continue
            v = sd[k].detach().reshape(-1).cpu().double().numpy()
            ref_norm = st["norms"][k]
            got = v[arrays["pick:" + k]]
            ref = arrays[f"step{step}:" + k].astype(np.float64)
            d = np.abs(got - ref)
            if kd in ("bn_rm", "bn_rv"):
                # running statistics: momentum 0.01 of batch statistics that agree to fp32 rounding in the first step; from the second
                # step on the two runs' parameters differ in the entries whose first Adam step flipped sign (2 lr each), and the
                # batch statistics of the deep, narrow levels follow (measured 9e-5 at step 1)
                assert d.max() <= (2e-5, 5e-4, 2e-3)[step] * max(1.0, float(np.abs(ref).max())), (step, k, float(d.max()))
                continue
            if k in noise:
                continue                      # conv biases in front of a training-mode BatchNorm: true gradient 0, Adam follows rounding noise
            assert d.max() <= 2.2 * lr * (step + 1) + 1e-6, (step, k, float(d.max()))
            # (a few per cent of the entries may sit 2 lr apart - sign flips of near-zero gradients: that bounds the norms' distance)
            assert abs(float(np.sqrt((v ** 2).sum())) - ref_norm) <= 1e-3 * ref_norm + 0.3 * lr * (step + 1) * np.sqrt(v.size), (step, k)
            dev_all.append(d)
        d = np.concatenate(dev_all)
        frac, flipped = float((d > 1e-6).mean()), float((d > 0.5 * lr).mean())
        print(f"step {step}: loss rel {rel:.1e}; sampled parameter entries off by > 1e-6: {100 * frac:.2f} %, by > lr/2: {100 * flipped:.2f} %, "
              f"median {np.median(d):.1e}, max {d.max():.1e}")
        if step == 0:
            # the first update is lr * sign(g) to 1e-8: identical wherever the two gradients agree in sign
            assert frac <= 0.03 and np.median(d) <= 1e-7, (step, frac)
        else:
            # later updates are lr * m / sqrt(v) of slightly different gradient histories: a smooth, small deviation in the bulk
            # (measured median 6e-6 at step 1) plus the entries whose sign flipped somewhere along the way
            assert np.median(d) <= 5e-5 * step and flipped <= 0.05 * (step + 1), (step, float(np.median(d)), flipped)
    nbt = [int(v) for k, v in net.state_dict().items() if k.endswith("num_batches_tracked") and "upsampling_path.4" not in k]
    assert set(nbt) == {len(meta["steps"])} and meta["steps"][-1]["nbt"] == len(meta["steps"])


def test_phiseg_vs_live_oracle_other_seed():
    """Same seeded inputs through the HIP path and the CPU oracle (fresh seed, ragged batch of 3)."""
    filters, hw, B = [4, 8, 8, 8, 8, 8, 8], 64, 3
    from unet_zoo_amd.models.phiseg import PHISeg, phiseg_spec
    sd = oracle.deterministic_state_dict(phiseg_spec(1, 2, filters), seed=77)
    net = PHISeg(1, 2, filters, image_size=(1, hw, hw))
    net.load_state_dict(sd)
    net.train()
    shapes = oracle.phiseg_eps_shapes(B, hw, hw)
    x, mask, eps = oracle.synthetic_batch(B, hw, hw, seed=5, eps_shapes=shapes + shapes)
    dev = torch.device("cuda", 0)
    s = net.forward(torch.from_numpy(x).to(dev), torch.from_numpy(mask).to(dev), training=True,
                    eps=[torch.from_numpy(e).to(dev) for e in eps])
    loss = net.loss(torch.from_numpy(mask).to(dev))
    loss.backward()
    lv = G.leaves(sd)
    e = [torch.from_numpy(a) for a in eps]
    out = oracle.phiseg_forward(lv, torch.from_numpy(x), torch.from_numpy(mask), dict(posterior=e[:5], prior=e[5:]))
    total, _ = oracle.phiseg_loss(out, torch.from_numpy(mask))
    total.backward()
    assert abs(float(loss) - float(total)) <= 2e-5 * abs(float(total))
    for l in range(5):
        assert G.maxabs(s[l].cpu().numpy(), out["s"][l].detach().numpy()) <= 1e-4
    noise = G.bn_shadowed_biases(lv.keys())
    for k, p in net.named_parameters():
        if k in noise:
            continue
        ref = lv[k].grad
        assert (p.grad is None) == (ref is None), k
        if ref is not None:
            assert G.maxabs(p.grad.cpu().numpy(), ref.numpy()) <= 5e-3 * (1e-3 + float(ref.abs().max())), k


def test_phiseg_b32_gradients_vs_fp64_reference():
    """Headline configuration (filters 32..192, 128x128, batch 32, default conv math = split-fp16 kernels on the large
    layers) against the REAL reference evaluated in double precision (tools/gen_golden.py `f64`: net.double(), same
    weights / inputs / noise), next to the reference's own fp32 run.  For every parameter tensor the fixture holds up to
    256 sampled gradient entries in fp64 and in the reference's fp32.  Gate: the HIP gradients must be as close to the
    real-valued gradient as the reference's fp32 arithmetic is.  Per tensor e = ||g - g64|| / ||g64|| over the sampled
    entries; compared are the DISTRIBUTIONS over the 368 tensors, because rounding noise through 30+ stacked batch
    normalisations is chaotic - for a given tensor either implementation can be the unlucky one (measured here: HIP median
    7.8e-4 / p90 4.3e-3 / max 6.3e-3, reference fp32 median 4.8e-4 / p90 1.1e-2 / max 2.8e-2: a narrower distribution with a
    higher centre).  Gates: median within 2.5x, 90th percentile and maximum within 1.5x of the reference's own; logits within
    1e-4 of fp64 and within 3x the reference's logit error."""
    arrays, meta = G.load("phiseg_full_b32_f64")
    net, _ = _model(meta)
    net.train()
    x, mask, eps = _inputs(meta, 0)
    s = net.forward(x, mask, training=True, eps=eps)
    loss = net.loss(mask)
    loss.backward()
    assert net.check_bounds() == 0
    assert abs(float(loss) - meta["loss64"]) <= 2e-5 * abs(meta["loss64"])
    idx = arrays["s_idx"]
    for l in range(5):
        mine = s[l].cpu().numpy().reshape(-1)[idx].astype(np.float64)
        e_hip, e_ref = np.abs(mine - arrays[f"s{l}_f64"]).max(), np.abs(arrays[f"s{l}_f32"].astype(np.float64) - arrays[f"s{l}_f64"]).max()
        assert e_hip <= 1e-4 and e_hip <= 3.0 * e_ref + 2e-5, (l, e_hip, e_ref)
    params = dict(net.named_parameters())
    noise = G.bn_shadowed_biases(params.keys())
    keys = [k[4:] for k in arrays.files if k.startswith("g64:") and k[4:] not in noise]
    eh, er = [], []
    for k in keys:
        g64 = arrays["g64:" + k]
        mine = params[k].grad.reshape(-1)[torch.from_numpy(arrays["i:" + k])].cpu().numpy().astype(np.float64)
        nrm = np.linalg.norm(g64) + 1e-300
        eh.append(np.linalg.norm(mine - g64) / nrm)
        er.append(np.linalg.norm(arrays["g32:" + k].astype(np.float64) - g64) / nrm)
    eh, er = np.array(eh), np.array(er)
    bad = [(keys[i], eh[i], er[i]) for i in range(len(keys)) if eh[i] > 3.0 * er[i] + 2e-4]
    print(f"b32 gradients vs fp64 reference: HIP median {np.median(eh):.2e} p90 {np.percentile(eh, 90):.2e} max {eh.max():.2e} | reference fp32 "
          f"median {np.median(er):.2e} p90 {np.percentile(er, 90):.2e} max {er.max():.2e} | tensors beyond 3x + 2e-4: {len(bad)} of {len(keys)}")
    assert np.median(eh) <= 2.5 * np.median(er), (np.median(eh), np.median(er))
    assert np.percentile(eh, 90) <= 1.5 * np.percentile(er, 90), (np.percentile(eh, 90), np.percentile(er, 90))
    assert eh.max() <= 1.5 * er.max(), (eh.max(), er.max())
    # Per tensor: a systematic error in ONE layer's gradient - which the norm gates of the digest tests (1e-2 of a
    # tensor's norm) would let through - shows here as a tensor far beyond what the reference's own fp32 arithmetic does to it.  Rounding
    # through 30+ stacked normalisations is chaotic, so for some tensors either implementation is the unlucky one.  Measured round 6, tensors
    # beyond 3x the reference's own error + 2e-4: 15 of 368 in the default mode, 50 with fp32 MFMA only, 52 with the split forced
    # everywhere; the worst single tensor 14.7x (a BatchNorm bias of the posterior's up path at 1.6e-2 of its norm against the reference's
    # 1.1e-3).  Gates: at most 15 % of the tensors beyond 3x + 2e-4, none beyond 25x + 2e-3 - a 1 % systematic error in a layer whose
    # reference error is 1e-4 is 100x.
    assert len(bad) <= 0.15 * len(keys), (len(bad), bad[:10])
    worst = [(keys[i], eh[i], er[i]) for i in range(len(keys)) if eh[i] > 25.0 * er[i] + 2e-3]
    assert not worst, worst


@pytest.mark.parametrize("fixture", ["phiseg_mid", "phiseg_full_digest"])
def test_phiseg_accuracy_vs_fp64_ground_truth(fixture):
    """Both the HIP path and the reference's fp32 CPU arithmetic are approximations of the same real-valued
    graph.  Against an fp64 evaluation of the oracle, the HIP path must be as accurate as the fp32 CPU
    path itself (logits and every parameter gradient) - i.e. the remaining HIP-vs-reference differences
    are fp32 rounding, not algorithmic.  `phiseg_full_digest` is the BASELINE architecture (filters 32..192, 128x128,
    batch 2): there the default mode routes the large layers to the split-fp16 kernels, so this is also their
    end-to-end accuracy gate at a batch size where the deepest normalisation sees 8 values per channel.  At that size the
    per-tensor errors of ANY fp32 implementation are dominated by chaotic amplification of rounding noise (measured at
    batch 2 / 8 with tools/diag_f64_full.py: the CPU reference is 10x worse than the HIP path on some sub-networks and
    10x better on others, changing with the batch), so the per-tensor bound lives in
    test_phiseg_b32_gradients_vs_fp64_reference (headline batch, real reference in fp64) and the gates here are the
    distribution's median and a cap on the worst tensor."""
    arrays, meta = G.load(fixture)
    net, sd0 = _model(meta)
    net.train()
    x, mask, eps = _inputs(meta, 0)
    s = net.forward(x, mask, training=True, eps=eps)
    loss = net.loss(mask)
    loss.backward()

    def cpu(dtype):
        lv = G.leaves({k: (v.to(dtype) if v.dtype.is_floating_point else v) for k, v in sd0.items()})
        e = [t.cpu().to(dtype) for t in eps]
        out = oracle.phiseg_forward(lv, x.cpu().to(dtype), mask.cpu().to(dtype), dict(posterior=e[:5], prior=e[5:]))
        total, _ = oracle.phiseg_loss(out, mask.cpu().to(dtype))
        total.backward()
        return out, {k: v.grad for k, v in lv.items() if v.requires_grad and v.grad is not None}

    o32, g32 = cpu(torch.float32)
    o64, g64 = cpu(torch.float64)
    for l in range(5):
        e_hip = float((s[l].cpu().double() - o64["s"][l]).abs().max())
        e_cpu = float((o32["s"][l].double() - o64["s"][l]).abs().max())
        assert e_hip <= 2.0 * e_cpu + 2e-5, (l, e_hip, e_cpu)
        assert e_hip <= 2e-4
    noise = G.bn_shadowed_biases(g64.keys())
    rh, rc, keys = [], [], []
    for k, p in net.named_parameters():
        if k in noise or k not in g64:
            continue
        sc = float(g64[k].abs().max()) + 1e-12
        rh.append(float((p.grad.cpu().double() - g64[k]).abs().max()) / sc)
        rc.append(float((g32[k].double() - g64[k]).abs().max()) / sc)
        keys.append(k)
    rh, rc = np.array(rh), np.array(rc)
    worst = int(np.argmax(rh / (3.0 * rc + 2e-5)))
    print(f"{fixture}: grad err vs fp64 rel. to tensor max  HIP median {np.median(rh):.2e} max {rh.max():.2e} | fp32 CPU median "
          f"{np.median(rc):.2e} max {rc.max():.2e} | worst per-tensor ratio {rh[worst] / (3.0 * rc[worst] + 2e-5):.2f} at {keys[worst]}")
    assert np.median(rh) <= 2.0 * np.median(rc) + 1e-6, (np.median(rh), np.median(rc))
    assert rh.max() <= max(3.0 * rc.max() + 1e-4, 0.15), (rh.max(), rc.max())


@pytest.mark.parametrize("lanes", ["1", "4"])
def test_graph_replay_is_bit_identical_to_eager(lanes, monkeypatch):
    """hipGraph replay (single lane and forked capture lanes) must reproduce the eager tape bit for
    bit: same kernels, same accumulation order, only the launch mechanism / overlap differs."""
    from unet_zoo_amd.optim import FusedAdam
    monkeypatch.setenv("UZ_LANES", lanes)
    _, meta = G.load("phiseg_mid")
    results = []
    for graphs in (False, True):
        net, _ = _model(meta)
        net.enable_graphs(graphs)
        opt = FusedAdam(net, lr=1e-3, weight_decay=1e-5)
        losses = []
        for step in range(5):                       # graph mode: step 0 eager warm-up, step 1 capture, 2.. replay
            x, mask, eps = _inputs(meta, step % 3)
            net.forward(x, mask, training=True, eps=eps)
            loss = net.loss(mask)
            net.zero_grad()
            loss.backward()
            opt.step()
            losses.append(float(loss))
        results.append((losses, {k: v.detach().cpu().clone() for k, v in net.state_dict().items()}))
    (l0, s0), (l1, s1) = results
    assert l0 == l1
    for k in s0:
        assert torch.equal(s0[k], s1[k]), k


@pytest.mark.parametrize("lanes", ["2", "3"])
def test_two_replayed_runs_at_the_headline_size_agree_bit_for_bit(lanes, monkeypatch):
    """Run-to-run determinism where the kernels really overlap: two identical PHiSeg 7/5 nets (filters 32..192, batch 32, 128 x 128)
    stepped in lockstep with hipGraph replay on several dependency lanes must hold identical parameters after every step.  (Round 4:
    one build of the bilinear backward kernel made exactly this fail - every gradient off by ~1e-4 after the first replayed step, only
    with kernels in flight beside it, never in isolation; tools/diag_dp_race.py is the long form of this test, with data parallelism.)"""
    from unet_zoo_amd.models.phiseg import PHISeg
    from unet_zoo_amd.optim import FusedAdam
    from unet_zoo_amd.synthetic import synthetic_batch
    monkeypatch.setenv("UZ_LANES", lanes)
    B = 32
    x, m, _ = synthetic_batch(B, 128, 128, seed=5)
    x, m = torch.from_numpy(x).to("cuda"), torch.from_numpy(m).to("cuda")
    g = torch.Generator(device="cuda").manual_seed(7)
    noise = [torch.randn(s_, generator=g, device="cuda") for s_ in [(B, 2, 2 << k, 2 << k) for k in range(5)] * 2]
    nets = []
    for _ in range(2):
        torch.manual_seed(1)
        net = PHISeg(1, 2, [32, 64, 128, 192, 192, 192, 192], latent_levels=5, image_size=(1, 128, 128))
        net.train()
        net.enable_graphs(True)
        nets.append((net, FusedAdam(net, lr=1e-3, weight_decay=1e-5)))
    for step in range(4):
        for net, opt in nets:
            net.forward(x, m, training=True, eps=noise)
            loss = net.loss(m)
            opt.zero_grad()
            loss.backward()
            opt.step()
        torch.cuda.synchronize()
        assert torch.equal(nets[0][0]._ptab.gflat, nets[1][0]._ptab.gflat), step
        assert torch.equal(nets[0][0]._ptab.pflat, nets[1][0]._ptab.pflat), step


def test_a_tuned_schedule_leaves_the_training_trajectory_bit_identical(monkeypatch):
    """Engine.tune_schedule at the headline size: the profile-guided rounds really produce other schedules (measured per-op costs on the
    plan's ops, a tape time per round), and a net stepped under the tuned schedule holds exactly the parameters and gradients of an
    untuned twin stepped in lockstep - a schedule only chooses among orders the DAG allows."""
    from unet_zoo_amd.models.phiseg import PHISeg
    from unet_zoo_amd.optim import FusedAdam
    from unet_zoo_amd.synthetic import synthetic_batch
    monkeypatch.setenv("UZ_LANES", "3")
    B = 32
    x, m, _ = synthetic_batch(B, 128, 128, seed=5)
    x, m = torch.from_numpy(x).to("cuda"), torch.from_numpy(m).to("cuda")
    g = torch.Generator(device="cuda").manual_seed(7)
    noise = [torch.randn(s_, generator=g, device="cuda") for s_ in [(B, 2, 2 << k, 2 << k) for k in range(5)] * 2]
    nets = []
    for _ in range(2):
        torch.manual_seed(1)
        net = PHISeg(1, 2, [32, 64, 128, 192, 192, 192, 192], latent_levels=5, image_size=(1, 128, 128))
        net.train()
        net.enable_graphs(True)
        nets.append((net, FusedAdam(net, lr=1e-3, weight_decay=1e-5)))
    if nets[0][0].replay_mode != "lanes":
        pytest.skip("tune_schedule() belongs to the lane replay")

    def one(net, opt, update=True):
        net.forward(x, m, training=True, eps=noise)
        loss = net.loss(m)
        opt.zero_grad()
        loss.backward()
        if update:
            opt.step()
    for net, opt in nets:
        one(net, opt)
    tuned, topt = nets[1]
    plan = tuned._cur
    before = {w: [(id(o), o["lane"]) for o in ops] for w, ops in (("fwd", plan.fwd_ops), ("bwd", plan.bwd_ops))}
    calls = [0]

    def tuned_step():                                 # no optimizer step: the twins stay in lockstep
        calls[0] += 1
        one(tuned, topt, update=False)
    res = tuned.tune_schedule(tuned_step, rounds=2, samples=2, validate=2)
    hist = res["tape_us"]
    assert set(hist) == {"fwd", "bwd"} and all(len(v) == 3 and min(v) > 100.0 for v in hist.values()), res
    assert res["kept"] in ("initial", "tuned") and set(res["step_ms"]) == {"initial", "tuned"}, res
    after = {w: [(id(o), o["lane"]) for o in ops] for w, ops in (("fwd", plan.fwd_ops), ("bwd", plan.bwd_ops))}
    print("tuning:", res, "schedule changed:", {w: before[w] != after[w] for w in before})
    for _ in range(calls[0]):                        # the twin sees the same batches (BatchNorm running statistics)
        one(nets[0][0], nets[0][1], update=False)
    for step in range(3):
        for net, opt in nets:
            one(net, opt)
        torch.cuda.synchronize()
        assert torch.equal(nets[0][0]._ptab.gflat, nets[1][0]._ptab.gflat), step
        assert torch.equal(nets[0][0]._ptab.pflat, nets[1][0]._ptab.pflat), step


def test_fused_latent_heads_leave_the_training_step_bit_identical(monkeypatch):
    """Plan.latent_heads (one op per direction for the two heads + sampling tail of every SampleZBlock, phiseg.py:95-105) against the
    plan with the separate ops (UZ_FUSE_HEADS=0): same loss, same gradients, same parameters after three Adam steps, bit for bit, and
    the same decoded sample - the fused kernels keep the arithmetic order of the ops they replace."""
    from unet_zoo_amd.models.phiseg import PHISeg
    from unet_zoo_amd.optim import FusedAdam
    from unet_zoo_amd.synthetic import synthetic_batch
    B = 4
    x, m, _ = synthetic_batch(B, 64, 64, seed=11)
    x, m = torch.from_numpy(x).to("cuda"), torch.from_numpy(m).to("cuda")
    g = torch.Generator(device="cuda").manual_seed(3)
    noise = [torch.randn(s_, generator=g, device="cuda") for s_ in [(B, 2, 1 << k, 1 << k) for k in range(5)] * 2]      # deepest level first: 1 x 1 ... 16 x 16
    runs = []
    monkeypatch.setenv("UZ_HEADS_PAR", "0")          # the sequential forward keeps the separate ops' arithmetic order (the channel-parallel form: test_ops_gpu.py)
    for fuse in ("1", "0"):
        monkeypatch.setenv("UZ_FUSE_HEADS", fuse)
        torch.manual_seed(1)
        net = PHISeg(1, 2, [8, 16, 32, 48, 48, 48, 48], latent_levels=5, image_size=(1, 64, 64))
        net.train()
        opt = FusedAdam(net, lr=1e-3, weight_decay=1e-5)
        losses = []
        for step in range(3):
            net.forward(x, m, training=True, eps=noise)
            loss = net.loss(m)
            opt.zero_grad()
            loss.backward()
            opt.step()
            losses.append(float(loss))
        codes = [o["code"] for o in net._cur.fwd_ops]
        assert ("UZ_OP_LATENT_HEADS_FWD" in codes) == (fuse == "1")
        torch.cuda.synchronize()
        runs.append((losses, net._ptab.gflat.clone(), net._ptab.pflat.clone(), [t.clone() for t in net.posterior_mu + net.posterior_sigma + net.prior_sigma]))
    assert runs[0][0] == runs[1][0]
    assert torch.equal(runs[0][1], runs[1][1]) and torch.equal(runs[0][2], runs[1][2])
    for a, b in zip(runs[0][3], runs[1][3]):
        assert torch.equal(a, b)


def test_a_tape_replayed_under_another_weight_gradient_grid_refuses_to_run():
    """uz_set_wgrad_target is process state that sizes the weight-gradient slab buffers when a plan is built AND the
    grids when its tape runs.  A caller that bypasses NativeModel._run (which restores the model's setting in front of every tape) and
    replays the backward tape under another target would write up to twice the allocated slabs and reduce the wrong number - silently.
    The op now carries the slab count its buffer was sized for (UZ_OP_CONV_BWD_WEIGHT i[12]) and the tape runner refuses."""
    from unet_zoo_amd import _ffi
    if os.environ.get("UZ_WGS_TARGET"):
        pytest.skip("UZ_WGS_TARGET pins the target")
    if _ffi.lib().uz_get_conv_math() in (0, 3):
        pytest.skip("only the split-path weight gradients are cut by the workgroup target")
    arrays, meta = G.load("phiseg_full_b32_digest")
    net, _ = _model(meta)
    net.train()
    x, mask, eps = _inputs(meta, 0)
    net.forward(x, mask, training=True, eps=eps)
    loss = net.loss(mask)
    loss.backward()
    torch.cuda.synchronize()
    plan, L = net._cur, _ffi.lib()
    sized = [o["i"][12] for o in plan.bwd_ops if o["code"] == "UZ_OP_CONV_BWD_WEIGHT" and o["i"][11]]
    assert len(sized) > 50 and all(n > 0 for n in sized)
    before = L.uz_get_wgrad_target()
    try:
        L.uz_set_wgrad_target(256 if before != 256 else 128)
        with pytest.raises(_ffi.UzError, match="the plan sized"):
            plan.run("bwd", net._stream())
    finally:
        L.uz_set_wgrad_target(before)
        torch.cuda.synchronize()
    plan.run("bwd", net._stream())                             # ... and runs again under the setting it was built with
    torch.cuda.synchronize()
    assert net.check_bounds() == 0
