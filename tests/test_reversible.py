"""Reversible blocks: ReversibleSequence (torchlayers.py:55-82) around revtorch's additive coupling, with
activation recomputation in the backward tape.  revtorch==0.2.0 is not vendored and cannot be installed here, so the oracle
restates its published algorithm and this row's PARITY IS UNPINNED by any reference output (oracle/refgraph.py:rev_sequence);
what is tested: the experiment surface (the reference's own *_rev_* files construct and train), the state_dict key scheme, the
memory saving, and HIP-vs-oracle agreement of forward, loss, gradients and BatchNorm buffers."""
import glob
import os

import numpy as np
import pytest
import torch

import oracle
import unet_zoo_amd  # noqa: F401
from tests import _golden as G
from unet_zoo_amd import train_model as TM

REF_EXP = "/root/reference/models/experiments"
NF7 = [32, 64, 128, 192, 192, 192, 192]


@pytest.mark.skipif(not os.path.isdir(REF_EXP), reason="reference tree not present")
def test_every_reversible_experiment_file_of_the_reference_constructs():
    files = sorted(f for f in glob.glob(os.path.join(REF_EXP, "*.py")) if "rev" in os.path.basename(f))
    assert len(files) >= 14
    built = 0
    for f in files:
        cfg = TM.load_experiment(f)
        assert cfg.use_reversible is True
        if not hasattr(cfg, "image_size"):                 # reversible_unet.py: the reference's harness cannot drive it either
            net = cfg.model(cfg.input_channels, cfg.n_classes, cfg.filter_channels, reversible=True, device="cpu")
        else:
            if max(cfg.image_size) > 256:                  # 384 / 512 px UZH configs: same code path, skip the big host buffers
                continue
            net = TM.UNetModel(cfg).net
        assert net.reversible and any(".sequence.reversible_blocks." in k for k in net.state_dict())
        built += 1
    assert built >= 12


def test_reversible_state_dict_keys_and_counts():
    from unet_zoo_amd.models import PHISeg
    net = PHISeg(1, 2, NF7, image_size=(1, 128, 128), reversible=True, device="cpu")
    keys = list(net.state_dict())
    p = "posterior.contracting_path.1.layers.1"
    assert p + ".inital_conv.convolution.0.weight" in keys                      # 32 -> 64: 1x1 widening unit (torchlayers.py:63-66)
    assert tuple(net.state_dict()[p + ".inital_conv.convolution.0.weight"].shape) == (64, 32, 1, 1)
    for i in range(3):
        for blk in ("f_block", "g_block"):
            w = net.state_dict()[f"{p}.sequence.reversible_blocks.{i}.{blk}.0.convolution.0.weight"]
            assert tuple(w.shape) == (32, 32, 3, 3)                               # F, G act on half the channels
    assert "posterior.contracting_path.0.layers.0.inital_conv.convolution.0.weight" in keys      # 3 -> 32
    assert "posterior.sample_z_path.0.conv.0.inital_conv.convolution.0.weight" not in keys        # 192 -> 192: nn.Identity
    assert "likelihood.likelihood_ups_path.0.inital_conv.convolution.0.weight" in keys
    assert "likelihood.likelihood_post_ups_path.0.1.convolution.0.convolution.0.weight" in keys   # increase_resolution stays a Conv2DSequence


def test_reversible_plan_saves_memory_and_keeps_every_dependency():
    """The forward pass of a reversible sequence stores only its output; block inputs are recomputed in the backward tape
    into pooled scratch (README.md:4-5: the paper's memory claim).  Arena of the headline architecture at the reference's
    batch size 12 (phiseg_rev_7_5_12.py): at least 10 % below the non-reversible plan."""
    from tests.test_host_cpu import _check_lane_schedule
    from unet_zoo_amd.models import PHISeg
    mb = {}
    for rev in (False, True):
        net = PHISeg(1, 2, NF7, image_size=(1, 128, 128), reversible=rev, device="cpu")
        plan = net._build(12, 128, 128, True, True)
        mb[rev] = plan.summary()["arena_MB"]
    assert mb[True] <= 0.90 * mb[False], mb
    small = PHISeg(1, 2, [4, 8, 8, 8, 8, 8, 8], image_size=(1, 64, 64), reversible=True, device="cpu")
    plan = small._build(2, 64, 64, True, True)
    for which, ops in (("fwd", plan.fwd_ops), ("bwd", plan.bwd_ops)):
        pairs, _, _ = _check_lane_schedule(plan, which, ops)
        assert pairs > 100
    assert sum(o["code"] == "UZ_OP_ADD_VIEWS" for o in plan.bwd_ops) > 50        # x2 = y2 - G(y1), x1 = y1 - F(x2) per block


def test_oracle_reversible_block_is_invertible():
    """The restated coupling is exactly invertible: x2 = y2 - G(y1), x1 = y1 - F(x2) (eval-mode BN: F, G deterministic)."""
    from unet_zoo_amd._modtree import rev_sequence_spec
    sd = oracle.deterministic_state_dict(rev_sequence_spec("s", 8, 8, 2), seed=5)
    x = torch.randn(2, 8, 6, 6)
    y = oracle.rev_sequence(sd, "s", x, bn_train=False)
    cur = y
    for i in reversed(range(2)):
        b = f"s.sequence.reversible_blocks.{i}"
        y1, y2 = torch.chunk(cur, 2, dim=1)
        x2 = y2 - oracle.refgraph.conv_unit(sd, b + ".g_block.0", y1, False)
        x1 = y1 - oracle.refgraph.conv_unit(sd, b + ".f_block.0", x2, False)
        cur = torch.cat([x1, x2], dim=1)
    assert float((cur - x).abs().max()) <= 1e-5 * float(x.abs().max())


# ----------------------------------------------------------------------------- device parity vs the oracle
def _small_weights(spec, seed):
    """deterministic weights with BN gammas scaled down: additive couplings grow activations by (1 + gamma) per block."""
    sd = oracle.deterministic_state_dict(spec, seed=seed)
    for k, v in sd.items():
        if k.endswith("convolution.1.weight"):
            sd[k] = v * 0.3
    return sd


@pytest.mark.gpu
def test_reversible_phiseg_vs_oracle_and_trains():
    from unet_zoo_amd.models.phiseg import PHISeg, phiseg_spec
    from unet_zoo_amd.optim import FusedAdam
    dev = torch.device("cuda", 0)
    filters, hw, B = [8, 16, 16, 16, 16, 16, 16], 128, 4
    sd0 = _small_weights(phiseg_spec(1, 2, filters, reversible=True), 31)
    net = PHISeg(1, 2, filters, image_size=(1, hw, hw), reversible=True)
    net.load_state_dict(sd0)
    net.train()
    shapes = oracle.phiseg_eps_shapes(B, hw, hw)
    x, mask, eps = oracle.synthetic_batch(B, hw, hw, seed=9, eps_shapes=shapes + shapes)
    xd, md = torch.from_numpy(x).to(dev), torch.from_numpy(mask).to(dev)
    s = net.forward(xd, md, training=True, eps=[torch.from_numpy(e).to(dev) for e in eps])
    loss = net.loss(md)
    bn_fwd = {k: v.detach().cpu().clone() for k, v in net.state_dict().items()}
    loss.backward()
    lv = G.leaves(sd0)
    e = [torch.from_numpy(a) for a in eps]
    out = oracle.phiseg_forward(lv, torch.from_numpy(x), torch.from_numpy(mask), dict(posterior=e[:5], prior=e[5:]))
    total, _ = oracle.phiseg_loss(out, torch.from_numpy(mask))
    total.backward()
    assert abs(float(loss) - float(total)) <= 5e-5 * abs(float(total)), (float(loss), float(total))
    for l in range(5):
        ref = out["s"][l].detach()
        assert G.maxabs(s[l].cpu().numpy(), ref.numpy()) <= 2e-4 * max(1.0, float(ref.abs().max())), l
    # BatchNorm buffers: one momentum update after forward (as the oracle's), a second one for the units of the reversible
    # blocks after backward (revtorch re-runs F and G while recomputing) and two tracked batches
    after = {k: v.detach().cpu() for k, v in net.state_dict().items()}
    sd1 = {k: v.detach() for k, v in lv.items()}
    sd2 = oracle.revtorch_second_bn_update(sd0, sd1)
    for k in sd0:
        if "running_" in k and "upsampling_path.4" not in k:
            sc = 1e-4 * max(1.0, float(sd1[k].abs().max()))
            assert G.maxabs(bn_fwd[k].numpy(), sd1[k].numpy()) <= sc, k
            assert G.maxabs(after[k].numpy(), sd2[k].numpy()) <= sc, k
        if k.endswith("num_batches_tracked") and "upsampling_path.4" not in k:
            assert int(after[k]) == (2 if ".reversible_blocks." in k else 1), k
    noise = G.bn_shadowed_biases(lv.keys())
    worst, wk = 0.0, None
    for k, p in net.named_parameters():
        ref = lv[k].grad
        assert (p.grad is None) == (ref is None), k
        if ref is None or k in noise:
            continue
        err = G.maxabs(p.grad.cpu().numpy(), ref.numpy()) / (1e-3 * float(ref.abs().max()) + float(ref.abs().max()) + 1e-30)
        if err > worst:
            worst, wk = err, k
    print(f"reversible PHiSeg: loss rel {abs(float(loss) - float(total)) / abs(float(total)):.1e}, worst gradient deviation {worst:.2e} at {wk}")
    assert worst <= 3e-2, (worst, wk)
    # trains: three more steps with hipGraph replay, loss finite and parameters moving
    net.enable_graphs(True)
    opt = FusedAdam(net, lr=1e-4, weight_decay=1e-5)
    before = net._ptab.pflat.clone()
    for _ in range(3):
        net.forward(xd, md, training=True)
        l2 = net.loss(md)
        opt.zero_grad()
        l2.backward()
        opt.step()
    assert torch.isfinite(l2) and not torch.equal(before, net._ptab.pflat)


@pytest.mark.gpu
def test_reversible_unet_vs_oracle():
    from unet_zoo_amd.models.unet import Unet, unet_spec
    dev = torch.device("cuda", 0)
    filters, B = [8, 16, 16, 16], 2
    sd0 = _small_weights(unet_spec(1, 2, filters, reversible=True), 33)
    net = Unet(1, 2, filters, reversible=True)
    net.load_state_dict(sd0)
    net.train()
    x, mask, _ = oracle.synthetic_batch(B, 128, 128, seed=4)
    pred = net.forward(torch.from_numpy(x).to(dev))
    loss = net.loss(torch.from_numpy(mask).to(dev))
    loss.backward()
    lv = G.leaves(sd0)
    ref = oracle.unet_forward(lv, torch.from_numpy(x), bn_train=True)
    rl = oracle.unet_loss(ref, torch.from_numpy(mask))
    rl.backward()
    assert G.maxabs(pred.cpu().numpy(), ref.detach().numpy()) <= 2e-4 * max(1.0, float(ref.abs().max()))
    assert abs(float(loss) - float(rl)) <= 5e-5 * abs(float(rl))
    noise = G.bn_shadowed_biases(lv.keys())
    for k, p in net.named_parameters():
        if k in noise:
            continue
        r = lv[k].grad
        assert G.maxabs(p.grad.cpu().numpy(), r.numpy()) <= 3e-2 * (1e-6 + float(r.abs().max())), k
