"""End-to-end parity of the native vanilla U-Net and Probabilistic U-Net against golden vectors generated
from the real reference (tools/gen_golden.py): predictions / features, loss, every parameter gradient,
which parameters keep grad None, and the parameters after three Adam steps."""
import os

import numpy as np
import pytest
import torch

import oracle
from tests import _golden as G

pytestmark = pytest.mark.gpu
DEV = lambda: torch.device("cuda", 0)  # noqa: E731


def test_unet_small_vs_reference_golden():
    from unet_zoo_amd.models.unet import Unet
    from unet_zoo_amd.optim import FusedAdam
    arrays, meta = G.load("unet_small")
    net = Unet(1, 2, meta["filters"])
    net.load_state_dict(oracle.deterministic_state_dict(G.spec_of(meta), seed=meta["weight_seed"]))
    assert list(net.state_dict().keys()) == [k for k, _, _ in G.spec_of(meta)]
    net.train()
    opt = FusedAdam(net, lr=1e-3, weight_decay=1e-5)
    theta0 = {k: v.detach().cpu().numpy().copy() for k, v in net.named_parameters()}
    for step, st in enumerate(meta["steps"]):
        x, mask, _ = oracle.synthetic_batch(meta["batch"], 128, 128, seed=20201004 + step)
        xd, md = torch.from_numpy(x).to(DEV()), torch.from_numpy(mask).to(DEV())
        pred = net.forward(xd)
        loss = net.loss(md)
        opt.zero_grad()
        loss.backward()
        assert abs(float(loss) - st["loss"]) <= (2e-5, 2e-4, 1e-3)[step] * abs(st["loss"]), (step, float(loss), st["loss"])
        if step == 0:
            assert G.maxabs(pred.cpu().numpy(), arrays["pred"]) <= 1e-4
            for k, p in net.named_parameters():
                ref = arrays["grad:" + k]
                assert G.maxabs(p.grad.cpu().numpy(), ref) <= 2e-3 * (1e-4 + float(np.abs(ref).max())), k
            g_hip = {k: p.grad.cpu().numpy().copy() for k, p in net.named_parameters()}
        opt.step()
        if step == 0:
            theta1 = {k: v.detach().cpu().numpy() for k, v in net.named_parameters()}
            G.check_first_adam_step(theta0, theta1, g_hip, {k: arrays["grad:" + k] for k in g_hip})
    for k, v in net.state_dict().items():
        d = np.abs(v.cpu().numpy().astype(np.float64) - arrays["final:" + k]).reshape(-1)
        assert d.max() <= 6.5e-3, k
        if d.size >= 64:
            assert np.median(d) <= 5e-4, (k, float(np.median(d)))


def test_probunet_small_vs_reference_golden():
    from unet_zoo_amd.models.probabilistic_unet import ProbabilisticUnet
    from unet_zoo_amd.optim import FusedAdam
    arrays, meta = G.load("probunet_small")
    net = ProbabilisticUnet(1, 2, meta["filters"], latent_dim=meta["latent_dim"], no_convs_fcomb=3, image_size=(1, 128, 128))
    net.load_state_dict(oracle.deterministic_state_dict(G.spec_of(meta), seed=meta["weight_seed"]))
    assert list(net.state_dict().keys()) == [k for k, _, _ in G.spec_of(meta)]
    net.train()
    opt = FusedAdam(net, lr=1e-3, weight_decay=1e-5)
    noise = G.bn_shadowed_biases(dict(net.named_parameters()).keys())
    theta0 = {k: v.detach().cpu().numpy().copy() for k, v in net.named_parameters()}
    for step, st in enumerate(meta["steps"]):
        x, mask, eps = oracle.synthetic_batch(meta["batch"], 128, 128, seed=20201004 + step,
                                              eps_shapes=[(meta["batch"], meta["latent_dim"])])
        xd, md = torch.from_numpy(x).to(DEV()), torch.from_numpy(mask).to(DEV())
        last = net.forward(xd, md, training=True)
        loss = net.loss(md, eps=torch.from_numpy(eps[0]).to(DEV()))
        opt.zero_grad()
        loss.backward()
        assert abs(float(loss) - st["loss"]) <= (2e-5, 3e-4, 3e-3)[step] * abs(st["loss"]), (step, float(loss), st["loss"])
        none = sorted(k for k, p in net.named_parameters() if p.grad is None)
        assert none == sorted(st["none_grads"])
        if step == 0:
            assert abs(float(net.kl_divergence_loss) - st["kl"]) <= 1e-3 * max(1.0, abs(st["kl"]))
            assert abs(float(net.reconstruction_loss) - st["recon"]) <= 2e-5 * abs(st["recon"])
            assert G.maxabs(last.cpu().numpy(), arrays["last_conv"]) <= 1e-4
            assert G.maxabs(net.unet_features.cpu().numpy(), arrays["unet_features"]) <= 1e-4
            assert G.maxabs(net.reconstruction.cpu().numpy(), arrays["reconstruction"]) <= 2e-4
            assert G.maxabs(net.posterior_latent_space.mean.cpu().numpy(), arrays["post_mu"]) <= 1e-4
            assert G.maxabs(net.posterior_latent_space.stddev.cpu().numpy(), arrays["post_sigma"]) <= 1e-4
            assert G.maxabs(net.prior_latent_space.mean.cpu().numpy(), arrays["prior_mu"]) <= 1e-4
            assert G.maxabs(net.prior_latent_space.stddev.cpu().numpy(), arrays["prior_sigma"]) <= 1e-4
            worst, wk = 0.0, None
            for k, p in net.named_parameters():
                if p.grad is not None and k not in noise:
                    ref = arrays["grad:" + k]
                    e = G.maxabs(p.grad.cpu().numpy(), ref) / (1e-3 + float(np.abs(ref).max()))
                    if e > worst:
                        worst, wk = e, k
            assert worst <= 2e-2, (worst, wk)
            g_hip = {k: p.grad.cpu().numpy().copy() for k, p in net.named_parameters() if p.grad is not None}
        opt.step()
        if step == 0:
            theta1 = {k: v.detach().cpu().numpy() for k, v in net.named_parameters()}
            G.check_first_adam_step(theta0, theta1, g_hip, {k: arrays["grad:" + k] for k in g_hip}, skip=noise)
            for k in st["none_grads"]:
                assert np.array_equal(theta1[k], theta0[k]), k
    for k, v in net.state_dict().items():
        if v.dtype.is_floating_point and k not in noise:
            assert G.maxabs(v.cpu().numpy(), arrays["final:" + k]) <= 6.5e-3, k
    # decode path: sample() / reconstruct() run the Fcomb tape on the cached features
    net.eval()
    with torch.no_grad():
        net.forward(xd, md, training=False)
        z = net.posterior_latent_space.mean
        rec = net.reconstruct(use_posterior_mean=True)
        sd = {k: v.detach().cpu().clone() for k, v in net.state_dict().items()}
        out = oracle.probunet_forward(sd, torch.from_numpy(x), torch.from_numpy(mask), bn_train=False)
        ref = oracle.probunet_fcomb(sd, out["unet_features"], out["posterior_mu"], bn_train=False)
    assert G.maxabs(z.cpu().numpy(), out["posterior_mu"].numpy()) <= 1e-4
    assert G.maxabs(rec.cpu().numpy(), ref.numpy()) <= 2e-4
    assert net.sample(testing=True).shape == rec.shape


@pytest.mark.parametrize("model", ["unet", "probunet"])
def test_graph_replay_is_bit_identical_to_eager(model):
    """hipGraph replay with dependency lanes reproduces the eager tapes bit for bit for the other two models too
    (different DAG shapes: skip connections only / two Gaussian encoders + a 1x1 fcomb chain + an L2 regulariser)."""
    from unet_zoo_amd.optim import FusedAdam
    name = "unet_small" if model == "unet" else "probunet_small"
    arrays, meta = G.load(name)
    results = []
    for graphs in (False, True):
        if model == "unet":
            from unet_zoo_amd.models.unet import Unet
            net = Unet(1, 2, meta["filters"])
        else:
            from unet_zoo_amd.models.probabilistic_unet import ProbabilisticUnet
            net = ProbabilisticUnet(1, 2, meta["filters"], latent_dim=meta["latent_dim"], no_convs_fcomb=3, image_size=(1, 128, 128))
        net.load_state_dict(oracle.deterministic_state_dict(G.spec_of(meta), seed=meta["weight_seed"]))
        net.train()
        net.enable_graphs(graphs)
        opt = FusedAdam(net, lr=1e-3, weight_decay=1e-5)
        losses = []
        for step in range(4):                       # graph mode: eager warm-up, capture, replay, replay
            if model == "unet":
                x, mask, _ = oracle.synthetic_batch(meta["batch"], 128, 128, seed=20201004 + step % 2)
                xd, md = torch.from_numpy(x).to(DEV()), torch.from_numpy(mask).to(DEV())
                net.forward(xd)
                loss = net.loss(md)
            else:
                x, mask, eps = oracle.synthetic_batch(meta["batch"], 128, 128, seed=20201004 + step % 2,
                                                      eps_shapes=[(meta["batch"], meta["latent_dim"])])
                xd, md = torch.from_numpy(x).to(DEV()), torch.from_numpy(mask).to(DEV())
                net.forward(xd, md, training=True)
                loss = net.loss(md, eps=torch.from_numpy(eps[0]).to(DEV()))
            opt.zero_grad()
            loss.backward()
            opt.step()
            losses.append(float(loss.detach()))
        results.append((losses, {k: v.detach().cpu().clone() for k, v in net.state_dict().items()}))
    (l0, s0), (l1, s1) = results
    assert l0 == l1
    for k in s0:
        assert torch.equal(s0[k], s1[k]), k


def test_models_with_different_weight_gradient_grids_alternate_in_one_process():
    """uz_set_wgrad_target is a PROCESS setting of the library that sizes slab buffers at plan time and grids at launch time; the Python
    face sets it per model (PHISeg 128, ProbabilisticUnet 192, Unet 256) in front of every plan build and every tape.  Two models with
    different values stepping in turn must each reproduce, bit for bit, what they compute alone (a stale setting would either trip the
    slab-count assertion of uz_conv_bwd_weight_ex or change the order of the slab sums)."""
    from unet_zoo_amd import _ffi
    from unet_zoo_amd.models.phiseg import PHISeg
    from unet_zoo_amd.models.unet import Unet
    from unet_zoo_amd.optim import FusedAdam
    assert PHISeg.wgrad_workgroups != Unet.wgrad_workgroups
    B = 8
    x, mask, _ = oracle.synthetic_batch(B, 128, 128, seed=77)
    xd, md = torch.from_numpy(x).to(DEV()), torch.from_numpy(mask).to(DEV())
    g = torch.Generator(device="cuda").manual_seed(5)
    noise = [torch.randn(B, 2, 2 << k, 2 << k, generator=g, device="cuda") for k in range(5)] * 2        # deepest level first: 2 x 2 ... 32 x 32

    def make(kind):
        torch.manual_seed(3)
        net = PHISeg(1, 2, [32, 64, 64, 64, 64, 64, 64], latent_levels=5, image_size=(1, 128, 128)) if kind == "phiseg" else Unet(1, 2, [32, 64, 64, 64])
        net.train(); net.enable_graphs(True)
        return net, FusedAdam(net, lr=1e-3, weight_decay=1e-5)

    def step(kind, net, opt):
        if kind == "phiseg":
            net.forward(xd, md, training=True, eps=noise)
        else:
            net.forward(xd)
        loss = net.loss(md)
        opt.zero_grad(); loss.backward(); opt.step()
        return float(loss.detach())
    alone = {}
    for kind in ("phiseg", "unet"):
        net, opt = make(kind)
        losses = [step(kind, net, opt) for _ in range(3)]
        torch.cuda.synchronize()
        alone[kind] = (losses, net._ptab.pflat.clone())
        assert _ffi.lib().uz_get_wgrad_target() == type(net).wgrad_workgroups or os.environ.get("UZ_WGS_TARGET")
    nets = {kind: make(kind) for kind in ("phiseg", "unet")}
    losses = {"phiseg": [], "unet": []}
    for _ in range(3):
        for kind in ("phiseg", "unet"):
            losses[kind].append(step(kind, *nets[kind]))
    torch.cuda.synchronize()
    for kind in ("phiseg", "unet"):
        assert losses[kind] == alone[kind][0], kind
        assert torch.equal(nets[kind][0]._ptab.pflat, alone[kind][1]), kind


def test_probunet_gradients_with_and_without_deferred_tables_agree_bit_for_bit(monkeypatch):
    """The table-driven deferred reductions (weight-gradient slabs, bias rows) add in the same order as the per-layer launches, so
    switching them off must not change one bit of the gradient - INCLUDING the regulariser's share: the L2 term adds to what the
    layers wrote (g += coeff w / |w|), the tables assign, so the tables have to run first (round 4 ran them last for a while and
    silently dropped the regulariser's gradient of every 3 x 3 weight; far below the gates of the golden tests)."""
    from unet_zoo_amd.models.probabilistic_unet import ProbabilisticUnet
    arrays, meta = G.load("probunet_small")
    grads = []
    for tables in ("1", "0"):
        monkeypatch.setenv("UZ_WGRAD_TABLE", tables)
        monkeypatch.setenv("UZ_DBIAS_TABLE", tables)
        net = ProbabilisticUnet(1, 2, meta["filters"], latent_dim=meta["latent_dim"], no_convs_fcomb=3, image_size=(1, 128, 128))
        net.load_state_dict(oracle.deterministic_state_dict(G.spec_of(meta), seed=meta["weight_seed"]))
        net.train()
        x, mask, eps = oracle.synthetic_batch(meta["batch"], 128, 128, seed=20201004, eps_shapes=[(meta["batch"], meta["latent_dim"])])
        xd, md = torch.from_numpy(x).to(DEV()), torch.from_numpy(mask).to(DEV())
        net.forward(xd, md, training=True)
        loss = net.loss(md, eps=torch.from_numpy(eps[0]).to(DEV()))
        net.zero_grad()
        loss.backward()
        torch.cuda.synchronize()
        codes = {o["code"] for o in net._cur.bwd_ops}
        assert ("UZ_OP_WGRAD_REDUCE_TABLE" in codes) == (tables == "1") and "UZ_OP_L2_NORMS_BWD" in codes
        grads.append(net._ptab.gflat.clone())
    assert torch.equal(grads[0], grads[1])
    # and the regulariser's share is really in there: without it the gradient differs
    assert float(grads[0].abs().sum()) > 0


def test_gradient_views_second_backward_and_foreign_gradients_like_torch():
    """Host semantics around the flat gradient buffer (train_model.py:160-179 relies on them like on any torch module): parameters'
    .grad are cached VIEWS of the buffer (same objects after every backward); a second backward() without zero_grad() accumulates;
    a gradient the user replaced by a tensor of their own is copied in by FusedAdam.step(), whose update then equals
    torch.optim.Adam's on the same numbers; zero_grad() detaches them again."""
    from unet_zoo_amd.models.unet import Unet
    from unet_zoo_amd.optim import FusedAdam
    torch.manual_seed(5)
    net = Unet(1, 2, [8, 16, 16, 16])
    net.train()
    x, mask, _ = oracle.synthetic_batch(2, 32, 32, seed=3)
    xd, md = torch.from_numpy(x).to(DEV()), torch.from_numpy(mask).to(DEV())
    opt = FusedAdam(net, lr=1e-3, weight_decay=1e-5)

    def backward():
        net.forward(xd)
        loss = net.loss(md)
        loss.backward()
    opt.zero_grad()
    backward()
    params = dict(net.named_parameters())
    views = {k: p.grad for k, p in params.items()}
    g1 = {k: v.clone() for k, v in views.items()}
    assert all(v.data_ptr() == net._ptab.gview(k).data_ptr() for k, v in views.items())
    backward()                                                  # no zero_grad: accumulates, like autograd
    for k, p in params.items():
        assert torch.allclose(p.grad, 2 * g1[k], rtol=1e-5, atol=1e-7), k
    opt.zero_grad()
    assert all(p.grad is None for p in params.values())
    backward()
    assert all(params[k].grad is views[k] for k in params)      # the cached view objects again
    for k in params:
        assert torch.equal(params[k].grad, g1[k]), k            # and the same numbers as the first pass (the tape overwrites)
    # a foreign gradient on one parameter; torch.optim.Adam on clones as the reference
    ref = {k: torch.nn.Parameter(p.detach().clone()) for k, p in params.items()}
    ropt = torch.optim.Adam(list(ref.values()), lr=1e-3, weight_decay=1e-5)
    key = next(k for k in params if k.endswith("weight") and params[k].dim() == 4)
    params[key].grad = g1[key] * 3.0                            # new storage, not a view of the flat buffer
    for k, p in ref.items():
        p.grad = (g1[k] * 3.0 if k == key else g1[k]).clone()
    opt.step()
    ropt.step()
    for k in params:
        assert torch.allclose(params[k].detach(), ref[k].detach(), rtol=1e-6, atol=1e-7), k
