"""The C twins of the C ABI (oracle/uz_cpu.c): scalar restatements with the HIP entry points' signatures minus the
stream.  CPU tier: every twin against the torch.nn.functional op / autograd formula the pinned oracle is made of.  GPU tier:
the HIP kernels against the twins through IDENTICAL argument lists (channel-slice views, accumulate flags and all)."""
import ctypes as C

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from tests import _twins as T


def rnd(*shape, seed=0, scale=1.0):
    return (np.random.default_rng(seed).standard_normal(shape) * scale).astype(np.float32)


def close(a, b, tol=2e-5):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return float(np.abs(a - b).max()) <= tol * max(1.0, float(np.abs(b).max()))


def test_twin_set_covers_the_hot_path():
    have = set(T.names())
    for n in ("conv_fwd", "conv_bwd_data", "conv_bwd_weight", "bn_relu_fwd", "bn_relu_bwd", "relu_bwd", "avgpool2_fwd", "avgpool2_bwd",
              "bilinear2x_fwd", "bilinear2x_bwd", "nearest_fwd", "nearest_bwd", "posterior_input", "latent_sample_fwd", "latent_sample_bwd",
              "kl_fwd", "kl_bwd", "residual_ce_fwd", "residual_ce_bwd", "sum_terms", "adam_step", "add_views"):
        assert "uz_cpu_" + n in have, n


@pytest.mark.parametrize("N,Cin,Cout,H,W,ks", [(2, 3, 5, 6, 7, 3), (1, 4, 2, 5, 5, 1)])
def test_conv_twins_vs_torch(N, Cin, Cout, H, W, ks):
    x, w, b = rnd(N, Cin + 2, H, W, seed=1), rnd(Cout, Cin, ks, ks, seed=2, scale=0.3), rnd(Cout, seed=3)
    xv = np.ascontiguousarray(x)                                           # the conv reads channels [1, 1 + Cin) of a wider buffer
    xt = torch.from_numpy(x[:, 1:1 + Cin].copy()).requires_grad_(True)
    wt, bt = torch.from_numpy(w).requires_grad_(True), torch.from_numpy(b).requires_grad_(True)
    yt = F.conv2d(xt, wt, bt, padding=ks // 2)
    dy = rnd(*yt.shape, seed=4)
    yt.backward(torch.from_numpy(dy))
    y = np.full((N, Cout + 1, H, W), np.nan, np.float32)
    xs = xv.reshape(-1)[H * W:]                                             # pointer to channel 1 of image 0
    T.call("uz_cpu_conv_fwd", xs, Cin, Cin + 2, w, b, y.reshape(-1)[H * W:], Cout, Cout + 1, N, H, W, ks, 0, None, None, None, None, 0)
    assert close(y[:, 1:], yt.detach().numpy()) and np.isnan(y[:, 0]).all()
    dx = np.zeros((N, Cin, H, W), np.float32)
    T.call("uz_cpu_conv_bwd_data", dy, Cout, Cout, w, dx, Cin, Cin, N, H, W, ks, 0, None, None, None, 0)
    T.call("uz_cpu_conv_bwd_data", dy, Cout, Cout, w, dx, Cin, Cin, N, H, W, ks, 1, None, None, None, 0)
    assert close(dx, 2 * xt.grad.numpy())
    dw, db = np.zeros_like(w), np.zeros_like(b)
    T.call("uz_cpu_conv_bwd_weight", xs, Cin, Cin + 2, dy, Cout, Cout, dw, db, N, H, W, ks, None, None, None, 0)
    assert close(dw, wt.grad.numpy()) and close(db, bt.grad.numpy())


@pytest.mark.parametrize("relu", [1, 0])
def test_bn_relu_twins_vs_torch(relu):
    N, Cc, H, W = 3, 4, 5, 6
    y, gam, bet = rnd(N, Cc, H, W, seed=1), rnd(Cc, seed=2) + 1.5, rnd(Cc, seed=3, scale=0.3)
    rm, rv = rnd(Cc, seed=4, scale=0.1), np.abs(rnd(Cc, seed=5)) + 0.5
    yt, gt, bt = (torch.from_numpy(v.copy()).requires_grad_(True) for v in (y, gam, bet))
    rmt, rvt = torch.from_numpy(rm.copy()), torch.from_numpy(rv.copy())
    at = F.batch_norm(yt, rmt, rvt, gt, bt, training=True, momentum=0.01, eps=1e-3)
    at = F.relu(at) if relu else at
    da = rnd(N, Cc, H, W, seed=6)
    at.backward(torch.from_numpy(da))
    a, save = np.zeros_like(y), np.zeros(2 * Cc, np.float32)
    T.call("uz_cpu_bn_relu_fwd", y, Cc, Cc, gam, bet, rm, rv, save, a, Cc, N, H, W, C.c_float(1e-3), C.c_float(0.01), 1, relu, None, None)
    assert close(a, at.detach().numpy()) and close(rm, rmt.numpy()) and close(rv, rvt.numpy())
    dyv, dg, dbt, dbias = np.zeros_like(y), np.zeros(Cc, np.float32), np.zeros(Cc, np.float32), np.zeros(Cc, np.float32)
    T.call("uz_cpu_bn_relu_bwd", da, Cc, y, Cc, Cc, gam, bet, save, dyv, Cc, dg, dbt, dbias, N, H, W, relu, None, None)
    assert close(dyv, yt.grad.numpy(), 1e-4) and close(dg, gt.grad.numpy(), 1e-4) and close(dbt, bt.grad.numpy(), 1e-4)
    assert float(np.abs(dbias).max()) <= 1e-3                                # sum of dy through a training-mode BN is zero up to rounding
    # eval mode normalises with the running statistics
    T.call("uz_cpu_bn_relu_fwd", y, Cc, Cc, gam, bet, rm, rv, save, a, Cc, N, H, W, C.c_float(1e-3), C.c_float(0.01), 0, relu, None, None)
    ev = F.batch_norm(torch.from_numpy(y), rmt, rvt, torch.from_numpy(gam), torch.from_numpy(bet), training=False, eps=1e-3)
    assert close(a, (F.relu(ev) if relu else ev).numpy())


@pytest.mark.parametrize("ac", [1, 0])
def test_resampling_twins_vs_torch(ac):
    N, Cc, H, W = 2, 3, 5, 7
    x = rnd(N, Cc, H, W, seed=1)
    xt = torch.from_numpy(x.copy()).requires_grad_(True)
    pt = F.avg_pool2d(xt, 2, 2, 0, ceil_mode=True)
    y = np.zeros(tuple(pt.shape), np.float32)
    T.call("uz_cpu_avgpool2_fwd", x, Cc, Cc, y, Cc, N, H, W, None, None)
    assert close(y, pt.detach().numpy())
    dy = rnd(*pt.shape, seed=2)
    pt.backward(torch.from_numpy(dy))
    dx = np.zeros_like(x)
    T.call("uz_cpu_avgpool2_bwd", dy, Cc, Cc, dx, Cc, N, H, W, 0)
    assert close(dx, xt.grad.numpy())
    xt.grad = None
    ut = F.interpolate(xt, scale_factor=2, mode="bilinear", align_corners=bool(ac))
    u = np.zeros(tuple(ut.shape), np.float32)
    T.call("uz_cpu_bilinear2x_fwd", x, Cc, Cc, u, Cc, N, H, W, ac, None, None)
    assert close(u, ut.detach().numpy())
    du = rnd(*ut.shape, seed=3)
    ut.backward(torch.from_numpy(du))
    T.call("uz_cpu_bilinear2x_bwd", du, Cc, Cc, dx, Cc, N, H, W, ac, 0)
    assert close(dx, xt.grad.numpy())
    xt.grad = None
    nt = F.interpolate(xt, size=[3 * H, 3 * W], mode="nearest")
    n = np.zeros(tuple(nt.shape), np.float32)
    T.call("uz_cpu_nearest_fwd", x, Cc, Cc, n, Cc, N, H, W, 3)
    assert np.array_equal(n, nt.detach().numpy())
    dn = rnd(*nt.shape, seed=4)
    nt.backward(torch.from_numpy(dn))
    T.call("uz_cpu_nearest_bwd", dn, Cc, Cc, dx, Cc, N, H, W, 3, 0)
    assert close(dx, xt.grad.numpy())


def test_latent_kl_ce_adam_twins_vs_oracle():
    import oracle
    N, per = 3, 40
    mu0, mu1 = rnd(N, per, seed=1), rnd(N, per, seed=2)
    p0, p1, eps = rnd(N, per, seed=3), rnd(N, per, seed=4), rnd(N, per, seed=5)
    for act in (0, 1):
        pt = torch.from_numpy(p0.copy()).requires_grad_(True)
        mt = torch.from_numpy(mu0.copy()).requires_grad_(True)
        st = torch.exp(pt) if act else F.softplus(pt)
        zt = mt + st * torch.from_numpy(eps)
        sig, z = np.zeros_like(p0), np.zeros_like(p0)
        T.call("uz_cpu_latent_sample_fwd", mu0, p0, eps, sig, z, N * per, act)
        assert close(sig, st.detach().numpy()) and close(z, zt.detach().numpy())
        dz, dsg = rnd(N, per, seed=6), rnd(N, per, seed=7)
        (zt * torch.from_numpy(dz)).sum().add((st * torch.from_numpy(dsg)).sum()).backward()
        dmu, dpre = np.zeros_like(p0), np.zeros_like(p0)
        T.call("uz_cpu_latent_sample_bwd", None, dsg, dz, eps, sig, dmu, dpre, N * per, act)
        assert close(dmu, mt.grad.numpy()) and close(dpre, pt.grad.numpy(), 1e-4)
    s0, s1 = np.abs(p0) + 0.1, np.abs(p1) + 0.1
    ts = [torch.from_numpy(v.copy()).requires_grad_(True) for v in (mu0, s0, mu1, s1)]
    kl = 4.0 * oracle.kl_two_gauss_with_diag_cov(*ts)
    out = np.zeros(1, np.float32)
    T.call("uz_cpu_kl_fwd", mu0, s0, mu1, s1, N, per, C.c_float(4.0), out)
    assert abs(out[0] - float(kl)) <= 1e-5 * abs(float(kl))
    (kl * 0.5).backward()
    g = [np.zeros_like(mu0) for _ in range(4)]
    T.call("uz_cpu_kl_bwd", mu0, s0, mu1, s1, N, per, C.c_float(4.0), np.array([0.5], np.float32), *g)
    for a, t in zip(g, ts):
        assert close(a, t.grad.numpy(), 1e-4)
    # residual multinoulli loss over 3 levels of logits, 3 classes
    L, K, Nb, H, W = 3, 3, 2, 4, 5
    logits = [rnd(Nb, K, H, W, seed=10 + l) for l in range(L)]
    mask = np.random.default_rng(3).integers(0, K, (Nb, 1, H, W)).astype(np.float32)
    lt = [torch.from_numpy(v.copy()).requires_grad_(True) for v in logits]
    acc, terms = None, [None] * L
    for l in reversed(range(L)):
        acc = lt[l] if acc is None else acc + lt[l]
        terms[l] = oracle.refgraph.multinoulli_loss(acc, torch.from_numpy(mask), K)
    ptr = (C.c_void_p * L)(*[v.ctypes.data for v in logits])
    losses = np.zeros(L, np.float32)
    T.call("uz_cpu_residual_ce_fwd", C.cast(ptr, C.c_void_p), L, K, mask, Nb, H, W, losses, None)
    assert close(losses, np.array([float(t) for t in terms]))
    (sum(terms) * 2.0).backward()
    grads = [np.zeros_like(v) for v in logits]
    gptr = (C.c_void_p * L)(*[v.ctypes.data for v in grads])
    T.call("uz_cpu_residual_ce_bwd", C.cast(ptr, C.c_void_p), C.cast(gptr, C.c_void_p), L, K, mask, Nb, H, W, np.array([2.0], np.float32))
    for a, t in zip(grads, lt):
        assert close(a, t.grad.numpy(), 1e-4)
    # Adam with L2 weight decay, two steps
    p, gr = rnd(50, seed=20), rnd(50, seed=21)
    pt = torch.nn.Parameter(torch.from_numpy(p.copy()))
    opt = torch.optim.Adam([pt], lr=1e-3, weight_decay=1e-5)
    m, v = np.zeros_like(p), np.zeros_like(p)
    for step in (1, 2):
        pt.grad = torch.from_numpy(gr * step)
        opt.step()
        T.call("uz_cpu_adam_step", p, (gr * step).astype(np.float32), m, v, 50, step, C.c_float(1e-3), C.c_float(0.9), C.c_float(0.999),
               C.c_float(1e-8), C.c_float(1e-5), C.c_float(1.0))
    assert close(p, pt.detach().numpy(), 1e-6)


# ----------------------------------------------------------------------------- HIP kernels vs the twins, same argument lists
def _both(name, args, outs):
    """Run uz_<name> on the device and uz_cpu_<name> on the host with the same argument list; numpy arrays are mirrored to the
    device, `outs` are the indices of the output arguments.  Returns [(hip, cpu)] per output."""
    from tests import _gpu as g
    dev_args, host_args = [], []
    for a in args:
        if isinstance(a, np.ndarray):
            dev_args.append(torch.from_numpy(a.copy()).to(g.dev()))
            host_args.append(a.copy())
        else:
            dev_args.append(a)
            host_args.append(a)
    g.call("uz_" + name, *dev_args)
    T.call("uz_cpu_" + name, *host_args)
    return [(dev_args[i].cpu().numpy(), host_args[i]) for i in outs]


@pytest.mark.gpu
def test_hip_conv_unit_vs_cpu_twins():
    from unet_zoo_amd import _ffi
    L = _ffi.lib()
    N, Cin, Cout, H, W = 2, 24, 40, 20, 16
    x, w, b, dy = rnd(N, Cin, H, W, seed=1), rnd(Cout, Cin, 3, 3, seed=2, scale=0.2), rnd(Cout, seed=3), rnd(N, Cout, H, W, seed=4)
    wsb = max(L.uz_conv_workspace(Cin, Cout, N, H, W, 3), L.uz_conv_bwd_weight_workspace(Cin, Cout, N, H, W, 3))
    ws = np.zeros(wsb // 4 + 16, np.float32)
    y = np.zeros((N, Cout, H, W), np.float32)
    (hy, cy), = _both("conv_fwd", [x, Cin, Cin, w, b, y, Cout, Cout, N, H, W, 3, 1, None, None, None, ws, wsb], [5])
    assert close(hy, cy)
    (hx, cx), = _both("conv_bwd_data", [dy, Cout, Cout, w, np.ones_like(x), Cin, Cin, N, H, W, 3, 1, None, None, ws, wsb], [4])
    assert close(hx, cx)
    (hw, cw), (hb, cb) = _both("conv_bwd_weight", [x, Cin, Cin, dy, Cout, Cout, np.zeros_like(w), np.zeros_like(b), N, H, W, 3, None, None, ws, wsb], [6, 7])
    assert close(hw, cw, 1e-4) and close(hb, cb, 1e-4)
    # BatchNorm + ReLU in training mode, forward and backward
    gam, bet = rnd(Cout, seed=5) + 1.5, rnd(Cout, seed=6, scale=0.3)
    rm, rv, save = np.zeros(Cout, np.float32), np.ones(Cout, np.float32), np.zeros(2 * Cout, np.float32)
    bws = np.zeros(L.uz_bn_workspace(Cout, N, H, W) // 4 + 16, np.float32)
    res = _both("bn_relu_fwd", [cy, Cout, Cout, gam, bet, rm, rv, save, np.zeros_like(cy), Cout, N, H, W, C.c_float(1e-3), C.c_float(0.01), 1, 1, None, bws], [8, 5, 6, 7])
    for h, c_ in res:
        assert close(h, c_, 1e-4)
    save_c = res[3][1]
    res = _both("bn_relu_bwd", [dy, Cout, cy, Cout, Cout, gam, bet, save_c, np.zeros_like(cy), Cout, np.zeros(Cout, np.float32), np.zeros(Cout, np.float32),
                                np.zeros(Cout, np.float32), N, H, W, 1, None, bws], [8, 10, 11])
    for h, c_ in res:
        assert close(h, c_, 2e-4)


@pytest.mark.gpu
def test_hip_resampling_and_losses_vs_cpu_twins():
    N, Cc, H, W = 3, 6, 12, 8
    x = rnd(N, Cc + 2, H, W, seed=1)
    xs = x.reshape(-1)
    for name, args, outs in (
        ("avgpool2_fwd", [xs, Cc, Cc + 2, np.zeros((N, Cc, H // 2, W // 2), np.float32), Cc, N, H, W, None, None], [3]),
        ("bilinear2x_fwd", [xs, Cc, Cc + 2, np.zeros((N, Cc, 2 * H, 2 * W), np.float32), Cc, N, H, W, 1, None, None], [3]),
        ("bilinear2x_bwd", [rnd(N, Cc, 2 * H, 2 * W, seed=2), Cc, Cc, np.ones((N, Cc + 2, H, W), np.float32), Cc + 2, N, H, W, 0, 1], [3]),
        ("nearest_bwd", [rnd(N, Cc, 4 * H, 4 * W, seed=3), Cc, Cc, np.zeros((N, Cc, H, W), np.float32), Cc, N, H, W, 4, 0], [3]),
    ):
        for h, c_ in _both(name, args, outs):
            assert close(h, c_, 1e-5), name
    per = 64
    mu0, s0, mu1, s1 = rnd(N, per, seed=4), np.abs(rnd(N, per, seed=5)) + 0.2, rnd(N, per, seed=6), np.abs(rnd(N, per, seed=7)) + 0.2
    (h, c_), = _both("kl_fwd", [mu0, s0, mu1, s1, N, per, C.c_float(16.0), np.zeros(1, np.float32)], [7])
    assert close(h, c_, 1e-5)
    res = _both("kl_bwd", [mu0, s0, mu1, s1, N, per, C.c_float(16.0), np.array([0.25], np.float32)] + [np.zeros_like(mu0) for _ in range(4)], [8, 9, 10, 11])
    for h, c_ in res:
        assert close(h, c_, 1e-4)


def _heads_case(N=2, Cin=12, ctot=15, L=2, H=6, W=8):
    h = rnd(N, ctot, H, W, seed=11)
    wm, ws_ = rnd(L, Cin, 1, 1, seed=12, scale=0.3), rnd(L, Cin, 1, 1, seed=13, scale=0.3)
    bm, bs = rnd(L, seed=14), rnd(L, seed=15)
    eps = rnd(N, L, H, W, seed=16)
    dmu, dpre = rnd(N, L, H, W, seed=17), rnd(N, L, H, W, seed=18)
    return h, wm, bm, ws_, bs, eps, dmu, dpre


@pytest.mark.parametrize("act", [0, 1])
def test_latent_heads_twins_vs_torch(act):
    """uz_cpu_latent_heads_* (oracle of uz_latent_heads_*: the tail of SampleZBlock.forward, phiseg.py:95-105, and its backward) against
    torch autograd on the reference's own expressions."""
    N, Cin, ctot, L, H, W = 2, 12, 15, 2, 6, 8
    h, wm, bm, ws_, bs, eps, dmu, dpre = _heads_case(N, Cin, ctot, L, H, W)
    hv = np.ascontiguousarray(h.reshape(-1)[1 * H * W:])                      # the heads read channels [1, 1 + Cin) of a wider buffer
    mu, pre, sg, z = (np.zeros((N, L, H, W), np.float32) for _ in range(4))
    T.call("uz_cpu_latent_heads_fwd", hv, Cin, ctot, wm, bm, ws_, bs, eps, mu, pre, sg, z, L, N, H, W, act)
    ht = torch.from_numpy(h[:, 1:1 + Cin].copy()).requires_grad_(True)
    tw = [torch.from_numpy(a.copy()).requires_grad_(True) for a in (wm, bm, ws_, bs)]
    mr = F.conv2d(ht, tw[0], tw[1])
    pr = F.conv2d(ht, tw[2], tw[3])
    sr = torch.exp(pr) if act else F.softplus(pr)
    zr = mr + sr * torch.from_numpy(eps)
    assert close(mu, mr.detach()) and close(sg, sr.detach()) and close(z, zr.detach()) and close(pre, pr.detach())
    (mr * torch.from_numpy(dmu) + pr * torch.from_numpy(dpre)).sum().backward()
    dh = np.full((N, ctot, H, W), 7.0, np.float32)
    dhv = dh.reshape(-1)[1 * H * W:]
    T.call("uz_cpu_latent_heads_bwd_data", dpre, dmu, L, ws_, wm, dhv, Cin, ctot, N, H, W, 0)
    assert close(dh[:, 1:1 + Cin], ht.grad) and np.all(dh[:, 0] == 7.0) and np.all(dh[:, 1 + Cin:] == 7.0)
    dws, dbs, dwm, dbm = np.zeros_like(ws_), np.zeros_like(bs), np.zeros_like(wm), np.zeros_like(bm)
    T.call("uz_cpu_latent_heads_bwd_weight", hv, Cin, ctot, dpre, dmu, L, dws, dbs, dwm, dbm, N, H, W, None, 0)
    for got, ref in ((dwm, tw[0].grad), (dbm, tw[1].grad), (dws, tw[2].grad), (dbs, tw[3].grad)):
        assert close(got, ref, 1e-4)


@pytest.mark.gpu
def test_hip_latent_heads_vs_cpu_twins():
    from unet_zoo_amd import _ffi
    N, Cin, ctot, L, H, W = 2, 12, 15, 2, 6, 8
    h, wm, bm, ws_, bs, eps, dmu, dpre = _heads_case(N, Cin, ctot, L, H, W)
    hv = np.ascontiguousarray(h.reshape(-1)[1 * H * W:])
    new = lambda: np.zeros((N, L, H, W), np.float32)
    for act in (0, 1):
        res = _both("latent_heads_fwd", [hv, Cin, ctot, wm, bm, ws_, bs, eps, new(), new(), new(), new(), L, N, H, W, act], [8, 9, 10, 11])
        for hh, cc in res:
            assert close(hh, cc)
    dh = np.full((N * ctot * H * W - H * W,), 3.0, np.float32)
    (hh, cc), = _both("latent_heads_bwd_data", [dpre, dmu, L, ws_, wm, dh, Cin, ctot, N, H, W, 1], [5])
    assert close(hh, cc)
    wsb = _ffi.lib().uz_latent_heads_bwd_weight_workspace(Cin, L, N, H, W)
    res = _both("latent_heads_bwd_weight", [hv, Cin, ctot, dpre, dmu, L, np.zeros_like(ws_), np.zeros_like(bs), np.zeros_like(wm), np.zeros_like(bm), N, H, W,
                                            np.zeros(wsb // 4 + 16, np.float32), wsb], [6, 7, 8, 9])
    for hh, cc in res:
        assert close(hh, cc, 1e-4)
