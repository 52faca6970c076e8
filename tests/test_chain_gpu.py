"""Deep-level chain launch (csrc/chain.hip, uz_chain_run; off by default, UZ_CHAIN=8192 switches it on): every sub-op through the C ABI
against an fp64 / fp32 PyTorch CPU reference of the op it mirrors, and the PHiSeg step with forward + backward chains against the
per-op tape and against the real reference's digest (tests/golden/phiseg_full_b32_digest)."""
import ctypes as C
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from tests import _gpu as T
from unet_zoo_amd import _ffi

pytestmark = pytest.mark.gpu
CODES = _ffi.chain_codes()


@pytest.fixture(autouse=True)
def _two_piece_mode_only():
    if _ffi.lib().uz_get_conv_math() in (0, 3):
        pytest.skip("the chain's convolutions are two-piece split-fp16: not built under UZ_CONV_MATH=f32 / bf16 (Plan._chain_limit)")


def _amax_slot(value):
    s = torch.zeros(256, device=T.dev())
    s[0] = float(value)
    return s


def _run_chain(subs, n_wgs=64):
    """subs: list of phases, each a list of (code, i, f, p) - p entries are tensors / None.  Runs them as one chain launch."""
    L = _ffi.lib()
    flat = [e for ph in subs for e in ph]
    arr = (_ffi.uz_chain_op * len(flat))()
    phases, k = [], 0
    for ph in subs:
        t0 = 0
        phases.append([k, len(ph)])
        for code, i, f, p in ph:
            a = arr[k]
            a.code = CODES[code]
            for j, v in enumerate(i):
                a.i[j] = int(v)
            for j, v in enumerate(f):
                a.f[j] = float(v)
            for j, v in enumerate(p):
                a.p[j] = 0 if v is None else (v.data_ptr() if isinstance(v, torch.Tensor) else int(v))
            nt = L.uz_chain_op_tiles(C.byref(a))
            assert nt > 0, (code, i)
            a.tile0, a.ntiles = t0 % n_wgs, nt
            t0 += nt
            k += 1
    ops = torch.frombuffer(bytearray(bytes(arr)), dtype=torch.uint8).clone().to(T.dev())
    pht = torch.tensor(phases, dtype=torch.int32).reshape(-1).to(T.dev())
    state = torch.zeros(L.uz_chain_state_bytes() // 4, dtype=torch.int32, device=T.dev())
    _ffi.check(L.uz_chain_run(ops.data_ptr(), pht.data_ptr(), len(subs), len(flat), n_wgs, state.data_ptr(), T.stream()), "chain_run")
    out = C.c_int(0)
    _ffi.check(L.uz_chain_status(state.data_ptr(), C.byref(out), T.stream()), "chain_status")
    assert out.value == 0, f"grid barrier of phase {out.value - 1} timed out"


def _pack(w, dgrad):
    """uz_chain_pack_weights of one layer: returns (image tensor, weight bound slot)."""
    L = _ffi.lib()
    cout, cin = w.shape[:2]
    mc, kc = (cin, cout) if dgrad else (cout, cin)
    img = torch.zeros(L.uz_chain_packed_bytes(kc, mc) // 4, device=T.dev())
    wd = w.to(T.dev()).contiguous()
    slot = _amax_slot(w.abs().max())
    tab = torch.tensor([wd.data_ptr(), img.data_ptr(), mc, kc, cin, int(dgrad), 0], dtype=torch.int64, device=T.dev())
    T.call("uz_chain_pack_weights", tab, 1, L.uz_chain_pack_blocks(kc, mc), slot)
    return img, slot, wd


@pytest.mark.parametrize("shape", [(4, 64, 64, 8, 8, 1), (32, 192, 192, 2, 2, 18), (8, 128, 192, 16, 16, 2), (3, 64, 96, 5, 7, 3), (32, 256, 256, 4, 4, 24)])
def test_chain_conv3_forward_and_data_gradient_against_fp64(shape):
    """UZ_CH_CONV3 (split-fp16 matrix pipe, operands staged straight from global memory) with the unit's BatchNorm adding the split-K slabs:
    y = conv(x) + bias against an fp64 CPU convolution, error <= 1e-5 of the output's maximum (the fp32-MFMA kernels sit at 2e-6, the
    split kernels of conv_split.hip at 1e-6 on the heaviest layer); ragged planes, half 64-channel blocks (96 outputs), S = 1 and S > 1;
    then the data gradient (the same sub-op on the flipped / transposed image) against autograd in fp64."""
    N, cin, cout, H, W, S = shape
    x = T.rnd(N, cin, H, W, seed=1).clamp_min(0)                      # post-ReLU activations
    w = T.rnd(cout, cin, 3, 3, seed=2, scale=(2.0 / (9 * cin)) ** 0.5)
    b = T.rnd(cout, seed=3, scale=0.1)
    img, wslot, _ = _pack(w, 0)
    xd = x.to(T.dev())
    xslot = _amax_slot(x.abs().max())
    y = torch.full((N, cout, H, W), float("nan"), device=T.dev())
    slabs = torch.zeros(S, N, cout, H, W, device=T.dev()) if S > 1 else None
    ref = F.conv2d(x.double(), w.double(), b.double(), padding=1)
    if S == 1:
        _run_chain([[("UZ_CH_CONV3", [cin, cin, cout, cout, N, H, W, 1, 0], [], [xd, img, b.to(T.dev()), y, None, xslot, wslot])]])
    else:
        # the consumer (here: a BatchNorm forward tile) adds bias + slabs in slab order and writes y
        gam, bet = torch.ones(cout, device=T.dev()), torch.zeros(cout, device=T.dev())
        save, a = torch.zeros(4 * cout, device=T.dev()), torch.zeros(N, cout, H, W, device=T.dev())
        aslot = torch.zeros(256, device=T.dev())
        _run_chain([[("UZ_CH_CONV3", [cin, cin, cout, cout, N, H, W, S, 0], [], [xd, img, None, None, slabs, xslot, wslot])],
                    [("UZ_CH_BN_FWD", [cout, cout, cout, N, H * W, 1, S], [1e-3, 0.01], [y, gam, bet, None, None, save, a, slabs, aslot, b.to(T.dev())])]])
        bn = F.batch_norm(ref.float(), None, None, gam.cpu(), bet.cpu(), True, 0.01, 1e-3).clamp_min(0)
        assert T.maxabs(a, bn) <= 2e-5 * float(bn.abs().max()) + 1e-6
        assert abs(float(aslot.max()) - float(a.abs().max())) <= 1e-6 * float(a.abs().max())
    assert T.relerr(y, ref) <= 1e-5
    # data gradient: dx = conv_transpose(dy, w)
    dy = T.rnd(N, cout, H, W, seed=4)
    xg = x.double().requires_grad_(True)
    F.conv2d(xg, w.double(), None, padding=1).backward(dy.double())
    if cout % 16 == 0 and cin % 32 == 0:
        imgd, wslot2, _ = _pack(w, 1)
        dx = torch.full((N, cin, H, W), float("nan"), device=T.dev())
        _run_chain([[("UZ_CH_CONV3", [cout, cout, cin, cin, N, H, W, 1, 0], [], [dy.to(T.dev()), imgd, None, dx, None, _amax_slot(dy.abs().max()), wslot2])]])
        assert T.relerr(dx, xg.grad) <= 1e-5


def test_chain_batchnorm_backward_pool_bilinear_heads_against_torch():
    """One chain of six phases on a 16 x 16 plane: pooling -> convolution of the 2-channel latent (vector pipe) -> BatchNorm forward ->
    interpolation -> the fused latent heads, then BatchNorm backward on the same unit - each against PyTorch on the CPU."""
    N, C, H, W = 8, 32, 16, 16
    x0 = T.rnd(N, 2, 2 * H, 2 * W, seed=5)
    w = T.rnd(C, 2, 3, 3, seed=6, scale=0.3)
    b = T.rnd(C, seed=7, scale=0.1)
    gam, bet = T.rnd(C, seed=8).abs() + 0.5, T.rnd(C, seed=9, scale=0.2)
    d = T.dev()
    pooled, y, a = (torch.full(s, float("nan"), device=d) for s in ((N, 2, H, W), (N, C, H, W), (N, C, H, W)))
    up = torch.full((N, C, 2 * H, 2 * W), float("nan"), device=d)
    save, rm, rv = torch.zeros(4 * C, device=d), torch.zeros(C, device=d), torch.ones(C, device=d)
    wm, ws = T.rnd(2, C, seed=10, scale=0.2), T.rnd(2, C, seed=11, scale=0.2)
    bm, bs = T.rnd(2, seed=12, scale=0.1), T.rnd(2, seed=13, scale=0.1)
    eps = T.rnd(N, 2, H, W, seed=14)
    mu, pre, sig, z = (torch.full((N, 2, H, W), float("nan"), device=d) for _ in range(4))
    _run_chain([
        [("UZ_CH_AVGPOOL_FWD", [2, 2, 2, N, 2 * H, 2 * W], [], [x0.to(d), pooled, None, None])],
        [("UZ_CH_CONV3_SMALL", [2, 2, C, C, N, H, W], [], [pooled, w.to(d), b.to(d), y])],
        [("UZ_CH_BN_FWD", [C, C, C, N, H * W, 1, 1], [1e-3, 0.01], [y, gam.to(d), bet.to(d), rm, rv, save, a, None, None, None])],
        [("UZ_CH_BILINEAR_FWD", [C, C, C, N, H, W, 1], [], [a, up, None, None]),
         ("UZ_CH_HEADS_FWD", [C, C, N, H * W, 0], [], [a, wm.to(d), bm.to(d), ws.to(d), bs.to(d), eps.to(d), mu, pre, sig, z])],
    ], n_wgs=48)
    rp = F.avg_pool2d(x0, 2, 2, ceil_mode=True)
    ry = F.conv2d(rp, w, b, padding=1)
    ra = F.batch_norm(ry, torch.zeros(C), torch.ones(C), gam, bet, True, 0.01, 1e-3).clamp_min(0)
    assert T.maxabs(pooled, rp) <= 1e-6 and T.relerr(y, ry) <= 2e-6 and T.relerr(a, ra) <= 5e-6
    assert T.relerr(up, F.interpolate(ra, scale_factor=2, mode="bilinear", align_corners=True)) <= 5e-6
    rmu = F.conv2d(ra, wm.reshape(2, C, 1, 1), bm)
    rpre = F.conv2d(ra, ws.reshape(2, C, 1, 1), bs)
    assert T.relerr(mu, rmu) <= 1e-5 and T.relerr(pre, rpre) <= 1e-5 and T.relerr(sig, F.softplus(rpre)) <= 1e-5
    assert T.relerr(z, rmu + F.softplus(rpre) * eps) <= 1e-5
    assert T.maxabs(rm, 0.01 * ry.mean((0, 2, 3))) <= 1e-6 and T.relerr(rv, 0.99 + 0.01 * ry.var((0, 2, 3), unbiased=True)) <= 1e-5
    # backward of the unit: dA -> dy, dgamma, dbeta, dbias; data gradient into the 2-channel input; pooling backward
    dA = T.rnd(N, C, H, W, seed=15)
    dyb = torch.full((N, C, H, W), float("nan"), device=d)
    dg, db, dbias = (torch.zeros(C, device=d) for _ in range(3))
    dxs = torch.full((N, 2, H, W), float("nan"), device=d)
    dx0 = torch.full((N, 2, 2 * H, 2 * W), float("nan"), device=d)
    dys = torch.zeros(256, device=d)
    _run_chain([
        [("UZ_CH_BN_BWD", [C, C, C, N, H * W, 1, 1], [], [dA.to(d), y, gam.to(d), save, dyb, dg, db, dbias, dys, None, bet.to(d)])],
        [("UZ_CH_CONV3_SMALL_BWD_DATA", [2, 2, C, C, N, H, W, 0], [], [dyb, w.to(d), dxs])],
        [("UZ_CH_AVGPOOL_BWD", [2, 2, 2, N, 2 * H, 2 * W, 0], [], [dxs, dx0])],
    ], n_wgs=32)
    x0g = x0.double().requires_grad_(True)
    wg, bg = w.double().requires_grad_(True), b.double().requires_grad_(True)
    gg, btg = gam.double().requires_grad_(True), bet.double().requires_grad_(True)
    yy = F.conv2d(F.avg_pool2d(x0g, 2, 2, ceil_mode=True), wg, bg, padding=1)
    yy.retain_grad()
    F.batch_norm(yy, None, None, gg, btg, True, 0.01, 1e-3).clamp_min(0).backward(dA.double())
    assert T.relerr(dyb, yy.grad) <= 2e-5 and T.relerr(dg, gg.grad) <= 2e-5 and T.relerr(db, btg.grad) <= 2e-5
    assert T.relerr(dx0, x0g.grad) <= 2e-5
    assert abs(float(dys.max()) - float(dyb.abs().max())) <= 1e-6 * float(dyb.abs().max())


def test_phiseg_step_with_forward_and_backward_chains(monkeypatch):
    """The headline PHiSeg step (7 / 5 levels, batch 32) with UZ_CHAIN=8192 - one forward chain, three backward chains - (a) against the
    real reference's digest with the gates of the per-op tape (tests/test_phiseg_gpu.py), (b) against the per-op tape on the same
    inputs: loss to 1e-6, logits to 5e-5, the flat gradient to 1e-2 in l2 (two fp32 implementations, knife-edge ReLU pixels included),
    (c) bit-identical when the step is run twice (a stale hand-off between workgroups would show as run-to-run noise)."""
    from tests import test_phiseg_gpu as P
    from tests import _golden as G
    monkeypatch.setenv("UZ_CHAIN", "8192")
    P.test_phiseg_full_size_digest_vs_reference_golden("phiseg_full_b32_digest")
    arrays, meta = G.load("phiseg_full_b32_digest")
    x, mask, eps = P._inputs(meta, 0)
    runs = {}
    for name, px in (("chain", "8192"), ("chain2", "8192"), ("per_op", "0")):
        monkeypatch.setenv("UZ_CHAIN", px)
        net, _ = P._model(meta)
        net.train()
        s = net.forward(x, mask, training=True, eps=eps)
        loss = net.loss(mask)
        loss.backward()
        torch.cuda.synchronize()
        info = net._cur.chain_info
        assert (len(info.get("bwd", [])) == 3 and len(info.get("fwd", [])) == 1) if px != "0" else not info
        assert net._cur.chain_status(net._stream()) == 0
        runs[name] = (float(loss.detach()), [t.clone() for t in s], net._ptab.gflat.clone())
    assert runs["chain"][0] == runs["chain2"][0] and torch.equal(runs["chain"][2], runs["chain2"][2])
    assert all(torch.equal(a, b) for a, b in zip(runs["chain"][1], runs["chain2"][1]))
    assert abs(runs["chain"][0] - runs["per_op"][0]) <= 1e-6 * abs(runs["per_op"][0])
    for a, b in zip(runs["chain"][1], runs["per_op"][1]):
        assert T.maxabs(a, b) <= 5e-5
    gc, gp = runs["chain"][2].double(), runs["per_op"][2].double()
    assert float((gc - gp).norm() / gp.norm()) <= 1e-2
