"""Round-4 entry points of the Conv2D unit (Conv2d -> BatchNorm2d -> ReLU, reference torchlayers.py:18-21) through the C ABI:

  * split storage - a producer that knows its tensor's bound beforehand (BatchNorm apply forward / backward, pooling, bilinear
    interpolation) writes each element as the two fp16 pieces of its scaled value; the split-fp16 convolutions read that directly;
  * the BatchNorm-backward reduction folded into the epilogue of the data gradient that writes dA.

Every case is checked against the SAME chain through the fp32-storage entry points (must agree to the split's own 2^-22) and against
a plain torch fp32 reference of the unit (gate 3e-5 of the largest magnitude, the convolution gate of test_ops_gpu.py)."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


def _g():
    from tests import _gpu
    return _gpu


def _lib():
    from unet_zoo_amd import _ffi
    return _ffi.lib()


def _need_split():
    if _lib().uz_get_conv_math() in (0, 3):
        pytest.skip("split storage is the two-piece fp16 format: not used in the fp32-only / bf16 math modes")


def _slot(v=0.0):
    t = torch.zeros(256, device=_g().dev())
    t[0] = v
    return t


def _unpack(packed, slot):
    g = _g()
    out = torch.empty_like(packed)
    g.call("uz_unpack_split", packed.contiguous(), out, packed.numel(), slot)
    return out


def test_pack_unpack_roundtrip():
    g = _g()
    x = g.rnd(3, 7, 33, 20, seed=4) * 3.0
    x[0, 0, 0, :4] = torch.tensor([0.0, -0.0, 1e-6, -2.5e-7])
    xd = x.to(g.dev())
    slot = _slot(float(x.abs().max()))
    pk = torch.empty_like(xd)
    g.call("uz_pack_split", xd, pk, xd.numel(), slot)
    back = _unpack(pk, slot)
    amax = float(x.abs().max())
    err = (back.cpu().double() - x.double()).abs()
    assert float((err - (2.0 ** -22) * x.double().abs()).max()) <= 2.0 ** -38 * amax       # |r| <= 2^-22 |v| (+ the fp16 subnormal floor)
    assert float(back[0, 0, 0, 0]) == 0.0 and float(back[0, 0, 0, 1]) == 0.0
    # packing what was unpacked reproduces the value (h1 + h2 is exactly representable in fp32)
    pk2 = torch.empty_like(xd)
    g.call("uz_pack_split", back, pk2, xd.numel(), slot)
    assert torch.equal(_unpack(pk2, slot), back)


UNIT_CASES = [(8, 64, 96, 64, 64), (32, 64, 64, 32, 32), (4, 96, 64, 37, 64), (2, 32, 32, 128, 128)]


@pytest.mark.parametrize("N,C1,C2,H,W", UNIT_CASES)
def test_two_units_forward_and_weight_gradient_with_split_storage(N, C1, C2, H, W):
    """conv1 (+ statistics in the epilogue) -> BatchNorm apply writing split storage -> conv2 forward and weight gradient reading it."""
    _need_split()
    g, L = _g(), _lib()
    C0 = 40
    if L.uz_conv_route(0, C1, C2, N, H, W, 3) != 1 or L.uz_conv_route(2, C1, C2, N, H, W, 3) != 1 or L.uz_conv_bn_partials(C0, C1, N, H, W, 3) <= 0:
        pytest.skip("shape off the split path in this math mode")
    x = g.rnd(N, C0, H, W, seed=1)
    w1, b1 = g.rnd(C1, C0, 3, 3, seed=2, scale=0.1), g.rnd(C1, seed=3)
    w2, b2 = g.rnd(C2, C1, 3, 3, seed=4, scale=0.1), g.rnd(C2, seed=5)
    gamma, beta = g.rnd(C1, seed=6).abs() + 0.5, g.rnd(C1, seed=7) * 0.3
    dy2 = g.rnd(N, C2, H, W, seed=8)
    # torch reference
    y1r = F.conv2d(x, w1, b1, padding=1)
    a1r = F.relu(F.batch_norm(y1r, None, None, gamma, beta, training=True, eps=1e-3)).requires_grad_(True)
    w2r = w2.clone().requires_grad_(True)
    y2r = F.conv2d(a1r, w2r, b2, padding=1)
    y2r.backward(dy2)
    d = g.dev()
    xd, w1d, b1d, w2d, b2d, gd, bd, dy2d = (t.to(d) for t in (x, w1, b1, w2, b2, gamma, beta, dy2))
    npart = L.uz_conv_bn_partials(C0, C1, N, H, W, 3)
    wsb = max(L.uz_conv_workspace(C0, C1, N, H, W, 3), L.uz_conv_workspace(C1, C2, N, H, W, 3), L.uz_conv_bwd_weight_workspace(C1, C2, N, H, W, 3))
    ws = torch.empty(wsb // 4 + 64, device=d)
    bws = torch.empty(L.uz_bn_workspace(C1, N, H, W) // 4 + 16, device=d)
    y1 = torch.empty(N, C1, H, W, device=d)
    part = torch.empty(npart * C1 * 4, device=d)
    g.call("uz_conv_fwd_bnstats", xd, C0, C0, w1d, b1d, y1, C1, C1, N, H, W, 3, 0, None, None, None, ws, wsb, None, part)
    res = {}
    for packed in (0, 1):
        a1 = torch.full((N, C1, H, W), float("nan"), device=d)
        save = torch.full((4 * C1,), float("nan"), device=d)
        slot = _slot()
        g.call("uz_bn_relu_fwd_ex", y1, C1, C1, gd, bd, None, None, save, a1, C1, N, H, W, 1e-3, 0.01, 1, 1, slot, bws, part, npart, packed)
        y2 = torch.empty(N, C2, H, W, device=d)
        g.call("uz_conv_fwd_ex", a1, C1, C1, w2d, b2d, y2, C2, C2, N, H, W, 3, 0, slot, None, None, ws, wsb, None, None, packed, None, 0)
        dw2 = torch.empty(C2, C1, 3, 3, device=d)
        g.call("uz_conv_bwd_weight_ex", a1, C1, C1, dy2d, C2, C2, dw2, None, N, H, W, 3, slot, None, ws, wsb, packed, None, 0, 0, None)
        res[packed] = (a1, y2, dw2, slot, save)
    a_f32, y2_f32, dw_f32, slot0, save0 = res[0]
    a_pk, y2_pk, dw_pk, slot1, save1 = res[1]
    assert torch.equal(slot0, slot1)                                   # same exact bound either way
    assert torch.isfinite(save1).all()                                # mean, rstd, alpha, beta'
    alpha = gamma.double() * save1[C1:2 * C1].cpu().double()
    assert float((save1[2 * C1:3 * C1].cpu().double() - alpha).abs().max()) <= 1e-6 * float(alpha.abs().max())
    # the stored words decode to the fp32 activation within the split's 2^-22
    dec = _unpack(a_pk, slot1)
    amax = float(a_f32.abs().max())
    assert float((dec - a_f32).abs().max()) <= 2.0 ** -21 * amax
    assert g.maxabs(dec, a1r) <= 3e-5 * max(1.0, amax)
    # consumers: the same operand pieces (up to ties of the first rounding) -> the same results
    assert g.relerr(y2_pk, y2_f32) <= 2e-6 and g.relerr(dw_pk, dw_f32) <= 2e-6
    assert g.relerr(y2_pk, y2r) <= 3e-5 and g.relerr(dw_pk, w2r.grad) <= 3e-5


def test_concat_buffer_with_two_scale_segments():
    """A concat buffer written by two producers, each scaling from its own bound (phiseg.py:315: torch.cat of the up-sampled coarse
    features and the level's own): BatchNorm apply into channels [0, 32), bilinear interpolation into [32, 80) with a 64 x larger
    magnitude; the consuming convolution switches scales at channel 32 (forward: accumulators rescaled at the chunk boundary;
    weight gradient: per input channel)."""
    _need_split()
    g, L = _g(), _lib()
    N, CA, CB, C2, H, W = 8, 32, 48, 64, 64, 64
    C = CA + CB
    if L.uz_conv_route(0, C, C2, N, H, W, 3) != 1 or L.uz_conv_route(2, C, C2, N, H, W, 3) != 1:
        pytest.skip("shape off the split path in this math mode")
    d = g.dev()
    C0 = 32
    x = g.rnd(N, C0, H, W, seed=11)
    w1 = g.rnd(CA, C0, 3, 3, seed=12, scale=0.1)
    gamma, beta = g.rnd(CA, seed=13).abs() + 0.5, g.rnd(CA, seed=14) * 0.3
    low = g.rnd(N, CB, H // 2, W // 2, seed=15) * 64.0
    w2, dy2 = g.rnd(C2, C, 3, 3, seed=16, scale=0.05), g.rnd(N, C2, H, W, seed=17)
    y1r = F.conv2d(x, w1, None, padding=1)
    ar = F.relu(F.batch_norm(y1r, None, None, gamma, beta, training=True, eps=1e-3))
    upr = F.interpolate(low, scale_factor=2, mode="bilinear", align_corners=True)
    catr = torch.cat([ar, upr], 1).requires_grad_(True)
    w2r = w2.clone().requires_grad_(True)
    y2r = F.conv2d(catr, w2r, None, padding=1)
    y2r.backward(dy2)

    npart = L.uz_conv_bn_partials(C0, CA, N, H, W, 3)
    assert npart > 0
    wsb = max(L.uz_conv_workspace(C0, CA, N, H, W, 3), L.uz_conv_workspace(C, C2, N, H, W, 3), L.uz_conv_bwd_weight_workspace(C, C2, N, H, W, 3))
    ws = torch.empty(wsb // 4 + 64, device=d)
    bws = torch.empty(L.uz_bn_workspace(CA, N, H, W) // 4 + 16, device=d)
    y1 = torch.empty(N, CA, H, W, device=d)
    part = torch.empty(npart * CA * 4, device=d)
    g.call("uz_conv_fwd_bnstats", x.to(d), C0, C0, w1.to(d), None, y1, CA, CA, N, H, W, 3, 0, None, None, None, ws, wsb, None, part)
    cat = torch.full((N, C, H, W), float("nan"), device=d)
    save = torch.empty(4 * CA, device=d)
    sA, sB, sLow = _slot(), _slot(), _slot(float(low.abs().max()))
    g.call("uz_bn_relu_fwd_ex", y1, CA, CA, gamma.to(d), beta.to(d), None, None, save, cat, C, N, H, W, 1e-3, 0.01, 1, 1, sA, bws, part, npart, 1)
    g.call("uz_bilinear2x_fwd_ex", low.to(d), CB, CB, cat[:, CA:], C, N, H // 2, W // 2, 1, sLow, sB, 1)
    assert float(sB.max()) == float(sLow.max()) and float(sB.max()) > 16 * float(sA.max())
    decA = _unpack(cat[:, :CA].contiguous(), sA)
    decB = _unpack(cat[:, CA:].contiguous(), sB)
    assert g.maxabs(decA, ar) <= 3e-5 * float(ar.abs().max()) and g.maxabs(decB, upr) <= 3e-5 * float(upr.abs().max())
    y2 = torch.empty(N, C2, H, W, device=d)
    g.call("uz_conv_fwd_ex", cat, C, C, w2.to(d), None, y2, C2, C2, N, H, W, 3, 0, sA, None, None, ws, wsb, None, None, 1, sB, CA)
    dw2 = torch.empty(C2, C, 3, 3, device=d)
    g.call("uz_conv_bwd_weight_ex", cat, C, C, dy2.to(d), C2, C2, dw2, None, N, H, W, 3, sA, None, ws, wsb, 1, sB, CA, 0, None)
    assert g.relerr(y2, y2r) <= 3e-5
    # per input-channel block: the small-magnitude segment keeps its own precision
    for lo, hi in ((0, CA), (CA, C)):
        assert g.relerr(dw2[:, lo:hi], w2r.grad[:, lo:hi]) <= 3e-5
    # the fp32-storage path on the decoded concat gives the same numbers
    catf = torch.cat([decA, decB], 1).contiguous()
    y2f = torch.empty_like(y2)
    g.call("uz_conv_fwd", catf, C, C, w2.to(d), None, y2f, C2, C2, N, H, W, 3, 0, None, None, None, ws, wsb)
    assert g.relerr(y2, y2f) <= 5e-6


@pytest.mark.parametrize("H,W", [(64, 64), (33, 36)])
def test_pooling_writes_split_storage(H, W):
    _need_split()
    g = _g()
    N, C = 4, 6
    x = g.rnd(N, C, H, W, seed=21) * 5
    ref = F.avg_pool2d(x, 2, 2, ceil_mode=True)
    d = g.dev()
    sx, sy = _slot(float(x.abs().max())), _slot()
    y = torch.empty(N, C, (H + 1) // 2, (W + 1) // 2, device=d)
    g.call("uz_avgpool2_fwd_ex", x.to(d), C, C, y, C, N, H, W, sx, sy, 1)
    assert float(sy.max()) == float(sx.max())
    assert g.maxabs(_unpack(y, sy), ref) <= 2.0 ** -21 * float(x.abs().max())


@pytest.mark.parametrize("N,C1,C2,H,W", UNIT_CASES)
def test_unit_backward_with_folded_reduction_and_split_storage(N, C1, C2, H, W):
    """Backward of unit 1 (Conv -> BN -> ReLU) behind unit 2's data gradient: the data gradient masks dA with unit 1's ReLU and leaves
    the BatchNorm-backward sums in its epilogue (uz_conv_bwd_data_ex), uz_bn_relu_bwd_ex finalises them and writes dy as split
    storage, unit 1's weight / data gradients read it - against the un-fused fp32-storage chain and torch autograd."""
    _need_split()
    g, L = _g(), _lib()
    C0 = 64
    rows = L.uz_conv_bwd_relu_partials(C1, C2, N, H, W, 3)
    if rows <= 0 or L.uz_conv_route(2, C0, C1, N, H, W, 3) != 1 or L.uz_conv_route(1, C0, C1, N, H, W, 3) != 1 or L.uz_conv_bn_partials(C0, C1, N, H, W, 3) <= 0:
        pytest.skip("shape off the split path in this math mode")
    d = g.dev()
    x = g.rnd(N, C0, H, W, seed=31)
    w1, w2 = g.rnd(C1, C0, 3, 3, seed=32, scale=0.1), g.rnd(C2, C1, 3, 3, seed=33, scale=0.1)
    gamma, beta = g.rnd(C1, seed=34).abs() + 0.5, g.rnd(C1, seed=35) * 0.3
    dy2 = g.rnd(N, C2, H, W, seed=36)
    xr, w1r = x.clone().requires_grad_(True), w1.clone().requires_grad_(True)
    gr, br = gamma.clone().requires_grad_(True), beta.clone().requires_grad_(True)
    y1r = F.conv2d(xr, w1r, None, padding=1)
    a1r = F.relu(F.batch_norm(y1r, None, None, gr, br, training=True, eps=1e-3))
    F.conv2d(a1r, w2, None, padding=1).backward(dy2)

    xd, w1d, w2d, gd, bd, dy2d = (t.to(d) for t in (x, w1, w2, gamma, beta, dy2))
    npart = L.uz_conv_bn_partials(C0, C1, N, H, W, 3)
    wsb = max(L.uz_conv_workspace(C0, C1, N, H, W, 3), L.uz_conv_workspace(C1, C2, N, H, W, 3), L.uz_conv_bwd_weight_workspace(C0, C1, N, H, W, 3))
    ws = torch.empty(wsb // 4 + 64, device=d)
    bws = torch.empty(L.uz_bn_workspace(C1, N, H, W) // 4 + 16, device=d)
    y1 = torch.empty(N, C1, H, W, device=d)
    part = torch.empty(npart * C1 * 4, device=d)
    g.call("uz_conv_fwd_bnstats", xd, C0, C0, w1d, None, y1, C1, C1, N, H, W, 3, 0, None, None, None, ws, wsb, None, part)
    a1 = torch.empty(N, C1, H, W, device=d)
    save = torch.empty(4 * C1, device=d)
    g.call("uz_bn_relu_fwd_ex", y1, C1, C1, gd, bd, None, None, save, a1, C1, N, H, W, 1e-3, 0.01, 1, 1, _slot(), bws, part, npart, 0)
    sdy2 = _slot(float(dy2.abs().max()))
    out = {}
    for fused in (0, 1):
        dA = torch.full((N, C1, H, W), float("nan"), device=d)
        bpart = torch.full((rows * C1 * 4,), float("nan"), device=d)
        if fused:
            g.call("uz_conv_bwd_data_ex", dy2d, C2, C2, w2d, dA, C1, C1, N, H, W, 3, 0, sdy2, None, ws, wsb, None, 0, y1, C1, save, 1, bpart)
        else:
            g.call("uz_conv_bwd_data", dy2d, C2, C2, w2d, dA, C1, C1, N, H, W, 3, 0, sdy2, None, ws, wsb)
        dy1 = torch.full((N, C1, H, W), float("nan"), device=d)
        dgm, dbt, dbias = (torch.empty(C1, device=d) for _ in range(3))
        sdy1 = _slot()
        g.call("uz_bn_relu_bwd_ex", dA, C1, y1, C1, C1, gd, bd, save, dy1, C1, dgm, dbt, dbias, N, H, W, 1, sdy1, bws, bpart if fused else None, rows if fused else 0, fused, None, None, 0)
        dw1 = torch.empty(C1, C0, 3, 3, device=d)
        g.call("uz_conv_bwd_weight_ex", xd, C0, C0, dy1, C1, C1, dw1, None, N, H, W, 3, None, sdy1, ws, wsb, 0, None, 0, fused, None)
        dx = torch.empty(N, C0, H, W, device=d)
        g.call("uz_conv_bwd_data_ex", dy1, C1, C1, w1d, dx, C0, C0, N, H, W, 3, 0, sdy1, None, ws, wsb, None, fused, None, 0, None, 0, None)
        out[fused] = (dA, dy1, dgm, dbt, dw1, dx, sdy1)
    dA0, dy0, dg0, db0, dw0, dx0, s0 = out[0]
    dA1, dy1p, dg1, db1, dw1_, dx1, s1 = out[1]
    mask = (a1 > 0).float()
    assert g.maxabs(dA1, dA0 * mask) <= 1e-6 * float(dA0.abs().max())                 # the folded launch stores dz = dA * [a > 0]
    dec = _unpack(dy1p, s1)
    assert float(s1.max()) >= float(dy0.abs().max()) * (1 - 1e-6)                     # the analytic bound covers the tensor
    assert g.maxabs(dec, dy0) <= 3e-6 * float(dy0.abs().max())
    assert g.relerr(dg1, dg0) <= 2e-5 and g.relerr(db1, db0) <= 2e-5
    assert g.relerr(dw1_, dw0) <= 5e-6 and g.relerr(dx1, dx0) <= 5e-6
    # ... and all of it against autograd
    assert g.relerr(dw1_, w1r.grad) <= 5e-5 and g.relerr(dx1, xr.grad) <= 5e-5
    assert g.relerr(dg1, gr.grad) <= 1e-4 and g.relerr(db1, br.grad) <= 1e-4


def test_deferred_conv_bias_sums():
    """uz_bn_relu_bwd_ex(dbias_partials) + uz_chan_sum_table == the conv-bias gradient uz_bn_relu_bwd sums in a launch of its own
    (the plans add all units' rows in ONE launch at the end of the backward tape)."""
    g, L = _g(), _lib()
    N, C, H, W = 3, 5, 128, 128
    assert N * H * W > L.uz_bn_bwd_fused_limit(H, W)
    d = g.dev()
    y, da = (g.rnd(N, C, H, W, seed=41) * 2 + 0.3).to(d), g.rnd(N, C, H, W, seed=42).to(d)
    gamma, beta = (g.rnd(C, seed=43).abs() + 0.5).to(d), (g.rnd(C, seed=44) * 0.3).to(d)
    ws = torch.empty(L.uz_bn_workspace(C, N, H, W) // 4 + 16, device=d)
    save = torch.empty(2 * C, device=d)
    a = torch.empty(N, C, H, W, device=d)
    g.call("uz_bn_relu_fwd", y, C, C, gamma, beta, None, None, save, a, C, N, H, W, 1e-3, 0.01, 1, 1, None, ws)
    dy0, dy1 = torch.empty_like(y), torch.empty_like(y)
    dg0, db0, dbias0, dg1, db1, dbias1 = (torch.empty(C, device=d) for _ in range(6))
    g.call("uz_bn_relu_bwd", da, C, y, C, C, gamma, beta, save, dy0, C, dg0, db0, dbias0, N, H, W, 1, None, ws)
    rows = L.uz_bn_bwd_dbias_rows(N, H, W)
    assert rows > 0
    part = torch.full((rows * C,), float("nan"), dtype=torch.float64, device=d)
    g.call("uz_bn_relu_bwd_ex", da, C, y, C, C, gamma, beta, save, dy1, C, dg1, db1, None, N, H, W, 1, None, ws, None, 0, 0, part, None, 0)
    table = torch.tensor([part.data_ptr(), dbias1.data_ptr(), rows, C, 1], dtype=torch.int64, device=d)
    g.call("uz_chan_sum_table", table, 1, C)
    assert torch.equal(dy0, dy1) and torch.equal(dg0, dg1) and torch.equal(db0, db1)
    assert torch.equal(dbias0, dbias1)


def test_weight_gradient_slabs_reduced_by_the_table_launch():
    """uz_conv_bwd_weight_ex(slabs_out) + uz_wgrad_reduce_table == uz_conv_bwd_weight BIT FOR BIT (same order of additions), for
    two layers of different kernel families sharing one table launch: a split-path layer (S > 64: two-stage order) and an
    fp32 small-plane layer."""
    g, L = _g(), _lib()
    d = g.dev()
    cases = [(8, 64, 64, 64, 64), (32, 48, 40, 8, 8)]
    rows, blk, keep = [], 0, []
    for k, (N, Cin, Cout, H, W) in enumerate(cases):
        x, dy = g.rnd(N, Cin, H, W, seed=60 + k).to(d), g.rnd(N, Cout, H, W, seed=70 + k).to(d)
        wsb = L.uz_conv_bwd_weight_workspace(Cin, Cout, N, H, W, 3)
        ws = torch.empty(wsb // 4 + 64, device=d)
        dw0 = torch.empty(Cout, Cin, 3, 3, device=d)
        g.call("uz_conv_bwd_weight", x, Cin, Cin, dy, Cout, Cout, dw0, None, N, H, W, 3, None, None, ws, wsb)
        S = L.uz_conv_bwd_weight_slabs(Cin, Cout, N, H, W, 3)
        assert S > 0
        slabs = torch.full((S * 9 * Cout * Cin,), float("nan"), device=d)
        dw1 = torch.full_like(dw0, float("nan"))
        g.call("uz_conv_bwd_weight_ex", x, Cin, Cin, dy, Cout, Cout, dw1, None, N, H, W, 3, None, None, ws, wsb, 0, None, 0, 0, slabs)
        assert torch.isnan(dw1).all()                      # the call stopped behind its main kernel
        rows += [slabs.data_ptr(), dw1.data_ptr(), S, Cout, Cin, 9, blk, 0]
        blk += L.uz_wgrad_reduce_blocks(Cin, Cout, 3)
        keep.append((dw0, dw1, slabs, S))
    # a depth window (Conv3d as the 2-D kernel over slices: 3 C view channels on a C-channel buffer, Cin = 3 C): the table row carries
    # C and the sum leaves in the Conv3d layout [Cout][C][3][3][3] like the call's own reduction
    for k, (D, C, Cout, H, W) in enumerate([(10, 32, 32, 64, 64), (6, 16, 24, 16, 16)]):
        vol = torch.zeros(D + 2, C, H, W, device=d)
        vol[1:-1] = g.rnd(D, C, H, W, seed=80 + k).to(d)
        dy = g.rnd(D, Cout, H, W, seed=90 + k).to(d)
        wsb = L.uz_conv_bwd_weight_workspace(3 * C, Cout, D, H, W, 3)
        ws = torch.empty(wsb // 4 + 64, device=d)
        dw0 = torch.empty(Cout, C, 3, 3, 3, device=d)
        g.call("uz_conv_bwd_weight", vol, 3 * C, C, dy, Cout, Cout, dw0, None, D, H, W, 3, None, None, ws, wsb)
        S = L.uz_conv_bwd_weight_slabs(3 * C, Cout, D, H, W, 3)
        assert S > 0
        slabs = torch.full((S * 9 * Cout * 3 * C,), float("nan"), device=d)
        dw1 = torch.full_like(dw0, float("nan"))
        g.call("uz_conv_bwd_weight_ex", vol, 3 * C, C, dy, Cout, Cout, dw1, None, D, H, W, 3, None, None, ws, wsb, 0, None, 0, 0, slabs)
        assert torch.isnan(dw1).all()
        rows += [slabs.data_ptr(), dw1.data_ptr(), S, Cout, 3 * C, 9, blk, C]
        blk += L.uz_wgrad_reduce_blocks(3 * C, Cout, 3)
        keep.append((dw0, dw1, slabs, S))
    table = torch.tensor(rows, dtype=torch.int64, device=d)
    g.call("uz_wgrad_reduce_table", table, len(keep), blk)
    for dw0, dw1, _, S in keep:
        assert torch.equal(dw0, dw1), S


@pytest.mark.parametrize("N,Cin,Cout,H,W,ks", [(32, 192, 192, 8, 8, 3), (32, 192, 192, 4, 4, 3), (32, 256, 256, 2, 2, 3), (7, 70, 50, 5, 3, 3)])
def test_data_gradient_slabs_folded_into_small_plane_batchnorm_backward(N, Cin, Cout, H, W, ks):
    """uz_conv_bwd_data_slabs + uz_bn_relu_bwd_ex(da_slabs) (the 8 x 8 ... 2 x 2 levels: the data gradient's split-K reduce folded into
    the one-workgroup-per-channel BatchNorm backward of the unit that produced the convolution's input) == uz_conv_bwd_data +
    uz_bn_relu_bwd BIT FOR BIT (same slab order)."""
    g, L = _g(), _lib()
    parts = L.uz_conv_bwd_splitk_parts(Cin, Cout, N, H, W, ks)
    if parts <= 1:
        pytest.skip("the data gradient of this shape is not split over workgroups in this math mode")
    d = g.dev()
    dy2 = g.rnd(N, Cout, H, W, seed=81).to(d)
    w = (g.rnd(Cout, Cin, ks, ks, seed=82) * 0.1).to(d)
    y1 = (g.rnd(N, Cin, H, W, seed=83) * 2 + 0.3).to(d)           # pre-normalisation output of the unit whose activation the convolution read
    gamma, beta = (g.rnd(Cin, seed=84).abs() + 0.5).to(d), (g.rnd(Cin, seed=85) * 0.3).to(d)
    bws = torch.empty(L.uz_bn_workspace(Cin, N, H, W) // 4 + 16, device=d)
    save, a1 = torch.empty(2 * Cin, device=d), torch.empty(N, Cin, H, W, device=d)
    g.call("uz_bn_relu_fwd", y1, Cin, Cin, gamma, beta, None, None, save, a1, Cin, N, H, W, 1e-3, 0.01, 1, 1, None, bws)
    wsb = L.uz_conv_workspace(Cin, Cout, N, H, W, ks)
    ws = torch.empty(wsb // 4 + 64, device=d)
    dA = torch.empty(N, Cin, H, W, device=d)
    g.call("uz_conv_bwd_data", dy2, Cout, Cout, w, dA, Cin, Cin, N, H, W, ks, 0, None, None, ws, wsb)
    dy0 = torch.empty_like(y1)
    dg0, db0, dbias0 = (torch.empty(Cin, device=d) for _ in range(3))
    g.call("uz_bn_relu_bwd", dA, Cin, y1, Cin, Cin, gamma, beta, save, dy0, Cin, dg0, db0, dbias0, N, H, W, 1, None, bws)
    slabs = torch.full((parts * N * Cin * H * W,), float("nan"), device=d)
    g.call("uz_conv_bwd_data_slabs", dy2, Cout, Cout, w, Cin, N, H, W, ks, slabs)
    dy1 = torch.full_like(y1, float("nan"))
    dg1, db1, dbias1 = (torch.full((Cin,), float("nan"), device=d) for _ in range(3))
    g.call("uz_bn_relu_bwd_ex", None, Cin, y1, Cin, Cin, gamma, beta, save, dy1, Cin, dg1, db1, dbias1, N, H, W, 1, None, bws, None, 0, 0, None, slabs, parts)
    assert torch.equal(dy0, dy1) and torch.equal(dg0, dg1) and torch.equal(db0, db1) and torch.equal(dbias0, dbias1)


@pytest.mark.parametrize("N,C1,C2,H,W", UNIT_CASES + [(3, 72, 64, 40, 64), (20, 64, 32, 16, 16)])
def test_forward_convolution_applies_the_producers_batchnorm_while_staging(N, C1, C2, H, W):
    """uz_conv_fwd_bn_ex (round 6): conv2 reads conv1's PRE-normalisation output y1 and the unit's statistics table and applies BatchNorm + ReLU in its
    staging.  Same values as conv2 over the activation the unit's apply pass stores in split storage (uz_bn_relu_fwd_ex, out_packed) - the operand
    pieces are formed from the same fp32 expression with the same scale - so the outputs agree to the last bit or two, and with torch within
    the split path's gate.  Covers ragged channel counts (K tail), a partial last tile row and the 16 x 16-pixel geometry."""
    _need_split()
    g, L = _g(), _lib()
    C0 = 40
    if L.uz_conv_route(0, C1, C2, N, H, W, 3) != 1 or L.uz_conv_bn_partials(C0, C1, N, H, W, 3) <= 0 or N * H * W <= 4096:
        pytest.skip("shape off the split path in this math mode (or a small plane: split storage starts above 4 096 values per channel)")
    x = g.rnd(N, C0, H, W, seed=1)
    w1, b1 = g.rnd(C1, C0, 3, 3, seed=2, scale=0.1), g.rnd(C1, seed=3)
    w2, b2 = g.rnd(C2, C1, 3, 3, seed=4, scale=0.1), g.rnd(C2, seed=5)
    gamma, beta = g.rnd(C1, seed=6).abs() + 0.5, g.rnd(C1, seed=7) * 0.3
    y1r = F.conv2d(x, w1, b1, padding=1)
    a1r = F.relu(F.batch_norm(y1r, None, None, gamma, beta, training=True, eps=1e-3))
    y2r = F.conv2d(a1r, w2, b2, padding=1)
    d = g.dev()
    xd, w1d, b1d, w2d, b2d, gd, bd = (t.to(d) for t in (x, w1, b1, w2, b2, gamma, beta))
    npart = L.uz_conv_bn_partials(C0, C1, N, H, W, 3)
    wsb = max(L.uz_conv_workspace(C0, C1, N, H, W, 3), L.uz_conv_workspace(C1, C2, N, H, W, 3))
    ws = torch.empty(wsb // 4 + 64, device=d)
    bws = torch.empty(L.uz_bn_workspace(C1, N, H, W) // 4 + 16, device=d)
    y1 = torch.empty(N, C1, H, W, device=d)
    part = torch.empty(npart * C1 * 4, device=d)
    g.call("uz_conv_fwd_bnstats", xd, C0, C0, w1d, b1d, y1, C1, C1, N, H, W, 3, 0, None, None, None, ws, wsb, None, part)
    a1 = torch.full((N, C1, H, W), float("nan"), device=d)
    save = torch.full((4 * C1,), float("nan"), device=d)
    slot = _slot()
    g.call("uz_bn_relu_fwd_ex", y1, C1, C1, gd, bd, None, None, save, a1, C1, N, H, W, 1e-3, 0.01, 1, 1, slot, bws, part, npart, 1)
    y2_stored = torch.empty(N, C2, H, W, device=d)
    g.call("uz_conv_fwd_ex", a1, C1, C1, w2d, b2d, y2_stored, C2, C2, N, H, W, 3, 0, slot, None, None, ws, wsb, None, None, 1, None, 0)
    y2_fused = torch.full((N, C2, H, W), float("nan"), device=d)
    g.call("uz_conv_fwd_bn_ex", y1, C1, C1, save, 1, w2d, b2d, y2_fused, C2, C2, N, H, W, 3, slot, None, None, ws, wsb, None, None)
    assert g.relerr(y2_fused, y2r) <= 3e-5
    assert g.maxabs(y2_fused.cpu(), y2_stored.cpu()) <= 2e-6 * float(y2r.abs().max())
    # without the ReLU (a bare BatchNorm in front): negative activations survive
    a1n = F.batch_norm(y1r, None, None, gamma, beta, training=True, eps=1e-3)
    slot_n = _slot(float(a1n.abs().max()))
    g.call("uz_conv_fwd_bn_ex", y1, C1, C1, save, 0, w2d, b2d, y2_fused, C2, C2, N, H, W, 3, slot_n, None, None, ws, wsb, None, None)
    assert g.relerr(y2_fused, F.conv2d(a1n, w2, b2, padding=1)) <= 3e-5


def test_phiseg_step_with_batchnorm_apply_off_the_chain(monkeypatch):
    """Plan._bn_offchain_pass (UZ_BN_OFFCHAIN=1, off by default - measured 1 % slower, profiles/NOTES_r6.md section 10): eight units of the headline plan run
    their BatchNorm as a statistics launch + an apply pass of its own group, and the one convolution that reads each activation applies the
    normalisation while staging (uz_conv_fwd_bn_ex).  The operand pieces are the ones the apply pass stores, so the step must agree with the default
    plan far inside the gates between two fp32 implementations: loss to 1e-6, logits to 5e-5, the flat gradient to 1e-3 in l2; and against the real
    reference's digest with the default gates."""
    _need_split()
    from tests import test_phiseg_gpu as P
    from tests import _golden as G
    monkeypatch.setenv("UZ_BN_OFFCHAIN", "1")
    P.test_phiseg_full_size_digest_vs_reference_golden("phiseg_full_b32_digest")
    arrays, meta = G.load("phiseg_full_b32_digest")
    x, mask, eps = P._inputs(meta, 0)
    runs = {}
    for name, flag in (("off_chain", "1"), ("default", "0")):
        monkeypatch.setenv("UZ_BN_OFFCHAIN", flag)
        net, _ = P._model(meta)
        net.train()
        s = net.forward(x, mask, training=True, eps=eps)
        loss = net.loss(mask)
        loss.backward()
        torch.cuda.synchronize()
        assert net._cur.bn_offchain["units"] == (8 if flag == "1" else 0)
        runs[name] = (float(loss.detach()), [t.clone() for t in s], net._ptab.gflat.clone())
    g = _g()
    assert abs(runs["off_chain"][0] - runs["default"][0]) <= 1e-6 * abs(runs["default"][0])
    for a, b in zip(runs["off_chain"][1], runs["default"][1]):
        assert g.maxabs(a, b) <= 5e-5
    ga, gb = runs["off_chain"][2].double(), runs["default"][2].double()
    assert float((ga - gb).norm() / gb.norm()) <= 1e-3
