"""bf16 STORAGE (include/uz_api.h, "bf16 storage"; BASELINE config 5: PHiSeg3D "bf16", phiseg3D.py:13-35) - op level, through the C ABI.

The oracle of every case is the fp32-STORAGE entry point of the same op in the same arithmetic mode (uz_set_conv_math(3): bf16 operands,
fp32 accumulation - itself pinned against torch in tests/test_ops_gpu.py::test_conv_bf16_arithmetic_mode) fed with the same, already
bf16-representable, values.  A bf16-stored INPUT must then give bit-identical results (the kernels stage the stored 16 bits as they
are instead of rounding fp32 values to them), a bf16-stored OUTPUT must be the round-to-nearest-even of the fp32-storage result, bit
for bit - so the gates here are equalities, not tolerances."""
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


def _rb(t):
    return t.to(torch.bfloat16).to(torch.float32)


@pytest.fixture()
def bf16_mode():
    L = _lib()
    L.uz_set_conv_math(3)
    yield L
    L.uz_set_conv_math(-1)


def _embed(t, ctot, c0, dtype):
    """NCHW tensor as channels [c0, c0 + C) of a wider device buffer of the given storage dtype (NaN canaries around it)."""
    g = _g()
    n, c, h, w = t.shape
    buf = torch.full((n, ctot, h, w), float("nan"), device=g.dev(), dtype=dtype)
    buf[:, c0:c0 + c] = t.to(g.dev()).to(dtype)
    return buf, buf[:, c0:]


@pytest.mark.parametrize("N,Cin,Cout,H,W,ctx,cty", [(8, 48, 64, 64, 64, 48, 64), (8, 40, 96, 48, 96, 53, 100), (4, 96, 32, 64, 128, 96, 32), (24, 64, 96, 64, 32, 70, 96), (40, 128, 64, 32, 32, 128, 64)])
def test_conv_forward_and_data_gradient_in_bf16_storage(bf16_mode, N, Cin, Cout, H, W, ctx, cty):
    g, L = _g(), bf16_mode
    d = g.dev()
    if L.uz_conv_route(0, Cin, Cout, N, H, W, 3) != 1 or L.uz_conv_route(1, Cin, Cout, N, H, W, 3) != 1:
        pytest.skip("shape not on the matrix-pipe path")
    x = _rb(g.rnd(N, Cin, H, W, seed=1))
    w = g.rnd(Cout, Cin, 3, 3, seed=2, scale=0.1).to(d)
    b = g.rnd(Cout, seed=3).to(d)
    dy = _rb(g.rnd(N, Cout, H, W, seed=4))
    wsb = L.uz_conv_workspace(Cin, Cout, N, H, W, 3)
    ws = torch.empty(wsb // 4 + 64, device=d)
    npart = L.uz_conv_bn_partials(Cin, Cout, N, H, W, 3)
    assert npart > 0 and L.uz_conv_split_parts(0, Cin, Cout, N, H, W) == 1 and L.uz_conv_split_parts(1, Cin, Cout, N, H, W) == 1
    x32b, x32 = _embed(x, ctx, ctx - Cin, torch.float32)
    x16b, x16 = _embed(x, ctx, ctx - Cin, torch.bfloat16)
    # ---- forward: fp32 storage (reference), bf16 input, bf16 input + output
    y_ref = torch.full((N, cty, H, W), float("nan"), device=d)
    p_ref = torch.full((npart * Cout * 4,), float("nan"), device=d)
    g.call("uz_conv_fwd_bnstats", x32, Cin, ctx, w, b, y_ref, Cout, cty, N, H, W, 3, 0, None, None, None, ws, wsb, None, p_ref)
    y_a = torch.full_like(y_ref, float("nan"))
    p_a = torch.full_like(p_ref, float("nan"))
    g.call("uz_conv_fwd_b16", x16, Cin, ctx, w, b, y_a, Cout, cty, N, H, W, 3, ws, wsb, None, p_a, 1, 0)
    assert torch.equal(y_a[:, :Cout], y_ref[:, :Cout]) and torch.equal(p_a, p_ref)
    assert torch.isnan(y_a[:, Cout:]).all()
    y_b = torch.full((N, cty, H, W), float("nan"), device=d, dtype=torch.bfloat16)
    p_b = torch.full_like(p_ref, float("nan"))
    g.call("uz_conv_fwd_b16", x16, Cin, ctx, w, b, y_b, Cout, cty, N, H, W, 3, ws, wsb, None, p_b, 1, 1)
    assert torch.equal(y_b[:, :Cout], y_ref[:, :Cout].to(torch.bfloat16))              # round to nearest even of the fp32-storage result
    assert torch.isnan(y_b[:, Cout:]).all()
    # the fused BatchNorm statistics are those of the STORED values: {sum, sum of squares, max, max(-y)} per (tile, channel)
    pb = p_b.view(npart, Cout, 4).double().cpu()
    ys = y_b[:, :Cout].float().double().cpu()
    assert torch.allclose(pb[:, :, 0].sum(0), ys.sum((0, 2, 3)), rtol=1e-5, atol=1e-2)
    assert torch.allclose(pb[:, :, 1].sum(0), (ys * ys).sum((0, 2, 3)), rtol=1e-5)
    assert torch.equal(pb[:, :, 2].max(0).values, ys.amax((0, 2, 3))) and torch.equal(pb[:, :, 3].max(0).values, (-ys).amax((0, 2, 3)))
    # bf16 output from an fp32 input (the first layer of a network reads the user's image)
    y_c = torch.full_like(y_b, float("nan"))
    g.call("uz_conv_fwd_b16", x32, Cin, ctx, w, b, y_c, Cout, cty, N, H, W, 3, ws, wsb, None, None, 0, 1)
    assert torch.equal(y_c[:, :Cout], y_b[:, :Cout])
    # ---- data gradient: overwrite and accumulate
    wsb2 = L.uz_conv_workspace(Cin, Cout, N, H, W, 3)
    dy32b, dy32 = _embed(dy, cty, 0, torch.float32)
    dy16b, dy16 = _embed(dy, cty, 0, torch.bfloat16)
    dx_ref = torch.full((N, ctx, H, W), float("nan"), device=d)
    g.call("uz_conv_bwd_data", dy32, Cout, cty, w, dx_ref[:, ctx - Cin:], Cin, ctx, N, H, W, 3, 0, None, None, ws, wsb2)
    dx_a = torch.full((N, ctx, H, W), float("nan"), device=d, dtype=torch.bfloat16)
    g.call("uz_conv_bwd_data_b16", dy16, Cout, cty, w, dx_a[:, ctx - Cin:], Cin, ctx, N, H, W, 3, 0, ws, wsb2, None, 1, 1)
    assert torch.equal(dx_a[:, ctx - Cin:], dx_ref[:, ctx - Cin:].to(torch.bfloat16))
    assert ctx == Cin or torch.isnan(dx_a[:, :ctx - Cin]).all()
    # accumulate onto a bf16 tensor: round(stored + new)
    base = _rb(g.rnd(N, Cin, H, W, seed=9)).to(d)
    dx_acc = torch.full((N, ctx, H, W), float("nan"), device=d, dtype=torch.bfloat16)
    dx_acc[:, ctx - Cin:] = base.to(torch.bfloat16)
    g.call("uz_conv_bwd_data_b16", dy16, Cout, cty, w, dx_acc[:, ctx - Cin:], Cin, ctx, N, H, W, 3, 1, ws, wsb2, None, 1, 1)
    assert torch.equal(dx_acc[:, ctx - Cin:], (dx_ref[:, ctx - Cin:] + base).to(torch.bfloat16))
    # sanity against torch (the arithmetic itself is pinned in test_ops_gpu.py)
    yt = F.conv2d(x, _rb(w.cpu()), b.cpu(), padding=1)
    assert g.relerr(y_ref[:, :Cout], yt) <= 2e-5


@pytest.mark.parametrize("N,Cin,Cout,H,W", [(2, 64, 64, 64, 64), (12, 32, 32, 64, 96), (8, 96, 32, 128, 128), (2, 72, 80, 48, 64), (8, 64, 128, 64, 32)])
def test_weight_gradient_with_operands_in_bf16_storage(bf16_mode, N, Cin, Cout, H, W):
    g, L = _g(), bf16_mode
    d = g.dev()
    if L.uz_conv_route(2, Cin, Cout, N, H, W, 3) != 1:
        pytest.skip("shape not on the matrix-pipe path")
    x = _rb(g.rnd(N, Cin, H, W, seed=11))
    dy = _rb(g.rnd(N, Cout, H, W, seed=12))
    wsb = L.uz_conv_bwd_weight_workspace(Cin, Cout, N, H, W, 3)
    ws = torch.empty(wsb // 4 + 64, device=d)
    x32b, x32 = _embed(x, Cin + 5, 5, torch.float32)
    x16b, x16 = _embed(x, Cin + 5, 5, torch.bfloat16)
    dy32b, dy32 = _embed(dy, Cout + 3, 0, torch.float32)
    dy16b, dy16 = _embed(dy, Cout + 3, 0, torch.bfloat16)
    dw_ref = torch.empty(Cout, Cin, 3, 3, device=d)
    g.call("uz_conv_bwd_weight", x32, Cin, Cin + 5, dy32, Cout, Cout + 3, dw_ref, None, N, H, W, 3, None, None, ws, wsb)
    ref = torch.nn.grad.conv2d_weight(x, (Cout, Cin, 3, 3), dy, padding=1)
    assert g.relerr(dw_ref, ref) <= 5e-5
    errs = {}
    for xb, db in ((0, 0), (1, 0), (0, 1), (1, 1)):
        dw = torch.full_like(dw_ref, float("nan"))
        g.call("uz_conv_bwd_weight_b16", x16 if xb else x32, Cin, Cin + 5, dy16 if db else dy32, Cout, Cout + 3, dw, N, H, W, 3, ws, wsb, xb, db, None)
        errs[(xb, db)] = (bool(torch.equal(dw, dw_ref)), g.relerr(dw, dw_ref))
    assert all(v[0] for v in errs.values()), errs


@pytest.mark.parametrize("N,C,H,W,relu", [(40, 24, 32, 32, 1), (3, 40, 128, 128, 1), (2, 16, 128, 192, 0)])
def test_batchnorm_in_bf16_storage(N, C, H, W, relu):
    """uz_bn_relu_fwd_b16 / uz_bn_relu_bwd_b16 against the fp32-storage entry points on the same (bf16-representable) tensors:
    statistics and parameter gradients agree to rounding of the summation order, outputs are the rounding of the fp32-storage
    outputs (one bf16 ulp where the two statistics differ in the last bit)."""
    g, L = _g(), _lib()
    d = g.dev()
    y = _rb(g.rnd(N, C, H, W, seed=21) * 1.7 + 0.4)
    da = _rb(g.rnd(N, C, H, W, seed=22))
    gamma, beta = (g.rnd(C, seed=23).abs() + 0.5).to(d), (g.rnd(C, seed=24) * 0.3).to(d)
    ws = torch.empty(L.uz_bn_workspace(C, N, H, W) // 4 + 64, device=d)
    y32b, y32 = _embed(y, C + 2, 2, torch.float32)
    y16b, y16 = _embed(y, C + 2, 2, torch.bfloat16)
    rm0, rv0 = torch.zeros(C, device=d), torch.ones(C, device=d)
    save0, a0 = torch.empty(2 * C, device=d), torch.empty(N, C, H, W, device=d)
    g.call("uz_bn_relu_fwd", y32, C, C + 2, gamma, beta, rm0, rv0, save0, a0, C, N, H, W, 1e-3, 0.01, 1, relu, None, ws)
    rm1, rv1 = torch.zeros(C, device=d), torch.ones(C, device=d)
    save1 = torch.empty(2 * C, device=d)
    a1 = torch.full((N, C + 1, H, W), float("nan"), device=d, dtype=torch.bfloat16)
    g.call("uz_bn_relu_fwd_b16", y16, C, C + 2, gamma, beta, rm1, rv1, save1, a1, C + 1, N, H, W, 1e-3, 0.01, 1, relu, ws, None, 0, 1, 1)
    assert torch.allclose(save1, save0, rtol=2e-6, atol=1e-7) and torch.allclose(rm1, rm0, rtol=1e-5, atol=1e-8) and torch.allclose(rv1, rv0, rtol=1e-5)
    assert torch.isnan(a1[:, C:]).all()
    a1f = a1[:, :C].float()
    ulp = a0.abs().clamp_min(1e-30) * 2.0 ** -7
    assert bool(((a1f - a0).abs() <= ulp).all())
    assert float((a1[:, :C] != a0.to(torch.bfloat16)).float().mean()) < 1e-3           # all but a sliver: exactly the rounding of the fp32-storage output
    # fp32 output from a bf16 input: no rounding at all
    a2 = torch.empty(N, C, H, W, device=d)
    save2 = torch.empty(2 * C, device=d)
    g.call("uz_bn_relu_fwd_b16", y16, C, C + 2, gamma, beta, None, None, save2, a2, C, N, H, W, 1e-3, 0.01, 1, relu, ws, None, 0, 1, 0)
    assert g.maxabs(a2, a0) <= 2e-6 * float(a0.abs().max())
    # ---- backward
    dy0 = torch.empty(N, C, H, W, device=d)
    dg0, db0, dbias0 = (torch.empty(C, device=d) for _ in range(3))
    g.call("uz_bn_relu_bwd", da.to(d), C, y32, C, C + 2, gamma, beta, save0, dy0, C, dg0, db0, dbias0, N, H, W, relu, None, ws)
    da16 = da.to(d).to(torch.bfloat16)
    dy1 = torch.full((N, C + 3, H, W), float("nan"), device=d, dtype=torch.bfloat16)
    dg1, db1, dbias1 = (torch.full((C,), float("nan"), device=d) for _ in range(3))
    g.call("uz_bn_relu_bwd_b16", da16, C, y16, C, C + 2, gamma, beta, save0, dy1[:, 3:], C + 3, dg1, db1, dbias1, N, H, W, relu, ws, 1, 1, 1)
    assert torch.allclose(dg1, dg0, rtol=1e-5, atol=1e-3) and torch.allclose(db1, db0, rtol=1e-5, atol=1e-3)
    assert torch.isnan(dy1[:, :3]).all()
    d1 = dy1[:, 3:].float()
    assert bool(((d1 - dy0).abs() <= dy0.abs() * 2.0 ** -7 + 1e-6 * float(dy0.abs().max())).all())
    # conv-bias gradient = the sum of the STORED dy
    assert torch.allclose(dbias1.double().cpu(), d1.double().sum((0, 2, 3)).cpu(), rtol=1e-6, atol=1e-3)
    # statistics from a convolution's partials (the plans' path): one synthetic partial row per channel
    part = torch.empty(1, C, 4, device=d)
    yd = y.to(d)
    part[0, :, 0], part[0, :, 1] = yd.sum((0, 2, 3)), (yd * yd).sum((0, 2, 3))
    part[0, :, 2], part[0, :, 3] = yd.amax((0, 2, 3)), (-yd).amax((0, 2, 3))
    save3 = torch.empty(2 * C, device=d)
    a3 = torch.empty(N, C, H, W, device=d, dtype=torch.bfloat16)
    g.call("uz_bn_relu_fwd_b16", y16, C, C + 2, gamma, beta, None, None, save3, a3, C, N, H, W, 1e-3, 0.01, 1, relu, ws, part, 1, 1, 1)
    assert torch.allclose(save3, save0, rtol=1e-4, atol=1e-5)
    assert bool(((a3.float() - a0).abs() <= a0.abs() * 2.0 ** -6 + 1e-4).all())


@pytest.mark.parametrize("C,D,H,W", [(5, 8, 32, 48), (3, 7, 16, 64)])
def test_pooling_and_depth_interpolation_in_bf16_storage(C, D, H, W):
    g, L = _g(), _lib()
    d = g.dev()
    x = _rb(g.rnd(D, C, H, W, seed=31))
    x32, x16 = x.to(d), x.to(d).to(torch.bfloat16)
    Do, Ho, Wo = (D + 1) // 2, H // 2, W // 2
    y0 = torch.empty(Do, C, Ho, Wo, device=d)
    g.call("uz_avgpool3d_fwd", x32, C, C, y0, C, D, H, W)
    y1 = torch.full((Do, C + 1, Ho, Wo), float("nan"), device=d, dtype=torch.bfloat16)
    g.call("uz_avgpool3d_fwd_b16", x16, C, C, y1[:, 1:], C + 1, D, H, W, 1, 1)
    assert torch.equal(y1[:, 1:], y0.to(torch.bfloat16)) and torch.isnan(y1[:, :1]).all()
    gy = _rb(g.rnd(Do, C, Ho, Wo, seed=32)).to(d)
    base = _rb(g.rnd(D, C, H, W, seed=33)).to(d)
    dx0 = base.clone()
    g.call("uz_avgpool3d_bwd", gy, C, C, dx0, C, D, H, W, 1)
    dx1 = base.to(torch.bfloat16)
    g.call("uz_avgpool3d_bwd_b16", gy.to(torch.bfloat16), C, C, dx1, C, D, H, W, 1, 1, 1)
    assert torch.equal(dx1, dx0.to(torch.bfloat16))
    # depth stage of the trilinear interpolation: fp32 in (the in-plane stage's output) -> bf16 out, and its backward
    z0 = torch.empty(2 * D, C, H, W, device=d)
    g.call("uz_depth_lerp2x_fwd", x32, C, C, z0, C, D, H, W)
    z1 = torch.empty(2 * D, C, H, W, device=d, dtype=torch.bfloat16)
    g.call("uz_depth_lerp2x_fwd_b16", x32, C, C, z1, C, D, H, W, 0, 1)
    assert torch.equal(z1, z0.to(torch.bfloat16))
    gz = _rb(g.rnd(2 * D, C, H, W, seed=34)).to(d)
    dxa = torch.empty(D, C, H, W, device=d)
    g.call("uz_depth_lerp2x_bwd", gz, C, C, dxa, C, D, H, W, 0)
    dxb = torch.empty(D, C, H, W, device=d)
    g.call("uz_depth_lerp2x_bwd_b16", gz.to(torch.bfloat16), C, C, dxb, C, D, H, W, 0, 1, 0)
    assert torch.equal(dxb, dxa)
    # conversions
    t = g.rnd(1000, seed=35).to(d)
    h16 = torch.empty(1000, device=d, dtype=torch.bfloat16)
    g.call("uz_cvt_f32_to_b16", t, h16, 1000)
    assert torch.equal(h16, t.to(torch.bfloat16))
    back = torch.empty(1000, device=d)
    g.call("uz_cvt_b16_to_f32", h16, back, 1000)
    assert torch.equal(back, h16.float())


@pytest.mark.parametrize("N,Cin,Cout,H,W", [(6, 96, 2, 64, 32), (3, 40, 3, 32, 48)])
def test_one_by_one_heads_with_the_wide_operand_in_bf16_storage(N, Cin, Cout, H, W):
    """mu_conv / sigma_conv / s_layer (phiseg3D.py:83-84): x (forward, weight gradient) / dx (data gradient) in bf16 storage, the
    2 - 3-channel side fp32.  Arithmetic is fp32 VALU on widened values: a bf16 INPUT gives bit-identical results, a bf16 OUTPUT the
    rounding of the fp32 one."""
    g, L = _g(), _lib()
    d = g.dev()
    x = _rb(g.rnd(N, Cin, H, W, seed=41))
    w = g.rnd(Cout, Cin, 1, 1, seed=42, scale=0.2).to(d)
    b = g.rnd(Cout, seed=43).to(d)
    dy = g.rnd(N, Cout, H, W, seed=44).to(d)
    x32b, x32 = _embed(x, Cin + 3, 3, torch.float32)
    x16b, x16 = _embed(x, Cin + 3, 3, torch.bfloat16)
    y0, y1 = torch.empty(N, Cout, H, W, device=d), torch.empty(N, Cout, H, W, device=d)
    g.call("uz_conv_fwd", x32, Cin, Cin + 3, w, b, y0, Cout, Cout, N, H, W, 1, 0, None, None, None, None, 0)
    g.call("uz_conv1x1_fwd_b16", x16, Cin, Cin + 3, w, b, y1, Cout, Cout, N, H, W, 1)
    assert torch.equal(y0, y1)
    assert g.relerr(y0, F.conv2d(x, w.cpu(), b.cpu())) <= 2e-6
    wsb = L.uz_conv_bwd_weight_workspace(Cin, Cout, N, H, W, 1)
    ws = torch.empty(wsb // 4 + 64, device=d)
    dw0, db0, dw1, db1 = torch.empty_like(w), torch.empty_like(b), torch.full_like(w, float("nan")), torch.full_like(b, float("nan"))
    g.call("uz_conv_bwd_weight", x32, Cin, Cin + 3, dy, Cout, Cout, dw0, db0, N, H, W, 1, None, None, ws, wsb)
    g.call("uz_conv1x1_bwd_weight_b16", x16, Cin, Cin + 3, dy, Cout, Cout, dw1, db1, N, H, W, ws, wsb, 1)
    assert torch.equal(dw0, dw1) and torch.equal(db0, db1)
    base = _rb(g.rnd(N, Cin, H, W, seed=45)).to(d)
    dx0 = torch.full((N, Cin + 3, H, W), float("nan"), device=d)
    dx0[:, 3:] = base
    g.call("uz_conv_bwd_data", dy, Cout, Cout, w, dx0[:, 3:], Cin, Cin + 3, N, H, W, 1, 1, None, None, None, 0)
    dx1 = torch.full((N, Cin + 3, H, W), float("nan"), device=d, dtype=torch.bfloat16)
    dx1[:, 3:] = base.to(torch.bfloat16)
    g.call("uz_conv1x1_bwd_data_b16", dy, Cout, Cout, w, dx1[:, 3:], Cin, Cin + 3, N, H, W, 1, 1)
    assert torch.equal(dx1[:, 3:], dx0[:, 3:].to(torch.bfloat16)) and torch.isnan(dx1[:, :3]).all()


@pytest.mark.parametrize("N,C,H,W,ac", [(5, 6, 32, 16, 1), (3, 4, 64, 32, 1), (2, 3, 16, 64, 0)])
def test_in_plane_interpolation_with_the_high_resolution_side_in_bf16_storage(N, C, H, W, ac):
    g, L = _g(), _lib()
    d = g.dev()
    x = g.rnd(N, C, H, W, seed=51).to(d)
    y0 = torch.empty(N, C, 2 * H, 2 * W, device=d)
    g.call("uz_bilinear2x_fwd", x, C, C, y0, C, N, H, W, ac, None, None)
    y1 = torch.full((N, C + 2, 2 * H, 2 * W), float("nan"), device=d, dtype=torch.bfloat16)
    g.call("uz_bilinear2x_fwd_b16", x, C, C, y1[:, 2:], C + 2, N, H, W, ac, 1)
    assert torch.equal(y1[:, 2:], y0.to(torch.bfloat16)) and torch.isnan(y1[:, :2]).all()
    gy = _rb(g.rnd(N, C, 2 * H, 2 * W, seed=52)).to(d)
    base = g.rnd(N, C, H, W, seed=53).to(d)
    dx0, dx1 = base.clone(), base.clone()
    g.call("uz_bilinear2x_bwd", gy, C, C, dx0, C, N, H, W, ac, 1)
    g.call("uz_bilinear2x_bwd_b16", gy.to(torch.bfloat16), C, C, dx1, C, N, H, W, ac, 1, 1)
    assert torch.equal(dx0, dx1)
