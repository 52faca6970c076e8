// Volumetric (PHiSeg3D, models/phiseg3D.py) support kernels.  A volume lives in HBM as [D + 2][C][H][W] (one sample; a zero
// slice before and after the D real ones), i.e. a batch of D channel-major 2-D images.  With that layout
//   * Conv3d(3x3x3, pad 1) (phiseg3D.py:24) IS a 2-D 3x3 convolution over the batch of slices whose input has 3 C channels:
//     the slices d-1, d, d+1 are adjacent in memory, so the view (pointer = slice d-1, Cin = 3 C, batch stride = ONE slice)
//     is the depth window, the zero slices are the depth padding, and the 2-D implicit-GEMM kernels (conv_split.hip,
//     conv_wgrad_split.hip, conv_mfma.hip) run it unchanged with K = 27 Cin.  Only the weights need a permutation, done by the
//     three small kernels below (forward: [co][kd][ci][3][3]; data gradient: [(j, co)][ci][3][3] with kd = 2 - j; weight
//     gradient: back from [co][kd][ci][9] to the parameter layout [co][ci][kd][9]);
//   * BatchNorm3d / ReLU / 1x1x1 heads / latent ops are their 2-D kernels over the batch of D slices;
//   * AvgPool3d(2, ceil_mode) (phiseg3D.py:101) and trilinear x2 (align_corners=True: phiseg3D.py:146,306,376) are the kernels
//     below (trilinear = the 2-D bilinear kernel per slice, then linear interpolation along the depth).
#include <type_traits>
#include "uz_common.h"
#include "split_f16.h"

namespace {

__global__ void absmax_copy_k(const float* src, float* dst) { uz::amax_publish_one(uz::amax_read(src), dst, 0u); }


// ---- weight permutations (all tiny: <= 27 * 256 * 256 floats)
// mode 0: out[co][kd][ci][9] = w[co][ci][kd][9]                    (forward, Cin' = 3 Cin)
// mode 1: out[j][co][ci][9]  = w[co][ci][2 - j][9]                 (data gradient: rows (j, co) = 3 Cout "output" channels)
// mode 2: dw[co][ci][kd][9]  = src[co][kd][ci][9]                  (weight gradient back to the parameter layout)
__global__ __launch_bounds__(256) void w3d_permute_k(const float* __restrict__ src, float* __restrict__ dst, int Cout, int Cin, int mode) {
    const int n = Cout * Cin * 27;
    for (int e = blockIdx.x * 256 + threadIdx.x; e < n; e += gridDim.x * 256) {
        const int t = e % 9, r = e / 9;
        if (mode == 0) { const int ci = r % Cin, r2 = r / Cin, kd = r2 % 3, co = r2 / 3; dst[e] = src[((co * Cin + ci) * 3 + kd) * 9 + t]; }
        else if (mode == 1) { const int ci = r % Cin, r2 = r / Cin, co = r2 % Cout, j = r2 / Cout; dst[e] = src[((co * Cin + ci) * 3 + (2 - j)) * 9 + t]; }
        else { const int kd = r % 3, r2 = r / 3, ci = r2 % Cin, co = r2 / Cin; dst[e] = src[((co * 3 + kd) * Cin + ci) * 9 + t]; }
    }
}

// ---- AvgPool3d(kernel 2, stride 2, ceil_mode): windows clipped at the far faces divide by the in-bounds count
__global__ __launch_bounds__(256) void avgpool3d_fwd_k(const float* __restrict__ x, int CtotX, float* __restrict__ y, int CtotY,
                                                        int C, int D, int H, int W, int Do, int Ho, int Wo) {
    const int od = blockIdx.z, c = blockIdx.y, n = Ho * Wo;
    const int d0 = 2 * od, d1 = min(d0 + 2, D);
    for (int q = blockIdx.x * 256 + threadIdx.x; q < n; q += gridDim.x * 256) {
        const int oy = q / Wo, ox = q - oy * Wo, y0 = 2 * oy, x0 = 2 * ox, y1 = min(y0 + 2, H), x1 = min(x0 + 2, W);
        float acc = 0.f;
        for (int d = d0; d < d1; ++d)
            for (int yy = y0; yy < y1; ++yy)
                for (int xx = x0; xx < x1; ++xx) acc += x[((size_t)d * CtotX + c) * H * W + yy * W + xx];
        y[((size_t)od * CtotY + c) * n + q] = acc / (float)((d1 - d0) * (y1 - y0) * (x1 - x0));
    }
}
// W % 4 == 0 (and 16-byte aligned views): a thread turns up to 2 x 2 float4 loads into one float2 of outputs
__global__ __launch_bounds__(256) void avgpool3d_fwd_v4(const float* __restrict__ x, int CtotX, float* __restrict__ y, int CtotY,
                                                         int C, int D, int H, int W, int Do, int Ho, int Wo) {
    const int od = blockIdx.z, c = blockIdx.y, W4 = W / 4, n2 = Ho * W4;
    const int d0 = 2 * od, d1 = min(d0 + 2, D);
    for (int q = blockIdx.x * 256 + threadIdx.x; q < n2; q += gridDim.x * 256) {
        const int oy = q / W4, j = q - oy * W4, y0 = 2 * oy, y1 = min(y0 + 2, H);
        float a0 = 0.f, a1 = 0.f;
        for (int d = d0; d < d1; ++d)
            for (int yy = y0; yy < y1; ++yy) {
                const float4 v = *reinterpret_cast<const float4*>(x + ((size_t)d * CtotX + c) * H * W + (size_t)yy * W + 4 * j);
                a0 += v.x + v.y; a1 += v.z + v.w;
            }
        const float inv = 1.f / (float)((d1 - d0) * (y1 - y0) * 2);
        *reinterpret_cast<float2*>(y + ((size_t)od * CtotY + c) * Ho * Wo + (size_t)oy * Wo + 2 * j) = make_float2(a0 * inv, a1 * inv);
    }
}
__global__ __launch_bounds__(256) void avgpool3d_bwd_k(const float* __restrict__ dy, int CtotDy, float* __restrict__ dx, int CtotDx,
                                                        int C, int D, int H, int W, int Do, int Ho, int Wo, int accumulate) {
    const int d = blockIdx.z, c = blockIdx.y, n = H * W, od = d >> 1;
    for (int q = blockIdx.x * 256 + threadIdx.x; q < n; q += gridDim.x * 256) {
        const int yy = q / W, xx = q - yy * W, oy = yy >> 1, ox = xx >> 1;
        const int cnt = (min(2 * od + 2, D) - 2 * od) * (min(2 * oy + 2, H) - 2 * oy) * (min(2 * ox + 2, W) - 2 * ox);
        const float v = dy[((size_t)od * CtotDy + c) * Ho * Wo + oy * Wo + ox] / (float)cnt;
        float* dst = dx + ((size_t)d * CtotDx + c) * n + q;
        *dst = accumulate ? *dst + v : v;
    }
}
__global__ __launch_bounds__(256) void avgpool3d_bwd_v4(const float* __restrict__ dy, int CtotDy, float* __restrict__ dx, int CtotDx,
                                                         int C, int D, int H, int W, int Do, int Ho, int Wo, int accumulate) {
    const int d = blockIdx.z, c = blockIdx.y, W4 = W / 4, n4 = H * W4, od = d >> 1;
    const int cd = min(2 * od + 2, D) - 2 * od;
    for (int q = blockIdx.x * 256 + threadIdx.x; q < n4; q += gridDim.x * 256) {
        const int yy = q / W4, j = q - yy * W4, oy = yy >> 1;
        const float inv = 1.f / (float)(cd * (min(2 * oy + 2, H) - 2 * oy) * 2);
        const float2 g = *reinterpret_cast<const float2*>(dy + ((size_t)od * CtotDy + c) * Ho * Wo + (size_t)oy * Wo + 2 * j);
        float4* dst = reinterpret_cast<float4*>(dx + ((size_t)d * CtotDx + c) * H * W + (size_t)yy * W + 4 * j);
        float4 v = make_float4(g.x * inv, g.x * inv, g.y * inv, g.y * inv);
        if (accumulate) { const float4 o = *dst; v.x += o.x; v.y += o.y; v.z += o.z; v.w += o.w; }
        *dst = v;
    }
}

// ---- depth stage of trilinear x2, align_corners=True: y[od] = (1 - t) x[d0] + t x[d0 + 1], source = od (D - 1) / (2 D - 1)
__device__ __forceinline__ void depth_src(int od, int D, int& d0, int& dp, float& t) {
    const float r = (2 * D > 1) ? (float)od * ((float)(D - 1) / (float)(2 * D - 1)) : 0.f;
    d0 = (int)r;
    if (d0 > D - 1) d0 = D - 1;
    dp = d0 < D - 1 ? 1 : 0;
    t = r - (float)d0;
}
__global__ __launch_bounds__(256) void depth_lerp_fwd_k(const float* __restrict__ x, int CtotX, float* __restrict__ y, int CtotY,
                                                         int C, int D, int HW) {
    const int od = blockIdx.z, c = blockIdx.y;
    int d0, dp; float t;
    depth_src(od, D, d0, dp, t);
    const float* a = x + ((size_t)d0 * CtotX + c) * HW;
    const float* b = x + ((size_t)(d0 + dp) * CtotX + c) * HW;
    float* o = y + ((size_t)od * CtotY + c) * HW;
    for (int q = blockIdx.x * 256 + threadIdx.x; q < HW; q += gridDim.x * 256) o[q] = (1.f - t) * a[q] + t * b[q];
}
__global__ __launch_bounds__(256) void depth_lerp_fwd_v4(const float* __restrict__ x, int CtotX, float* __restrict__ y, int CtotY,
                                                          int C, int D, int HW) {
    const int od = blockIdx.z, c = blockIdx.y;
    int d0, dp; float t;
    depth_src(od, D, d0, dp, t);
    const float4* a = reinterpret_cast<const float4*>(x + ((size_t)d0 * CtotX + c) * HW);
    const float4* b = reinterpret_cast<const float4*>(x + ((size_t)(d0 + dp) * CtotX + c) * HW);
    float4* o = reinterpret_cast<float4*>(y + ((size_t)od * CtotY + c) * HW);
    const float s = 1.f - t;
    for (int q = blockIdx.x * 256 + threadIdx.x; q < HW / 4; q += gridDim.x * 256) {
        const float4 u = a[q], v = b[q];
        o[q] = make_float4(s * u.x + t * v.x, s * u.y + t * v.y, s * u.z + t * v.z, s * u.w + t * v.w);
    }
}
__global__ __launch_bounds__(256) void depth_lerp_bwd_v4(const float* __restrict__ dy, int CtotDy, float* __restrict__ dx, int CtotDx,
                                                          int C, int D, int HW, int accumulate) {
    const int d = blockIdx.z, c = blockIdx.y;
    float w[6]; int ods[6], nw = 0;
    for (int od = max(0, 2 * d - 2); od <= min(2 * D - 1, 2 * d + 3); ++od) {
        int d0, dp; float t;
        depth_src(od, D, d0, dp, t);
        const float ww = (d0 == d ? 1.f - t : 0.f) + (d0 + dp == d ? t : 0.f);
        if (ww != 0.f) { w[nw] = ww; ods[nw] = od; ++nw; }
    }
    float4* o = reinterpret_cast<float4*>(dx + ((size_t)d * CtotDx + c) * HW);
    for (int q = blockIdx.x * 256 + threadIdx.x; q < HW / 4; q += gridDim.x * 256) {
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
        for (int k = 0; k < nw; ++k) {
            const float4 g = reinterpret_cast<const float4*>(dy + ((size_t)ods[k] * CtotDy + c) * HW)[q];
            acc.x += w[k] * g.x; acc.y += w[k] * g.y; acc.z += w[k] * g.z; acc.w += w[k] * g.w;
        }
        if (accumulate) { const float4 p = o[q]; acc.x += p.x; acc.y += p.y; acc.z += p.z; acc.w += p.w; }
        o[q] = acc;
    }
}
__global__ __launch_bounds__(256) void depth_lerp_bwd_k(const float* __restrict__ dy, int CtotDy, float* __restrict__ dx, int CtotDx,
                                                         int C, int D, int HW, int accumulate) {
    const int d = blockIdx.z, c = blockIdx.y;
    // output slices whose footprint touches d: od in a small window around 2 d (weights recomputed exactly as in the forward)
    float w[6]; int ods[6], nw = 0;
    for (int od = max(0, 2 * d - 2); od <= min(2 * D - 1, 2 * d + 3); ++od) {
        int d0, dp; float t;
        depth_src(od, D, d0, dp, t);
        const float ww = (d0 == d ? 1.f - t : 0.f) + (d0 + dp == d ? t : 0.f);
        if (ww != 0.f) { w[nw] = ww; ods[nw] = od; ++nw; }
    }
    float* o = dx + ((size_t)d * CtotDx + c) * HW;
    for (int q = blockIdx.x * 256 + threadIdx.x; q < HW; q += gridDim.x * 256) {
        float acc = 0.f;
        for (int k = 0; k < nw; ++k) acc += w[k] * dy[((size_t)ods[k] * CtotDy + c) * HW + q];
        o[q] = accumulate ? o[q] + acc : acc;
    }
}

// ---- nearest resize of a volume by integer factors (fz, f, f): the 3-D counterpart of phiseg.py:321 (the reference's own
// 3-D call, phiseg3D.py:398, passes a 2-element size to a 5-D tensor and raises; this is the evident intent)
__global__ __launch_bounds__(256) void nearest3d_fwd_k(const float* __restrict__ x, int CtotX, float* __restrict__ y, int CtotY,
                                                        int H, int W, int f, int fz) {
    const int od = blockIdx.z, c = blockIdx.y, Ho = H * f, Wo = W * f, n = Ho * Wo;
    const float* s = x + ((size_t)(od / fz) * CtotX + c) * H * W;
    float* o = y + ((size_t)od * CtotY + c) * n;
    for (int q = blockIdx.x * 256 + threadIdx.x; q < n; q += gridDim.x * 256) { const int oy = q / Wo, ox = q - oy * Wo; o[q] = s[(oy / f) * W + ox / f]; }
}
__global__ __launch_bounds__(256) void nearest3d_bwd_k(const float* __restrict__ dy, int CtotDy, float* __restrict__ dx, int CtotDx,
                                                        int H, int W, int f, int fz, int accumulate) {
    const int d = blockIdx.z, c = blockIdx.y, Ho = H * f, Wo = W * f, n = H * W;
    float* o = dx + ((size_t)d * CtotDx + c) * n;
    for (int q = blockIdx.x * 256 + threadIdx.x; q < n; q += gridDim.x * 256) {
        const int yy = q / W, xx = q - yy * W;
        float acc = 0.f;
        for (int k = 0; k < fz; ++k)
            for (int i = 0; i < f; ++i)
                for (int j = 0; j < f; ++j) acc += dy[((size_t)(d * fz + k) * CtotDy + c) * Ho * Wo + (yy * f + i) * Wo + xx * f + j];
        o[q] = accumulate ? o[q] + acc : acc;
    }
}

// Large factors (the deep levels' logits resized to the full volume: 16 x 16 x 16 = 4 096 children per element of an 8 x 8 x 4
// tensor): one WAVE per low-resolution element - lanes stride over the children (x fastest: coalesced runs of f floats), ordered
// butterfly at the end - instead of one thread walking 4 096 strided values (397 us for 768 threads' worth of work).
__global__ __launch_bounds__(256) void nearest3d_bwd_wave_k(const float* __restrict__ dy, int CtotDy, float* __restrict__ dx, int CtotDx,
                                                             int H, int W, int f, int fz, int accumulate) {
    const int d = blockIdx.z, c = blockIdx.y, Ho = H * f, Wo = W * f, n = H * W;
    const int q = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (q >= n) return;
    const int yy = q / W, xx = q - yy * W, m = f * f * fz;
    float acc = 0.f;
    for (int e = lane; e < m; e += 64) {
        const int j = e % f, r = e / f, i = r % f, k = r / f;
        acc += dy[((size_t)(d * fz + k) * CtotDy + c) * Ho * Wo + (size_t)(yy * f + i) * Wo + xx * f + j];
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o, 64);
    if (lane == 0) {
        float* o = dx + ((size_t)d * CtotDx + c) * n + q;
        *o = accumulate ? *o + acc : acc;
    }
}

// ---- bf16 STORAGE variants (include/uz_api.h, "bf16 storage"): the float4 kernels above with each tensor either fp32 or bf16
// (xb / yb: workgroup-uniform flags), arithmetic in fp32, values rounded to nearest even where the output is bf16
__global__ __launch_bounds__(256) void avgpool3d_fwd_st(const float* __restrict__ x, int CtotX, float* __restrict__ y, int CtotY,
                                                         int C, int D, int H, int W, int Do, int Ho, int Wo, int xb, int yb) {
    const int od = blockIdx.z, c = blockIdx.y, W4 = W / 4, n2 = Ho * W4;
    const int d0 = 2 * od, d1 = min(d0 + 2, D);
    for (int q = blockIdx.x * 256 + threadIdx.x; q < n2; q += gridDim.x * 256) {
        const int oy = q / W4, j = q - oy * W4, y0 = 2 * oy, y1 = min(y0 + 2, H);
        float a0 = 0.f, a1 = 0.f;
        for (int d = d0; d < d1; ++d)
            for (int yy = y0; yy < y1; ++yy) {
                const uz::f32x4 v = uz::ld_elem4(x, ((size_t)d * CtotX + c) * H * W + (size_t)yy * W + 4 * j, xb);
                a0 += v.x + v.y; a1 += v.z + v.w;
            }
        const float inv = 1.f / (float)((d1 - d0) * (y1 - y0) * 2);
        const size_t o = ((size_t)od * CtotY + c) * Ho * Wo + (size_t)oy * Wo + 2 * j;
        if (yb) *reinterpret_cast<unsigned*>(reinterpret_cast<unsigned short*>(y) + o) = uz::pack_bf16x2(a0 * inv, a1 * inv);
        else *reinterpret_cast<float2*>(y + o) = make_float2(a0 * inv, a1 * inv);
    }
}
__global__ __launch_bounds__(256) void avgpool3d_bwd_st(const float* __restrict__ dy, int CtotDy, float* __restrict__ dx, int CtotDx,
                                                         int C, int D, int H, int W, int Do, int Ho, int Wo, int accumulate, int dyb, int dxb) {
    const int d = blockIdx.z, c = blockIdx.y, W4 = W / 4, n4 = H * W4, od = d >> 1;
    const int cd = min(2 * od + 2, D) - 2 * od;
    for (int q = blockIdx.x * 256 + threadIdx.x; q < n4; q += gridDim.x * 256) {
        const int yy = q / W4, j = q - yy * W4, oy = yy >> 1;
        const float inv = 1.f / (float)(cd * (min(2 * oy + 2, H) - 2 * oy) * 2);
        const size_t go = ((size_t)od * CtotDy + c) * Ho * Wo + (size_t)oy * Wo + 2 * j;
        float g0, g1;
        if (dyb) { const unsigned w = *reinterpret_cast<const unsigned*>(reinterpret_cast<const unsigned short*>(dy) + go); g0 = uz::bf16_lo(w); g1 = uz::bf16_hi(w); }
        else { const float2 g = *reinterpret_cast<const float2*>(dy + go); g0 = g.x; g1 = g.y; }
        const size_t o = ((size_t)d * CtotDx + c) * H * W + (size_t)yy * W + 4 * j;
        uz::f32x4 v = {g0 * inv, g0 * inv, g1 * inv, g1 * inv};
        if (accumulate) v += uz::ld_elem4(dx, o, dxb);
        uz::st_elem4(dx, o, v, dxb);
    }
}
__global__ __launch_bounds__(256) void depth_lerp_fwd_st(const float* __restrict__ x, int CtotX, float* __restrict__ y, int CtotY,
                                                          int C, int D, int HW, int xb, int yb) {
    const int od = blockIdx.z, c = blockIdx.y;
    int d0, dp; float t;
    depth_src(od, D, d0, dp, t);
    const size_t ra = ((size_t)d0 * CtotX + c) * HW, rb = ((size_t)(d0 + dp) * CtotX + c) * HW, ro = ((size_t)od * CtotY + c) * HW;
    const float s = 1.f - t;
    for (int q = blockIdx.x * 256 + threadIdx.x; q < HW / 4; q += gridDim.x * 256) {
        const uz::f32x4 u = uz::ld_elem4(x, ra + 4 * (size_t)q, xb), v = uz::ld_elem4(x, rb + 4 * (size_t)q, xb);
        uz::st_elem4(y, ro + 4 * (size_t)q, uz::f32x4{s * u.x + t * v.x, s * u.y + t * v.y, s * u.z + t * v.z, s * u.w + t * v.w}, yb);
    }
}
__global__ __launch_bounds__(256) void depth_lerp_bwd_st(const float* __restrict__ dy, int CtotDy, float* __restrict__ dx, int CtotDx,
                                                          int C, int D, int HW, int accumulate, int dyb, int dxb) {
    const int d = blockIdx.z, c = blockIdx.y;
    float w[6]; int ods[6], nw = 0;
    for (int od = max(0, 2 * d - 2); od <= min(2 * D - 1, 2 * d + 3); ++od) {
        int d0, dp; float t;
        depth_src(od, D, d0, dp, t);
        const float ww = (d0 == d ? 1.f - t : 0.f) + (d0 + dp == d ? t : 0.f);
        if (ww != 0.f) { w[nw] = ww; ods[nw] = od; ++nw; }
    }
    const size_t ro = ((size_t)d * CtotDx + c) * HW;
    // the number of contributing slices (3 ... 5, block-uniform) becomes a compile-time count: all of a thread's loads are issued before
    // the first is used (with a run-time count the weights and slice numbers were indexed dynamically and each load waited for
    // the one before it: 173 us on 192 ch 32 -> 64 slices of 128 x 128)
    auto sweep = [&](auto nwc) {
        constexpr int NW = decltype(nwc)::value;
        for (int q = blockIdx.x * 256 + threadIdx.x; q < HW / 4; q += gridDim.x * 256) {
            uz::f32x4 v[NW];
#pragma unroll
            for (int k = 0; k < NW; ++k) v[k] = uz::ld_elem4(dy, ((size_t)ods[k] * CtotDy + c) * HW + 4 * (size_t)q, dyb);
            uz::f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int k = 0; k < NW; ++k) acc += w[k] * v[k];
            if (accumulate) acc += uz::ld_elem4(dx, ro + 4 * (size_t)q, dxb);
            uz::st_elem4(dx, ro + 4 * (size_t)q, acc, dxb);
        }
    };
    if (nw == 0) { w[0] = 0.f; ods[0] = 0; nw = 1; }
    switch (nw) {
        case 1: sweep(std::integral_constant<int, 1>{}); break;
        case 2: sweep(std::integral_constant<int, 2>{}); break;
        case 3: sweep(std::integral_constant<int, 3>{}); break;
        case 4: sweep(std::integral_constant<int, 4>{}); break;
        case 5: sweep(std::integral_constant<int, 5>{}); break;
        default: sweep(std::integral_constant<int, 6>{}); break;
    }
}

inline int gx(int n) { int g = (n + 255) / 256; return g < 1 ? 1 : (g > 64 ? 64 : g); }
inline bool a16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

}  // namespace

extern "C" int uz_absmax_copy(const float* src_slot, float* dst_slot, void* stream) {
    UZ_REQUIRE(src_slot && dst_slot, "absmax_copy: null slot");
    hipLaunchKernelGGL(absmax_copy_k, dim3(1), dim3(1), 0, uz::S(stream), src_slot, dst_slot);
    return uz::check_launch("absmax_copy_k");
}
extern "C" int uz_w3d_permute(const float* src, float* dst, int Cout, int Cin, int mode, void* stream) {
    UZ_REQUIRE(src && dst && Cout > 0 && Cin > 0 && mode >= 0 && mode <= 2, "w3d_permute: bad arguments");
    // (up to 27 x 256 x 256 elements: a grid capped at 64 workgroups walked 64 elements per thread and took 26 us a launch, 72 launches a step)
    int g = (Cout * Cin * 27 + 255) / 256;
    g = g < 1 ? 1 : (g > 2048 ? 2048 : g);
    hipLaunchKernelGGL(w3d_permute_k, dim3(g), dim3(256), 0, uz::S(stream), src, dst, Cout, Cin, mode);
    return uz::check_launch("w3d_permute_k");
}
extern "C" int uz_avgpool3d_fwd(const float* x, int C, int CtotX, float* y, int CtotY, int D, int H, int W, void* stream) {
    UZ_REQUIRE(C > 0 && D > 0 && H > 0 && W > 0 && C <= 65535 && D <= 65535, "avgpool3d_fwd: bad sizes");
    const int Do = (D + 1) / 2, Ho = (H + 1) / 2, Wo = (W + 1) / 2;
    if (W % 4 == 0 && a16(x) && a16(y) && (H * W) % 4 == 0 && (Ho * Wo) % 2 == 0)
        hipLaunchKernelGGL(avgpool3d_fwd_v4, dim3(gx(Ho * W / 4), C, Do), dim3(256), 0, uz::S(stream), x, CtotX, y, CtotY, C, D, H, W, Do, Ho, Wo);
    else
        hipLaunchKernelGGL(avgpool3d_fwd_k, dim3(gx(Ho * Wo), C, Do), dim3(256), 0, uz::S(stream), x, CtotX, y, CtotY, C, D, H, W, Do, Ho, Wo);
    return uz::check_launch("avgpool3d_fwd_k");
}
extern "C" int uz_avgpool3d_bwd(const float* dy, int C, int CtotDy, float* dx, int CtotDx, int D, int H, int W, int accumulate, void* stream) {
    UZ_REQUIRE(C > 0 && D > 0 && H > 0 && W > 0 && C <= 65535 && D <= 65535, "avgpool3d_bwd: bad sizes");
    const int Do = (D + 1) / 2, Ho = (H + 1) / 2, Wo = (W + 1) / 2;
    if (W % 4 == 0 && a16(dx) && a16(dy) && (Ho * Wo) % 2 == 0)
        hipLaunchKernelGGL(avgpool3d_bwd_v4, dim3(gx(H * W / 4), C, D), dim3(256), 0, uz::S(stream), dy, CtotDy, dx, CtotDx, C, D, H, W, Do, Ho, Wo, accumulate);
    else
        hipLaunchKernelGGL(avgpool3d_bwd_k, dim3(gx(H * W), C, D), dim3(256), 0, uz::S(stream), dy, CtotDy, dx, CtotDx, C, D, H, W, Do, Ho, Wo, accumulate);
    return uz::check_launch("avgpool3d_bwd_k");
}
// bf16 STORAGE variants: x / y (dy / dx) each fp32 or bf16; W % 4 == 0 (pooling) or H*W % 4 == 0 (depth stage), 16-byte aligned views
extern "C" int uz_avgpool3d_fwd_b16(const void* x, int C, int CtotX, void* y, int CtotY, int D, int H, int W, int x_b16, int y_b16, void* stream) {
    UZ_REQUIRE(C > 0 && D > 0 && H > 0 && W > 0 && C <= 65535 && D <= 65535, "avgpool3d_fwd_b16: bad sizes");
    UZ_REQUIRE(W % 4 == 0 && H % 2 == 0 && a16(x) && a16(y), "avgpool3d_fwd_b16: needs W %% 4 == 0, even H and 16-byte aligned views");
    const int Do = (D + 1) / 2, Ho = (H + 1) / 2, Wo = (W + 1) / 2;
    hipLaunchKernelGGL(avgpool3d_fwd_st, dim3(gx(Ho * W / 4), C, Do), dim3(256), 0, uz::S(stream), static_cast<const float*>(x), CtotX, static_cast<float*>(y), CtotY,
                       C, D, H, W, Do, Ho, Wo, x_b16, y_b16);
    return uz::check_launch("avgpool3d_fwd_st");
}
extern "C" int uz_avgpool3d_bwd_b16(const void* dy, int C, int CtotDy, void* dx, int CtotDx, int D, int H, int W, int accumulate, int dy_b16, int dx_b16, void* stream) {
    UZ_REQUIRE(C > 0 && D > 0 && H > 0 && W > 0 && C <= 65535 && D <= 65535, "avgpool3d_bwd_b16: bad sizes");
    UZ_REQUIRE(W % 4 == 0 && H % 2 == 0 && a16(dx) && a16(dy), "avgpool3d_bwd_b16: needs W %% 4 == 0, even H and 16-byte aligned views");
    const int Do = (D + 1) / 2, Ho = (H + 1) / 2, Wo = (W + 1) / 2;
    hipLaunchKernelGGL(avgpool3d_bwd_st, dim3(gx(H * W / 4), C, D), dim3(256), 0, uz::S(stream), static_cast<const float*>(dy), CtotDy, static_cast<float*>(dx), CtotDx,
                       C, D, H, W, Do, Ho, Wo, accumulate, dy_b16, dx_b16);
    return uz::check_launch("avgpool3d_bwd_st");
}
extern "C" int uz_depth_lerp2x_fwd_b16(const void* x, int C, int CtotX, void* y, int CtotY, int D, int H, int W, int x_b16, int y_b16, void* stream) {
    UZ_REQUIRE(C > 0 && D > 0 && H > 0 && W > 0 && C <= 65535 && 2 * D <= 65535, "depth_lerp2x_fwd_b16: bad sizes");
    UZ_REQUIRE((H * W) % 4 == 0 && a16(x) && a16(y), "depth_lerp2x_fwd_b16: needs H*W %% 4 == 0 and 16-byte aligned views");
    hipLaunchKernelGGL(depth_lerp_fwd_st, dim3(gx(H * W / 4), C, 2 * D), dim3(256), 0, uz::S(stream), static_cast<const float*>(x), CtotX, static_cast<float*>(y), CtotY, C, D, H * W, x_b16, y_b16);
    return uz::check_launch("depth_lerp_fwd_st");
}
extern "C" int uz_depth_lerp2x_bwd_b16(const void* dy, int C, int CtotDy, void* dx, int CtotDx, int D, int H, int W, int accumulate, int dy_b16, int dx_b16, void* stream) {
    UZ_REQUIRE(C > 0 && D > 0 && H > 0 && W > 0 && C <= 65535 && D <= 65535, "depth_lerp2x_bwd_b16: bad sizes");
    UZ_REQUIRE((H * W) % 4 == 0 && a16(dx) && a16(dy), "depth_lerp2x_bwd_b16: needs H*W %% 4 == 0 and 16-byte aligned views");
    hipLaunchKernelGGL(depth_lerp_bwd_st, dim3(gx(H * W / 4), C, D), dim3(256), 0, uz::S(stream), static_cast<const float*>(dy), CtotDy, static_cast<float*>(dx), CtotDx, C, D, H * W, accumulate, dy_b16, dx_b16);
    return uz::check_launch("depth_lerp_bwd_st");
}
extern "C" int uz_depth_lerp2x_fwd(const float* x, int C, int CtotX, float* y, int CtotY, int D, int H, int W, void* stream) {
    UZ_REQUIRE(C > 0 && D > 0 && H > 0 && W > 0 && C <= 65535 && 2 * D <= 65535, "depth_lerp2x_fwd: bad sizes");
    if ((H * W) % 4 == 0 && a16(x) && a16(y))
        hipLaunchKernelGGL(depth_lerp_fwd_v4, dim3(gx(H * W / 4), C, 2 * D), dim3(256), 0, uz::S(stream), x, CtotX, y, CtotY, C, D, H * W);
    else
        hipLaunchKernelGGL(depth_lerp_fwd_k, dim3(gx(H * W), C, 2 * D), dim3(256), 0, uz::S(stream), x, CtotX, y, CtotY, C, D, H * W);
    return uz::check_launch("depth_lerp_fwd_k");
}
extern "C" int uz_depth_lerp2x_bwd(const float* dy, int C, int CtotDy, float* dx, int CtotDx, int D, int H, int W, int accumulate, void* stream) {
    UZ_REQUIRE(C > 0 && D > 0 && H > 0 && W > 0 && C <= 65535 && D <= 65535, "depth_lerp2x_bwd: bad sizes");
    if ((H * W) % 4 == 0 && a16(dx) && a16(dy))
        hipLaunchKernelGGL(depth_lerp_bwd_v4, dim3(gx(H * W / 4), C, D), dim3(256), 0, uz::S(stream), dy, CtotDy, dx, CtotDx, C, D, H * W, accumulate);
    else
        hipLaunchKernelGGL(depth_lerp_bwd_k, dim3(gx(H * W), C, D), dim3(256), 0, uz::S(stream), dy, CtotDy, dx, CtotDx, C, D, H * W, accumulate);
    return uz::check_launch("depth_lerp_bwd_k");
}
extern "C" int uz_nearest3d_fwd(const float* x, int C, int CtotX, float* y, int CtotY, int D, int H, int W, int f, int fz, void* stream) {
    UZ_REQUIRE(C > 0 && D > 0 && f >= 1 && fz >= 1 && C <= 65535 && D * fz <= 65535, "nearest3d_fwd: bad sizes");
    hipLaunchKernelGGL(nearest3d_fwd_k, dim3(gx(H * f * W * f), C, D * fz), dim3(256), 0, uz::S(stream), x, CtotX, y, CtotY, H, W, f, fz);
    return uz::check_launch("nearest3d_fwd_k");
}
extern "C" int uz_nearest3d_bwd(const float* dy, int C, int CtotDy, float* dx, int CtotDx, int D, int H, int W, int f, int fz, int accumulate, void* stream) {
    UZ_REQUIRE(C > 0 && D > 0 && f >= 1 && fz >= 1 && C <= 65535 && D <= 65535, "nearest3d_bwd: bad sizes");
    if (f * f * fz >= 64 && (H * W + 3) / 4 <= 65535)
        hipLaunchKernelGGL(nearest3d_bwd_wave_k, dim3((H * W + 3) / 4, C, D), dim3(256), 0, uz::S(stream), dy, CtotDy, dx, CtotDx, H, W, f, fz, accumulate);
    else
        hipLaunchKernelGGL(nearest3d_bwd_k, dim3(gx(H * W), C, D), dim3(256), 0, uz::S(stream), dy, CtotDy, dx, CtotDx, H, W, f, fz, accumulate);
    return uz::check_launch("nearest3d_bwd_k");
}

// ---- storage-format conversions (bf16 <-> fp32), contiguous
namespace {
__global__ __launch_bounds__(256) void cvt_f32_b16_k(const float* __restrict__ src, unsigned short* __restrict__ dst, size_t n) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) dst[i] = (unsigned short)(uz::pack_bf16x2(src[i], 0.f) & 0xFFFFu);
}
__global__ __launch_bounds__(256) void cvt_b16_f32_k(const unsigned short* __restrict__ src, float* __restrict__ dst, size_t n) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) dst[i] = __builtin_bit_cast(float, (unsigned)src[i] << 16);
}
}  // namespace
extern "C" int uz_cvt_f32_to_b16(const float* src, void* dst, size_t n, void* stream) {
    UZ_REQUIRE(src && dst, "cvt_f32_to_b16: null argument");
    if (n == 0) return 0;
    size_t g = (n + 255) / 256; if (g > 65535) g = 65535;
    hipLaunchKernelGGL(cvt_f32_b16_k, dim3((unsigned)g), dim3(256), 0, uz::S(stream), src, static_cast<unsigned short*>(dst), n);
    return uz::check_launch("cvt_f32_b16_k");
}
extern "C" int uz_cvt_b16_to_f32(const void* src, float* dst, size_t n, void* stream) {
    UZ_REQUIRE(src && dst, "cvt_b16_to_f32: null argument");
    if (n == 0) return 0;
    size_t g = (n + 255) / 256; if (g > 65535) g = 65535;
    hipLaunchKernelGGL(cvt_b16_f32_k, dim3((unsigned)g), dim3(256), 0, uz::S(stream), static_cast<const unsigned short*>(src), dst, n);
    return uz::check_launch("cvt_b16_f32_k");
}
