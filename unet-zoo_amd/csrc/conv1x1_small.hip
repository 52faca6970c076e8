// 1x1 convolutions with very few output channels (<= 8): the latent / logit heads of the models
// (mu_conv, sigma_conv phiseg.py:95-96; s_layer phiseg.py:281-284; Unet.last_layer unet.py:122;
// Fcomb.last_layer; AxisAlignedConvGaussian.conv_layer probabilistic_unet.py:95).
// With 2..8 outputs there is no dense contraction to put on the matrix cores: the op is a streaming
// read of the input planes (HBM-bound), so these are float4 VALU kernels:
//   fwd        y[b,n,p]  = bias[n] + sum_c w[n][c] x[b,c,p]          reads x once
//   bwd_data   dx[b,c,p] (+)= sum_n w[n][c] dy[b,n,p]                writes dx once
//   bwd_weight dw[n][c] = sum_{b,p} dy[b,n,p] x[b,c,p], db[n] = sum dy   reads x once, fp64 ordered partials
#include "uz_common.h"
#include "split_f16.h"

namespace {

constexpr int MAXN = 8;
constexpr int PIX = 1024;        // pixels per workgroup (256 threads x float4)

struct C1P {
    const float* x; const float* w; const float* bias; const float* dy; float* y; float* dx; double* part;
    int Cin, CinTot, Cout, CoutTot, N, HW, nchunk, accumulate, cgroup;
    int xb16;        // the many-channel tensor (x; dx in the data gradient) holds 2-byte bf16 elements (float4 paths only); y / dy stay fp32
};

template <int NO, bool VEC>
__global__ __launch_bounds__(256) void c1_fwd(const C1P p) {
    __shared__ float ws[MAXN * 512];
    const int b = blockIdx.y;
    for (int i = threadIdx.x; i < NO * p.Cin; i += 256) ws[i] = p.w[i];
    __syncthreads();
    const float* xb = p.x + (size_t)b * p.CinTot * p.HW;
    float* yb = p.y + (size_t)b * p.CoutTot * p.HW;
    if (VEC) {
        const int q = (blockIdx.x * 256 + threadIdx.x) * 4;
        if (q >= p.HW) return;
        float4 acc[NO];
#pragma unroll
        for (int n = 0; n < NO; ++n) { const float bv = p.bias ? p.bias[n] : 0.f; acc[n] = make_float4(bv, bv, bv, bv); }
#pragma unroll 8
        for (int c = 0; c < p.Cin; ++c) {
            const uz::f32x4 v = uz::ld_elem4(p.x, ((size_t)b * p.CinTot + c) * p.HW + q, p.xb16);
#pragma unroll
            for (int n = 0; n < NO; ++n) {
                const float wv = ws[n * p.Cin + c];
                acc[n].x = fmaf(wv, v.x, acc[n].x); acc[n].y = fmaf(wv, v.y, acc[n].y);
                acc[n].z = fmaf(wv, v.z, acc[n].z); acc[n].w = fmaf(wv, v.w, acc[n].w);
            }
        }
#pragma unroll
        for (int n = 0; n < NO; ++n) *reinterpret_cast<float4*>(yb + (size_t)n * p.HW + q) = acc[n];
    } else {
        for (int q = blockIdx.x * PIX + threadIdx.x; q < min(p.HW, (int)(blockIdx.x + 1) * PIX); q += 256) {
            float acc[NO];
#pragma unroll
            for (int n = 0; n < NO; ++n) acc[n] = p.bias ? p.bias[n] : 0.f;
            for (int c = 0; c < p.Cin; ++c) {
                const float v = xb[(size_t)c * p.HW + q];
#pragma unroll
                for (int n = 0; n < NO; ++n) acc[n] = fmaf(ws[n * p.Cin + c], v, acc[n]);
            }
#pragma unroll
            for (int n = 0; n < NO; ++n) yb[(size_t)n * p.HW + q] = acc[n];
        }
    }
}

template <int NO, bool VEC>
__global__ __launch_bounds__(256) void c1_bwd_data(const C1P p) {
    __shared__ float ws[MAXN * 512];
    const int b = blockIdx.y;
    for (int i = threadIdx.x; i < NO * p.Cin; i += 256) ws[i] = p.w[i];
    __syncthreads();
    const float* db = p.dy + (size_t)b * p.CoutTot * p.HW;
    float* xb = p.dx + (size_t)b * p.CinTot * p.HW;
    const int c_lo = blockIdx.z * p.cgroup, c_hi = min(p.Cin, c_lo + p.cgroup);   // input-channel group of this workgroup
    if (VEC) {
        const int q = (blockIdx.x * 256 + threadIdx.x) * 4;
        if (q >= p.HW) return;
        float4 g[NO];
#pragma unroll
        for (int n = 0; n < NO; ++n) g[n] = *reinterpret_cast<const float4*>(db + (size_t)n * p.HW + q);
#pragma unroll 4
        for (int c = c_lo; c < c_hi; ++c) {
            const size_t e = ((size_t)b * p.CinTot + c) * p.HW + q;
            uz::f32x4 r = p.accumulate ? uz::ld_elem4(p.dx, e, p.xb16) : uz::f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int n = 0; n < NO; ++n) {
                const float wv = ws[n * p.Cin + c];
                r.x = fmaf(wv, g[n].x, r.x); r.y = fmaf(wv, g[n].y, r.y); r.z = fmaf(wv, g[n].z, r.z); r.w = fmaf(wv, g[n].w, r.w);
            }
            uz::st_elem4(p.dx, e, r, p.xb16);
        }
    } else {
        for (int q = blockIdx.x * PIX + threadIdx.x; q < min(p.HW, (int)(blockIdx.x + 1) * PIX); q += 256) {
            float g[NO];
#pragma unroll
            for (int n = 0; n < NO; ++n) g[n] = db[(size_t)n * p.HW + q];
            for (int c = c_lo; c < c_hi; ++c) {
                float* dst = xb + (size_t)c * p.HW + q;
                float r = p.accumulate ? *dst : 0.f;
#pragma unroll
                for (int n = 0; n < NO; ++n) r = fmaf(ws[n * p.Cin + c], g[n], r);
                *dst = r;
            }
        }
    }
}

// grid (Cin, nchunk): block (c, k) reduces its slice of the N*HW pixels for channel c against all NO outputs
template <int NO>
__global__ __launch_bounds__(256) void c1_bwd_weight_partial(const C1P p) {
    __shared__ double sm[4 * 2 * MAXN];
    const int c = blockIdx.x, k = blockIdx.y;
    const bool do_bias = (c == 0);            // the c == 0 blocks also sum dy for the bias gradient
    const long long total = (long long)p.N * p.HW;
    const long long per = ((total + p.nchunk - 1) / p.nchunk + 1023) / 1024 * 1024;     // multiple of the float4 sweep
    const long long lo = k * per, hi = min(total, lo + per);
    double dacc[2 * NO];
#pragma unroll
    for (int n = 0; n < 2 * NO; ++n) dacc[n] = 0.0;
    // fp32 running sums are flushed into fp64 every 32 steps; (b, q) advance incrementally (no per-step division)
    if (p.HW % 4 == 0 && per % 4 == 0) {
        float4 acc[NO], bacc[NO];
#pragma unroll
        for (int n = 0; n < NO; ++n) { acc[n] = make_float4(0.f, 0.f, 0.f, 0.f); bacc[n] = make_float4(0.f, 0.f, 0.f, 0.f); }
        long long i = lo + 4 * threadIdx.x;
        int b = (int)(i / p.HW), q = (int)(i - (long long)b * p.HW), cnt = 0;
        for (; i < hi; i += 1024) {
            const uz::f32x4 xv = uz::ld_elem4(p.x, ((size_t)b * p.CinTot + c) * p.HW + q, p.xb16);
#pragma unroll
            for (int n = 0; n < NO; ++n) {
                const float4 g = *reinterpret_cast<const float4*>(p.dy + ((size_t)b * p.CoutTot + n) * p.HW + q);
                acc[n].x = fmaf(g.x, xv.x, acc[n].x); acc[n].y = fmaf(g.y, xv.y, acc[n].y);
                acc[n].z = fmaf(g.z, xv.z, acc[n].z); acc[n].w = fmaf(g.w, xv.w, acc[n].w);
                if (do_bias) { bacc[n].x += g.x; bacc[n].y += g.y; bacc[n].z += g.z; bacc[n].w += g.w; }
            }
            q += 1024;
            while (q >= p.HW) { q -= p.HW; ++b; }
            if (++cnt == 32) {
#pragma unroll
                for (int n = 0; n < NO; ++n) {
                    dacc[n] += (double)((acc[n].x + acc[n].y) + (acc[n].z + acc[n].w));
                    dacc[NO + n] += (double)((bacc[n].x + bacc[n].y) + (bacc[n].z + bacc[n].w));
                    acc[n] = make_float4(0.f, 0.f, 0.f, 0.f); bacc[n] = make_float4(0.f, 0.f, 0.f, 0.f);
                }
                cnt = 0;
            }
        }
#pragma unroll
        for (int n = 0; n < NO; ++n) {
            dacc[n] += (double)((acc[n].x + acc[n].y) + (acc[n].z + acc[n].w));
            dacc[NO + n] += (double)((bacc[n].x + bacc[n].y) + (bacc[n].z + bacc[n].w));
        }
    } else {
        float acc[NO];
#pragma unroll
        for (int n = 0; n < NO; ++n) acc[n] = 0.f;
        long long i = lo + threadIdx.x;
        int b = (int)(i / p.HW), q = (int)(i - (long long)b * p.HW), cnt = 0;
        for (; i < hi; i += 256) {
            const float xv = p.x[((size_t)b * p.CinTot + c) * p.HW + q];
#pragma unroll
            for (int n = 0; n < NO; ++n) {
                const float g = p.dy[((size_t)b * p.CoutTot + n) * p.HW + q];
                acc[n] = fmaf(g, xv, acc[n]);
                if (do_bias) dacc[NO + n] += g;
            }
            q += 256;
            while (q >= p.HW) { q -= p.HW; ++b; }
            if (++cnt == 64) {
#pragma unroll
                for (int n = 0; n < NO; ++n) { dacc[n] += acc[n]; acc[n] = 0.f; }
                cnt = 0;
            }
        }
#pragma unroll
        for (int n = 0; n < NO; ++n) dacc[n] += acc[n];
    }
    uz::block_sum_d<2 * NO>(dacc, sm);
    if (threadIdx.x == 0) {
#pragma unroll
        for (int n = 0; n < NO; ++n) p.part[((size_t)k * NO + n) * p.Cin + c] = dacc[n];
        if (do_bias) {
            double* pb = p.part + (size_t)p.nchunk * NO * p.Cin;
#pragma unroll
            for (int n = 0; n < NO; ++n) pb[k * NO + n] = dacc[NO + n];
        }
    }
}
template <int NO>
__global__ __launch_bounds__(256) void c1_bwd_weight_final(const double* __restrict__ part, int nchunk, int Cin, float* __restrict__ dw,
                                                            float* __restrict__ db) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < NO * Cin) {
        double s = 0.0;
        for (int k = 0; k < nchunk; ++k) s += part[(size_t)k * NO * Cin + i];
        dw[i] = (float)s;
    }
    if (db && i < NO) {
        const double* pb = part + (size_t)nchunk * NO * Cin;
        double s = 0.0;
        for (int k = 0; k < nchunk; ++k) s += pb[k * NO + i];
        db[i] = (float)s;
    }
}
inline bool vec4(int HW, const void* a, const void* b) {
    return HW % 4 == 0 && (reinterpret_cast<uintptr_t>(a) & 15) == 0 && (reinterpret_cast<uintptr_t>(b) & 15) == 0;
}

}  // namespace

namespace uz {

bool conv1x1_small_ok(int Cin, int Cout) { return Cout <= MAXN && Cin <= 512; }

#define C1_DISPATCH(KERN, VECFLAG, GRID)                                                                     \
    switch (p.Cout) {                                                                                        \
        case 1: if (VECFLAG) hipLaunchKernelGGL((KERN<1, true>), GRID, dim3(256), 0, st, p); else hipLaunchKernelGGL((KERN<1, false>), GRID, dim3(256), 0, st, p); break; \
        case 2: if (VECFLAG) hipLaunchKernelGGL((KERN<2, true>), GRID, dim3(256), 0, st, p); else hipLaunchKernelGGL((KERN<2, false>), GRID, dim3(256), 0, st, p); break; \
        case 3: if (VECFLAG) hipLaunchKernelGGL((KERN<3, true>), GRID, dim3(256), 0, st, p); else hipLaunchKernelGGL((KERN<3, false>), GRID, dim3(256), 0, st, p); break; \
        case 4: if (VECFLAG) hipLaunchKernelGGL((KERN<4, true>), GRID, dim3(256), 0, st, p); else hipLaunchKernelGGL((KERN<4, false>), GRID, dim3(256), 0, st, p); break; \
        case 6: if (VECFLAG) hipLaunchKernelGGL((KERN<6, true>), GRID, dim3(256), 0, st, p); else hipLaunchKernelGGL((KERN<6, false>), GRID, dim3(256), 0, st, p); break; \
        case 8: if (VECFLAG) hipLaunchKernelGGL((KERN<8, true>), GRID, dim3(256), 0, st, p); else hipLaunchKernelGGL((KERN<8, false>), GRID, dim3(256), 0, st, p); break; \
        default: return -2;                                                                                  \
    }

// returns -2 when this (Cout) is not covered (caller falls back to the MFMA kernel)
int conv1x1_small_fwd(const float* x, int Cin, int CinTot, const float* w, const float* bias, float* y, int Cout, int CoutTot,
                      int N, int H, int W, hipStream_t st, int x_b16) {
    C1P p = {}; p.xb16 = x_b16; p.x = x; p.w = w; p.bias = bias; p.y = y; p.Cin = Cin; p.CinTot = CinTot; p.Cout = Cout; p.CoutTot = CoutTot; p.N = N; p.HW = H * W;
    const bool v = vec4(p.HW, x, y);
    if (x_b16 && !v) return fail("conv1x1 forward: bf16 storage needs H*W %% 4 == 0 and 16-byte aligned views");
    const dim3 grid(ceil_div(p.HW, PIX), N);
    C1_DISPATCH(c1_fwd, v, grid)
    return check_launch("c1_fwd");
}
int conv1x1_small_bwd_data(const float* dy, int Cout, int CoutTot, const float* w, float* dx, int Cin, int CinTot,
                           int N, int H, int W, int accumulate, hipStream_t st, int dx_b16) {
    C1P p = {}; p.xb16 = dx_b16; p.dy = dy; p.w = w; p.dx = dx; p.Cin = Cin; p.CinTot = CinTot; p.Cout = Cout; p.CoutTot = CoutTot; p.N = N; p.HW = H * W;
    p.accumulate = accumulate;
    const bool v = vec4(p.HW, dy, dx);
    if (dx_b16 && !v) return fail("conv1x1 data gradient: bf16 storage needs H*W %% 4 == 0 and 16-byte aligned views");
    // low-resolution planes have few pixel blocks: split the input channels over grid.z until ~1024 workgroups exist
    const int pixblk = ceil_div(p.HW, PIX) * N;
    int groups = ceil_div(1024, pixblk);
    if (groups > ceil_div(Cin, 8)) groups = ceil_div(Cin, 8);
    if (groups < 1) groups = 1;
    p.cgroup = ceil_div(Cin, groups);
    const dim3 grid(ceil_div(p.HW, PIX), N, ceil_div(Cin, p.cgroup));
    C1_DISPATCH(c1_bwd_data, v, grid)
    return check_launch("c1_bwd_data");
}
size_t conv1x1_small_bwd_weight_ws(int Cin, int Cout, int N, int H, int W) {
    const long long total = (long long)N * H * W;
    int nchunk = (int)((total + 16383) / 16384);
    if (nchunk > 64) nchunk = 64;
    return ((size_t)nchunk * Cout * Cin + (size_t)nchunk * Cout) * sizeof(double);
}
int conv1x1_small_bwd_weight(const float* x, int Cin, int CinTot, const float* dy, int Cout, int CoutTot, float* dw, float* db,
                             int N, int H, int W, void* ws, hipStream_t st, int x_b16) {
    if (x_b16 && ((H * W) % 4 != 0 || (reinterpret_cast<uintptr_t>(x) & 15) != 0)) return fail("conv1x1 weight gradient: bf16 storage needs H*W %% 4 == 0 and a 16-byte aligned view");
    C1P p = {}; p.xb16 = x_b16; p.x = x; p.dy = dy; p.Cin = Cin; p.CinTot = CinTot; p.Cout = Cout; p.CoutTot = CoutTot; p.N = N; p.HW = H * W;
    const long long total = (long long)N * H * W;
    p.nchunk = (int)((total + 16383) / 16384);
    if (p.nchunk > 64) p.nchunk = 64;
    p.part = static_cast<double*>(ws);
    const dim3 grid(Cin, p.nchunk);
    const int fg = ceil_div(Cout * Cin, 256);
#define C1_W(NO_)                                                                                              \
    case NO_:                                                                                                  \
        hipLaunchKernelGGL(c1_bwd_weight_partial<NO_>, grid, dim3(256), 0, st, p);                            \
        hipLaunchKernelGGL(c1_bwd_weight_final<NO_>, dim3(fg), dim3(256), 0, st, p.part, p.nchunk, Cin, dw, db); \
        break;
    switch (Cout) {
        C1_W(1) C1_W(2) C1_W(3) C1_W(4) C1_W(6) C1_W(8)
        default: return -2;
    }
    return check_launch("c1_bwd_weight");
}

}  // namespace uz

// ---- bf16 STORAGE of the many-channel operand of a 1x1(x1) head (include/uz_api.h, "bf16 storage"): x (forward, weight gradient) / dx
// (data gradient) hold 2-byte bf16 elements, the 1 .. 8-channel side (y, dy) stays fp32.  Cout in {1, 2, 3, 4, 6, 8}, Cin <= 512.
extern "C" int uz_conv1x1_fwd_b16(const void* x, int Cin, int CinTot, const float* w, const float* bias, float* y, int Cout, int CoutTot,
                                  int N, int H, int W, int x_b16, void* stream) {
    UZ_REQUIRE(uz::conv1x1_small_ok(Cin, Cout) && N > 0 && H > 0 && W > 0, "conv1x1_fwd_b16: shape not covered by the streaming 1x1 kernels");
    const int rc = uz::conv1x1_small_fwd(static_cast<const float*>(x), Cin, CinTot, w, bias, y, Cout, CoutTot, N, H, W, uz::S(stream), x_b16 != 0);
    return rc == -2 ? uz::fail("conv1x1_fwd_b16: %d outputs not covered", Cout) : rc;
}
extern "C" int uz_conv1x1_bwd_data_b16(const float* dy, int Cout, int CoutTot, const float* w, void* dx, int Cin, int CinTot,
                                       int N, int H, int W, int accumulate, int dx_b16, void* stream) {
    UZ_REQUIRE(uz::conv1x1_small_ok(Cin, Cout) && N > 0 && H > 0 && W > 0, "conv1x1_bwd_data_b16: shape not covered by the streaming 1x1 kernels");
    const int rc = uz::conv1x1_small_bwd_data(dy, Cout, CoutTot, w, static_cast<float*>(dx), Cin, CinTot, N, H, W, accumulate, uz::S(stream), dx_b16 != 0);
    return rc == -2 ? uz::fail("conv1x1_bwd_data_b16: %d outputs not covered", Cout) : rc;
}
extern "C" int uz_conv1x1_bwd_weight_b16(const void* x, int Cin, int CinTot, const float* dy, int Cout, int CoutTot, float* dw, float* db,
                                         int N, int H, int W, void* workspace, size_t workspace_bytes, int x_b16, void* stream) {
    UZ_REQUIRE(uz::conv1x1_small_ok(Cin, Cout) && N > 0 && H > 0 && W > 0, "conv1x1_bwd_weight_b16: shape not covered by the streaming 1x1 kernels");
    UZ_REQUIRE(workspace && workspace_bytes >= uz::conv1x1_small_bwd_weight_ws(Cin, Cout, N, H, W), "conv1x1_bwd_weight_b16: workspace too small");
    const int rc = uz::conv1x1_small_bwd_weight(static_cast<const float*>(x), Cin, CinTot, dy, Cout, CoutTot, dw, db, N, H, W, workspace, uz::S(stream), x_b16 != 0);
    return rc == -2 ? uz::fail("conv1x1_bwd_weight_b16: %d outputs not covered", Cout) : rc;
}
