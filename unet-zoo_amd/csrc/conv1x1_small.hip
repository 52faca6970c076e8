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

// ---------------------------------------------------------------- the two heads of a SampleZBlock as ONE op per direction
// mu = mu_conv(h), pre = sigma_conv(h), sigma = softplus(pre), z = mu + sigma eps (phiseg.py:95-105): three launches and two reads
// of h in the forward, six launches, three reads of h and a read-modify-write of dh in the backward - on the serial chain of the
// latent hierarchy (level k + 1 waits for z_k).  Here: one forward launch (h read once; mu, pre, sigma, z written from registers),
// one data-gradient launch (dh written once from both heads' dy) and one weight-gradient launch pair (h read once).  Every output
// keeps the arithmetic ORDER of the separate kernels above (per-output fmaf chains over the input channels; the data gradient adds
// head A's rows before head B's, as the two accumulating launches did; the weight gradient's fp32 / fp64 partial sums are per output
// row), so results are bit-identical to the three-op path - tests/test_ops_gpu.py compares them.
namespace {

struct HP {
    const float* x; int Cin, CinTot, N, HW;
    const float* wA; const float* bA; const float* wB; const float* bB;          // head A's rows come first, then head B's
    const float* eps; float* mu; float* pre; float* sigma; float* z; int act;    // forward: A = mu head, B = sigma head
    const float* dyA; const float* dyB; float* dx; int accumulate, cgroup;       // data gradient: dy of either head, [N][L][HW]
    double* part; int nchunk;                                                    // weight gradient
};

__device__ __forceinline__ float head_softplus(float x) { return x > 20.f ? x : log1pf(expf(x)); }   // = pointwise.hip softplus_f
__device__ __forceinline__ float head_sigma(float pre, int act) { return act ? expf(pre) : head_softplus(pre); }

template <int LL>
__device__ __forceinline__ void heads_weights(const HP& p, float* ws) {
    const int half = LL * p.Cin;
    for (int i = threadIdx.x; i < 2 * half; i += 256) ws[i] = i < half ? p.wA[i] : p.wB[i - half];
    __syncthreads();
}

template <int LL, bool VEC>
__global__ __launch_bounds__(256) void heads_fwd(const HP p) {
    constexpr int NO = 2 * LL;
    __shared__ float ws[MAXN * 512];
    const int b = blockIdx.y;
    heads_weights<LL>(p, ws);
    float bias[NO];
#pragma unroll
    for (int n = 0; n < LL; ++n) { bias[n] = p.bA ? p.bA[n] : 0.f; bias[LL + n] = p.bB ? p.bB[n] : 0.f; }
    if (VEC) {
        const int q = (blockIdx.x * 256 + threadIdx.x) * 4;
        if (q >= p.HW) return;
        float4 acc[NO];
#pragma unroll
        for (int n = 0; n < NO; ++n) acc[n] = make_float4(bias[n], bias[n], bias[n], bias[n]);
#pragma unroll 8
        for (int c = 0; c < p.Cin; ++c) {
            const float4 v = *reinterpret_cast<const float4*>(p.x + ((size_t)b * p.CinTot + c) * p.HW + q);
#pragma unroll
            for (int n = 0; n < NO; ++n) {
                const float wv = ws[n * p.Cin + c];
                acc[n].x = fmaf(wv, v.x, acc[n].x); acc[n].y = fmaf(wv, v.y, acc[n].y);
                acc[n].z = fmaf(wv, v.z, acc[n].z); acc[n].w = fmaf(wv, v.w, acc[n].w);
            }
        }
#pragma unroll
        for (int n = 0; n < LL; ++n) {
            const size_t e = ((size_t)b * LL + n) * p.HW + q;
            const float4 m = acc[n], pr = acc[LL + n];
            const float4 s = make_float4(head_sigma(pr.x, p.act), head_sigma(pr.y, p.act), head_sigma(pr.z, p.act), head_sigma(pr.w, p.act));
            *reinterpret_cast<float4*>(p.mu + e) = m;
            *reinterpret_cast<float4*>(p.pre + e) = pr;
            *reinterpret_cast<float4*>(p.sigma + e) = s;
            if (p.z) {
                const float4 ev = *reinterpret_cast<const float4*>(p.eps + e);
                *reinterpret_cast<float4*>(p.z + e) = make_float4(m.x + s.x * ev.x, m.y + s.y * ev.y, m.z + s.z * ev.z, m.w + s.w * ev.w);
            }
        }
    } else {
        for (int q = blockIdx.x * PIX + threadIdx.x; q < min(p.HW, (int)(blockIdx.x + 1) * PIX); q += 256) {
            float acc[NO];
#pragma unroll
            for (int n = 0; n < NO; ++n) acc[n] = bias[n];
            for (int c = 0; c < p.Cin; ++c) {
                const float v = p.x[((size_t)b * p.CinTot + c) * p.HW + q];
#pragma unroll
                for (int n = 0; n < NO; ++n) acc[n] = fmaf(ws[n * p.Cin + c], v, acc[n]);
            }
#pragma unroll
            for (int n = 0; n < LL; ++n) {
                const size_t e = ((size_t)b * LL + n) * p.HW + q;
                const float s = head_sigma(acc[LL + n], p.act);
                p.mu[e] = acc[n]; p.pre[e] = acc[LL + n]; p.sigma[e] = s;
                if (p.z) p.z[e] = acc[n] + s * p.eps[e];
            }
        }
    }
}

// Channel-parallel form of the forward for the latent hierarchy's own planes (round 5).  heads_fwd above gives every thread four pixels
// and walks the input channels one after the other: on a 2 x 2 ... 32 x 32 plane that is ONE workgroup per image whose threads wait for
// 192 dependent-by-queue loads (28 - 60 us per launch, ten launches on the forward's critical chain - level k + 1 waits for z_k).  Here a
// workgroup owns QPB (<= 64) pixel quads of one image and CG = 256 / QPB channel groups: thread (g, q) accumulates channels g, g + CG, ...
// for quad q, the CG partial sums are added in a fixed binary tree through LDS (deterministic; NOT the sequential order of heads_fwd:
// results agree to rounding, tests/test_ops_gpu.py), and the threads of group 0 finish bias, softplus and sampling.  UZ_HEADS_PAR=0
// keeps heads_fwd everywhere.
template <int LL>
__global__ __launch_bounds__(256) void heads_fwd_par(const HP p, const int qpb_log2) {
    constexpr int NO = 2 * LL;
    __shared__ float ws[MAXN * 512];
    __shared__ float4 red[256 * NO];
    const int b = blockIdx.y;
    heads_weights<LL>(p, ws);
    const int QPB = 1 << qpb_log2, CG = 256 >> qpb_log2;
    const int g = threadIdx.x >> qpb_log2, q = (blockIdx.x << qpb_log2) + (threadIdx.x & (QPB - 1));       // channel group, pixel quad of the image
    float4 acc[NO];
#pragma unroll
    for (int n = 0; n < NO; ++n) acc[n] = make_float4(0.f, 0.f, 0.f, 0.f);
    const float* src = p.x + (size_t)b * p.CinTot * p.HW + 4 * q;
#pragma unroll 4
    for (int c = g; c < p.Cin; c += CG) {
        const float4 v = *reinterpret_cast<const float4*>(src + (size_t)c * p.HW);
#pragma unroll
        for (int n = 0; n < NO; ++n) {
            const float wv = ws[n * p.Cin + c];
            acc[n].x = fmaf(wv, v.x, acc[n].x); acc[n].y = fmaf(wv, v.y, acc[n].y);
            acc[n].z = fmaf(wv, v.z, acc[n].z); acc[n].w = fmaf(wv, v.w, acc[n].w);
        }
    }
    for (int stride = CG >> 1; stride >= 1; stride >>= 1) {
        if (g >= stride && g < 2 * stride) {
#pragma unroll
            for (int n = 0; n < NO; ++n) red[threadIdx.x * NO + n] = acc[n];
        }
        __syncthreads();
        if (g < stride) {
            const int o = threadIdx.x + (stride << qpb_log2);
#pragma unroll
            for (int n = 0; n < NO; ++n) {
                const float4 r = red[o * NO + n];
                acc[n].x += r.x; acc[n].y += r.y; acc[n].z += r.z; acc[n].w += r.w;
            }
        }
        __syncthreads();
    }
    if (g != 0) return;
#pragma unroll
    for (int n = 0; n < LL; ++n) {
        const float bm = p.bA ? p.bA[n] : 0.f, bs = p.bB ? p.bB[n] : 0.f;
        const size_t e = ((size_t)b * LL + n) * p.HW + 4 * q;
        const float4 m = make_float4(acc[n].x + bm, acc[n].y + bm, acc[n].z + bm, acc[n].w + bm);
        const float4 pr = make_float4(acc[LL + n].x + bs, acc[LL + n].y + bs, acc[LL + n].z + bs, acc[LL + n].w + bs);
        const float4 sg = make_float4(head_sigma(pr.x, p.act), head_sigma(pr.y, p.act), head_sigma(pr.z, p.act), head_sigma(pr.w, p.act));
        *reinterpret_cast<float4*>(p.mu + e) = m;
        *reinterpret_cast<float4*>(p.pre + e) = pr;
        *reinterpret_cast<float4*>(p.sigma + e) = sg;
        if (p.z) {
            const float4 ev = *reinterpret_cast<const float4*>(p.eps + e);
            *reinterpret_cast<float4*>(p.z + e) = make_float4(m.x + sg.x * ev.x, m.y + sg.y * ev.y, m.z + sg.z * ev.z, m.w + sg.w * ev.w);
        }
    }
}

template <int LL, bool VEC>
__global__ __launch_bounds__(256) void heads_bwd_data(const HP p) {
    constexpr int NO = 2 * LL;
    __shared__ float ws[MAXN * 512];
    const int b = blockIdx.y;
    heads_weights<LL>(p, ws);
    const int c_lo = blockIdx.z * p.cgroup, c_hi = min(p.Cin, c_lo + p.cgroup);
    if (VEC) {
        const int q = (blockIdx.x * 256 + threadIdx.x) * 4;
        if (q >= p.HW) return;
        float4 g[NO];
#pragma unroll
        for (int n = 0; n < LL; ++n) {
            g[n] = *reinterpret_cast<const float4*>(p.dyA + ((size_t)b * LL + n) * p.HW + q);
            g[LL + n] = *reinterpret_cast<const float4*>(p.dyB + ((size_t)b * LL + n) * p.HW + q);
        }
#pragma unroll 4
        for (int c = c_lo; c < c_hi; ++c) {
            float4* dst = reinterpret_cast<float4*>(p.dx + ((size_t)b * p.CinTot + c) * p.HW + q);
            float4 r = p.accumulate ? *dst : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
            for (int n = 0; n < NO; ++n) {
                const float wv = ws[n * p.Cin + c];
                r.x = fmaf(wv, g[n].x, r.x); r.y = fmaf(wv, g[n].y, r.y); r.z = fmaf(wv, g[n].z, r.z); r.w = fmaf(wv, g[n].w, r.w);
            }
            *dst = r;
        }
    } else {
        for (int q = blockIdx.x * PIX + threadIdx.x; q < min(p.HW, (int)(blockIdx.x + 1) * PIX); q += 256) {
            float g[NO];
#pragma unroll
            for (int n = 0; n < LL; ++n) { g[n] = p.dyA[((size_t)b * LL + n) * p.HW + q]; g[LL + n] = p.dyB[((size_t)b * LL + n) * p.HW + q]; }
            for (int c = c_lo; c < c_hi; ++c) {
                float* dst = p.dx + ((size_t)b * p.CinTot + c) * p.HW + q;
                float r = p.accumulate ? *dst : 0.f;
#pragma unroll
                for (int n = 0; n < NO; ++n) r = fmaf(ws[n * p.Cin + c], g[n], r);
                *dst = r;
            }
        }
    }
}

// grid (Cin, nchunk) as c1_bwd_weight_partial: block (c, k) reduces its slice of the N * HW pixels for channel c against the 2 L rows
template <int LL>
__global__ __launch_bounds__(256) void heads_bwd_weight_partial(const HP p) {
    constexpr int NO = 2 * LL;
    __shared__ double sm[4 * 2 * MAXN];
    const int c = blockIdx.x, k = blockIdx.y;
    const bool do_bias = (c == 0);
    const long long total = (long long)p.N * p.HW;
    const long long per = ((total + p.nchunk - 1) / p.nchunk + 1023) / 1024 * 1024;
    const long long lo = k * per, hi = min(total, lo + per);
    double dacc[2 * NO];
#pragma unroll
    for (int n = 0; n < 2 * NO; ++n) dacc[n] = 0.0;
    if (p.HW % 4 == 0 && per % 4 == 0) {
        float4 acc[NO], bacc[NO];
#pragma unroll
        for (int n = 0; n < NO; ++n) { acc[n] = make_float4(0.f, 0.f, 0.f, 0.f); bacc[n] = make_float4(0.f, 0.f, 0.f, 0.f); }
        long long i = lo + 4 * threadIdx.x;
        int b = (int)(i / p.HW), q = (int)(i - (long long)b * p.HW), cnt = 0;
        for (; i < hi; i += 1024) {
            const float4 xv = *reinterpret_cast<const float4*>(p.x + ((size_t)b * p.CinTot + c) * p.HW + q);
#pragma unroll
            for (int n = 0; n < NO; ++n) {
                const float* dy = n < LL ? p.dyA : p.dyB;
                const float4 g = *reinterpret_cast<const float4*>(dy + ((size_t)b * LL + (n < LL ? n : n - LL)) * p.HW + q);
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
                const float* dy = n < LL ? p.dyA : p.dyB;
                const float g = dy[((size_t)b * LL + (n < LL ? n : n - LL)) * p.HW + q];
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
template <int LL>
__global__ __launch_bounds__(256) void heads_bwd_weight_final(const double* __restrict__ part, int nchunk, int Cin, float* __restrict__ dwA, float* __restrict__ dbA,
                                                               float* __restrict__ dwB, float* __restrict__ dbB) {
    constexpr int NO = 2 * LL;
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < NO * Cin) {
        double s = 0.0;
        for (int k = 0; k < nchunk; ++k) s += part[(size_t)k * NO * Cin + i];
        if (i < LL * Cin) dwA[i] = (float)s; else dwB[i - LL * Cin] = (float)s;
    }
    if (i < NO) {
        const double* pb = part + (size_t)nchunk * NO * Cin;
        double s = 0.0;
        for (int k = 0; k < nchunk; ++k) s += pb[k * NO + i];
        if (i < LL) { if (dbA) dbA[i] = (float)s; } else if (dbB) dbB[i - LL] = (float)s;
    }
}

inline bool al16(const void* a) { return (reinterpret_cast<uintptr_t>(a) & 15) == 0; }
inline int heads_nchunk(int N, int H, int W) {
    const long long total = (long long)N * H * W;
    const int nchunk = (int)((total + 16383) / 16384);
    return nchunk > 64 ? 64 : nchunk;
}

#define HEADS_DISPATCH(KERN, VECFLAG, GRID)                                                                                     \
    switch (L) {                                                                                                                 \
        case 1: if (VECFLAG) hipLaunchKernelGGL((KERN<1, true>), GRID, dim3(256), 0, st, p); else hipLaunchKernelGGL((KERN<1, false>), GRID, dim3(256), 0, st, p); break; \
        case 2: if (VECFLAG) hipLaunchKernelGGL((KERN<2, true>), GRID, dim3(256), 0, st, p); else hipLaunchKernelGGL((KERN<2, false>), GRID, dim3(256), 0, st, p); break; \
        case 3: if (VECFLAG) hipLaunchKernelGGL((KERN<3, true>), GRID, dim3(256), 0, st, p); else hipLaunchKernelGGL((KERN<3, false>), GRID, dim3(256), 0, st, p); break; \
        case 4: if (VECFLAG) hipLaunchKernelGGL((KERN<4, true>), GRID, dim3(256), 0, st, p); else hipLaunchKernelGGL((KERN<4, false>), GRID, dim3(256), 0, st, p); break; \
        default: return uz::fail("latent heads: %d latent channels per head not covered (1 .. 4)", L);                          \
    }

}  // namespace

extern "C" int uz_latent_heads_ok(int Cin, int L) { return L >= 1 && L <= 4 && Cin >= 1 && Cin <= 512 ? 1 : 0; }

extern "C" int uz_latent_heads_fwd(const float* h, int Cin, int CinTot, const float* w_mu, const float* b_mu, const float* w_sigma, const float* b_sigma,
                                   const float* eps, float* mu, float* pre_sigma, float* sigma, float* z, int L, int N, int H, int W, int act, void* stream) {
    UZ_REQUIRE(uz_latent_heads_ok(Cin, L) && N > 0 && H > 0 && W > 0 && CinTot >= Cin, "latent_heads_fwd: shape not covered (1 <= L <= 4, Cin <= 512)");
    UZ_REQUIRE(h && w_mu && w_sigma && mu && pre_sigma && sigma, "latent_heads_fwd: null operand");
    UZ_REQUIRE(!z || eps, "latent_heads_fwd: z needs eps");
    hipStream_t st = uz::S(stream);
    HP p = {}; p.x = h; p.Cin = Cin; p.CinTot = CinTot; p.N = N; p.HW = H * W; p.wA = w_mu; p.bA = b_mu; p.wB = w_sigma; p.bB = b_sigma;
    p.eps = eps; p.mu = mu; p.pre = pre_sigma; p.sigma = sigma; p.z = z; p.act = act;
    const bool v = p.HW % 4 == 0 && al16(h) && al16(mu) && al16(pre_sigma) && al16(sigma) && (!z || (al16(z) && al16(eps)));
    // the latent hierarchy's own planes (a power-of-two number of pixel quads, at most 1024 pixels per image at moderate batch): channel-parallel form
    const int quads = p.HW / 4;
    const char* par_env = getenv("UZ_HEADS_PAR");
    if (v && quads >= 1 && (quads & (quads - 1)) == 0 && (long long)N * quads <= 16384 && !(par_env && atoi(par_env) == 0)) {
        int lg = 0;
        while ((1 << lg) < quads && lg < 6) ++lg;               // QPB = min(quads, 64)
        const dim3 gp(quads >> lg, N);
        switch (L) {
            case 1: hipLaunchKernelGGL(heads_fwd_par<1>, gp, dim3(256), 0, st, p, lg); break;
            case 2: hipLaunchKernelGGL(heads_fwd_par<2>, gp, dim3(256), 0, st, p, lg); break;
            case 3: hipLaunchKernelGGL(heads_fwd_par<3>, gp, dim3(256), 0, st, p, lg); break;
            case 4: hipLaunchKernelGGL(heads_fwd_par<4>, gp, dim3(256), 0, st, p, lg); break;
            default: return uz::fail("latent heads: %d latent channels per head not covered (1 .. 4)", L);
        }
        return uz::check_launch("heads_fwd_par");
    }
    const dim3 grid(uz::ceil_div(p.HW, PIX), N);
    HEADS_DISPATCH(heads_fwd, v, grid)
    return uz::check_launch("heads_fwd");
}

extern "C" int uz_latent_heads_bwd_data(const float* dy_a, const float* dy_b, int L, const float* w_a, const float* w_b, float* dh, int Cin, int CinTot,
                                        int N, int H, int W, int accumulate, void* stream) {
    UZ_REQUIRE(uz_latent_heads_ok(Cin, L) && N > 0 && H > 0 && W > 0 && CinTot >= Cin, "latent_heads_bwd_data: shape not covered (1 <= L <= 4, Cin <= 512)");
    UZ_REQUIRE(dy_a && dy_b && w_a && w_b && dh, "latent_heads_bwd_data: null operand");
    hipStream_t st = uz::S(stream);
    HP p = {}; p.dyA = dy_a; p.dyB = dy_b; p.wA = w_a; p.wB = w_b; p.dx = dh; p.Cin = Cin; p.CinTot = CinTot; p.N = N; p.HW = H * W; p.accumulate = accumulate;
    const bool v = p.HW % 4 == 0 && al16(dy_a) && al16(dy_b) && al16(dh);
    const int pixblk = uz::ceil_div(p.HW, PIX) * N;            // as conv1x1_small_bwd_data: channel groups over grid.z on the small planes
    int groups = uz::ceil_div(1024, pixblk);
    if (groups > uz::ceil_div(Cin, 8)) groups = uz::ceil_div(Cin, 8);
    if (groups < 1) groups = 1;
    p.cgroup = uz::ceil_div(Cin, groups);
    const dim3 grid(uz::ceil_div(p.HW, PIX), N, uz::ceil_div(Cin, p.cgroup));
    HEADS_DISPATCH(heads_bwd_data, v, grid)
    return uz::check_launch("heads_bwd_data");
}

extern "C" size_t uz_latent_heads_bwd_weight_workspace(int Cin, int L, int N, int H, int W) {
    const int nchunk = heads_nchunk(N, H, W);
    return ((size_t)nchunk * 2 * L * Cin + (size_t)nchunk * 2 * L) * sizeof(double);
}

extern "C" int uz_latent_heads_bwd_weight(const float* h, int Cin, int CinTot, const float* dy_a, const float* dy_b, int L, float* dw_a, float* db_a,
                                          float* dw_b, float* db_b, int N, int H, int W, void* workspace, size_t workspace_bytes, void* stream) {
    UZ_REQUIRE(uz_latent_heads_ok(Cin, L) && N > 0 && H > 0 && W > 0 && CinTot >= Cin, "latent_heads_bwd_weight: shape not covered (1 <= L <= 4, Cin <= 512)");
    UZ_REQUIRE(h && dy_a && dy_b && dw_a && dw_b, "latent_heads_bwd_weight: null operand");
    UZ_REQUIRE(workspace && workspace_bytes >= uz_latent_heads_bwd_weight_workspace(Cin, L, N, H, W), "latent_heads_bwd_weight: workspace too small");
    hipStream_t st = uz::S(stream);
    HP p = {}; p.x = h; p.dyA = dy_a; p.dyB = dy_b; p.Cin = Cin; p.CinTot = CinTot; p.N = N; p.HW = H * W;
    p.nchunk = heads_nchunk(N, H, W);
    p.part = static_cast<double*>(workspace);
    const dim3 grid(Cin, p.nchunk);
    const int fg = uz::ceil_div(2 * L * Cin, 256);
#define HEADS_W(LL_)                                                                                                           \
    case LL_:                                                                                                                  \
        hipLaunchKernelGGL(heads_bwd_weight_partial<LL_>, grid, dim3(256), 0, st, p);                                         \
        hipLaunchKernelGGL(heads_bwd_weight_final<LL_>, dim3(fg), dim3(256), 0, st, p.part, p.nchunk, Cin, dw_a, db_a, dw_b, db_b); \
        break;
    switch (L) {
        HEADS_W(1) HEADS_W(2) HEADS_W(3) HEADS_W(4)
        default: return uz::fail("latent_heads_bwd_weight: %d latent channels per head not covered", L);
    }
    return uz::check_launch("heads_bwd_weight");
}
