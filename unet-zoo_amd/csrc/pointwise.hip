// Latent heads, losses, optimiser and vector helpers (all HBM-bound or tiny):
//   posterior input  = cat(patch, onehot(mask) - 0.5)      utils.py:289-311, phiseg.py:178-183
//   latent sampling  = softplus + reparameterisation        phiseg.py:100-105
//   KL(q||p) with the reference's sigma1*sigma0 quirk       phiseg.py:436-453, probabilistic_unet.py:291-308
//   residual multinoulli cross entropy                      phiseg.py:481-513 (unet.py:159-165 for L=1)
//   accumulate_output + softmax + argmax                    phiseg.py:428-434, train_model.py:186,195
//   torch.optim.Adam step (L2 weight decay)                 train_model.py:49,122
//   sum of 2-norms regulariser                              utils.py:93-101
#include "uz_common.h"
#include "split_f16.h"

namespace {

constexpr int MAXL = 8;

// ---------------------------------------------------------------- additive coupling of the reversible blocks
// y = (accumulate ? y : 0) + a + alpha * b on channel-slice views (b nullable).  V = 4: float4 sweep of 16-byte aligned planes.
template <int V>
__global__ __launch_bounds__(256) void add_views_k(const float* __restrict__ a, int CtotA, const float* __restrict__ b, int CtotB,
                                                    float* __restrict__ y, int CtotY, int C, int HW, float alpha, int accumulate,
                                                    const float* a_amax, const float* b_amax, float* y_amax) {
    if (y_amax && a_amax && (b_amax || !b) && blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0)
        uz::amax_publish_one(uz::amax_read(a_amax) + (b ? fabsf(alpha) * uz::amax_read(b_amax) : 0.f), y_amax, 0u);   // |y| <= |a| + |alpha| |b|
    const int bi = blockIdx.y;
    const size_t per = (size_t)C * HW;
    const float* as = a + (size_t)bi * CtotA * HW;
    const float* bs = b ? b + (size_t)bi * CtotB * HW : nullptr;
    float* ys = y + (size_t)bi * CtotY * HW;
    for (size_t i = ((size_t)blockIdx.x * 256 + threadIdx.x) * V; i < per; i += (size_t)gridDim.x * 256 * V) {
        if (V == 4) {
            float4 v = *reinterpret_cast<const float4*>(as + i);
            if (bs) { const float4 w = *reinterpret_cast<const float4*>(bs + i); v.x = fmaf(alpha, w.x, v.x); v.y = fmaf(alpha, w.y, v.y); v.z = fmaf(alpha, w.z, v.z); v.w = fmaf(alpha, w.w, v.w); }
            if (accumulate) { const float4 o = *reinterpret_cast<const float4*>(ys + i); v.x += o.x; v.y += o.y; v.z += o.z; v.w += o.w; }
            *reinterpret_cast<float4*>(ys + i) = v;
        } else {
            float v = as[i];
            if (bs) v = fmaf(alpha, bs[i], v);
            if (accumulate) v += ys[i];
            ys[i] = v;
        }
    }
}

// ---------------------------------------------------------------- posterior input
__global__ __launch_bounds__(256) void posterior_input_k(const float* __restrict__ patch, int in_ch, const float* __restrict__ mask,
                                                          int nlabels, float* __restrict__ out, int HW) {
    const int b = blockIdx.z, c = blockIdx.y, ctot = in_ch + nlabels;
    float* d = out + ((size_t)b * ctot + c) * HW;
    if (c < in_ch) {
        const float* s = patch + ((size_t)b * in_ch + c) * HW;
        for (int i = blockIdx.x * 256 + threadIdx.x; i < HW; i += gridDim.x * 256) d[i] = s[i];
    } else {
        const float lab = (float)(c - in_ch);
        const float* m = mask + (size_t)b * HW;
        for (int i = blockIdx.x * 256 + threadIdx.x; i < HW; i += gridDim.x * 256) d[i] = (m[i] == lab ? 1.f : 0.f) - 0.5f;
    }
}

// ---------------------------------------------------------------- latent sampling
__device__ __forceinline__ float softplus_f(float x) { return x > 20.f ? x : log1pf(expf(x)); }   // F.softplus(beta=1, threshold=20)

__global__ __launch_bounds__(256) void latent_fwd_k(const float* __restrict__ mu, const float* __restrict__ pre, const float* __restrict__ eps,
                                                     float* __restrict__ sigma, float* __restrict__ z, size_t n, int act) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        const float s = act ? expf(pre[i]) : softplus_f(pre[i]);
        sigma[i] = s;
        if (z) z[i] = mu[i] + s * eps[i];
    }
}
__global__ __launch_bounds__(256) void latent_bwd_k(const float* __restrict__ dmu, const float* __restrict__ dsigma, const float* __restrict__ dz,
                                                     const float* __restrict__ eps, const float* __restrict__ sigma,
                                                     float* __restrict__ dmu_pre, float* __restrict__ dpre, size_t n, int act) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        const float gz = dz ? dz[i] : 0.f;
        const float gm = (dmu ? dmu[i] : 0.f) + gz;
        const float gs = (dsigma ? dsigma[i] : 0.f) + gz * eps[i];
        dmu_pre[i] = gm;
        // d softplus(x)/dx = sigmoid(x) = 1 - exp(-softplus(x)); identity above the threshold
        const float s = sigma[i];
        dpre[i] = gs * (act ? s : (s > 20.f ? 1.f : (1.f - expf(-s))));
    }
}

// ---------------------------------------------------------------- KL
// one workgroup of 1024 threads (the result is a single ordered fp64 sum; the largest level has 1M elements)
__global__ __launch_bounds__(1024) void kl_fwd_k(const float* __restrict__ mu0, const float* __restrict__ s0, const float* __restrict__ mu1,
                                                  const float* __restrict__ s1, int total, int N, float weight, float* __restrict__ out) {
    __shared__ double sm[16];
    double v = 0.0;
    for (int i = threadIdx.x; i < total; i += 1024) {
        const float a0 = s0[i], a1 = s1[i];
        const float s0fs = a0 * a0, s1fs = a1 * a0;
        const float d = mu1[i] - mu0[i];
        const float t = (s0fs + d * d) / (s1fs + 1e-10f) + logf(s1fs + 1e-10f) - logf(s0fs + 1e-10f) - 1.f;
        v += t;
    }
    v = uz::wave_sum_d(v);
    if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = v;
    __syncthreads();
    if (threadIdx.x == 0) {
        double s = 0.0;
        for (int w = 0; w < 16; ++w) s += sm[w];
        out[0] = (float)((double)weight * 0.5 * s / N);
    }
}
// Large tensors (a volume's full-resolution latent level: 2 M elements in ONE sample - 1.3 ms in the single workgroup above): the
// same ordered fp64 sum in two stages - G workgroups over fixed contiguous ranges, then one thread over the G partials.
__global__ __launch_bounds__(1024) void kl_fwd_part_k(const float* __restrict__ mu0, const float* __restrict__ s0, const float* __restrict__ mu1,
                                                       const float* __restrict__ s1, int total, int chunk, double* __restrict__ part) {
    __shared__ double sm[16];
    const int lo = blockIdx.x * chunk, hi = min(total, lo + chunk);
    double v = 0.0;
    for (int i = lo + threadIdx.x; i < hi; i += 1024) {
        const float a0 = s0[i], a1 = s1[i];
        const float s0fs = a0 * a0, s1fs = a1 * a0;
        const float d = mu1[i] - mu0[i];
        v += (s0fs + d * d) / (s1fs + 1e-10f) + logf(s1fs + 1e-10f) - logf(s0fs + 1e-10f) - 1.f;
    }
    v = uz::wave_sum_d(v);
    if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = v;
    __syncthreads();
    if (threadIdx.x == 0) {
        double s = 0.0;
        for (int w = 0; w < 16; ++w) s += sm[w];
        part[blockIdx.x] = s;
    }
}
__global__ __launch_bounds__(64) void kl_fwd_final_k(const double* __restrict__ part, int G, int N, float weight, float* __restrict__ out) {
    if (threadIdx.x == 0) {
        double s = 0.0;
        for (int g = 0; g < G; ++g) s += part[g];
        out[0] = (float)((double)weight * 0.5 * s / N);
    }
}
__global__ __launch_bounds__(256) void kl_bwd_k(const float* __restrict__ mu0, const float* __restrict__ s0, const float* __restrict__ mu1,
                                                 const float* __restrict__ s1, int total, float k, const float* __restrict__ scale,
                                                 float* __restrict__ dmu0, float* __restrict__ ds0, float* __restrict__ dmu1, float* __restrict__ ds1) {
    const float kk = k * (scale ? scale[0] : 1.f);
    for (int i = blockIdx.x * 256 + threadIdx.x; i < total; i += gridDim.x * 256) {
        const float a0 = s0[i], a1 = s1[i];
        const float A = a0 * a0 + (mu1[i] - mu0[i]) * (mu1[i] - mu0[i]);
        const float B = a1 * a0 + 1e-10f, C0 = a0 * a0 + 1e-10f;
        const float d = mu1[i] - mu0[i];
        const float g = 2.f * d / B;
        if (dmu0) dmu0[i] = -kk * g;
        if (dmu1) dmu1[i] = kk * g;
        if (ds0) ds0[i] = kk * (2.f * a0 / B - A * a1 / (B * B) + a1 / B - 2.f * a0 / C0);
        if (ds1) ds1[i] = kk * (-A * a0 / (B * B) + a0 / B);
    }
}

// ---------------------------------------------------------------- residual multinoulli CE
struct CeP {
    const float* const* s; float* const* ds; const float* mask; double* part; float* out; const float* scale;
    int L, N, HW, nblk;
};

template <int K>
__global__ __launch_bounds__(256) void ce_fwd_k(const CeP p) {
    __shared__ double sm[4 * MAXL];
    const int b = blockIdx.y;
    double acc_l[MAXL];
#pragma unroll
    for (int l = 0; l < MAXL; ++l) acc_l[l] = 0.0;
    for (int q = blockIdx.x * 256 + threadIdx.x; q < p.HW; q += gridDim.x * 256) {
        const int t = (int)p.mask[(size_t)b * p.HW + q];
        float a[K];
#pragma unroll
        for (int k = 0; k < K; ++k) a[k] = 0.f;
#pragma unroll
        for (int l = MAXL - 1; l >= 0; --l) {
            if (l < p.L) {
                const float* s = p.s[l] + (size_t)b * K * p.HW + q;
                float mx = -INFINITY, at = 0.f;
#pragma unroll
                for (int k = 0; k < K; ++k) {
                    a[k] = (l == p.L - 1) ? s[(size_t)k * p.HW] : a[k] + s[(size_t)k * p.HW];
                    mx = fmaxf(mx, a[k]);
                    if (k == t) at = a[k];
                }
                float se = 0.f;
#pragma unroll
                for (int k = 0; k < K; ++k) se += expf(a[k] - mx);
                acc_l[l] += (double)(mx + logf(se) - at);
            }
        }
    }
    uz::block_sum_d<MAXL>(acc_l, sm);
    if (threadIdx.x == 0) {
        double* o = p.part + (size_t)(b * gridDim.x + blockIdx.x) * MAXL;
#pragma unroll
        for (int l = 0; l < MAXL; ++l) o[l] = acc_l[l];
    }
}
// one wave per level: lane j adds partials j, j + 64, ... in order, then a fixed butterfly (a volume has 4 096 partials per level:
// the former one-thread-per-level loop took 150 us of the loss tape)
__global__ __launch_bounds__(64 * MAXL) void ce_finalize_k(const CeP p) {
    const int l = threadIdx.x >> 6, lane = threadIdx.x & 63;
    if (l >= p.L) return;
    double s = 0.0;
    for (int i = lane; i < p.nblk; i += 64) s += p.part[(size_t)i * MAXL + l];
    s = uz::wave_sum_d(s);
    if (lane == 0) p.out[l] = (float)(s / p.N);
}
template <int K>
__global__ __launch_bounds__(256) void ce_bwd_k(const CeP p) {
    const int b = blockIdx.y;
    const float k0 = (p.scale ? p.scale[0] : 1.f) / (float)p.N;
    for (int q = blockIdx.x * 256 + threadIdx.x; q < p.HW; q += gridDim.x * 256) {
        const int t = (int)p.mask[(size_t)b * p.HW + q];
        float a[K], g[MAXL][K];
#pragma unroll
        for (int k = 0; k < K; ++k) a[k] = 0.f;
#pragma unroll
        for (int l = MAXL - 1; l >= 0; --l) {
            if (l < p.L) {
                const float* s = p.s[l] + (size_t)b * K * p.HW + q;
                float mx = -INFINITY;
#pragma unroll
                for (int k = 0; k < K; ++k) {
                    a[k] = (l == p.L - 1) ? s[(size_t)k * p.HW] : a[k] + s[(size_t)k * p.HW];
                    mx = fmaxf(mx, a[k]);
                }
                float e[K], se = 0.f;
#pragma unroll
                for (int k = 0; k < K; ++k) { e[k] = expf(a[k] - mx); se += e[k]; }
#pragma unroll
                for (int k = 0; k < K; ++k) g[l][k] = e[k] / se - (k == t ? 1.f : 0.f);
            }
        }
        float cum[K];
#pragma unroll
        for (int k = 0; k < K; ++k) cum[k] = 0.f;
#pragma unroll
        for (int l = 0; l < MAXL; ++l) {
            if (l < p.L) {
                float* d = p.ds[l] + (size_t)b * K * p.HW + q;
#pragma unroll
                for (int k = 0; k < K; ++k) { cum[k] += g[l][k]; d[(size_t)k * p.HW] = cum[k] * k0; }
            }
        }
    }
}

__global__ void sum_terms_k(const float* __restrict__ t, int n, float* __restrict__ total) {
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        float s = 0.f;
        for (int i = 0; i < n; ++i) s += t[i];
        total[0] = s;
    }
}

template <int K>
__global__ __launch_bounds__(256) void acc_softmax_argmax_k(const float* const* __restrict__ sp, int L, int HW,
                                                             float* __restrict__ acc, float* __restrict__ soft, uint8_t* __restrict__ label) {
    const int b = blockIdx.y;
    for (int q = blockIdx.x * 256 + threadIdx.x; q < HW; q += gridDim.x * 256) {
        float a[K];
        const size_t base = (size_t)b * K * HW + q;
#pragma unroll
        for (int k = 0; k < K; ++k) a[k] = sp[L - 1][base + (size_t)k * HW];        // s_accum = output_list[-1]
        for (int l = 0; l < L - 1; ++l)
#pragma unroll
            for (int k = 0; k < K; ++k) a[k] += sp[l][base + (size_t)k * HW];       // += output_list[i]
        float mx = a[0];
#pragma unroll
        for (int k = 1; k < K; ++k) mx = fmaxf(mx, a[k]);
        float e[K], se = 0.f;
#pragma unroll
        for (int k = 0; k < K; ++k) { e[k] = expf(a[k] - mx); se += e[k]; }
        int best = 0; float bv = e[0] / se;
#pragma unroll
        for (int k = 0; k < K; ++k) {
            const float sv = e[k] / se;
            if (acc) acc[base + (size_t)k * HW] = a[k];
            if (soft) soft[base + (size_t)k * HW] = sv;
            if (sv > bv) { bv = sv; best = k; }
        }
        if (label) label[(size_t)b * HW + q] = (uint8_t)best;
    }
}

// ---------------------------------------------------------------- Adam / vector ops
__global__ __launch_bounds__(256) void adam_k(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m, float* __restrict__ v,
                                               size_t n, float lr_over_bc1, float inv_sqrt_bc2, float beta1, float beta2, float eps, float wd, float gs) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        const float pi = p[i];
        const float gi = g[i] * gs + wd * pi;
        const float mi = m[i] + (gi - m[i]) * (1.f - beta1);            // exp_avg.lerp_(grad, 1 - beta1)
        const float vi = v[i] * beta2 + (1.f - beta2) * gi * gi;        // mul_(beta2).addcmul_(grad, grad, 1 - beta2)
        m[i] = mi; v[i] = vi;
        const float denom = sqrtf(vi) * inv_sqrt_bc2 + eps;
        p[i] = pi - lr_over_bc1 * (mi / denom);
    }
}
__global__ __launch_bounds__(256) void axpy_k(float* __restrict__ y, const float* __restrict__ x, float a, size_t n) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) y[i] += a * x[i];
}
__global__ __launch_bounds__(256) void scale_k(float* __restrict__ y, float a, size_t n) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) y[i] *= a;
}
// One 1024-thread workgroup per tensor, float4 body between scalar head / tail (tensor offsets in the flat buffer are only
// 4-byte aligned): the largest tensors (192 x 192 x 9) set the duration of the launch - 256 threads and scalar loads took 220 us.
__global__ __launch_bounds__(1024) void l2_norms_k(const float* __restrict__ flat, const int64_t* __restrict__ oc, float* __restrict__ out) {
    __shared__ double sm[16];
    const int64_t off = oc[2 * blockIdx.x], cnt = oc[2 * blockIdx.x + 1];
    const float* p = flat + off;
    int64_t head = (int64_t)((16 - (reinterpret_cast<uintptr_t>(p) & 15)) & 15) / 4;
    if (head > cnt) head = cnt;
    const int64_t nv = (cnt - head) / 4;
    double v = 0.0;
    for (int64_t i = threadIdx.x; i < head; i += 1024) { const double t = p[i]; v += t * t; }
    const float4* p4 = reinterpret_cast<const float4*>(p + head);
    for (int64_t i = threadIdx.x; i < nv; i += 1024) {
        const float4 q = p4[i];
        v += (double)q.x * q.x + (double)q.y * q.y + (double)q.z * q.z + (double)q.w * q.w;
    }
    for (int64_t i = head + 4 * nv + threadIdx.x; i < cnt; i += 1024) { const double t = p[i]; v += t * t; }
    v = uz::wave_sum_d(v);
    if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = v;
    __syncthreads();
    if (threadIdx.x == 0) {
        double s = 0.0;
        for (int w = 0; w < 16; ++w) s += sm[w];
        out[blockIdx.x] = (float)sqrt(s);
    }
}
__global__ __launch_bounds__(1024) void l2_norms_bwd_k(const float* __restrict__ flat, const int64_t* __restrict__ oc, const float* __restrict__ norms,
                                                        const float* __restrict__ scale, float* __restrict__ grad) {
    const int64_t off = oc[2 * blockIdx.x], cnt = oc[2 * blockIdx.x + 1];
    const float nrm = norms[blockIdx.x];
    const float k = nrm > 0.f ? scale[0] / nrm : 0.f;
    const float* p = flat + off;
    float* g = grad + off;                                   // same offset in a buffer of the same alignment class
    int64_t head = (int64_t)((16 - (reinterpret_cast<uintptr_t>(p) & 15)) & 15) / 4;
    if (head > cnt) head = cnt;
    const bool vec = (reinterpret_cast<uintptr_t>(g + head) & 15) == 0;
    const int64_t nv = vec ? (cnt - head) / 4 : 0;
    for (int64_t i = threadIdx.x; i < head; i += 1024) g[i] += k * p[i];
    const float4* p4 = reinterpret_cast<const float4*>(p + head);
    float4* g4 = reinterpret_cast<float4*>(g + head);
    for (int64_t i = threadIdx.x; i < nv; i += 1024) {
        const float4 q = p4[i];
        float4 r = g4[i];
        r.x += k * q.x; r.y += k * q.y; r.z += k * q.z; r.w += k * q.w;
        g4[i] = r;
    }
    for (int64_t i = head + 4 * nv + threadIdx.x; i < cnt; i += 1024) g[i] += k * p[i];
}

inline int vgrid(size_t n) { size_t g = (n + 255) / 256; if (g > 2048) g = 2048; if (g < 1) g = 1; return (int)g; }

}  // namespace

extern "C" int uz_posterior_input(const float* patch, int in_ch, const float* mask, int nlabels, float* out, int N, int H, int W, void* stream) {
    UZ_REQUIRE(in_ch > 0 && nlabels > 0 && N > 0 && H > 0 && W > 0 && N <= 65535, "posterior_input: bad sizes");
    const int HW = H * W;
    hipLaunchKernelGGL(posterior_input_k, dim3(uz::ceil_div(HW, 1024), in_ch + nlabels, N), dim3(256), 0, uz::S(stream), patch, in_ch, mask, nlabels, out, HW);
    return uz::check_launch("posterior_input_k");
}
extern "C" int uz_latent_sample_fwd(const float* mu, const float* pre_sigma, const float* eps, float* sigma, float* z, size_t n, int act, void* stream) {
    UZ_REQUIRE(n > 0 && pre_sigma && sigma, "latent_sample_fwd: bad arguments");
    UZ_REQUIRE(!z || (mu && eps), "latent_sample_fwd: z needs mu and eps");
    hipLaunchKernelGGL(latent_fwd_k, dim3(vgrid(n)), dim3(256), 0, uz::S(stream), mu, pre_sigma, eps, sigma, z, n, act);
    return uz::check_launch("latent_fwd_k");
}
extern "C" int uz_latent_sample_bwd(const float* dmu, const float* dsigma, const float* dz, const float* eps, const float* sigma,
                                    float* dmu_pre, float* dpre_sigma, size_t n, int act, void* stream) {
    UZ_REQUIRE(n > 0 && eps && sigma && dmu_pre && dpre_sigma, "latent_sample_bwd: bad arguments");
    hipLaunchKernelGGL(latent_bwd_k, dim3(vgrid(n)), dim3(256), 0, uz::S(stream), dmu, dsigma, dz, eps, sigma, dmu_pre, dpre_sigma, n, act);
    return uz::check_launch("latent_bwd_k");
}
extern "C" int uz_kl_fwd(const float* mu0, const float* s0, const float* mu1, const float* s1, int N, int per_sample, float weight,
                         float* loss_out, void* stream) {
    UZ_REQUIRE(N > 0 && per_sample > 0, "kl_fwd: empty tensor");
    hipLaunchKernelGGL(kl_fwd_k, dim3(1), dim3(1024), 0, uz::S(stream), mu0, s0, mu1, s1, N * per_sample, N, weight, loss_out);
    return uz::check_launch("kl_fwd_k");
}
extern "C" int uz_kl_fwd_ws(const float* mu0, const float* s0, const float* mu1, const float* s1, int N, int per_sample, float weight,
                            float* loss_out, void* workspace, void* stream) {
    UZ_REQUIRE(N > 0 && per_sample > 0, "kl_fwd: empty tensor");
    const long long total = (long long)N * per_sample;
    UZ_REQUIRE(total < (1ll << 31), "kl_fwd: tensor too large");
    if (!workspace || total <= 131072) return uz_kl_fwd(mu0, s0, mu1, s1, N, per_sample, weight, loss_out, stream);
    const int chunk = 65536;
    int G = (int)((total + chunk - 1) / chunk);
    const int ck = G > 64 ? (int)(((total + 63) / 64 + 1023) / 1024 * 1024) : chunk;      // at most 64 partials (512 bytes of workspace)
    G = (int)((total + ck - 1) / ck);
    double* part = static_cast<double*>(workspace);
    hipLaunchKernelGGL(kl_fwd_part_k, dim3(G), dim3(1024), 0, uz::S(stream), mu0, s0, mu1, s1, (int)total, ck, part);
    if (int rc = uz::check_launch("kl_fwd_part_k")) return rc;
    hipLaunchKernelGGL(kl_fwd_final_k, dim3(1), dim3(64), 0, uz::S(stream), part, G, N, weight, loss_out);
    return uz::check_launch("kl_fwd_final_k");
}
extern "C" int uz_kl_bwd(const float* mu0, const float* s0, const float* mu1, const float* s1, int N, int per_sample, float weight,
                         const float* loss_scale, float* dmu0, float* ds0, float* dmu1, float* ds1, void* stream) {
    UZ_REQUIRE(N > 0 && per_sample > 0, "kl_bwd: empty tensor");
    const int total = N * per_sample;
    hipLaunchKernelGGL(kl_bwd_k, dim3(vgrid(total)), dim3(256), 0, uz::S(stream), mu0, s0, mu1, s1, total, weight * 0.5f / (float)N, loss_scale,
                       dmu0, ds0, dmu1, ds1);
    return uz::check_launch("kl_bwd_k");
}

static int ce_blocks(int HW) { int g = uz::ceil_div(HW, 1024); return g < 1 ? 1 : (g > 64 ? 64 : g); }

extern "C" size_t uz_ce_workspace(int N, int H, int W, int L) {
    (void)L;
    return (size_t)N * ce_blocks(H * W) * MAXL * sizeof(double);
}
extern "C" int uz_residual_ce_fwd(const float* const* s_ptrs, int L, int K, const float* mask, int N, int H, int W,
                                  float* loss_out, void* workspace, void* stream) {
    UZ_REQUIRE(L >= 1 && L <= MAXL, "residual_ce_fwd: 1..%d levels supported", MAXL);
    UZ_REQUIRE(K >= 2 && K <= 4, "residual_ce_fwd: 2..4 classes supported (got %d)", K);
    UZ_REQUIRE(N > 0 && N <= 65535 && workspace, "residual_ce_fwd: bad arguments");
    CeP p = {}; p.s = s_ptrs; p.mask = mask; p.part = static_cast<double*>(workspace); p.out = loss_out; p.L = L; p.N = N; p.HW = H * W;
    const int gx = ce_blocks(p.HW); p.nblk = N * gx;
    const dim3 grid(gx, N);
    hipStream_t st = uz::S(stream);
    if (K == 2) hipLaunchKernelGGL(ce_fwd_k<2>, grid, dim3(256), 0, st, p);
    else if (K == 3) hipLaunchKernelGGL(ce_fwd_k<3>, grid, dim3(256), 0, st, p);
    else hipLaunchKernelGGL(ce_fwd_k<4>, grid, dim3(256), 0, st, p);
    if (int rc = uz::check_launch("ce_fwd_k")) return rc;
    hipLaunchKernelGGL(ce_finalize_k, dim3(1), dim3(64 * MAXL), 0, st, p);
    return uz::check_launch("ce_finalize_k");
}
extern "C" int uz_residual_ce_bwd(const float* const* s_ptrs, float* const* ds_ptrs, int L, int K, const float* mask, int N, int H, int W,
                                  const float* loss_scale, void* stream) {
    UZ_REQUIRE(L >= 1 && L <= MAXL, "residual_ce_bwd: 1..%d levels supported", MAXL);
    UZ_REQUIRE(K >= 2 && K <= 4, "residual_ce_bwd: 2..4 classes supported (got %d)", K);
    UZ_REQUIRE(N > 0 && N <= 65535, "residual_ce_bwd: bad arguments");
    CeP p = {}; p.s = s_ptrs; p.ds = ds_ptrs; p.mask = mask; p.scale = loss_scale; p.L = L; p.N = N; p.HW = H * W;
    const dim3 grid(ce_blocks(p.HW), N);
    hipStream_t st = uz::S(stream);
    if (K == 2) hipLaunchKernelGGL(ce_bwd_k<2>, grid, dim3(256), 0, st, p);
    else if (K == 3) hipLaunchKernelGGL(ce_bwd_k<3>, grid, dim3(256), 0, st, p);
    else hipLaunchKernelGGL(ce_bwd_k<4>, grid, dim3(256), 0, st, p);
    return uz::check_launch("ce_bwd_k");
}
extern "C" int uz_sum_terms(const float* terms, int n, float* total, void* stream) {
    hipLaunchKernelGGL(sum_terms_k, dim3(1), dim3(64), 0, uz::S(stream), terms, n, total);
    return uz::check_launch("sum_terms_k");
}
extern "C" int uz_accumulate_softmax_argmax(const float* const* s_ptrs, int L, int K, int N, int H, int W,
                                            float* acc, float* soft, uint8_t* label, void* stream) {
    UZ_REQUIRE(L >= 1 && K >= 2 && K <= 4 && N > 0 && N <= 65535, "accumulate_softmax_argmax: bad arguments");
    const int HW = H * W;
    const dim3 grid(ce_blocks(HW), N);
    hipStream_t st = uz::S(stream);
    if (K == 2) hipLaunchKernelGGL(acc_softmax_argmax_k<2>, grid, dim3(256), 0, st, s_ptrs, L, HW, acc, soft, label);
    else if (K == 3) hipLaunchKernelGGL(acc_softmax_argmax_k<3>, grid, dim3(256), 0, st, s_ptrs, L, HW, acc, soft, label);
    else hipLaunchKernelGGL(acc_softmax_argmax_k<4>, grid, dim3(256), 0, st, s_ptrs, L, HW, acc, soft, label);
    return uz::check_launch("acc_softmax_argmax_k");
}
extern "C" int uz_adam_step(float* params, const float* grads, float* exp_avg, float* exp_avg_sq, size_t n, int64_t step,
                            float lr, float beta1, float beta2, float eps, float weight_decay, float grad_scale, void* stream) {
    UZ_REQUIRE(step >= 1, "adam_step: step must be >= 1");
    if (n == 0) return 0;
    const double bc1 = 1.0 - pow((double)beta1, (double)step), bc2 = 1.0 - pow((double)beta2, (double)step);
    hipLaunchKernelGGL(adam_k, dim3(vgrid(n)), dim3(256), 0, uz::S(stream), params, grads, exp_avg, exp_avg_sq, n,
                       (float)((double)lr / bc1), (float)(1.0 / sqrt(bc2)), beta1, beta2, eps, weight_decay, grad_scale);
    return uz::check_launch("adam_k");
}
extern "C" int uz_add_views(const float* a, int CtotA, const float* b, int CtotB, float* y, int CtotY, int C, int N, int H, int W,
                            float alpha, int accumulate, const float* a_amax, const float* b_amax, float* y_amax, void* stream) {
    UZ_REQUIRE(a && y && C > 0 && N > 0 && H > 0 && W > 0, "add_views: bad arguments");
    UZ_REQUIRE(N <= 65535, "add_views: N exceeds grid limits");
    const int HW = H * W;
    const size_t per = (size_t)C * HW;
    auto al = [](const void* q) { return q == nullptr || (reinterpret_cast<uintptr_t>(q) & 15) == 0; };
    const bool v4 = HW % 4 == 0 && al(a) && al(b) && al(y);
    int gx = (int)((per / (v4 ? 4 : 1) + 255) / 256);
    gx = gx < 1 ? 1 : (gx > 1024 ? 1024 : gx);
    if (v4) hipLaunchKernelGGL(add_views_k<4>, dim3(gx, N), dim3(256), 0, uz::S(stream), a, CtotA, b, CtotB, y, CtotY, C, HW, alpha, accumulate, a_amax, b_amax, y_amax);
    else hipLaunchKernelGGL(add_views_k<1>, dim3(gx, N), dim3(256), 0, uz::S(stream), a, CtotA, b, CtotB, y, CtotY, C, HW, alpha, accumulate, a_amax, b_amax, y_amax);
    return uz::check_launch("add_views_k");
}
extern "C" int uz_axpy(float* y, const float* x, float alpha, size_t n, void* stream) {
    if (n == 0) return 0;
    hipLaunchKernelGGL(axpy_k, dim3(vgrid(n)), dim3(256), 0, uz::S(stream), y, x, alpha, n);
    return uz::check_launch("axpy_k");
}
// Zero-fill / copy as KERNEL launches: under stream capture they become kernel nodes like every other op of a tape (memset /
// memcpy nodes inside multi-branch hipGraphs crashed hipGraphLaunch on ROCm 7.2 at some sizes).
__global__ __launch_bounds__(256) void zero_k(float* __restrict__ p, size_t n) {
    for (size_t e = (size_t)blockIdx.x * 256 + threadIdx.x; e < n; e += (size_t)gridDim.x * 256) p[e] = 0.f;
}
__global__ __launch_bounds__(256) void copy_k(float* __restrict__ d, const float* __restrict__ s, size_t n) {
    for (size_t e = (size_t)blockIdx.x * 256 + threadIdx.x; e < n; e += (size_t)gridDim.x * 256) d[e] = s[e];
}
// ---- split storage <-> fp32 (include/uz_api.h, round 4; tests and tools - the model plans never convert whole tensors)
namespace {
__global__ __launch_bounds__(256) void pack_split_k(const float* __restrict__ x, unsigned* __restrict__ out, size_t n, const float* __restrict__ amax, int* flags) {
    const float s = uz::split_scale(uz::amax_read(amax));
    bool bad = false;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) out[i] = uz::pack_split(x[i], s, bad);
    uz::raise_flag(flags, bad, uz::FLAG_X_BOUND);
}
__global__ __launch_bounds__(256) void unpack_split_k(const unsigned* __restrict__ in, float* __restrict__ x, size_t n, const float* __restrict__ amax) {
    const float inv = uz::split_inv_scale(uz::amax_read(amax));
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) x[i] = uz::unpack_split(in[i], inv);
}
inline int stream_grid(size_t n) { size_t g = (n + 256 * 8 - 1) / (256 * 8); return (int)(g < 1 ? 1 : (g > 4096 ? 4096 : g)); }
}  // namespace
extern "C" int uz_pack_split(const float* x, float* packed, size_t n, const float* amax, void* stream) {
    UZ_REQUIRE(x && packed && amax, "pack_split: null argument");
    if (n == 0) return 0;
    hipLaunchKernelGGL(pack_split_k, dim3(stream_grid(n)), dim3(256), 0, uz::S(stream), x, reinterpret_cast<unsigned*>(packed), n, amax, uz::dev_flags_ptr());
    return uz::check_launch("pack_split_k");
}
extern "C" int uz_unpack_split(const float* packed, float* x, size_t n, const float* amax, void* stream) {
    UZ_REQUIRE(x && packed && amax, "unpack_split: null argument");
    if (n == 0) return 0;
    hipLaunchKernelGGL(unpack_split_k, dim3(stream_grid(n)), dim3(256), 0, uz::S(stream), reinterpret_cast<const unsigned*>(packed), x, n, amax);
    return uz::check_launch("unpack_split_k");
}

extern "C" int uz_zero_f32(float* p, size_t n, void* stream) {
    if (n == 0) return 0;
    hipLaunchKernelGGL(zero_k, dim3(vgrid(n)), dim3(256), 0, uz::S(stream), p, n);
    return uz::check_launch("zero_k");
}
extern "C" int uz_copy_f32(float* dst, const float* src, size_t n, void* stream) {
    if (n == 0) return 0;
    hipLaunchKernelGGL(copy_k, dim3(vgrid(n)), dim3(256), 0, uz::S(stream), dst, src, n);
    return uz::check_launch("copy_k");
}
extern "C" int uz_scale(float* y, float alpha, size_t n, void* stream) {
    if (n == 0) return 0;
    hipLaunchKernelGGL(scale_k, dim3(vgrid(n)), dim3(256), 0, uz::S(stream), y, alpha, n);
    return uz::check_launch("scale_k");
}
extern "C" int uz_l2_norms(const float* flat, const int64_t* offs_counts, int n_tensors, float* out, void* stream) {
    if (n_tensors <= 0) return 0;
    hipLaunchKernelGGL(l2_norms_k, dim3(n_tensors), dim3(1024), 0, uz::S(stream), flat, offs_counts, out);
    return uz::check_launch("l2_norms_k");
}
extern "C" int uz_l2_norms_bwd(const float* flat, const int64_t* offs_counts, int n_tensors, const float* norms, const float* scale,
                               float* grad_flat, void* stream) {
    if (n_tensors <= 0) return 0;
    hipLaunchKernelGGL(l2_norms_bwd_k, dim3(n_tensors), dim3(1024), 0, uz::S(stream), flat, offs_counts, norms, scale, grad_flat);
    return uz::check_launch("l2_norms_bwd_k");
}

// ---------------------------------------------------------------- latent noise and step counters
// eps ~ N(0, 1) for the reparameterised samples (phiseg.py:104 `torch.randn_like`, probabilistic_unet.py rsample): counter-based
// Philox4x32-10 + Box-Muller, one 128-bit counter per four outputs.  The stream is (seed, offset) in device memory, so a captured
// or replayed launch draws fresh numbers every time: uz_step_counters advances the offset behind the fill (and counts the
// BatchNorm batches, num_batches_tracked, in the same launch).  The reference draws on the host generator; no two devices share
// a random stream, so the draw ORDER is not part of the contract - determinism for a given seed is.
namespace {
__device__ __forceinline__ void philox_round(unsigned (&c)[4], unsigned k0, unsigned k1) {
    const unsigned long long p0 = (unsigned long long)0xD2511F53u * c[0], p1 = (unsigned long long)0xCD9E8D57u * c[2];
    const unsigned n0 = (unsigned)(p1 >> 32) ^ c[1] ^ k0, n1 = (unsigned)p1, n2 = (unsigned)(p0 >> 32) ^ c[3] ^ k1, n3 = (unsigned)p0;
    c[0] = n0; c[1] = n1; c[2] = n2; c[3] = n3;
}
__global__ __launch_bounds__(256) void randn_fill_k(float* __restrict__ dst, size_t n, const unsigned long long* __restrict__ state) {
    const unsigned long long seed = state[0], base = state[1];
    const size_t quads = (n + 3) / 4;
    for (size_t q = (size_t)blockIdx.x * 256 + threadIdx.x; q < quads; q += (size_t)gridDim.x * 256) {
        const unsigned long long ctr = base + q;
        unsigned c[4] = {(unsigned)ctr, (unsigned)(ctr >> 32), 0x5A5A5A5Au, 0u};
        unsigned k0 = (unsigned)seed, k1 = (unsigned)(seed >> 32);
#pragma unroll
        for (int r = 0; r < 10; ++r) { philox_round(c, k0, k1); k0 += 0x9E3779B9u; k1 += 0xBB67AE85u; }
        float o[4];
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const float u1 = ((float)(c[2 * h] >> 8) + 0.5f) * (1.0f / 16777216.0f);        // (0, 1): 24 bits, never 0
            const float u2 = ((float)(c[2 * h + 1] >> 8) + 0.5f) * (1.0f / 16777216.0f);
            const float r = sqrtf(-2.0f * logf(u1));
            float sn, cs;
            sincosf(6.283185307179586f * u2, &sn, &cs);
            o[2 * h] = r * cs; o[2 * h + 1] = r * sn;
        }
#pragma unroll
        for (int e = 0; e < 4; ++e)
            if (4 * q + e < n) dst[4 * q + e] = o[e];
    }
}
__global__ __launch_bounds__(256) void step_counters_k(long long* __restrict__ counters, const long long* __restrict__ idx, int n_idx,
                                                       unsigned long long* __restrict__ rng_state, unsigned long long advance) {
    const int e = blockIdx.x * 256 + threadIdx.x;
    if (e < n_idx) counters[idx[e]] += 1;                 // indices are distinct (one BatchNorm each)
    if (e == 0 && rng_state) rng_state[1] += advance;
}
}  // namespace
extern "C" int uz_randn_fill(float* dst, size_t n, const void* rng_state, void* stream) {
    UZ_REQUIRE(dst && rng_state, "randn_fill: null argument");
    if (n == 0) return 0;
    hipLaunchKernelGGL(randn_fill_k, dim3(vgrid((n + 3) / 4)), dim3(256), 0, uz::S(stream), dst, n, reinterpret_cast<const unsigned long long*>(rng_state));
    return uz::check_launch("randn_fill_k");
}
extern "C" int uz_step_counters(int64_t* counters, const int64_t* idx, int n_idx, void* rng_state, int64_t advance, void* stream) {
    UZ_REQUIRE(n_idx >= 0 && (n_idx == 0 || (counters && idx)), "step_counters: null argument");
    if (n_idx == 0 && !rng_state) return 0;
    hipLaunchKernelGGL(step_counters_k, dim3(uz::ceil_div(n_idx > 0 ? n_idx : 1, 256)), dim3(256), 0, uz::S(stream),
                       reinterpret_cast<long long*>(counters), reinterpret_cast<const long long*>(idx), n_idx,
                       reinterpret_cast<unsigned long long*>(rng_state), (unsigned long long)advance);
    return uz::check_launch("step_counters_k");
}
