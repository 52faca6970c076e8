// BatchNorm2d(eps, momentum) + ReLU, training and eval, forward and backward.
// Replaces nn.BatchNorm2d(eps=1e-3, momentum=0.01) -> nn.ReLU of the reference's Conv2D unit
// (torchlayers.py:20-21), i.e. aten::native_batch_norm(+_backward) and threshold_backward.
//
// Memory-bound streaming kernels.  Two regimes:
//  * large planes (N*H*W > SMALL_LIMIT): one workgroup per (image, channel, plane chunk), float4
//    loads, wave-shuffle + LDS reduction to fp64 partial sums, then an ordered per-channel
//    finalise (bitwise reproducible; no atomics), then the streaming apply pass.
//  * small planes: ONE workgroup per channel does statistics and apply in a single launch (the
//    second read of the <= 32 KB channel comes from L1/L2).
// Statistics are accumulated in fp64 so that E[y^2]-E[y]^2 carries no cancellation error into the
// 30+ stacked normalisations of PHiSeg.  One qualification, stated here because the sentence above is not literally true
// (ADVICE r3): the first stage is fp32 - a thread's float4 sums in the streaming pass, and the pairwise sums over the 256 pixels one
// wave holds in the partials that come out of a convolution's epilogue (bn_finalize_conv_partials; conv_split.hip); the accumulation
// ACROSS threads, waves, tiles and images is fp64.  A partial's
// sum of squares carries ~1.5e-7 of its own size; against the variance's share of it that is 1.5e-7 (mean/std)^2 per partial and
// 1/sqrt(#partials) of that in the channel's variance: <= 2e-5 at |mean|/std = 30, 2e-3 at 1000 (measured, 8 x 64 x 64 channel) -
// tests/test_ops_gpu.py::test_fused_bn_statistics_of_a_channel_with_a_large_offset holds both paths to those figures against fp64.
#include <stdlib.h>
#include <type_traits>
#include "uz_common.h"
#include "split_f16.h"

namespace {

constexpr int SMALL_LIMIT = 4096;    // N*H*W at or below which the fused single-launch path is used
constexpr int CHUNK = 16384;         // plane elements per workgroup on the large path

struct BnP {
    const float* y; const float* da; const float* gamma; const float* beta;
    float* rmean; float* rvar; float* save;          // save[0..C) mean, save[C..2C) rstd
    float* out;                                        // a (fwd) or dy (bwd)
    float* dgamma; float* dbeta; float* dbias;
    double* part; double* part2; double* chan;        // workspace
    float* mm;                                         // workspace: per-(group, chunk, channel) {max, max of the negated} partials for the output bound
    int C, CtotY, CtotDa, CtotOut, N, HW, parts;
    int nb, ngrp;                                      // reduction kernels: images per workgroup, number of image groups
    float eps, momentum;
    int training, relu;
    int pre;                                           // training statistics already in save[] (finalised from the convolution's partials)
    const float* cpart; int ncpart;                    // the producing convolution's per-tile partials {sum, sum sq, max, max(-y)} [ncpart][C]
    const float* slab; int nslab; const float* cbias; float* ywr;     // small path: y := conv bias + sum of the split-K slabs (uz_bn_relu_fwd_slabs)
    float* amax;                                       // nullable: atomic max of |out| (bound for a following split-fp16 convolution)
    int save4;                                         // save[] holds 4 C floats: mean, rstd and (written by the finalise kernel) alpha = gamma rstd, beta' = beta - mean alpha
    int out_packed;                                    // out is written as split storage (split_f16.h: two fp16 pieces of v * split_scale(*amax) per word)
    float* chanf;                                      // backward, finalised: [C] m1 = mean(dz), [C] m2 = mean(dz x_hat)
    int yb, dab, outb;                                 // bf16 STORAGE of y / da / out (2-byte elements; the *_st kernels)
    int nba;                                           // *_st apply kernels: images per workgroup
    int* flags;                                        // device flag word (uz_device_flags): raised when a value written as split storage exceeds its bound
};

// block-wide maxima of two floats (blockDim.x == 256); result valid in thread 0
__device__ __forceinline__ void block_max2(float& a, float& b, float* smf /* >= 8 */) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { a = fmaxf(a, __shfl_xor(a, o, 64)); b = fmaxf(b, __shfl_xor(b, o, 64)); }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane == 0) { smf[wave * 2] = a; smf[wave * 2 + 1] = b; }
    __syncthreads();
    if (threadIdx.x == 0) {
        a = fmaxf(fmaxf(smf[0], smf[2]), fmaxf(smf[4], smf[6]));
        b = fmaxf(fmaxf(smf[1], smf[3]), fmaxf(smf[5], smf[7]));
    }
}

__device__ __forceinline__ void alpha_beta(const BnP& p, int c, float& alpha, float& beta_, float& mean, float& rstd) {
    if (p.training) { mean = p.save[c]; rstd = p.save[p.C + c]; }
    else { mean = p.rmean[c]; rstd = 1.0f / sqrtf(p.rvar[c] + p.eps); }
    const float g = p.gamma ? p.gamma[c] : 1.f, b = p.beta ? p.beta[c] : 0.f;
    alpha = g * rstd;
    beta_ = b - mean * alpha;
}

// ------------------------------------------------------------------ forward, large path
template <bool VEC>
__global__ __launch_bounds__(256) void bn_stats_partial(const BnP p) {
    __shared__ double sm[8];
    __shared__ float smf[8];
    const int c = blockIdx.y, grp = blockIdx.z, part = blockIdx.x;
    const int lo = part * CHUNK, hi = min(p.HW, lo + CHUNK);
    double v2[2] = {0.0, 0.0};
    float vmx = -INFINITY, vmn = -INFINITY;             // max(y), max(-y): the normalised output's range follows from them exactly
    // one workgroup sweeps the chunk of p.nb consecutive images before it pays for the block reduction
    for (int b = grp * p.nb; b < min(p.N, (grp + 1) * p.nb); ++b) {
        const float* src = p.y + ((size_t)b * p.CtotY + c) * p.HW;
        float s = 0.f, ss = 0.f;
        if (VEC) {
            const float4* s4 = reinterpret_cast<const float4*>(src);
#pragma unroll 4
            for (int i = lo / 4 + threadIdx.x; i < hi / 4; i += 256) {
                const float4 v = s4[i];
                s += (v.x + v.y) + (v.z + v.w);
                ss += (v.x * v.x + v.y * v.y) + (v.z * v.z + v.w * v.w);
                vmx = fmaxf(fmaxf(vmx, fmaxf(v.x, v.y)), fmaxf(v.z, v.w));
                vmn = fmaxf(fmaxf(vmn, fmaxf(-v.x, -v.y)), fmaxf(-v.z, -v.w));
            }
        } else {
            for (int i = lo + threadIdx.x; i < hi; i += 256) { const float v = src[i]; s += v; ss += v * v; vmx = fmaxf(vmx, v); vmn = fmaxf(vmn, -v); }
        }
        v2[0] += (double)s; v2[1] += (double)ss;
    }
    uz::block_sum_d<2>(v2, sm);
    block_max2(vmx, vmn, smf);
    if (threadIdx.x == 0) {
        const size_t e = (size_t)(grp * p.parts + part) * p.C + c;
        p.part[e * 2] = v2[0]; p.part[e * 2 + 1] = v2[1];
        p.mm[e * 2] = vmx; p.mm[e * 2 + 1] = vmn;
    }
}

// Ordered sum of the per-(image, chunk) fp64 partials of channel c, done by the first wave of every
// workgroup that needs it (identical order everywhere, so every workgroup gets the same bits) - this
// replaces a separate "finalize" launch per layer.
__device__ __forceinline__ void channel_totals(const BnP& p, int c, double* red, double& t0, double& t1, float* redf = nullptr) {
    if (threadIdx.x < 64) {
        const int P = p.ngrp * p.parts;
        double s = 0.0, ss = 0.0;
        float ma = -INFINITY, mb = -INFINITY;
        for (int i = threadIdx.x; i < P; i += 64) {
            const double* o = p.part + ((size_t)i * p.C + c) * 2;
            s += o[0]; ss += o[1];
            if (redf) { ma = fmaxf(ma, p.mm[((size_t)i * p.C + c) * 2]); mb = fmaxf(mb, p.mm[((size_t)i * p.C + c) * 2 + 1]); }
        }
        s = uz::wave_sum_d(s); ss = uz::wave_sum_d(ss);
        if (redf) {
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) { ma = fmaxf(ma, __shfl_xor(ma, o, 64)); mb = fmaxf(mb, __shfl_xor(mb, o, 64)); }
        }
        if (threadIdx.x == 0) { red[0] = s; red[1] = ss; if (redf) { redf[0] = ma; redf[1] = mb; } }
    }
    __syncthreads();
    t0 = red[0]; t1 = red[1];
}

// Statistics from the partials the convolution's epilogue wrote while the tile was still in registers (conv_split.hip): one
// workgroup per channel adds them in a fixed order in fp64 and publishes mean / rstd, the running statistics and the output
// bound - the streaming statistics pass over y (a third of the forward BatchNorm traffic) disappears.
__global__ __launch_bounds__(256) void bn_finalize_conv_partials(const BnP p) {
    __shared__ double sm[8];
    __shared__ float smf[8];
    const int c = blockIdx.x;
    double v2[2] = {0.0, 0.0};
    float vmx = -INFINITY, vmn = -INFINITY;
    for (int i = threadIdx.x; i < p.ncpart; i += 256) {
        const float4 q = *reinterpret_cast<const float4*>(p.cpart + ((size_t)i * p.C + c) * 4);
        v2[0] += (double)q.x; v2[1] += (double)q.y;
        vmx = fmaxf(vmx, q.z); vmn = fmaxf(vmn, q.w);
    }
    uz::block_sum_d<2>(v2, sm);
    block_max2(vmx, vmn, smf);
    if (threadIdx.x == 0) {
        const double n = (double)p.N * p.HW;
        const double m = v2[0] / n;
        double var = v2[1] / n - m * m;
        if (var < 0.0) var = 0.0;
        const float mean = (float)m, rstd = (float)(1.0 / sqrt(var + (double)p.eps));
        p.save[c] = mean;
        p.save[p.C + c] = rstd;
        if (p.rmean) {
            const double unb = n > 1.0 ? var * n / (n - 1.0) : var;
            p.rmean[c] = (float)((1.0 - p.momentum) * p.rmean[c] + p.momentum * m);
            p.rvar[c] = (float)((1.0 - p.momentum) * p.rvar[c] + p.momentum * unb);
        }
        if (p.save4) {       // the affine map of this channel, as alpha_beta() forms it: consumers (a data gradient folding this unit's backward reduction) read it from here
            const float g = p.gamma ? p.gamma[c] : 1.f, bb = p.beta ? p.beta[c] : 0.f;
            p.save[2 * p.C + c] = g * rstd;
            p.save[3 * p.C + c] = bb - mean * (g * rstd);
        }
        if (p.amax) {
            const float g = p.gamma ? p.gamma[c] : 1.f, bb = p.beta ? p.beta[c] : 0.f;
            const float alpha = g * rstd, beta_ = bb - mean * alpha;
            const float floor0 = p.relu ? 0.f : -INFINITY;
            const float e0 = fmaxf(fmaf(vmx, alpha, beta_), floor0), e1 = fmaxf(fmaf(-vmn, alpha, beta_), floor0);
            uz::amax_publish_one(fmaxf(fabsf(e0), fabsf(e1)), p.amax, (unsigned)c);
        }
    }
}

template <bool VEC, bool PK = false>
__global__ __launch_bounds__(256) void bn_apply(const BnP p) {
    __shared__ double red[2];
    __shared__ float redf[2];
    const int c = blockIdx.y, b = blockIdx.z, part = blockIdx.x;
    float alpha, beta_, mean, rstd;
    if (!PK && p.training && !p.pre) {
        double s, ss;
        channel_totals(p, c, red, s, ss, redf);
        const double n = (double)p.N * p.HW;
        const double m = s / n;
        double var = ss / n - m * m;
        if (var < 0.0) var = 0.0;
        mean = (float)m;
        rstd = (float)(1.0 / sqrt(var + (double)p.eps));
        if (b == 0 && part == 0 && threadIdx.x == 0) {          // one workgroup per channel publishes the statistics
            p.save[c] = mean;
            p.save[p.C + c] = rstd;
            if (p.rmean) {
                const double unb = n > 1.0 ? var * n / (n - 1.0) : var;
                p.rmean[c] = (float)((1.0 - p.momentum) * p.rmean[c] + p.momentum * m);
                p.rvar[c] = (float)((1.0 - p.momentum) * p.rvar[c] + p.momentum * unb);
            }
        }
        const float g = p.gamma ? p.gamma[c] : 1.f, bb = p.beta ? p.beta[c] : 0.f;
        alpha = g * rstd;
        beta_ = bb - mean * alpha;
        if (p.amax && b == 0 && part == 0 && threadIdx.x == 0) {
            // the output is affine in y: its extremes sit at max(y) and min(y) = -max(-y) of the channel (exact; one
            // publication per channel instead of one per workgroup)
            const float floor0 = p.relu ? 0.f : -INFINITY;
            const float e0 = fmaxf(fmaf(redf[0], alpha, beta_), floor0), e1 = fmaxf(fmaf(-redf[1], alpha, beta_), floor0);
            uz::amax_publish_one(fmaxf(fabsf(e0), fabsf(e1)), p.amax, (unsigned)c);
        }
    } else {
        alpha_beta(p, c, alpha, beta_, mean, rstd);
    }
    const bool track = p.amax && !p.training;            // eval mode: no statistics pass to take the range from
    const float* src = p.y + ((size_t)b * p.CtotY + c) * p.HW;
    float* dst = p.out + ((size_t)b * p.CtotOut + c) * p.HW;
    const int lo = part * CHUNK, hi = min(p.HW, lo + CHUNK);
    const float floor_ = p.relu ? 0.f : -INFINITY;
    float vmax = 0.f;
    if (PK) {
        // split storage: the finalise launch in front of this one published the tensor's exact bound, so every workgroup derives
        // the same scale and the words it writes are the operand pieces the consuming convolution would have formed itself
        const float s = uz::split_scale(uz::amax_read(p.amax));
        const float4* s4 = reinterpret_cast<const float4*>(src);
        uint4* d4 = reinterpret_cast<uint4*>(dst);
        bool bad = false;
#pragma unroll 4
        for (int i = lo / 4 + threadIdx.x; i < hi / 4; i += 256) {
            const float4 v = s4[i];
            uint4 o;
            o.x = uz::pack_split(fmaxf(fmaf(v.x, alpha, beta_), floor_), s, bad); o.y = uz::pack_split(fmaxf(fmaf(v.y, alpha, beta_), floor_), s, bad);
            o.z = uz::pack_split(fmaxf(fmaf(v.z, alpha, beta_), floor_), s, bad); o.w = uz::pack_split(fmaxf(fmaf(v.w, alpha, beta_), floor_), s, bad);
            d4[i] = o;
        }
        uz::raise_flag(p.flags, bad, uz::FLAG_X_BOUND);
        return;
    }
    // (round 4, measured and dropped: non-temporal loads of y here - the step gained 0.5 %, inside the noise between boxes, and the
    //  launch itself lost 8 % when its input was still in the memory-side cache)
    if (VEC) {
        const float4* s4 = reinterpret_cast<const float4*>(src);
        float4* d4 = reinterpret_cast<float4*>(dst);
#pragma unroll 4
        for (int i = lo / 4 + threadIdx.x; i < hi / 4; i += 256) {
            float4 v = s4[i];
            v.x = fmaxf(fmaf(v.x, alpha, beta_), floor_); v.y = fmaxf(fmaf(v.y, alpha, beta_), floor_);
            v.z = fmaxf(fmaf(v.z, alpha, beta_), floor_); v.w = fmaxf(fmaf(v.w, alpha, beta_), floor_);
            d4[i] = v;
            vmax = fmaxf(fmaxf(vmax, fmaxf(fabsf(v.x), fabsf(v.y))), fmaxf(fabsf(v.z), fabsf(v.w)));
        }
    } else {
        for (int i = lo + threadIdx.x; i < hi; i += 256) {
            const float v = fmaxf(fmaf(src[i], alpha, beta_), floor_);
            dst[i] = v;
            vmax = fmaxf(vmax, fabsf(v));
        }
    }
    if (track) uz::amax_publish(vmax, p.amax);
}

// ------------------------------------------------------------------ forward, small path (one WG / channel)
// EPT elements per thread: a channel's whole batch lives in registers.  Three instances (2 / 8 / 16 elements: N*H*W <= 512 / 2048 / 4096,
// i.e. the 2x2 + 4x4 / 8x8 / larger planes at batch 32) instead of one sized for the largest: the deep levels' instances stay under
// the registers a device-filling convolution leaves free on a SIMD (2 waves x 184 of 512), so their workgroups start beside it instead
// of waiting for one of its workgroups to retire (tools/bench_coresidency.py: 5 us alone, 65 us beside the convolution at 156 VGPRs).
// Same element-to-thread mapping and the same order of every sum in all three: bit-identical results.
constexpr int small_ept(int total) { return total <= 2 * 256 ? 2 : total <= 8 * 256 ? 8 : SMALL_LIMIT / 256; }
template <int EPT>
__global__ __launch_bounds__(256) void bn_fused_small_fwd(const BnP p) {
    __shared__ double sm[8];
    __shared__ float bc[2];
    const int c = blockIdx.x;
    const int total = p.N * p.HW;
    // One trip to memory: the channel's N*HW <= 4096 values are loaded once (all loads of a thread in flight together), the
    // statistics and the normalised output are formed from registers.  (Three dependent passes over global memory made this
    // launch 8 - 14 us on the 8x8 ... 2x2 levels, whose chains are the step's exposed latency.)
    float v[EPT];
    int bq[EPT];                               // image index << 12 | pixel (both < 4096 here)
#pragma unroll
    for (int j = 0; j < EPT; ++j) {
        const int i = threadIdx.x + 256 * j;
        const int b = i / p.HW;
        bq[j] = (b << 12) | (i - b * p.HW);
    }
    if (p.slab) {
        // The producing convolution left its split-K partial sums [nslab][N][C][HW]: finish it here (bias + slabs in slab order =
        // conv_splitk_reduce's arithmetic, bit for bit) instead of in a launch of its own; y is written for the backward pass.
        const size_t n = (size_t)p.N * p.C * p.HW;
        const float bv = p.cbias ? p.cbias[c] : 0.f;
        const float* __restrict__ slab = p.slab;
        float* __restrict__ ywr = p.ywr;
#pragma unroll
        for (int j = 0; j < EPT; ++j) v[j] = bv;
        // four slabs' loads in flight per round (the rounds are dependent trips to L2 / HBM: 12 slabs one at a time made this launch
        // 20 us); added in slab order.  E = elements per thread in use, as a compile-time bound (8 / 2 / 1 on 8x8 / 4x4 / 2x2 at batch 32)
        {
            constexpr int E = EPT, U = E <= 8 ? 4 : 1;
            int k = 0;
            for (; k + U <= p.nslab; k += U) {
                float t[U][E];
#pragma unroll
                for (int kk = 0; kk < U; ++kk)
#pragma unroll
                    for (int j = 0; j < E; ++j)
                        t[kk][j] = threadIdx.x + 256 * j < total ? slab[(size_t)(k + kk) * n + ((size_t)(bq[j] >> 12) * p.C + c) * p.HW + (bq[j] & 4095)] : 0.f;
#pragma unroll
                for (int kk = 0; kk < U; ++kk)
#pragma unroll
                    for (int j = 0; j < E; ++j) v[j] += t[kk][j];
            }
            for (; k < p.nslab; ++k) {
                float t[E];
#pragma unroll
                for (int j = 0; j < E; ++j)
                    t[j] = threadIdx.x + 256 * j < total ? slab[(size_t)k * n + ((size_t)(bq[j] >> 12) * p.C + c) * p.HW + (bq[j] & 4095)] : 0.f;
#pragma unroll
                for (int j = 0; j < E; ++j) v[j] += t[j];
            }
        }
#pragma unroll
        for (int j = 0; j < EPT; ++j)
            if (threadIdx.x + 256 * j < total) ywr[((size_t)(bq[j] >> 12) * p.CtotY + c) * p.HW + (bq[j] & 4095)] = v[j];
    } else {
#pragma unroll
        for (int j = 0; j < EPT; ++j)
            v[j] = threadIdx.x + 256 * j < total ? p.y[((size_t)(bq[j] >> 12) * p.CtotY + c) * p.HW + (bq[j] & 4095)] : 0.f;
    }
    if (p.training) {
        double v2[2] = {0.0, 0.0};
#pragma unroll
        for (int j = 0; j < EPT; ++j)
            if (threadIdx.x + 256 * j < total) { const double d = v[j]; v2[0] += d; v2[1] += d * d; }
        uz::block_sum_d<2>(v2, sm);
        if (threadIdx.x == 0) {
            const double n = (double)total;
            const double mean = v2[0] / n;
            double var = v2[1] / n - mean * mean;
            if (var < 0.0) var = 0.0;
            p.save[c] = (float)mean;
            p.save[p.C + c] = (float)(1.0 / sqrt(var + (double)p.eps));
            if (p.rmean) {
                const double unb = n > 1.0 ? var * n / (n - 1.0) : var;
                p.rmean[c] = (float)((1.0 - p.momentum) * p.rmean[c] + p.momentum * mean);
                p.rvar[c] = (float)((1.0 - p.momentum) * p.rvar[c] + p.momentum * unb);
            }
            bc[0] = p.save[c]; bc[1] = p.save[p.C + c];
        }
        __syncthreads();
    }
    float alpha, beta_, mean, rstd;
    if (p.training) {
        mean = bc[0]; rstd = bc[1];
        const float g = p.gamma ? p.gamma[c] : 1.f, bb = p.beta ? p.beta[c] : 0.f;
        alpha = g * rstd; beta_ = bb - mean * alpha;
    } else {
        alpha_beta(p, c, alpha, beta_, mean, rstd);
    }
    const float floor_ = p.relu ? 0.f : -INFINITY;
    float vmax = 0.f;
#pragma unroll
    for (int j = 0; j < EPT; ++j)
        if (threadIdx.x + 256 * j < total) {
            const float r = fmaxf(fmaf(v[j], alpha, beta_), floor_);
            p.out[((size_t)(bq[j] >> 12) * p.CtotOut + c) * p.HW + (bq[j] & 4095)] = r;
            vmax = fmaxf(vmax, fabsf(r));
        }
    if (p.amax) uz::amax_publish(vmax, p.amax);
}

// ------------------------------------------------------------------ backward, large path
template <bool VEC>
__global__ __launch_bounds__(256) void bn_bwd_reduce_partial(const BnP p) {
    __shared__ double sm[8];
    __shared__ float smf[8];
    const int c = blockIdx.y, grp = blockIdx.z, part = blockIdx.x;
    float alpha, beta_, mean, rstd;
    alpha_beta(p, c, alpha, beta_, mean, rstd);
    const int lo = part * CHUNK, hi = min(p.HW, lo + CHUNK);
    double v2[2] = {0.0, 0.0};
    float mdz = 0.f, mxh = 0.f;                          // max |dz|, max |x_hat|: bound of the gradient this layer hands down
    for (int b = grp * p.nb; b < min(p.N, (grp + 1) * p.nb); ++b) {
        const float* ys = p.y + ((size_t)b * p.CtotY + c) * p.HW;
        const float* ds = p.da + ((size_t)b * p.CtotDa + c) * p.HW;
        float s1 = 0.f, s2 = 0.f;
        auto one = [&](float yv, float dv) {
            const float dz = (!p.relu || fmaf(yv, alpha, beta_) > 0.f) ? dv : 0.f;
            const float xh = (yv - mean) * rstd;
            s1 += dz; s2 += dz * xh;
            mdz = fmaxf(mdz, fabsf(dz)); mxh = fmaxf(mxh, fabsf(xh));
        };
        if (VEC) {
            const float4* y4 = reinterpret_cast<const float4*>(ys);
            const float4* d4 = reinterpret_cast<const float4*>(ds);
#pragma unroll 4
            for (int i = lo / 4 + threadIdx.x; i < hi / 4; i += 256) {
                const float4 yv = y4[i], dv = d4[i];
                one(yv.x, dv.x); one(yv.y, dv.y); one(yv.z, dv.z); one(yv.w, dv.w);
            }
        } else {
            for (int i = lo + threadIdx.x; i < hi; i += 256) one(ys[i], ds[i]);
        }
        v2[0] += (double)s1; v2[1] += (double)s2;
    }
    uz::block_sum_d<2>(v2, sm);
    block_max2(mdz, mxh, smf);
    if (threadIdx.x == 0) {
        const size_t e = (size_t)(grp * p.parts + part) * p.C + c;
        p.part[e * 2] = v2[0]; p.part[e * 2 + 1] = v2[1];
        p.mm[e * 2] = mdz; p.mm[e * 2 + 1] = mxh;
    }
}

// Backward statistics as a launch of their own (one workgroup per channel, fixed order, fp64): from the fp64 partials of
// bn_bwd_reduce_partial, or from the float4 partials {sum dz, sum dz x_hat, max |dz|, max |x_hat|} that the data gradient which
// wrote dA last left in its epilogue (conv_split.hip, MK == 2: the reduction pass over dA and y disappears).  Publishes dgamma, dbeta,
// m1, m2 and the bound of dy - BEFORE the apply pass starts, which is what lets that pass write split storage.
__global__ __launch_bounds__(256) void bn_bwd_finalize(const BnP p) {
    __shared__ double sm[8];
    __shared__ float smf[8];
    const int c = blockIdx.x;
    double v2[2] = {0.0, 0.0};
    float mdz = 0.f, mxh = 0.f;
    if (p.cpart) {
        for (int i = threadIdx.x; i < p.ncpart; i += 256) {
            const float4 q = *reinterpret_cast<const float4*>(p.cpart + ((size_t)i * p.C + c) * 4);
            v2[0] += (double)q.x; v2[1] += (double)q.y;
            mdz = fmaxf(mdz, q.z); mxh = fmaxf(mxh, q.w);
        }
    } else {
        const int P = p.ngrp * p.parts;
        for (int i = threadIdx.x; i < P; i += 256) {
            const double* o = p.part + ((size_t)i * p.C + c) * 2;
            v2[0] += o[0]; v2[1] += o[1];
            mdz = fmaxf(mdz, p.mm[((size_t)i * p.C + c) * 2]); mxh = fmaxf(mxh, p.mm[((size_t)i * p.C + c) * 2 + 1]);
        }
    }
    uz::block_sum_d<2>(v2, sm);
    block_max2(mdz, mxh, smf);
    if (threadIdx.x == 0) {
        float alpha, beta_, mean, rstd;
        alpha_beta(p, c, alpha, beta_, mean, rstd);
        const double n = (double)p.N * p.HW;
        const float m1 = (float)(v2[0] / n), m2 = (float)(v2[1] / n);
        p.chanf[c] = m1; p.chanf[p.C + c] = m2;
        if (p.dbeta) p.dbeta[c] = (float)v2[0];
        if (p.dgamma) p.dgamma[c] = (float)v2[1];
        if (p.amax) uz::amax_publish_one(fabsf(alpha) * (mdz + fabsf(m1) + mxh * fabsf(m2)), p.amax, (unsigned)c);
    }
}

template <bool VEC, bool PRE = false, bool PK = false>
__global__ __launch_bounds__(256) void bn_bwd_apply(const BnP p) {
    __shared__ double sm[4];
    __shared__ double red[2];
    __shared__ float redf[2];
    const int c = blockIdx.y, b = blockIdx.z, part = blockIdx.x;
    float alpha, beta_, mean, rstd;
    alpha_beta(p, c, alpha, beta_, mean, rstd);
    float m1, m2;
    if (PRE) {
        m1 = p.chanf[c]; m2 = p.chanf[p.C + c];
        const float* ys = p.y + ((size_t)b * p.CtotY + c) * p.HW;
        const float* ds = p.da + ((size_t)b * p.CtotDa + c) * p.HW;
        const int lo = part * CHUNK, hi = min(p.HW, lo + CHUNK);
        float sd = 0.f;
        auto one = [&](float yv, float dv) -> float {
            const float dz = (!p.relu || fmaf(yv, alpha, beta_) > 0.f) ? dv : 0.f;
            const float r = alpha * (dz - m1 - ((yv - mean) * rstd) * m2);
            sd += r;
            return r;
        };
        const float4* y4 = reinterpret_cast<const float4*>(ys);
        const float4* d4 = reinterpret_cast<const float4*>(ds);
        if (PK) {
            const float s = uz::split_scale(uz::amax_read(p.amax));
            uint4* o4 = reinterpret_cast<uint4*>(p.out + ((size_t)b * p.CtotOut + c) * p.HW);
            bool bad = false;
            for (int i = lo / 4 + threadIdx.x; i < hi / 4; i += 256) {
                const float4 yv = y4[i], dv = d4[i];
                uint4 o;
                o.x = uz::pack_split(one(yv.x, dv.x), s, bad); o.y = uz::pack_split(one(yv.y, dv.y), s, bad);
                o.z = uz::pack_split(one(yv.z, dv.z), s, bad); o.w = uz::pack_split(one(yv.w, dv.w), s, bad);
                o4[i] = o;
            }
            uz::raise_flag(p.flags, bad, uz::FLAG_DY_BOUND);
        } else if (VEC) {
            float4* o4 = reinterpret_cast<float4*>(p.out + ((size_t)b * p.CtotOut + c) * p.HW);
            for (int i = lo / 4 + threadIdx.x; i < hi / 4; i += 256) {
                const float4 yv = y4[i], dv = d4[i];
                float4 r;
                r.x = one(yv.x, dv.x); r.y = one(yv.y, dv.y); r.z = one(yv.z, dv.z); r.w = one(yv.w, dv.w);
                o4[i] = r;
            }
        } else {
            float* dst = p.out + ((size_t)b * p.CtotOut + c) * p.HW;
            for (int i = lo + threadIdx.x; i < hi; i += 256) dst[i] = one(ys[i], ds[i]);
        }
        if (p.dbias) {
            double v1[1] = {(double)sd};
            uz::block_sum_d<1>(v1, sm);
            if (threadIdx.x == 0) p.part2[(size_t)(b * p.parts + part) * p.C + c] = v1[0];
        }
        return;
    }
    double s1, s2;
    channel_totals(p, c, red, s1, s2, redf);
    const double n = (double)p.N * p.HW;
    m1 = (float)(s1 / n); m2 = (float)(s2 / n);
    if (b == 0 && part == 0 && threadIdx.x == 0) {
        if (p.dbeta) p.dbeta[c] = (float)s1;
        if (p.dgamma) p.dgamma[c] = (float)s2;
        // |dy| = |alpha| |dz - m1 - x_hat m2| <= |alpha| (max|dz| + |m1| + max|x_hat| |m2|): an upper bound within a small factor
        // of the true maximum, one publication per channel (the split-fp16 consumers need a bound, not the maximum)
        if (p.amax) uz::amax_publish_one(fabsf(alpha) * (redf[0] + fabsf(m1) + redf[1] * fabsf(m2)), p.amax, (unsigned)c);
    }
    const float* ys = p.y + ((size_t)b * p.CtotY + c) * p.HW;
    const float* ds = p.da + ((size_t)b * p.CtotDa + c) * p.HW;
    float* dst = p.out + ((size_t)b * p.CtotOut + c) * p.HW;
    const int lo = part * CHUNK, hi = min(p.HW, lo + CHUNK);
    float sd = 0.f;
    auto one = [&](float yv, float dv) -> float {
        const float dz = (!p.relu || fmaf(yv, alpha, beta_) > 0.f) ? dv : 0.f;
        const float r = alpha * (dz - m1 - ((yv - mean) * rstd) * m2);
        sd += r;
        return r;
    };
    if (VEC) {
        const float4* y4 = reinterpret_cast<const float4*>(ys);
        const float4* d4 = reinterpret_cast<const float4*>(ds);
        float4* o4 = reinterpret_cast<float4*>(dst);
        for (int i = lo / 4 + threadIdx.x; i < hi / 4; i += 256) {
            const float4 yv = y4[i], dv = d4[i];
            float4 r;
            r.x = one(yv.x, dv.x); r.y = one(yv.y, dv.y); r.z = one(yv.z, dv.z); r.w = one(yv.w, dv.w);
            o4[i] = r;
        }
    } else {
        for (int i = lo + threadIdx.x; i < hi; i += 256) dst[i] = one(ys[i], ds[i]);
    }
    if (p.dbias) {
        double v1[1] = {(double)sd};
        uz::block_sum_d<1>(v1, sm);
        if (threadIdx.x == 0) p.part2[(size_t)(b * p.parts + part) * p.C + c] = v1[0];
    }
}

// out[c] = sum_i part[i][c] (ordered), one wave per channel
__global__ __launch_bounds__(256) void chan_partial_sum(const double* __restrict__ part, int P, int C, float* __restrict__ out) {
    const int c = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (c >= C) return;
    double s = 0.0;
    for (int i = lane; i < P; i += 64) s += part[(size_t)i * C + c];
    s = uz::wave_sum_d(s);
    if (lane == 0) out[c] = (float)s;
}

// ------------------------------------------------------------------ bf16 STORAGE (round 4, the volume path: BASELINE config 5)
// The large-path kernels once more with every tensor operand either fp32 or bf16 (p.yb / p.dab / p.outb, workgroup-uniform branches
// around the loads and stores): half the bytes where a tensor is bf16, arithmetic and statistics in fp32 / fp64 as before.  H*W % 4
// == 0 (8-byte rows).  The statistics are those of the STORED (rounded) y, so forward and backward see the same numbers.
__global__ __launch_bounds__(256) void bn_stats_partial_st(const BnP p) {
    __shared__ double sm[8];
    __shared__ float smf[8];
    const int c = blockIdx.y, grp = blockIdx.z, part = blockIdx.x;
    const int lo = part * CHUNK, hi = min(p.HW, lo + CHUNK);
    double v2[2] = {0.0, 0.0};
    float vmx = -INFINITY, vmn = -INFINITY;
    for (int b = grp * p.nb; b < min(p.N, (grp + 1) * p.nb); ++b) {
        const size_t row = ((size_t)b * p.CtotY + c) * p.HW;
        float s = 0.f, ss = 0.f;
        auto one4 = [&](const uz::f32x4 v) {
            s += (v.x + v.y) + (v.z + v.w);
            ss += (v.x * v.x + v.y * v.y) + (v.z * v.z + v.w * v.w);
            vmx = fmaxf(fmaxf(vmx, fmaxf(v.x, v.y)), fmaxf(v.z, v.w));
            vmn = fmaxf(fmaxf(vmn, fmaxf(-v.x, -v.y)), fmaxf(-v.z, -v.w));
        };
        if (p.HW % 8 == 0) {
#pragma unroll 2
            for (int i = lo / 8 + threadIdx.x; i < hi / 8; i += 256) {
                uz::f32x4 a, b2;
                uz::ld_elem8(p.y, row + 8 * (size_t)i, p.yb, a, b2);
                one4(a); one4(b2);
            }
        } else {
#pragma unroll 4
            for (int i = lo / 4 + threadIdx.x; i < hi / 4; i += 256) one4(uz::ld_elem4(p.y, row + 4 * (size_t)i, p.yb));
        }
        v2[0] += (double)s; v2[1] += (double)ss;
    }
    uz::block_sum_d<2>(v2, sm);
    block_max2(vmx, vmn, smf);
    if (threadIdx.x == 0) {
        const size_t e = (size_t)(grp * p.parts + part) * p.C + c;
        p.part[e * 2] = v2[0]; p.part[e * 2 + 1] = v2[1];
        p.mm[e * 2] = vmx; p.mm[e * 2 + 1] = vmn;
    }
}
// apply pass; the statistics are in save[] already (bn_finalize_conv_partials) or come from the partials of bn_stats_partial_st
__global__ __launch_bounds__(256) void bn_apply_st(const BnP p) {
    __shared__ double red[2];
    const int c = blockIdx.y, b = blockIdx.z, part = blockIdx.x;
    float alpha, beta_, mean, rstd;
    if (p.training && !p.pre) {
        double s, ss;
        channel_totals(p, c, red, s, ss);
        const double n = (double)p.N * p.HW;
        const double m = s / n;
        double var = ss / n - m * m;
        if (var < 0.0) var = 0.0;
        mean = (float)m;
        rstd = (float)(1.0 / sqrt(var + (double)p.eps));
        if (b == 0 && part == 0 && threadIdx.x == 0) {
            p.save[c] = mean;
            p.save[p.C + c] = rstd;
            if (p.rmean) {
                const double unb = n > 1.0 ? var * n / (n - 1.0) : var;
                p.rmean[c] = (float)((1.0 - p.momentum) * p.rmean[c] + p.momentum * m);
                p.rvar[c] = (float)((1.0 - p.momentum) * p.rvar[c] + p.momentum * unb);
            }
        }
        const float g = p.gamma ? p.gamma[c] : 1.f, bb = p.beta ? p.beta[c] : 0.f;
        alpha = g * rstd;
        beta_ = bb - mean * alpha;
    } else {
        alpha_beta(p, c, alpha, beta_, mean, rstd);
    }
    const int lo = part * CHUNK, hi = min(p.HW, lo + CHUNK);
    const float floor_ = p.relu ? 0.f : -INFINITY;
    auto map4 = [&](uz::f32x4 v) -> uz::f32x4 {
        v.x = fmaxf(fmaf(v.x, alpha, beta_), floor_); v.y = fmaxf(fmaf(v.y, alpha, beta_), floor_);
        v.z = fmaxf(fmaf(v.z, alpha, beta_), floor_); v.w = fmaxf(fmaf(v.w, alpha, beta_), floor_);
        return v;
    };
    // a workgroup walks p.nba consecutive images (grid.z = image groups): a 128 x 64 plane in bf16 is 16 KB - one plane per workgroup
    // left the launch at 4.3 TB/s, the per-workgroup prologue (statistics, channel constants) weighing as much as the streaming
    for (int bb = b * p.nba; bb < min(p.N, (b + 1) * p.nba); ++bb) {
        const size_t yrow = ((size_t)bb * p.CtotY + c) * p.HW, orow = ((size_t)bb * p.CtotOut + c) * p.HW;
        if (p.HW % 8 == 0) {
#pragma unroll 2
            for (int i = lo / 8 + threadIdx.x; i < hi / 8; i += 256) {
                uz::f32x4 a, b2;
                uz::ld_elem8(p.y, yrow + 8 * (size_t)i, p.yb, a, b2);
                uz::st_elem8(p.out, orow + 8 * (size_t)i, map4(a), map4(b2), p.outb);
            }
        } else {
            for (int i = lo / 4 + threadIdx.x; i < hi / 4; i += 256)
                uz::st_elem4(p.out, orow + 4 * (size_t)i, map4(uz::ld_elem4(p.y, yrow + 4 * (size_t)i, p.yb)), p.outb);
        }
    }
}
__global__ __launch_bounds__(256) void bn_bwd_reduce_partial_st(const BnP p) {
    __shared__ double sm[8];
    __shared__ float smf[8];
    const int c = blockIdx.y, grp = blockIdx.z, part = blockIdx.x;
    float alpha, beta_, mean, rstd;
    alpha_beta(p, c, alpha, beta_, mean, rstd);
    const int lo = part * CHUNK, hi = min(p.HW, lo + CHUNK);
    double v2[2] = {0.0, 0.0};
    float mdz = 0.f, mxh = 0.f;
    for (int b = grp * p.nb; b < min(p.N, (grp + 1) * p.nb); ++b) {
        const size_t yrow = ((size_t)b * p.CtotY + c) * p.HW, drow = ((size_t)b * p.CtotDa + c) * p.HW;
        float s1 = 0.f, s2 = 0.f;
        auto one = [&](float yv, float dv) {
            const float dz = (!p.relu || fmaf(yv, alpha, beta_) > 0.f) ? dv : 0.f;
            const float xh = (yv - mean) * rstd;
            s1 += dz; s2 += dz * xh;
            mdz = fmaxf(mdz, fabsf(dz)); mxh = fmaxf(mxh, fabsf(xh));
        };
        if (p.HW % 8 == 0) {
#pragma unroll 2
            for (int i = lo / 8 + threadIdx.x; i < hi / 8; i += 256) {
                uz::f32x4 y0, y1, d0, d1;
                uz::ld_elem8(p.y, yrow + 8 * (size_t)i, p.yb, y0, y1);
                uz::ld_elem8(p.da, drow + 8 * (size_t)i, p.dab, d0, d1);
                one(y0.x, d0.x); one(y0.y, d0.y); one(y0.z, d0.z); one(y0.w, d0.w);
                one(y1.x, d1.x); one(y1.y, d1.y); one(y1.z, d1.z); one(y1.w, d1.w);
            }
        } else {
#pragma unroll 4
            for (int i = lo / 4 + threadIdx.x; i < hi / 4; i += 256) {
                const uz::f32x4 yv = uz::ld_elem4(p.y, yrow + 4 * (size_t)i, p.yb), dv = uz::ld_elem4(p.da, drow + 4 * (size_t)i, p.dab);
                one(yv.x, dv.x); one(yv.y, dv.y); one(yv.z, dv.z); one(yv.w, dv.w);
            }
        }
        v2[0] += (double)s1; v2[1] += (double)s2;
    }
    uz::block_sum_d<2>(v2, sm);
    block_max2(mdz, mxh, smf);
    if (threadIdx.x == 0) {
        const size_t e = (size_t)(grp * p.parts + part) * p.C + c;
        p.part[e * 2] = v2[0]; p.part[e * 2 + 1] = v2[1];
        p.mm[e * 2] = mdz; p.mm[e * 2 + 1] = mxh;
    }
}
__global__ __launch_bounds__(256) void bn_bwd_apply_st(const BnP p) {
    __shared__ double sm[4];
    __shared__ double red[2];
    const int c = blockIdx.y, b = blockIdx.z, part = blockIdx.x;
    float alpha, beta_, mean, rstd;
    alpha_beta(p, c, alpha, beta_, mean, rstd);
    double s1, s2;
    channel_totals(p, c, red, s1, s2);
    const double n = (double)p.N * p.HW;
    const float m1 = (float)(s1 / n), m2 = (float)(s2 / n);
    if (b == 0 && part == 0 && threadIdx.x == 0) {
        if (p.dbeta) p.dbeta[c] = (float)s1;
        if (p.dgamma) p.dgamma[c] = (float)s2;
    }
    const int lo = part * CHUNK, hi = min(p.HW, lo + CHUNK);
    float sd = 0.f;
    auto one = [&](float yv, float dv) -> float {
        const float dz = (!p.relu || fmaf(yv, alpha, beta_) > 0.f) ? dv : 0.f;
        float r = alpha * (dz - m1 - ((yv - mean) * rstd) * m2);
        if (p.outb) r = uz::bf16_round(r);            // (the conv-bias sum below is that of the stored values)
        sd += r;
        return r;
    };
    double sdd = 0.0;
    for (int bb = b * p.nba; bb < min(p.N, (b + 1) * p.nba); ++bb) {      // image group of this workgroup (see bn_apply_st)
    const size_t yrow = ((size_t)bb * p.CtotY + c) * p.HW, drow = ((size_t)bb * p.CtotDa + c) * p.HW, orow = ((size_t)bb * p.CtotOut + c) * p.HW;
    if (p.HW % 8 == 0) {
#pragma unroll 2
        for (int i = lo / 8 + threadIdx.x; i < hi / 8; i += 256) {
            uz::f32x4 y0, y1, d0, d1, r0, r1;
            uz::ld_elem8(p.y, yrow + 8 * (size_t)i, p.yb, y0, y1);
            uz::ld_elem8(p.da, drow + 8 * (size_t)i, p.dab, d0, d1);
            r0.x = one(y0.x, d0.x); r0.y = one(y0.y, d0.y); r0.z = one(y0.z, d0.z); r0.w = one(y0.w, d0.w);
            r1.x = one(y1.x, d1.x); r1.y = one(y1.y, d1.y); r1.z = one(y1.z, d1.z); r1.w = one(y1.w, d1.w);
            uz::st_elem8(p.out, orow + 8 * (size_t)i, r0, r1, p.outb);
        }
    } else {
        for (int i = lo / 4 + threadIdx.x; i < hi / 4; i += 256) {
            const uz::f32x4 yv = uz::ld_elem4(p.y, yrow + 4 * (size_t)i, p.yb), dv = uz::ld_elem4(p.da, drow + 4 * (size_t)i, p.dab);
            uz::f32x4 r;
            r.x = one(yv.x, dv.x); r.y = one(yv.y, dv.y); r.z = one(yv.z, dv.z); r.w = one(yv.w, dv.w);
            uz::st_elem4(p.out, orow + 4 * (size_t)i, r, p.outb);
        }
    }
    sdd += (double)sd; sd = 0.f;
    }
    if (p.dbias) {
        double v1[1] = {sdd};
        uz::block_sum_d<1>(v1, sm);
        if (threadIdx.x == 0) p.part2[(size_t)(b * p.parts + part) * p.C + c] = v1[0];      // one row per (image group, chunk)
    }
}

// ------------------------------------------------------------------ backward, small path
template <int EPT>
__global__ __launch_bounds__(256) void bn_fused_small_bwd(const BnP p) {
    __shared__ double sm[8];
    __shared__ float bc[2];
    const int c = blockIdx.x;
    const int total = p.N * p.HW;
    float alpha, beta_, mean, rstd;
    alpha_beta(p, c, alpha, beta_, mean, rstd);
    // register-resident like the forward: y and da are read once
    float dz[EPT], xh[EPT];
    int bq[EPT];
    double v2[2] = {0.0, 0.0};
#pragma unroll
    for (int j = 0; j < EPT; ++j) {
        const int i = threadIdx.x + 256 * j;
        const int b = i / p.HW;
        bq[j] = (b << 12) | (i - b * p.HW);
    }
    float yv[EPT], dv[EPT];
#pragma unroll
    for (int j = 0; j < EPT; ++j) {
        const bool ok = threadIdx.x + 256 * j < total;
        yv[j] = ok ? p.y[((size_t)(bq[j] >> 12) * p.CtotY + c) * p.HW + (bq[j] & 4095)] : 0.f;
        dv[j] = (ok && !p.slab) ? p.da[((size_t)(bq[j] >> 12) * p.CtotDa + c) * p.HW + (bq[j] & 4095)] : 0.f;
    }
    if (p.slab) {
        // dA is the sum of the split-K partial sums the data gradient left (uz_conv_bwd_data_slabs): added here in slab order -
        // conv_splitk_reduce's arithmetic - instead of in a reduction launch of its own; dA itself is never materialised
        const size_t n = (size_t)p.N * p.C * p.HW;
        const float* __restrict__ slab = p.slab;
        for (int k = 0; k < p.nslab; k += 4) {
            float t[4][EPT];
#pragma unroll
            for (int kk = 0; kk < 4; ++kk)
#pragma unroll
                for (int j = 0; j < EPT; ++j)
                    t[kk][j] = (k + kk < p.nslab && threadIdx.x + 256 * j < total)
                                   ? slab[(size_t)(k + kk) * n + ((size_t)(bq[j] >> 12) * p.C + c) * p.HW + (bq[j] & 4095)] : 0.f;
#pragma unroll
            for (int kk = 0; kk < 4; ++kk)
#pragma unroll
                for (int j = 0; j < EPT; ++j) dv[j] += t[kk][j];
        }
    }
#pragma unroll
    for (int j = 0; j < EPT; ++j)
        if (threadIdx.x + 256 * j < total) {
            dz[j] = (!p.relu || fmaf(yv[j], alpha, beta_) > 0.f) ? dv[j] : 0.f;
            xh[j] = (yv[j] - mean) * rstd;
            v2[0] += dz[j]; v2[1] += (double)(dz[j] * xh[j]);
        }
    uz::block_sum_d<2>(v2, sm);
    if (threadIdx.x == 0) {
        if (p.dbeta) p.dbeta[c] = (float)v2[0];
        if (p.dgamma) p.dgamma[c] = (float)v2[1];
        bc[0] = (float)(v2[0] / total); bc[1] = (float)(v2[1] / total);
    }
    __syncthreads();
    const float m1 = bc[0], m2 = bc[1];
    double sd[1] = {0.0};
    float vmax = 0.f;
#pragma unroll
    for (int j = 0; j < EPT; ++j)
        if (threadIdx.x + 256 * j < total) {
            const float r = alpha * (dz[j] - m1 - xh[j] * m2);
            p.out[((size_t)(bq[j] >> 12) * p.CtotOut + c) * p.HW + (bq[j] & 4095)] = r;
            sd[0] += r;
            vmax = fmaxf(vmax, fabsf(r));
        }
    if (p.amax) uz::amax_publish(vmax, p.amax);
    if (p.dbias) {
        uz::block_sum_d<1>(sd, sm);
        if (threadIdx.x == 0) p.dbias[c] = (float)sd[0];
    }
}

// ------------------------------------------------------------------ forward, mid path (round 4)
// 4096 < N*H*W <= MID_LIMIT, training mode: ONE 1024-thread workgroup per channel reads the channel's batch once into registers,
// forms the statistics (fp64 block reduction, fixed order) and the normalised output from them - one launch instead of
// [statistics | finalise] + apply, and no dependence on the convolution's partials.
// PK (split storage): a workgroup only knows its own channel's range, and the tensor-wide bound is needed BEFORE the first word is
// written - so the scale comes from an A-PRIORI bound every workgroup can form alone: |x_hat| <= sqrt(n - 1) for any n samples
// (Samuelson), hence |a| <= max_c (|gamma_c| sqrt(n - 1) + |beta_c|).  That bound is loose by sqrt(n) / (actual max |x_hat|) ~ 2^5:
// an element is then stored to max(2^-22 |a|, 2^-33 A) instead of max(2^-22 |a|, 2^-38 A) - still 2^9 below fp32's own rounding of
// the tensor's large elements.
// NT threads x E4 float4 per thread (round 5): <1024, 8> up to 32 768 values per channel; <512, 4> up to 8 192 (the 16 x 16 level at
// batch 32) - two waves per SIMD under 72 VGPRs, which start beside a device-filling convolution's two waves instead of waiting for
// a whole CU's registers (see bn_fused_small_fwd).
template <bool PK, int NT, int E4>
__global__ __launch_bounds__(NT) void bn_fused_mid_fwd(const BnP p) {
    __shared__ double smd[(NT / 64) * 2];
    __shared__ float bc[3];
    const int c = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int hw4 = p.HW / 4, total4 = p.N * hw4;
    float4 yv[E4];
#pragma unroll
    for (int j = 0; j < E4; ++j) {
        const int e = tid + j * NT;
        const int b = e / hw4, q = e - b * hw4;
        yv[j] = e < total4 ? *reinterpret_cast<const float4*>(p.y + ((size_t)b * p.CtotY + c) * p.HW + 4 * q) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
    double v2[2] = {0.0, 0.0};
#pragma unroll
    for (int j = 0; j < E4; ++j) {
        const float s = (yv[j].x + yv[j].y) + (yv[j].z + yv[j].w);
        const float ss = (yv[j].x * yv[j].x + yv[j].y * yv[j].y) + (yv[j].z * yv[j].z + yv[j].w * yv[j].w);
        v2[0] += (double)s; v2[1] += (double)ss;                     // (padding elements are zeros)
    }
    v2[0] = uz::wave_sum_d(v2[0]); v2[1] = uz::wave_sum_d(v2[1]);
    if (lane == 0) { smd[wave * 2] = v2[0]; smd[wave * 2 + 1] = v2[1]; }
    float apriori = 0.f;
    if (PK && wave == 1) {                                            // the a-priori bound of the whole tensor, from the parameters alone
        const float root = sqrtf((float)(p.N * p.HW - 1));
        for (int k = lane; k < p.C; k += 64) apriori = fmaxf(apriori, fabsf(p.gamma ? p.gamma[k] : 1.f) * root + fabsf(p.beta ? p.beta[k] : 0.f));
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) apriori = fmaxf(apriori, __shfl_xor(apriori, o, 64));
        if (lane == 0) bc[2] = apriori;
    }
    __syncthreads();
    if (tid == 0) {
        double a = 0.0, b = 0.0;
        for (int w = 0; w < NT / 64; ++w) { a += smd[w * 2]; b += smd[w * 2 + 1]; }
        const double n = (double)p.N * p.HW;
        const double m = a / n;
        double var = b / n - m * m;
        if (var < 0.0) var = 0.0;
        const float mean = (float)m, rstd = (float)(1.0 / sqrt(var + (double)p.eps));
        p.save[c] = mean; p.save[p.C + c] = rstd;
        if (p.rmean) {
            const double unb = n > 1.0 ? var * n / (n - 1.0) : var;
            p.rmean[c] = (float)((1.0 - p.momentum) * p.rmean[c] + p.momentum * m);
            p.rvar[c] = (float)((1.0 - p.momentum) * p.rvar[c] + p.momentum * unb);
        }
        const float g = p.gamma ? p.gamma[c] : 1.f, bb = p.beta ? p.beta[c] : 0.f;
        const float alpha = g * rstd, beta_ = bb - mean * alpha;
        if (p.save4) { p.save[2 * p.C + c] = alpha; p.save[3 * p.C + c] = beta_; }
        bc[0] = alpha; bc[1] = beta_;
    }
    __syncthreads();
    const float alpha = bc[0], beta_ = bc[1], floor_ = p.relu ? 0.f : -INFINITY;
    const float s = PK ? uz::split_scale(bc[2]) : 1.f;
    float vmax = 0.f;
#pragma unroll
    for (int j = 0; j < E4; ++j) {
        const int e = tid + j * NT;
        if (e < total4) {
            const int b = e / hw4, q = e - b * hw4;
            float4 r;
            r.x = fmaxf(fmaf(yv[j].x, alpha, beta_), floor_); r.y = fmaxf(fmaf(yv[j].y, alpha, beta_), floor_);
            r.z = fmaxf(fmaf(yv[j].z, alpha, beta_), floor_); r.w = fmaxf(fmaf(yv[j].w, alpha, beta_), floor_);
            float* dst = p.out + ((size_t)b * p.CtotOut + c) * p.HW + 4 * q;
            if (PK) {
                uint4 o;
                bool bad = false;
                o.x = uz::pack_split(r.x, s, bad); o.y = uz::pack_split(r.y, s, bad); o.z = uz::pack_split(r.z, s, bad); o.w = uz::pack_split(r.w, s, bad);
                uz::raise_flag(p.flags, bad, uz::FLAG_X_BOUND);
                *reinterpret_cast<uint4*>(dst) = o;
            } else {
                *reinterpret_cast<float4*>(dst) = r;
                vmax = fmaxf(fmaxf(vmax, fmaxf(fabsf(r.x), fabsf(r.y))), fmaxf(fabsf(r.z), fabsf(r.w)));
            }
        }
    }
    if (p.amax) {
        if (PK) { if (c == 0 && tid == 0) uz::amax_publish_one(bc[2], p.amax, 0u); }
        else uz::amax_publish(vmax, p.amax);
    }
}

// ------------------------------------------------------------------ backward, mid path (round 4)
// 4096 < N*H*W <= MID_LIMIT (the 16 x 16 and 32 x 32 levels at batch 32: 40 of PHiSeg's 106 units): ONE 1024-thread workgroup per
// channel keeps the channel's y and dA in registers (<= 32 + 32 values per thread) - both are read once, the sums, the gradient and
// the conv-bias sum come from the registers.  One launch and three passes over memory instead of three launches and five passes
// (reduce: dA, y; apply: dA, y, dy; bias-gradient summation): 192 ch @ 32 x 32: 32 -> 13 us.  The price: the tensor-wide bound of dy
// is only known when every workgroup is done, so dy cannot be written as split storage here - its bound is atomic-max accumulated
// like the small path's (the plans pack dy on the larger planes only).
constexpr int MID_LIMIT = 32768, MID_HALF_LIMIT = 8192;      // <1024 threads, 8 float4 each> | <512, 4> (see bn_fused_mid_fwd)
template <int MID_NT, int MID_EPT4>
__global__ __launch_bounds__(MID_NT) __attribute__((amdgpu_waves_per_eu(MID_NT == 512 ? 7 : 4, 8))) void bn_fused_mid_bwd(const BnP p) {
    __shared__ double smd[(MID_NT / 64) * 2];
    __shared__ float bc[2];
    const int c = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int hw4 = p.HW / 4, total4 = p.N * hw4;
    float alpha, beta_, mean, rstd;
    alpha_beta(p, c, alpha, beta_, mean, rstd);
    float4 yv[MID_EPT4], dv[MID_EPT4];
#pragma unroll
    for (int j = 0; j < MID_EPT4; ++j) {
        const int e = tid + j * MID_NT;
        const int b = e / hw4, q = e - b * hw4;
        const bool ok = e < total4;
        yv[j] = ok ? *reinterpret_cast<const float4*>(p.y + ((size_t)b * p.CtotY + c) * p.HW + 4 * q) : make_float4(0.f, 0.f, 0.f, 0.f);
        dv[j] = ok ? *reinterpret_cast<const float4*>(p.da + ((size_t)b * p.CtotDa + c) * p.HW + 4 * q) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
    // dz (ReLU mask) in place of dA, x_hat in place of y
    float s1 = 0.f, s2 = 0.f;
    auto prep = [&](float& yy, float& dd) {
        dd = (!p.relu || fmaf(yy, alpha, beta_) > 0.f) ? dd : 0.f;
        yy = (yy - mean) * rstd;
        s1 += dd; s2 += dd * yy;
    };
    double v2[2] = {0.0, 0.0};
#pragma unroll
    for (int j = 0; j < MID_EPT4; ++j) {
        s1 = 0.f; s2 = 0.f;
        prep(yv[j].x, dv[j].x); prep(yv[j].y, dv[j].y); prep(yv[j].z, dv[j].z); prep(yv[j].w, dv[j].w);
        v2[0] += (double)s1; v2[1] += (double)s2;     // (padding elements are zeros: they add nothing)
    }
    v2[0] = uz::wave_sum_d(v2[0]); v2[1] = uz::wave_sum_d(v2[1]);
    if (lane == 0) { smd[wave * 2] = v2[0]; smd[wave * 2 + 1] = v2[1]; }
    __syncthreads();
    if (tid == 0) {
        double a = 0.0, b = 0.0;
        for (int w = 0; w < MID_NT / 64; ++w) { a += smd[w * 2]; b += smd[w * 2 + 1]; }      // fixed order
        if (p.dbeta) p.dbeta[c] = (float)a;
        if (p.dgamma) p.dgamma[c] = (float)b;
        const double n = (double)p.N * p.HW;
        bc[0] = (float)(a / n); bc[1] = (float)(b / n);
    }
    __syncthreads();
    const float m1 = bc[0], m2 = bc[1];
    float sd = 0.f, vmax = 0.f;
#pragma unroll
    for (int j = 0; j < MID_EPT4; ++j) {
        const int e = tid + j * MID_NT;
        if (e < total4) {
            float4 r;
            r.x = alpha * (dv[j].x - m1 - yv[j].x * m2); r.y = alpha * (dv[j].y - m1 - yv[j].y * m2);
            r.z = alpha * (dv[j].z - m1 - yv[j].z * m2); r.w = alpha * (dv[j].w - m1 - yv[j].w * m2);
            const int b = e / hw4, q = e - b * hw4;
            *reinterpret_cast<float4*>(p.out + ((size_t)b * p.CtotOut + c) * p.HW + 4 * q) = r;
            sd += (r.x + r.y) + (r.z + r.w);
            vmax = fmaxf(fmaxf(vmax, fmaxf(fabsf(r.x), fabsf(r.y))), fmaxf(fabsf(r.z), fabsf(r.w)));
        }
    }
    if (p.amax) uz::amax_publish(vmax, p.amax);
    if (p.dbias) {
        double t = uz::wave_sum_d((double)sd);
        __syncthreads();
        if (lane == 0) smd[wave] = t;
        __syncthreads();
        if (tid == 0) {
            double a = 0.0;
            for (int w = 0; w < MID_NT / 64; ++w) a += smd[w];
            p.dbias[c] = (float)a;
        }
    }
}

// UZ_BN_MID_HALF=0: the 1 024-thread instances also for <= 8 192 values per channel (A/B only)
static bool mid_half_on() {
    static const bool on = !(getenv("UZ_BN_MID_HALF") && atoi(getenv("UZ_BN_MID_HALF")) == 0);
    return on;
}

// (round 4, measured and dropped: a "cluster" variant for the 64 x 64 / 128 x 128 levels - G = N*H*W / 32768 workgroups per channel,
//  each keeping its slice of y and dA in registers, meeting on a device-scope counter, adding the G partials in a fixed order - i.e.
//  one launch and three passes instead of three launches and five.  128 ch @ 128 x 128 ran 697 us against 266 us: the agent-scope
//  fences around the counter write back / invalidate the XCD's whole L2 while the other workgroups' dy stores are in flight; and
//  two such launches on the two dependency lanes starve each other's workgroups until the queue scheduler time-slices them
//  (3.7 images/s).  Cross-workgroup hand-offs inside a streaming kernel are not worth a kernel boundary on this chip.)

// ------------------------------------------------------------------ ReLU-only backward (vanilla U-Net units)
template <bool VEC>
__global__ __launch_bounds__(256) void relu_bwd_kernel(const float* __restrict__ da, int CtotDa, const float* __restrict__ a, int CtotA,
                                                        float* __restrict__ dy, int CtotDy, double* __restrict__ part2,
                                                        int C, int HW, int parts, float* __restrict__ amax) {
    __shared__ double sm[4];
    const int c = blockIdx.y, b = blockIdx.z, part = blockIdx.x;
    const float* as = a + ((size_t)b * CtotA + c) * HW;
    const float* ds = da + ((size_t)b * CtotDa + c) * HW;
    float* dst = dy + ((size_t)b * CtotDy + c) * HW;
    const int lo = part * CHUNK, hi = min(HW, lo + CHUNK);
    float sd = 0.f, vmax = 0.f;
    if (VEC) {
        const float4* a4 = reinterpret_cast<const float4*>(as);
        const float4* d4 = reinterpret_cast<const float4*>(ds);
        float4* o4 = reinterpret_cast<float4*>(dst);
        for (int i = lo / 4 + threadIdx.x; i < hi / 4; i += 256) {
            const float4 av = a4[i], dv = d4[i];
            float4 r;
            r.x = av.x > 0.f ? dv.x : 0.f; r.y = av.y > 0.f ? dv.y : 0.f;
            r.z = av.z > 0.f ? dv.z : 0.f; r.w = av.w > 0.f ? dv.w : 0.f;
            sd += (r.x + r.y) + (r.z + r.w);
            o4[i] = r;
            vmax = fmaxf(fmaxf(vmax, fmaxf(fabsf(r.x), fabsf(r.y))), fmaxf(fabsf(r.z), fabsf(r.w)));
        }
    } else {
        for (int i = lo + threadIdx.x; i < hi; i += 256) { const float r = as[i] > 0.f ? ds[i] : 0.f; sd += r; dst[i] = r; vmax = fmaxf(vmax, fabsf(r)); }
    }
    if (amax) uz::amax_publish(vmax, amax);
    if (part2) {
        double v1[1] = {(double)sd};
        uz::block_sum_d<1>(v1, sm);
        if (threadIdx.x == 0) part2[(size_t)(b * parts + part) * C + c] = v1[0];
    }
}

inline bool vec_ok(int HW, const void* a, const void* b, const void* c) {
    auto al = [](const void* q) { return q == nullptr || (reinterpret_cast<uintptr_t>(q) & 15) == 0; };
    return (HW % 4 == 0) && al(a) && al(b) && al(c);
}

}  // namespace

extern "C" size_t uz_bn_workspace(int C, int N, int H, int W) {
    const int parts = uz::ceil_div(H * W, CHUNK);
    const size_t P = (size_t)N * parts;
    return (P * C * 3 + (size_t)2 * C) * sizeof(double) + P * C * 2 * sizeof(float) + 256;
}

namespace {
// images per reduction workgroup: keep about 2048 workgroups (8 per CU) so that the fp64 block reduction is
// amortised over several planes without starving the chip
void reduction_groups(BnP& p) {
    long long nb = (long long)p.N * p.parts * p.C / 2048;
    if (nb < 1) nb = 1;
    if (nb > 8) nb = 8;
    p.nb = (int)nb;
    p.ngrp = uz::ceil_div(p.N, p.nb);
}
// images per workgroup of the *_st apply kernels.  Measured on 96 ch x 128 x (128 x 64) in bf16: groups of 6 - 8 planes (64 k elements
// per workgroup) ran the apply launches 4 - 8 % SLOWER than one 16 KB plane per workgroup (4.17 vs 4.35 TB/s forward, 4.39 vs 4.76
// backward) - the planes are not prologue-bound, and a 4x unroll of the 8-wide loops changed nothing either: what is left is the fixed cost of the 2 - 3 launches per unit
int apply_group(const BnP&) { return 1; }
void carve(BnP& p, void* ws) {
    const size_t P = (size_t)p.N * p.parts;
    p.part = static_cast<double*>(ws);
    p.part2 = p.part + P * p.C * 2;
    p.chan = p.part2 + P * p.C;
    p.mm = reinterpret_cast<float*>(p.chan + 2 * p.C);
}
}  // namespace

extern "C" int uz_bn_relu_fwd(const float* y, int C, int CtotY, const float* gamma, const float* beta,
                              float* running_mean, float* running_var, float* save_mean_rstd,
                              float* a, int CtotA, int N, int H, int W, float eps, float momentum,
                              int training, int relu, float* a_amax, void* workspace, void* stream) {
    return uz_bn_relu_fwd_pre(y, C, CtotY, gamma, beta, running_mean, running_var, save_mean_rstd, a, CtotA, N, H, W, eps, momentum,
                              training, relu, a_amax, workspace, nullptr, 0, stream);
}

static int bn_relu_fwd_impl(const float* y, int C, int CtotY, const float* gamma, const float* beta,
                            float* running_mean, float* running_var, float* save_mean_rstd,
                            float* a, int CtotA, int N, int H, int W, float eps, float momentum,
                            int training, int relu, float* a_amax, void* workspace,
                            const float* conv_partials, int n_partials, const float* slabs, int n_slabs, const float* conv_bias, void* stream,
                            int save4 = 0, int out_packed = 0, int phase = 0) {
    UZ_REQUIRE(C > 0 && N > 0 && H > 0 && W > 0, "bn_relu_fwd: empty tensor");
    const bool mid = training && !conv_partials && !slabs && (size_t)N * H * W > SMALL_LIMIT && (size_t)N * H * W <= (size_t)uz_bn_fwd_fused_limit(H, W) &&
                     vec_ok(H * W, y, a, nullptr);
    UZ_REQUIRE(!out_packed || (training && (conv_partials || mid || phase == 2) && a_amax && (size_t)N * H * W > SMALL_LIMIT),
               "bn_relu_fwd_ex: split storage needs the output's bound before the apply pass - training mode, a bound slot, and statistics from the convolution's partials (or the one-launch mid path)");
    UZ_REQUIRE(!conv_partials || (training && n_partials > 0 && (size_t)N * H * W > SMALL_LIMIT), "bn_relu_fwd: convolution partials only serve the training-mode large-plane path");
    UZ_REQUIRE(N <= 65535 && C <= 65535, "bn_relu_fwd: N or C exceeds grid limits");
    UZ_REQUIRE(!training || save_mean_rstd, "bn_relu_fwd: training needs save_mean_rstd");
    UZ_REQUIRE(training || (running_mean && running_var), "bn_relu_fwd: eval needs running statistics");
    UZ_REQUIRE(!training || (size_t)N * H * W > 1, "bn_relu_fwd: Expected more than 1 value per channel when training");
    hipStream_t st = uz::S(stream);
    BnP p = {}; p.flags = uz::dev_flags_ptr();
    p.y = y; p.gamma = gamma; p.beta = beta; p.rmean = running_mean; p.rvar = running_var; p.save = save_mean_rstd;
    p.out = a; p.C = C; p.CtotY = CtotY; p.CtotOut = CtotA; p.N = N; p.HW = H * W;
    p.parts = uz::ceil_div(p.HW, CHUNK);
    p.eps = eps; p.momentum = momentum; p.training = training; p.relu = relu; p.amax = a_amax;
    p.save4 = save4; p.out_packed = out_packed;
    UZ_REQUIRE(!slabs || (n_slabs > 1 && (size_t)N * p.HW <= SMALL_LIMIT), "bn_relu_fwd_slabs: split-K slabs only serve the small-plane path (N*H*W <= 4096)");
    if ((size_t)N * p.HW <= SMALL_LIMIT) {
        p.slab = slabs; p.nslab = n_slabs; p.cbias = conv_bias; p.ywr = const_cast<float*>(y);
        switch (small_ept(N * H * W)) {
            case 2: hipLaunchKernelGGL(bn_fused_small_fwd<2>, dim3(C), dim3(256), 0, st, p); break;
            case 8: hipLaunchKernelGGL(bn_fused_small_fwd<8>, dim3(C), dim3(256), 0, st, p); break;
            default: hipLaunchKernelGGL(bn_fused_small_fwd<SMALL_LIMIT / 256>, dim3(C), dim3(256), 0, st, p);
        }
        return uz::check_launch("bn_fused_small_fwd");
    }
    if (mid) {
        const bool half = (size_t)N * p.HW <= MID_HALF_LIMIT && mid_half_on();
        if (half) {
            if (out_packed) hipLaunchKernelGGL((bn_fused_mid_fwd<true, 512, 4>), dim3(C), dim3(512), 0, st, p);
            else hipLaunchKernelGGL((bn_fused_mid_fwd<false, 512, 4>), dim3(C), dim3(512), 0, st, p);
        } else {
            if (out_packed) hipLaunchKernelGGL((bn_fused_mid_fwd<true, 1024, 8>), dim3(C), dim3(1024), 0, st, p);
            else hipLaunchKernelGGL((bn_fused_mid_fwd<false, 1024, 8>), dim3(C), dim3(1024), 0, st, p);
        }
        return uz::check_launch("bn_fused_mid_fwd");
    }
    const bool vec = vec_ok(p.HW, y, a, nullptr);
    const dim3 grid(p.parts, C, N);
    reduction_groups(p);
    const dim3 rgrid(p.parts, C, p.ngrp);
    // phase (uz_bn_relu_fwd_phase): 1 = the statistics launch only (table + bound; y and a are not touched), 2 = the apply pass only (the table holds
    // this step's statistics already), 0 = both
    if (phase == 2) p.pre = 1;
    else if (training && conv_partials) {
        p.pre = 1; p.cpart = conv_partials; p.ncpart = n_partials;
        hipLaunchKernelGGL(bn_finalize_conv_partials, dim3(C), dim3(256), 0, st, p);
        if (int rc = uz::check_launch("bn_finalize_conv_partials")) return rc;
        if (phase == 1) return 0;
    } else if (training) {
        UZ_REQUIRE(workspace, "bn_relu_fwd: workspace required");
        carve(p, workspace);
        if (vec) hipLaunchKernelGGL(bn_stats_partial<true>, rgrid, dim3(256), 0, st, p);
        else hipLaunchKernelGGL(bn_stats_partial<false>, rgrid, dim3(256), 0, st, p);
        if (int rc = uz::check_launch("bn_stats_partial")) return rc;
    }
    if (out_packed) {
        UZ_REQUIRE(vec, "bn_relu_fwd_ex: split storage needs H*W %% 4 == 0 and 16-byte aligned views");
        hipLaunchKernelGGL((bn_apply<true, true>), grid, dim3(256), 0, st, p);
    } else if (vec) hipLaunchKernelGGL(bn_apply<true>, grid, dim3(256), 0, st, p);
    else hipLaunchKernelGGL(bn_apply<false>, grid, dim3(256), 0, st, p);
    return uz::check_launch("bn_apply");
}

extern "C" int uz_bn_relu_fwd_pre(const float* y, int C, int CtotY, const float* gamma, const float* beta,
                                  float* running_mean, float* running_var, float* save_mean_rstd,
                                  float* a, int CtotA, int N, int H, int W, float eps, float momentum,
                                  int training, int relu, float* a_amax, void* workspace,
                                  const float* conv_partials, int n_partials, void* stream) {
    return bn_relu_fwd_impl(y, C, CtotY, gamma, beta, running_mean, running_var, save_mean_rstd, a, CtotA, N, H, W, eps, momentum,
                            training, relu, a_amax, workspace, conv_partials, n_partials, nullptr, 0, nullptr, stream);
}
// uz_bn_relu_fwd_pre with (a) save_mean_rstd_ab holding 4 C floats - mean, rstd, and alpha = gamma rstd, beta' = beta - mean alpha,
// written where the statistics come from the convolution's partials (a data gradient that folds this unit's backward reduction reads
// them, uz_conv_bwd_data_bn) - and (b) out_packed: `a` is written as split storage (include/uz_api.h, "Split storage").
extern "C" int uz_bn_relu_fwd_ex(const float* y, int C, int CtotY, const float* gamma, const float* beta,
                                 float* running_mean, float* running_var, float* save_mean_rstd_ab,
                                 float* a, int CtotA, int N, int H, int W, float eps, float momentum,
                                 int training, int relu, float* a_amax, void* workspace,
                                 const float* conv_partials, int n_partials, int out_packed, void* stream) {
    return bn_relu_fwd_impl(y, C, CtotY, gamma, beta, running_mean, running_var, save_mean_rstd_ab, a, CtotA, N, H, W, eps, momentum,
                            training, relu, a_amax, workspace, conv_partials, n_partials, nullptr, 0, nullptr, stream, 1, out_packed);
}
// uz_bn_relu_fwd_ex in two launches' worth of calls (large planes, training mode, statistics from the convolution's partials, 4 C statistics table):
// phase 1 = finalise the statistics (table, running buffers, the activation's bound) - neither y nor a is touched (may be null);
// phase 2 = the apply pass alone, from the table and the bound phase 1 left (conv_partials unused).  A following convolution that applies
// the normalisation itself (uz_conv_fwd_bn_ex) depends on phase 1 only.
extern "C" int uz_bn_relu_fwd_phase(const float* y, int C, int CtotY, const float* gamma, const float* beta,
                                    float* running_mean, float* running_var, float* save_mean_rstd_ab,
                                    float* a, int CtotA, int N, int H, int W, float eps, float momentum,
                                    int relu, float* a_amax, const float* conv_partials, int n_partials, int out_packed, int phase, void* stream) {
    UZ_REQUIRE(phase == 1 || phase == 2, "bn_relu_fwd_phase: phase 1 (statistics) or 2 (apply)");
    UZ_REQUIRE((size_t)N * H * W > (size_t)uz_bn_fwd_fused_limit(H, W), "bn_relu_fwd_phase: large planes only (beyond uz_bn_fwd_fused_limit)");
    UZ_REQUIRE(phase == 2 || (conv_partials && n_partials > 0), "bn_relu_fwd_phase: the statistics phase finalises the convolution's partials");
    UZ_REQUIRE(phase == 1 || (y && a), "bn_relu_fwd_phase: the apply phase needs y and a");
    return bn_relu_fwd_impl(y, C, CtotY, gamma, beta, running_mean, running_var, save_mean_rstd_ab, a, CtotA, N, H, W, eps, momentum,
                            1, relu, a_amax, nullptr, phase == 1 ? conv_partials : nullptr, phase == 1 ? n_partials : 0, nullptr, 0, nullptr, stream, 1, phase == 2 ? out_packed : 0, phase);
}
// Conv2d (split-K, fp32 kernel) + BatchNorm + ReLU on the small planes with the convolution's reduce folded in: `slabs` are the
// n_slabs partial-sum tensors [n_slabs][N][C][H*W] uz_conv_fwd_slabs left in its workspace, conv_bias the convolution's bias
// (nullable); y (written here: conv output, read by the backward pass) = bias + slabs in order.  N*H*W <= 4096.
extern "C" int uz_bn_relu_fwd_slabs(const float* slabs, int n_slabs, const float* conv_bias, float* y, int C, int CtotY,
                                    const float* gamma, const float* beta, float* running_mean, float* running_var, float* save_mean_rstd,
                                    float* a, int CtotA, int N, int H, int W, float eps, float momentum,
                                    int training, int relu, float* a_amax, void* stream) {
    UZ_REQUIRE(slabs && n_slabs > 1 && y, "bn_relu_fwd_slabs: needs at least two slabs and the output view");
    return bn_relu_fwd_impl(y, C, CtotY, gamma, beta, running_mean, running_var, save_mean_rstd, a, CtotA, N, H, W, eps, momentum,
                            training, relu, a_amax, nullptr, nullptr, 0, slabs, n_slabs, conv_bias, stream);
}

extern "C" int uz_chan_sum_partials_d(const double* partials, int n_rows, int C, float* out, void* stream) {
    UZ_REQUIRE(partials && out && n_rows > 0 && C > 0, "chan_sum_partials_d: bad arguments");
    hipLaunchKernelGGL(chan_partial_sum, dim3(uz::ceil_div(C, 4)), dim3(256), 0, uz::S(stream), partials, n_rows, C, out);
    return uz::check_launch("chan_partial_sum");
}
extern "C" int uz_bn_relu_bwd(const float* da, int CtotDa, const float* y, int C, int CtotY,
                              const float* gamma, const float* beta, const float* save_mean_rstd,
                              float* dy, int CtotDy, float* dgamma, float* dbeta, float* dbias,
                              int N, int H, int W, int relu, float* dy_amax, void* workspace, void* stream) {
    return uz_bn_relu_bwd_ex(da, CtotDa, y, C, CtotY, gamma, beta, save_mean_rstd, dy, CtotDy, dgamma, dbeta, dbias, N, H, W, relu, dy_amax, workspace,
                             nullptr, 0, 0, nullptr, nullptr, 0, stream);
}
// uz_bn_relu_bwd with (a) conv_partials: n_partials rows [C][4] of {sum dz, sum dz x_hat, max |dz|, max |x_hat|} left by the data
// gradient that wrote dA last (uz_conv_bwd_data_bn; dA then already carries the ReLU mask) - the reduction pass over dA and y is
// replaced by a one-workgroup-per-channel finalise - and (b) out_packed: dy is written as split storage (needs dy_amax; the
// finalise launch publishes the bound before the apply pass starts), (c) dbias_partials: uz_bn_bwd_dbias_rows() x C doubles that
// receive the per-workgroup sums of dy instead of a summation launch per unit (uz_chan_sum_table adds all units' rows in ONE launch
// at the end of a tape; dbias must then be NULL).  Large planes only (N*H*W > 4096).
extern "C" int uz_bn_relu_bwd_ex(const float* da, int CtotDa, const float* y, int C, int CtotY,
                                 const float* gamma, const float* beta, const float* save_mean_rstd,
                                 float* dy, int CtotDy, float* dgamma, float* dbeta, float* dbias,
                                 int N, int H, int W, int relu, float* dy_amax, void* workspace,
                                 const float* conv_partials, int n_partials, int out_packed, double* dbias_partials,
                                 const float* da_slabs, int n_da_slabs, void* stream) {
    UZ_REQUIRE(C > 0 && N > 0 && H > 0 && W > 0, "bn_relu_bwd: empty tensor");
    UZ_REQUIRE(!da_slabs || (n_da_slabs > 1 && (size_t)N * H * W <= SMALL_LIMIT), "bn_relu_bwd_ex: da_slabs only serve the small-plane path (N*H*W <= 4096)");
    UZ_REQUIRE(!dbias_partials || (!dbias && (size_t)N * H * W > SMALL_LIMIT), "bn_relu_bwd_ex: dbias_partials replaces dbias on the large-plane path");
    UZ_REQUIRE(!(conv_partials || out_packed) || (size_t)N * H * W > SMALL_LIMIT, "bn_relu_bwd_ex: folded statistics / split storage only serve the large-plane path");
    UZ_REQUIRE(!conv_partials || n_partials > 0, "bn_relu_bwd_ex: conv_partials without rows");
    UZ_REQUIRE(!out_packed || dy_amax, "bn_relu_bwd_ex: split storage needs the bound slot");
    UZ_REQUIRE(N <= 65535 && C <= 65535, "bn_relu_bwd: N or C exceeds grid limits");
    UZ_REQUIRE(save_mean_rstd, "bn_relu_bwd: needs the saved batch statistics");
    hipStream_t st = uz::S(stream);
    BnP p = {}; p.flags = uz::dev_flags_ptr();
    p.y = y; p.da = da; p.gamma = gamma; p.beta = beta; p.save = const_cast<float*>(save_mean_rstd);
    p.out = dy; p.dgamma = dgamma; p.dbeta = dbeta; p.dbias = dbias;
    p.C = C; p.CtotY = CtotY; p.CtotDa = CtotDa; p.CtotOut = CtotDy; p.N = N; p.HW = H * W;
    p.parts = uz::ceil_div(p.HW, CHUNK);
    p.training = 1; p.relu = relu; p.amax = dy_amax;
    if ((size_t)N * p.HW <= SMALL_LIMIT) {
        p.slab = da_slabs; p.nslab = n_da_slabs;
        switch (small_ept(N * H * W)) {
            case 2: hipLaunchKernelGGL(bn_fused_small_bwd<2>, dim3(C), dim3(256), 0, st, p); break;
            case 8: hipLaunchKernelGGL(bn_fused_small_bwd<8>, dim3(C), dim3(256), 0, st, p); break;
            default: hipLaunchKernelGGL(bn_fused_small_bwd<SMALL_LIMIT / 256>, dim3(C), dim3(256), 0, st, p);
        }
        return uz::check_launch("bn_fused_small_bwd");
    }
    static const bool mid_on = !(getenv("UZ_BN_MID") && atoi(getenv("UZ_BN_MID")) == 0);
    if (mid_on && !conv_partials && !out_packed && !dbias_partials && (size_t)N * p.HW <= MID_LIMIT && vec_ok(p.HW, y, da, dy)) {
        if ((size_t)N * p.HW <= MID_HALF_LIMIT && mid_half_on()) hipLaunchKernelGGL((bn_fused_mid_bwd<512, 4>), dim3(C), dim3(512), 0, st, p);
        else hipLaunchKernelGGL((bn_fused_mid_bwd<1024, 8>), dim3(C), dim3(1024), 0, st, p);
        return uz::check_launch("bn_fused_mid_bwd");
    }
    UZ_REQUIRE(workspace, "bn_relu_bwd: workspace required");
    carve(p, workspace);
    if (dbias_partials) {      // the conv-bias gradient's per-workgroup sums go to the caller's rows; one table-driven launch adds them later (uz_chan_sum_table)
        p.part2 = dbias_partials;
        p.dbias = reinterpret_cast<float*>(dbias_partials);      // (non-null: "write the partials")
    }
    const bool vec = vec_ok(p.HW, y, da, dy);
    const dim3 grid(p.parts, C, N);
    reduction_groups(p);
    const dim3 rgrid(p.parts, C, p.ngrp);
    const bool pre = conv_partials || out_packed;
    p.cpart = conv_partials; p.ncpart = n_partials; p.out_packed = out_packed;
    p.chanf = reinterpret_cast<float*>(p.chan);
    if (!conv_partials) {
        if (vec) hipLaunchKernelGGL(bn_bwd_reduce_partial<true>, rgrid, dim3(256), 0, st, p);
        else hipLaunchKernelGGL(bn_bwd_reduce_partial<false>, rgrid, dim3(256), 0, st, p);
        if (int rc = uz::check_launch("bn_bwd_reduce_partial")) return rc;
    }
    if (pre) {
        hipLaunchKernelGGL(bn_bwd_finalize, dim3(C), dim3(256), 0, st, p);
        if (int rc = uz::check_launch("bn_bwd_finalize")) return rc;
        if (out_packed) {
            UZ_REQUIRE(vec, "bn_relu_bwd_ex: split storage needs H*W %% 4 == 0 and 16-byte aligned views");
            hipLaunchKernelGGL((bn_bwd_apply<true, true, true>), grid, dim3(256), 0, st, p);
        } else if (vec) hipLaunchKernelGGL((bn_bwd_apply<true, true, false>), grid, dim3(256), 0, st, p);
        else hipLaunchKernelGGL((bn_bwd_apply<false, true, false>), grid, dim3(256), 0, st, p);
    } else if (vec) hipLaunchKernelGGL(bn_bwd_apply<true>, grid, dim3(256), 0, st, p);
    else hipLaunchKernelGGL(bn_bwd_apply<false>, grid, dim3(256), 0, st, p);
    if (int rc = uz::check_launch("bn_bwd_apply")) return rc;
    if (dbias && !dbias_partials) {
        hipLaunchKernelGGL(chan_partial_sum, dim3(uz::ceil_div(C, 4)), dim3(256), 0, st, p.part2, N * p.parts, C, dbias);
        if (int rc = uz::check_launch("chan_partial_sum")) return rc;
    }
    return 0;
}
// rows of dbias_partials a large-plane uz_bn_relu_bwd_ex call writes ([rows][C] doubles)
extern "C" int uz_bn_bwd_dbias_rows(int N, int H, int W) { return (size_t)N * H * W > (size_t)uz_bn_bwd_fused_limit(H, W) ? N * uz::ceil_div(H * W, CHUNK) : 0; }
// largest N*H*W whose training-mode FORWARD runs as one launch (small or mid path; the plans ask the convolution for no statistics
// partials there)
extern "C" int uz_bn_fwd_fused_limit(int H, int W) {
    static const bool mid_on = !(getenv("UZ_BN_MID_FWD") && atoi(getenv("UZ_BN_MID_FWD")) == 0);
    return (mid_on && (H * W) % 4 == 0) ? MID_LIMIT : SMALL_LIMIT;
}
// largest N*H*W whose backward runs as ONE launch with the channel's batch held on chip (no tensor-wide bound before the first
// write: dy of such a unit is fp32, never split storage): the small path's limit, or the mid path's where H*W allows float4 rows
extern "C" int uz_bn_bwd_fused_limit(int H, int W) {
    static const bool mid_on = !(getenv("UZ_BN_MID") && atoi(getenv("UZ_BN_MID")) == 0);
    return (mid_on && (H * W) % 4 == 0) ? MID_LIMIT : SMALL_LIMIT;
}

// ---- bf16 STORAGE entry points (include/uz_api.h, "bf16 storage").  y / a / da / dy each either fp32 or bf16 (flags), large-plane
// path only (N*H*W > 32 768, H*W % 4 == 0); statistics from the convolution's partials (conv_partials) or from a streaming pass.
extern "C" int uz_bn_relu_fwd_b16(const void* y, int C, int CtotY, const float* gamma, const float* beta,
                                  float* running_mean, float* running_var, float* save_mean_rstd,
                                  void* a, int CtotA, int N, int H, int W, float eps, float momentum, int training, int relu,
                                  void* workspace, const float* conv_partials, int n_partials, int y_b16, int a_b16, void* stream) {
    UZ_REQUIRE(C > 0 && N > 0 && H > 0 && W > 0, "bn_relu_fwd_b16: empty tensor");
    UZ_REQUIRE((size_t)N * H * W > MID_LIMIT && (H * W) % 4 == 0, "bn_relu_fwd_b16: the bf16-storage kernels serve the large-plane path (N*H*W > 32768, H*W %% 4 == 0)");
    UZ_REQUIRE((reinterpret_cast<uintptr_t>(y) & 15) == 0 && (reinterpret_cast<uintptr_t>(a) & 15) == 0, "bn_relu_fwd_b16: views must be 16-byte aligned");
    UZ_REQUIRE(N <= 65535 && C <= 65535, "bn_relu_fwd_b16: N or C exceeds grid limits");
    UZ_REQUIRE(!training || save_mean_rstd, "bn_relu_fwd_b16: training needs save_mean_rstd");
    UZ_REQUIRE(training || (running_mean && running_var), "bn_relu_fwd_b16: eval needs running statistics");
    UZ_REQUIRE(!conv_partials || (training && n_partials > 0), "bn_relu_fwd_b16: convolution partials only serve training mode");
    hipStream_t st = uz::S(stream);
    BnP p = {}; p.flags = uz::dev_flags_ptr();
    p.y = static_cast<const float*>(y); p.gamma = gamma; p.beta = beta; p.rmean = running_mean; p.rvar = running_var; p.save = save_mean_rstd;
    p.out = static_cast<float*>(a); p.C = C; p.CtotY = CtotY; p.CtotOut = CtotA; p.N = N; p.HW = H * W;
    p.parts = uz::ceil_div(p.HW, CHUNK);
    p.eps = eps; p.momentum = momentum; p.training = training; p.relu = relu;
    p.yb = y_b16 != 0; p.outb = a_b16 != 0;
    p.nba = apply_group(p);
    const dim3 grid(p.parts, C, uz::ceil_div(N, p.nba));
    reduction_groups(p);
    if (training && conv_partials) {
        p.pre = 1; p.cpart = conv_partials; p.ncpart = n_partials;
        hipLaunchKernelGGL(bn_finalize_conv_partials, dim3(C), dim3(256), 0, st, p);
        if (int rc = uz::check_launch("bn_finalize_conv_partials")) return rc;
    } else if (training) {
        UZ_REQUIRE(workspace, "bn_relu_fwd_b16: workspace required");
        carve(p, workspace);
        hipLaunchKernelGGL(bn_stats_partial_st, dim3(p.parts, C, p.ngrp), dim3(256), 0, st, p);
        if (int rc = uz::check_launch("bn_stats_partial_st")) return rc;
    }
    hipLaunchKernelGGL(bn_apply_st, grid, dim3(256), 0, st, p);
    return uz::check_launch("bn_apply_st");
}
extern "C" int uz_bn_relu_bwd_b16(const void* da, int CtotDa, const void* y, int C, int CtotY,
                                  const float* gamma, const float* beta, const float* save_mean_rstd,
                                  void* dy, int CtotDy, float* dgamma, float* dbeta, float* dbias,
                                  int N, int H, int W, int relu, void* workspace, int da_b16, int y_b16, int dy_b16, void* stream) {
    UZ_REQUIRE(C > 0 && N > 0 && H > 0 && W > 0, "bn_relu_bwd_b16: empty tensor");
    UZ_REQUIRE((size_t)N * H * W > MID_LIMIT && (H * W) % 4 == 0, "bn_relu_bwd_b16: the bf16-storage kernels serve the large-plane path (N*H*W > 32768, H*W %% 4 == 0)");
    UZ_REQUIRE((reinterpret_cast<uintptr_t>(y) & 15) == 0 && (reinterpret_cast<uintptr_t>(da) & 15) == 0 && (reinterpret_cast<uintptr_t>(dy) & 15) == 0, "bn_relu_bwd_b16: views must be 16-byte aligned");
    UZ_REQUIRE(N <= 65535 && C <= 65535, "bn_relu_bwd_b16: N or C exceeds grid limits");
    UZ_REQUIRE(save_mean_rstd && workspace, "bn_relu_bwd_b16: needs the saved statistics and a workspace");
    hipStream_t st = uz::S(stream);
    BnP p = {}; p.flags = uz::dev_flags_ptr();
    p.y = static_cast<const float*>(y); p.da = static_cast<const float*>(da); p.gamma = gamma; p.beta = beta; p.save = const_cast<float*>(save_mean_rstd);
    p.out = static_cast<float*>(dy); p.dgamma = dgamma; p.dbeta = dbeta; p.dbias = dbias;
    p.C = C; p.CtotY = CtotY; p.CtotDa = CtotDa; p.CtotOut = CtotDy; p.N = N; p.HW = H * W;
    p.parts = uz::ceil_div(p.HW, CHUNK);
    p.training = 1; p.relu = relu;
    p.yb = y_b16 != 0; p.dab = da_b16 != 0; p.outb = dy_b16 != 0;
    carve(p, workspace);
    reduction_groups(p);
    hipLaunchKernelGGL(bn_bwd_reduce_partial_st, dim3(p.parts, C, p.ngrp), dim3(256), 0, st, p);
    if (int rc = uz::check_launch("bn_bwd_reduce_partial_st")) return rc;
    p.nba = apply_group(p);
    const int nga = uz::ceil_div(N, p.nba);
    hipLaunchKernelGGL(bn_bwd_apply_st, dim3(p.parts, C, nga), dim3(256), 0, st, p);
    if (int rc = uz::check_launch("bn_bwd_apply_st")) return rc;
    if (dbias) {
        hipLaunchKernelGGL(chan_partial_sum, dim3(uz::ceil_div(C, 4)), dim3(256), 0, st, p.part2, nga * p.parts, C, dbias);
        if (int rc = uz::check_launch("chan_partial_sum")) return rc;
    }
    return 0;
}

extern "C" int uz_relu_bwd(const float* da, int CtotDa, const float* a, int C, int CtotA,
                           float* dy, int CtotDy, float* dbias, int N, int H, int W, float* dy_amax, void* workspace, void* stream) {
    UZ_REQUIRE(C > 0 && N > 0 && H > 0 && W > 0, "relu_bwd: empty tensor");
    UZ_REQUIRE(N <= 65535 && C <= 65535, "relu_bwd: N or C exceeds grid limits");
    UZ_REQUIRE(!dbias || workspace, "relu_bwd: workspace required for dbias");
    hipStream_t st = uz::S(stream);
    const int HW = H * W, parts = uz::ceil_div(HW, CHUNK);
    double* part2 = dbias ? static_cast<double*>(workspace) : nullptr;
    const dim3 grid(parts, C, N);
    if (vec_ok(HW, da, a, dy)) hipLaunchKernelGGL(relu_bwd_kernel<true>, grid, dim3(256), 0, st, da, CtotDa, a, CtotA, dy, CtotDy, part2, C, HW, parts, dy_amax);
    else hipLaunchKernelGGL(relu_bwd_kernel<false>, grid, dim3(256), 0, st, da, CtotDa, a, CtotA, dy, CtotDy, part2, C, HW, parts, dy_amax);
    if (int rc = uz::check_launch("relu_bwd_kernel")) return rc;
    if (dbias) {
        hipLaunchKernelGGL(chan_partial_sum, dim3(uz::ceil_div(C, 4)), dim3(256), 0, st, part2, N * parts, C, dbias);
        if (int rc = uz::check_launch("chan_partial_sum")) return rc;
    }
    return 0;
}
