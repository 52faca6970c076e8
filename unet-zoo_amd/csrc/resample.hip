// Resampling ops of the encoder-decoder: 2x2 average pooling (ceil_mode), bilinear x2 (both
// align_corners modes), nearest integer up-sampling, global spatial mean, channel broadcast.
// Reference call sites: phiseg.py:23,66,213-216,305-309,321; unet.py:22,67;
// probabilistic_unet.py:56,114-115,172-197.  All are HBM-bound streaming kernels: one workgroup
// per (plane chunk, channel, image), consecutive lanes on consecutive x (coalesced), backward
// passes written as deterministic gathers (no atomics).
#include "uz_common.h"
#include "split_f16.h"

namespace {

constexpr int PCH = 4096;   // output plane elements per workgroup

struct RsP {
    const float* src; float* dst;
    int C, CtotS, CtotD, N, H, W;      // H, W: sizes of the LOW-resolution side
    int Ho, Wo;                         // sizes of the HIGH-resolution side (pool input / upsample output)
    int ac, factor, accumulate;
    float sh, sw;
    const float* x_amax; float* y_amax;   // forward pooling / interpolation are convex combinations: |y| <= bound of |x|
    int pack;                             // forward pooling / interpolation: write dst as split storage scaled from *x_amax (split_f16.h)
    // backward kernels, optional: dst is the gradient w.r.t. the OUTPUT A of a Conv -> ReLU unit (vanilla U-Net blocks) and this launch
    // is its last writer - apply the unit's ReLU mask (a > 0), leave per-workgroup sums for its bias gradient, publish max |dst|
    const float* mask; int CtotM; double* part; float* m_amax;
    int* flags;                           // device flag word (uz_device_flags), raised by out_word
    int hb16;                             // bilinear band kernels: the HIGH-resolution tensor (forward: dst, backward: src) holds 2-byte bf16 elements
};
struct ReluFold { float sd = 0.f, vmax = 0.f; };
__device__ __forceinline__ float fold_value(const RsP& p, const float* mplane, int q, float v, ReluFold& f) {
    if (p.mask) { v = mplane[q] > 0.f ? v : 0.f; f.sd += v; f.vmax = fmaxf(f.vmax, fabsf(v)); }
    return v;
}
// (every thread of the workgroup calls this: it contains barriers) partial row = image * gridDim.x + blockIdx.x
__device__ __forceinline__ void fold_finish(const RsP& p, int c, int b, const ReluFold& f) {
    if (!p.mask) return;
    __shared__ double fsm[4];
    double v1[1] = {(double)f.sd};
    uz::block_sum_d<1>(v1, fsm);
    if (threadIdx.x == 0) p.part[((size_t)b * gridDim.x + blockIdx.x) * p.C + c] = v1[0];
    if (p.m_amax) uz::amax_publish(f.vmax, p.m_amax);
}

// output value as stored: the fp32 value, or (split storage) the word holding its two fp16 pieces
// (a value beyond the bound the scale was derived from - pooling and interpolation forward their INPUT's bound, so only a wrong
//  bound handed in through the C ABI can do that - is clamped and raises the device flag word)
__device__ __forceinline__ float out_word(const RsP& p, float v, float s) {
    if (!p.pack) return v;
    bool bad = false;
    const unsigned w = uz::pack_split(v, s, bad);
    uz::raise_flag(p.flags, bad, uz::FLAG_X_BOUND);
    return __builtin_bit_cast(float, w);
}
__device__ __forceinline__ float pack_scale(const RsP& p) { return p.pack ? uz::split_scale(uz::amax_read(p.x_amax)) : 1.f; }
// the output bound of a pooling / interpolation pass is its input's bound (one lane of the grid forwards it)
__device__ __forceinline__ void forward_bound(const RsP& p) {
    if (p.y_amax && p.x_amax && blockIdx.x == 0 && blockIdx.y == 0 && blockIdx.z == 0 && threadIdx.x == 0)
        uz::amax_publish_one(uz::amax_read(p.x_amax), p.y_amax, 0u);
}

// ---------------------------------------------------------------- avg pool 2x2 stride 2 ceil_mode
// here (Ho,Wo) is the pool INPUT (high-res) and (H,W) the pool OUTPUT
__global__ __launch_bounds__(256) void avgpool_fwd_k(const RsP p) {
    forward_bound(p);
    const int c = blockIdx.y, b = blockIdx.z;
    const float* s = p.src + ((size_t)b * p.CtotS + c) * p.Ho * p.Wo;
    float* d = p.dst + ((size_t)b * p.CtotD + c) * p.H * p.W;
    const int n = p.H * p.W;
    const float ps = pack_scale(p);
    for (int q = blockIdx.x * PCH + threadIdx.x; q < min(n, (int)(blockIdx.x + 1) * PCH); q += 256) {
        const int oy = q / p.W, ox = q - oy * p.W;
        const int y0 = 2 * oy, x0 = 2 * ox;
        const int y1 = min(y0 + 2, p.Ho), x1 = min(x0 + 2, p.Wo);
        float acc = 0.f;
        for (int yy = y0; yy < y1; ++yy)
            for (int xx = x0; xx < x1; ++xx) acc += s[yy * p.Wo + xx];
        d[q] = out_word(p, acc / (float)((y1 - y0) * (x1 - x0)), ps);
    }
}
__global__ __launch_bounds__(256) void avgpool_bwd_k(const RsP p) {   // src = dy (low res), dst = dx (high res)
    const int c = blockIdx.y, b = blockIdx.z;
    const float* s = p.src + ((size_t)b * p.CtotS + c) * p.H * p.W;
    float* d = p.dst + ((size_t)b * p.CtotD + c) * p.Ho * p.Wo;
    const int n = p.Ho * p.Wo;
    const float* mp = p.mask ? p.mask + ((size_t)b * p.CtotM + c) * n : nullptr;
    ReluFold f;
    for (int q = blockIdx.x * PCH + threadIdx.x; q < min(n, (int)(blockIdx.x + 1) * PCH); q += 256) {
        const int y = q / p.Wo, x = q - y * p.Wo;
        const int oy = y >> 1, ox = x >> 1;
        const int cnt = (min(2 * oy + 2, p.Ho) - 2 * oy) * (min(2 * ox + 2, p.Wo) - 2 * ox);
        const float v = s[oy * p.W + ox] / (float)cnt;
        d[q] = fold_value(p, mp, q, p.accumulate ? d[q] + v : v, f);
    }
    fold_finish(p, c, b, f);
}

// Even planes (Ho even, Wo % 4 == 0, 16-byte aligned views: every plane of the models at their usual sizes): a thread turns two float4 of
// the input rows 2 oy, 2 oy + 1 into one float2 of outputs / one float2 of dy into two float4 of dx - a quarter of the memory instructions
// and no per-element integer division (the scalar kernels above ran at 2.6 - 3.1 TB/s on 32 ch @ 128 x 128).
__global__ __launch_bounds__(256) void avgpool_fwd_v4(const RsP p) {
    forward_bound(p);
    const int c = blockIdx.y, b = blockIdx.z;
    const float* s = p.src + ((size_t)b * p.CtotS + c) * p.Ho * p.Wo;
    float* d = p.dst + ((size_t)b * p.CtotD + c) * p.H * p.W;
    const int w2 = p.W / 2, n2 = p.H * w2;               // float2 outputs per plane
    const float ps = pack_scale(p);
    for (int q = blockIdx.x * (PCH / 2) + threadIdx.x; q < min(n2, (int)(blockIdx.x + 1) * (PCH / 2)); q += 256) {
        const int oy = q / w2, j = q - oy * w2;
        const float4 r0 = *reinterpret_cast<const float4*>(s + (size_t)(2 * oy) * p.Wo + 4 * j);
        const float4 r1 = *reinterpret_cast<const float4*>(s + (size_t)(2 * oy + 1) * p.Wo + 4 * j);
        const float a0 = (((r0.x + r0.y) + r1.x) + r1.y) / 4.f, a1 = (((r0.z + r0.w) + r1.z) + r1.w) / 4.f;      // (same order of additions and the same division as the scalar kernel)
        *reinterpret_cast<float2*>(d + (size_t)oy * p.W + 2 * j) = make_float2(out_word(p, a0, ps), out_word(p, a1, ps));
    }
}
__global__ __launch_bounds__(256) void avgpool_bwd_v4(const RsP p) {   // src = dy (low res), dst = dx (high res)
    const int c = blockIdx.y, b = blockIdx.z;
    const float* s = p.src + ((size_t)b * p.CtotS + c) * p.H * p.W;
    float* d = p.dst + ((size_t)b * p.CtotD + c) * p.Ho * p.Wo;
    const int n = p.Ho * p.Wo, w4 = p.Wo / 4, n4 = p.Ho * w4;
    const float* mp = p.mask ? p.mask + ((size_t)b * p.CtotM + c) * n : nullptr;
    ReluFold f;
    for (int q = blockIdx.x * (PCH / 4) + threadIdx.x; q < min(n4, (int)(blockIdx.x + 1) * (PCH / 4)); q += 256) {
        const int y = q / w4, j = q - y * w4;
        const float2 g = *reinterpret_cast<const float2*>(s + (size_t)(y >> 1) * p.W + 2 * j);
        const float v0 = g.x / 4.f, v1 = g.y / 4.f;
        const int e = y * p.Wo + 4 * j;
        float4 o = make_float4(v0, v0, v1, v1);
        if (p.accumulate) { const float4 t = *reinterpret_cast<const float4*>(d + e); o.x += t.x; o.y += t.y; o.z += t.z; o.w += t.w; }
        o.x = fold_value(p, mp, e, o.x, f); o.y = fold_value(p, mp, e + 1, o.y, f); o.z = fold_value(p, mp, e + 2, o.z, f); o.w = fold_value(p, mp, e + 3, o.w, f);
        *reinterpret_cast<float4*>(d + e) = o;
    }
    fold_finish(p, c, b, f);
}

// ---------------------------------------------------------------- bilinear x2
__device__ __forceinline__ void src_index(int o, float scale, int ac, int in, int& i0, int& ip, float& l0, float& l1) {
    float r;
    if (ac) r = scale * (float)o;                                 // area_pixel_compute_source_index, align_corners
    else { r = scale * ((float)o + 0.5f) - 0.5f; if (r < 0.f) r = 0.f; }
    i0 = (int)r;
    if (i0 > in - 1) i0 = in - 1;
    ip = (i0 < in - 1) ? 1 : 0;
    l1 = r - (float)i0;
    l0 = 1.f - l1;
}
__global__ __launch_bounds__(256) void bilinear_fwd_k(const RsP p) {
    forward_bound(p);
    const int c = blockIdx.y, b = blockIdx.z;
    const float* s = p.src + ((size_t)b * p.CtotS + c) * p.H * p.W;
    float* d = p.dst + ((size_t)b * p.CtotD + c) * p.Ho * p.Wo;
    const int n = p.Ho * p.Wo;
    const float ps = pack_scale(p);
    for (int q = blockIdx.x * PCH + threadIdx.x; q < min(n, (int)(blockIdx.x + 1) * PCH); q += 256) {
        const int oy = q / p.Wo, ox = q - oy * p.Wo;
        int h1, hp, w1, wp; float h0l, h1l, w0l, w1l;
        src_index(oy, p.sh, p.ac, p.H, h1, hp, h0l, h1l);
        src_index(ox, p.sw, p.ac, p.W, w1, wp, w0l, w1l);
        const float* r0 = s + h1 * p.W + w1;
        const float* r1 = r0 + hp * p.W;
        d[q] = out_word(p, h0l * (w0l * r0[0] + w1l * r0[wp]) + h1l * (w0l * r1[0] + w1l * r1[wp]), ps);
    }
}
// gather form of upsample_bilinear2d_backward: every low-res pixel sums the <= 7x7 high-res pixels
// whose interpolation footprint touches it (weights recomputed exactly as in the forward).  The
// per-row / per-column tap weights depend only on the coordinate, so a workgroup tabulates them once
// in LDS (planes up to 128 x 128 low-res; larger planes compute them per pixel).
__device__ __forceinline__ float tap_weight(int o, int osize, float scale, int ac, int isize, int i) {
    if (o < 0 || o >= osize) return 0.f;
    int i0, ip; float l0, l1;
    src_index(o, scale, ac, isize, i0, ip, l0, l1);
    return (i0 == i ? l0 : 0.f) + (i0 + ip == i ? l1 : 0.f);
}
constexpr int BTAB = 128;
__global__ __launch_bounds__(256) void bilinear_bwd_k(const RsP p) {   // src = dy (high res), dst = dx (low res)
    __shared__ float wyT[BTAB * 7], wxT[BTAB * 7];
    const int c = blockIdx.y, b = blockIdx.z;
    const float* s = p.src + ((size_t)b * p.CtotS + c) * p.Ho * p.Wo;
    float* d = p.dst + ((size_t)b * p.CtotD + c) * p.H * p.W;
    const int n = p.H * p.W;
    const float* mpl = p.mask ? p.mask + ((size_t)b * p.CtotM + c) * n : nullptr;
    ReluFold fo;
    const bool tab = p.H <= BTAB && p.W <= BTAB;
    if (tab) {
        for (int e = threadIdx.x; e < (p.H + p.W) * 7; e += 256) {
            if (e < p.H * 7) { const int i = e / 7, k = e - i * 7; wyT[e] = tap_weight(2 * i - 2 + k, p.Ho, p.sh, p.ac, p.H, i); }
            else { const int e2 = e - p.H * 7, i = e2 / 7, k = e2 - i * 7; wxT[e2] = tap_weight(2 * i - 2 + k, p.Wo, p.sw, p.ac, p.W, i); }
        }
        __syncthreads();
    }
    for (int q = blockIdx.x * PCH + threadIdx.x; q < min(n, (int)(blockIdx.x + 1) * PCH); q += 256) {
        const int iy = q / p.W, ix = q - iy * p.W;
        float wy[7], wx[7];
        const int oy0 = 2 * iy - 2, ox0 = 2 * ix - 2;
#pragma unroll
        for (int k = 0; k < 7; ++k) {
            wy[k] = tab ? wyT[iy * 7 + k] : tap_weight(oy0 + k, p.Ho, p.sh, p.ac, p.H, iy);
            wx[k] = tab ? wxT[ix * 7 + k] : tap_weight(ox0 + k, p.Wo, p.sw, p.ac, p.W, ix);
        }
        float acc = 0.f;
#pragma unroll
        for (int ky = 0; ky < 7; ++ky) {
            if (wy[ky] != 0.f) {
                const float* row = s + (oy0 + ky) * p.Wo + ox0;
                float ra = 0.f;
#pragma unroll
                for (int kx = 0; kx < 7; ++kx)
                    if (wx[kx] != 0.f) ra += wx[kx] * row[kx];
                acc += wy[ky] * ra;
            }
        }
        d[q] = fold_value(p, mpl, q, p.accumulate ? d[q] + acc : acc, fo);
    }
    fold_finish(p, c, b, fo);
}

// Band kernels for planes whose rows fit LDS (the layers that carry the traffic).  Bilinear interpolation is separable,
// Y = Ry X Rx^T, hence dX = Ry^T dY Rx: a workgroup stages the high-res rows that LB low-res rows can touch (coalesced
// float4 reads, each dY element leaves HBM ~1.2 times), reduces them along x into LDS and then along y - 2 x KT
// multiply-adds per output instead of a KT x KT gather.  Low-res index i only receives from high-res 2i-1 .. 2i+2 (both
// align_corners modes, any size - enumerated); the tap tables cover 2i-2 .. 2i+3 and hold exactly the forward's weights.
constexpr int LB = 16, BWMAX = 128, KT = 6, BROWS = 2 * LB + KT - 2;
// The six-tap sums of the two band kernels as explicit fused multiply-add chains: the pair and the float4 kernel serve the same
// call by shape / alignment, so they must produce the same bits - left to the compiler, the same source expression was contracted
// differently in the two kernels.  (the compiler-contracted build of the float4 kernel also made the
// replayed training step non-deterministic under concurrency, cause not found; these chains are what the tests pin.)
__device__ __forceinline__ float dot6(const float* w, float a0, float a1, float a2, float a3, float a4, float a5) {
    float t = w[0] * a0;
    t = fmaf(w[1], a1, t); t = fmaf(w[2], a2, t); t = fmaf(w[3], a3, t); t = fmaf(w[4], a4, t); t = fmaf(w[5], a5, t);
    return t;
}
__global__ __launch_bounds__(256) void bilinear_bwd_sep_k(const RsP p) {   // src = dy (high res), dst = dx (low res)
    __shared__ float wyT[LB * KT];
    __shared__ float tx[BROWS * (BWMAX / 2)];                               // dY reduced along x
    const int c = blockIdx.y, b = blockIdx.z;
    const float* s = p.src + ((size_t)b * p.CtotS + c) * p.Ho * p.Wo;
    float* d = p.dst + ((size_t)b * p.CtotD + c) * p.H * p.W;
    const float* mpl = p.mask ? p.mask + ((size_t)b * p.CtotM + c) * p.H * p.W : nullptr;
    ReluFold fo;
    // x pass straight from HBM: the KT taps of low-res column ix are the three aligned float2 at high-res columns 2 ix - 2 ..
    // 2 ix + 3 (a pair is entirely inside or outside the row because Wo is even); consecutive lanes read consecutive pairs,
    // so each of the three loads of a wave is one contiguous 512-byte run and the overlap is served by the vector L1
    // (host guarantees 256 % W == 0: a thread keeps one column, its KT weights live in registers).
    // A workgroup walks ALL bands of its plane (grid.x = 1): the column weights, the index arithmetic (W is a run-time value:
    // every / and % is a ~40-instruction sequence) and the launch overhead are paid once per plane, not once per 16 rows -
    // at 17 KB of input per band they were most of the kernel.
    const int ix = threadIdx.x % p.W, rstep = 256 / p.W, r0 = threadIdx.x / p.W;
    float wx[KT];
#pragma unroll
    for (int k = 0; k < KT; ++k) wx[k] = tap_weight(2 * ix - 2 + k, p.Wo, p.sw, p.ac, p.W, ix);
    const bool v0 = ix > 0, v2 = 2 * ix + 3 < p.Wo;              // first / last pair inside the row (the middle one always is)
    for (int iy0 = blockIdx.x * LB; iy0 < p.H; iy0 += gridDim.x * LB) {
        const int nrow = min(LB, p.H - iy0);
        const int ob = 2 * iy0 - 2;                              // high-res row of band row 0
        const int nb = 2 * nrow + KT - 2;                        // band rows in use
        for (int e = threadIdx.x; e < nrow * KT; e += 256) {
            const int i = e / KT, k = e - i * KT;
            wyT[e] = tap_weight(2 * (iy0 + i) - 2 + k, p.Ho, p.sh, p.ac, p.H, iy0 + i);
        }
#pragma unroll 4
        for (int r = r0; r < nb; r += rstep) {
            const int oy = ob + r;
            float acc = 0.f;
            if (oy >= 0 && oy < p.Ho) {
                float2 a, m, z;
                if (p.hb16) {                                 // bf16 storage: a pair is one aligned dword
                    const unsigned* row = reinterpret_cast<const unsigned*>(reinterpret_cast<const unsigned short*>(p.src) + ((size_t)b * p.CtotS + c) * p.Ho * p.Wo + (size_t)oy * p.Wo + 2 * ix - 2);
                    const unsigned wa = v0 ? row[0] : 0u, wm = row[1], wz = v2 ? row[2] : 0u;
                    a = make_float2(uz::bf16_lo(wa), uz::bf16_hi(wa)); m = make_float2(uz::bf16_lo(wm), uz::bf16_hi(wm)); z = make_float2(uz::bf16_lo(wz), uz::bf16_hi(wz));
                } else {
                    // ONE aligned float2 per lane (high-res columns 2 ix, 2 ix + 1); the pairs to its left and right are the neighbouring
                    // lanes' loads (a wave holds whole rows: 256 % W == 0 and W <= 64) - a third of the load instructions, no re-reads
                    // through the vector L1 (three overlapping float2 per lane: 150 - 160 us on 192 ch 64 x 64 -> 128 x 128, this way 140 - 147 us = 3.4 - 3.6 TB/s;
                    // 128 ch: 98 -> 87 us)
                    m = *reinterpret_cast<const float2*>(s + (size_t)oy * p.Wo + 2 * ix);
                    const float ax = __shfl_up(m.x, 1), ay = __shfl_up(m.y, 1), zx = __shfl_down(m.x, 1), zy = __shfl_down(m.y, 1);
                    a = v0 ? make_float2(ax, ay) : make_float2(0.f, 0.f);
                    z = v2 ? make_float2(zx, zy) : make_float2(0.f, 0.f);
                }
                acc = dot6(wx, a.x, a.y, m.x, m.y, z.x, z.y);
            }
            tx[r * p.W + ix] = acc;
        }
        __syncthreads();
        for (int il = r0; il < nrow; il += rstep) {               // same (row group, column) split as the x pass: no divisions
            const float* col = tx + (2 * il) * p.W + ix;
            const float* wy = wyT + il * KT;
            float acc = 0.f;
#pragma unroll
            for (int k = 0; k < KT; ++k) acc = fmaf(wy[k], col[k * p.W], acc);
            float* dst = d + (size_t)(iy0 + il) * p.W + ix;
            *dst = fold_value(p, mpl, (iy0 + il) * p.W + ix, p.accumulate ? *dst + acc : acc, fo);
        }
        __syncthreads();                                          // tx / wyT are rewritten by the next band
    }
    fold_finish(p, c, b, fo);
}

// Two low-resolution columns per lane: ONE aligned float4 of a high-res row per lane (columns 4 j .. 4 j + 3), so a wave's load is a
// contiguous 1 KB run and there are half as many load instructions per byte; the pair to the left of low-res column 2 j and the pair to
// the right of column 2 j + 1 are the neighbouring lanes' halves (a wave holds whole rows: 64 % (W / 2) == 0).  Same fused
// multiply-add chains (dot6, fmaf) as bilinear_bwd_sep_k: identical bits (tests/test_ops_gpu.py compares the two kernels' outputs).
__global__ __launch_bounds__(256) void bilinear_bwd_sep4_k(const RsP p) {
    __shared__ float wyT[LB * KT];
    __shared__ __attribute__((aligned(8))) float tx[BROWS * (BWMAX / 2)];
    const int c = blockIdx.y, b = blockIdx.z;
    const float* s = p.src + ((size_t)b * p.CtotS + c) * p.Ho * p.Wo;
    float* d = p.dst + ((size_t)b * p.CtotD + c) * p.H * p.W;
    const float* mpl = p.mask ? p.mask + ((size_t)b * p.CtotM + c) * p.H * p.W : nullptr;
    ReluFold fo;
    constexpr int MAXR = (BROWS + 7) / 8;           // band rows per thread: rstep = 256 / (W / 2) >= 8
    const int W2 = p.W >> 1, j = threadIdx.x % W2, rstep = 256 / W2, r0 = threadIdx.x / W2, ix = 2 * j;
    float wa[KT], wb[KT];
#pragma unroll
    for (int k = 0; k < KT; ++k) {
        wa[k] = tap_weight(2 * ix - 2 + k, p.Wo, p.sw, p.ac, p.W, ix);
        wb[k] = tap_weight(2 * ix + k, p.Wo, p.sw, p.ac, p.W, ix + 1);
    }
    const bool v0 = j > 0, v2 = j + 1 < W2;
    for (int iy0 = blockIdx.x * LB; iy0 < p.H; iy0 += gridDim.x * LB) {
        const int nrow = min(LB, p.H - iy0);
        const int ob = 2 * iy0 - 2, nb = 2 * nrow + KT - 2;
        for (int e = threadIdx.x; e < nrow * KT; e += 256) {
            const int i = e / KT, k = e - i * KT;
            wyT[e] = tap_weight(2 * (iy0 + i) - 2 + k, p.Ho, p.sh, p.ac, p.H, iy0 + i);
        }
        // all of a thread's band rows (<= 5: 36 rows over >= 8 row groups) are loaded before the first is used - the shuffles keep
        // the compiler from unrolling a loop around them, and one load in flight per thread was what held the pair kernel at 3 TB/s
        float4 m[MAXR];
#pragma unroll
        for (int u = 0; u < MAXR; ++u) {
            const int r = r0 + u * rstep, oy = ob + r;
            m[u] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (r < nb && oy >= 0 && oy < p.Ho) {
                if (p.hb16) {                              // bf16 storage: the lane's four columns are one aligned 8-byte load
                    const uint2 w2 = *reinterpret_cast<const uint2*>(reinterpret_cast<const unsigned short*>(p.src) + ((size_t)b * p.CtotS + c) * p.Ho * p.Wo + (size_t)oy * p.Wo + 4 * j);
                    m[u] = make_float4(uz::bf16_lo(w2.x), uz::bf16_hi(w2.x), uz::bf16_lo(w2.y), uz::bf16_hi(w2.y));
                } else {
                    m[u] = *reinterpret_cast<const float4*>(s + (size_t)oy * p.Wo + 4 * j);
                }
            }
        }
#pragma unroll
        for (int u = 0; u < MAXR; ++u) {
            const int r = r0 + u * rstep;
            const float pz = __shfl_up(m[u].z, 1), pw = __shfl_up(m[u].w, 1), nx = __shfl_down(m[u].x, 1), ny = __shfl_down(m[u].y, 1);
            const float ax = v0 ? pz : 0.f, ay = v0 ? pw : 0.f, zx = v2 ? nx : 0.f, zy = v2 ? ny : 0.f;
            float2 acc;
            acc.x = dot6(wa, ax, ay, m[u].x, m[u].y, m[u].z, m[u].w);
            acc.y = dot6(wb, m[u].x, m[u].y, m[u].z, m[u].w, zx, zy);
            if (r < nb) *reinterpret_cast<float2*>(tx + r * p.W + ix) = acc;
        }
        __syncthreads();
        for (int il = r0; il < nrow; il += rstep) {
            const float* col = tx + (2 * il) * p.W + ix;
            const float* wy = wyT + il * KT;
            float2 acc = make_float2(0.f, 0.f);
#pragma unroll
            for (int k = 0; k < KT; ++k) {
                const float2 t = *reinterpret_cast<const float2*>(col + k * p.W);
                acc.x = fmaf(wy[k], t.x, acc.x); acc.y = fmaf(wy[k], t.y, acc.y);
            }
            const int q = (iy0 + il) * p.W + ix;
            float2* dst = reinterpret_cast<float2*>(d + q);
            if (p.accumulate) { const float2 o = *dst; acc.x = o.x + acc.x; acc.y = o.y + acc.y; }
            acc.x = fold_value(p, mpl, q, acc.x, fo);
            acc.y = fold_value(p, mpl, q + 1, acc.y, fo);
            *dst = acc;
        }
        __syncthreads();
    }
    fold_finish(p, c, b, fo);
}

// Forward: a workgroup produces OB output rows of one plane from the <= OB / 2 + 2 source rows they touch (staged in LDS by
// float4), four consecutive outputs per thread and one float4 store each; same expression as bilinear_fwd_k.
constexpr int OB = 32, FWMAX = 128, FROWS = OB / 2 + 3;
__global__ __launch_bounds__(256) void bilinear_fwd_band_k(const RsP p) {
    __shared__ __attribute__((aligned(16))) float srow[FROWS * FWMAX];
    __shared__ int xi[2 * FWMAX];
    __shared__ float xl[2 * FWMAX];
    forward_bound(p);
    const int c = blockIdx.y, b = blockIdx.z, oy0 = blockIdx.x * OB;
    const float* s = p.src + ((size_t)b * p.CtotS + c) * p.H * p.W;
    float* d = p.dst + ((size_t)b * p.CtotD + c) * p.Ho * p.Wo;
    const int nout = min(OB, p.Ho - oy0);
    int first, last, ip; float l0, l1;
    src_index(oy0, p.sh, p.ac, p.H, first, ip, l0, l1);
    src_index(oy0 + nout - 1, p.sh, p.ac, p.H, last, ip, l0, l1);
    last += ip;
    for (int e = threadIdx.x; e < p.Wo; e += 256) {
        int i0, jp; float a0, a1;
        src_index(e, p.sw, p.ac, p.W, i0, jp, a0, a1);
        xi[e] = i0 | (jp << 16);
        xl[e] = a1;
    }
    const int w4 = p.W / 4;
    for (int e = threadIdx.x; e < (last - first + 1) * w4; e += 256) {
        const int r = e / w4, q = e - r * w4;
        *reinterpret_cast<float4*>(srow + r * p.W + 4 * q) = *reinterpret_cast<const float4*>(s + (size_t)(first + r) * p.W + 4 * q);
    }
    __syncthreads();
    const int o4 = p.Wo / 4;
    const float ps = pack_scale(p);
    for (int e = threadIdx.x; e < nout * o4; e += 256) {
        const int r = e / o4, q = e - r * o4, oy = oy0 + r;
        int h1, hp; float h0l, h1l;
        src_index(oy, p.sh, p.ac, p.H, h1, hp, h0l, h1l);
        const float* r0 = srow + (h1 - first) * p.W;
        const float* r1 = r0 + hp * p.W;
        float v[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int ox = 4 * q + j, w1 = xi[ox] & 0xFFFF, wp = xi[ox] >> 16;
            const float w1l = xl[ox], w0l = 1.f - w1l;
            v[j] = h0l * (w0l * r0[w1] + w1l * r0[w1 + wp]) + h1l * (w0l * r1[w1] + w1l * r1[w1 + wp]);
        }
        if (p.hb16)
            *reinterpret_cast<uint2*>(reinterpret_cast<unsigned short*>(p.dst) + ((size_t)b * p.CtotD + c) * p.Ho * p.Wo + (size_t)oy * p.Wo + 4 * q) =
                make_uint2(uz::pack_bf16x2(v[0], v[1]), uz::pack_bf16x2(v[2], v[3]));
        else
            *reinterpret_cast<float4*>(d + (size_t)oy * p.Wo + 4 * q) = make_float4(out_word(p, v[0], ps), out_word(p, v[1], ps), out_word(p, v[2], ps), out_word(p, v[3], ps));
    }
}

// ---------------------------------------------------------------- nearest, integer factor
__global__ __launch_bounds__(256) void nearest_fwd_k(const RsP p) {
    const int c = blockIdx.y, b = blockIdx.z;
    const float* s = p.src + ((size_t)b * p.CtotS + c) * p.H * p.W;
    float* d = p.dst + ((size_t)b * p.CtotD + c) * p.Ho * p.Wo;
    const int n = p.Ho * p.Wo;
    for (int q = blockIdx.x * PCH + threadIdx.x; q < min(n, (int)(blockIdx.x + 1) * PCH); q += 256) {
        const int oy = q / p.Wo, ox = q - oy * p.Wo;
        d[q] = s[(oy / p.factor) * p.W + ox / p.factor];
    }
}
__global__ __launch_bounds__(256) void nearest_bwd_k(const RsP p) {   // src = dy (high res), dst = dx (low res)
    const int c = blockIdx.y, b = blockIdx.z;
    const float* s = p.src + ((size_t)b * p.CtotS + c) * p.Ho * p.Wo;
    float* d = p.dst + ((size_t)b * p.CtotD + c) * p.H * p.W;
    const int n = p.H * p.W;
    for (int q = blockIdx.x * PCH + threadIdx.x; q < min(n, (int)(blockIdx.x + 1) * PCH); q += 256) {
        const int iy = q / p.W, ix = q - iy * p.W;
        float acc = 0.f;
        for (int yy = 0; yy < p.factor; ++yy)
            for (int xx = 0; xx < p.factor; ++xx) acc += s[(iy * p.factor + yy) * p.Wo + ix * p.factor + xx];
        d[q] = p.accumulate ? d[q] + acc : acc;
    }
}

// ---------------------------------------------------------------- spatial mean / broadcast
// mean over H (dim 2) then over W (dim 3), as the reference does (probabilistic_unet.py:114-115)
__global__ __launch_bounds__(256) void spatial_mean_fwd_k(const float* __restrict__ x, int C, int Ctot, float* __restrict__ y,
                                                           int H, int W) {
    __shared__ double sm[4];
    const int c = blockIdx.x, b = blockIdx.y;
    const float* s = x + ((size_t)b * Ctot + c) * H * W;
    double v[1] = {0.0};
    for (int i = threadIdx.x; i < H * W; i += 256) v[0] += s[i];
    uz::block_sum_d<1>(v, sm);
    if (threadIdx.x == 0) y[(size_t)b * C + c] = (float)(v[0] / H / W);
}
__global__ __launch_bounds__(256) void spatial_mean_bwd_k(const float* __restrict__ dy, int C, float* __restrict__ dx, int Ctot,
                                                           int H, int W, int accumulate) {
    const int c = blockIdx.x, b = blockIdx.y;
    float* d = dx + ((size_t)b * Ctot + c) * H * W;
    const float v = dy[(size_t)b * C + c] / (float)H / (float)W;
    for (int i = threadIdx.x; i < H * W; i += 256) d[i] = accumulate ? d[i] + v : v;
}
__global__ __launch_bounds__(256) void bcast_fwd_k(const float* __restrict__ z, int L, float* __restrict__ y, int Ctot, int HW) {
    const int l = blockIdx.y, b = blockIdx.z;
    float* d = y + ((size_t)b * Ctot + l) * HW;
    const float v = z[(size_t)b * L + l];
    for (int i = blockIdx.x * 256 + threadIdx.x; i < HW; i += gridDim.x * 256) d[i] = v;
}
__global__ __launch_bounds__(256) void bcast_bwd_k(const float* __restrict__ dy, int Ctot, int L, float* __restrict__ dz, int HW) {
    __shared__ double sm[4];
    const int l = blockIdx.x, b = blockIdx.y;
    const float* s = dy + ((size_t)b * Ctot + l) * HW;
    double v[1] = {0.0};
    for (int i = threadIdx.x; i < HW; i += 256) v[0] += s[i];
    uz::block_sum_d<1>(v, sm);
    if (threadIdx.x == 0) dz[(size_t)b * L + l] = (float)v[0];
}

int check_dims(const char* op, int C, int N, int H, int W) {
    UZ_REQUIRE(C > 0 && N > 0 && H > 0 && W > 0, "%s: empty tensor", op);
    UZ_REQUIRE(N <= 65535 && C <= 65535, "%s: N or C exceeds grid limits", op);
    return 0;
}

}  // namespace

#define RS_LAUNCH(kern, nplane)                                                                      \
    hipLaunchKernelGGL(kern, dim3(uz::ceil_div((nplane), PCH), C, N), dim3(256), 0, uz::S(stream), p); \
    return uz::check_launch(#kern)

extern "C" int uz_avgpool2_fwd(const float* x, int C, int CtotX, float* y, int CtotY, int N, int H, int W,
                               const float* x_amax, float* y_amax, void* stream) {
    return uz_avgpool2_fwd_ex(x, C, CtotX, y, CtotY, N, H, W, x_amax, y_amax, 0, stream);
}
extern "C" int uz_avgpool2_fwd_ex(const float* x, int C, int CtotX, float* y, int CtotY, int N, int H, int W,
                                  const float* x_amax, float* y_amax, int out_packed, void* stream) {
    if (int rc = check_dims("avgpool2_fwd", C, N, H, W)) return rc;
    UZ_REQUIRE(!out_packed || (x_amax && y_amax), "avgpool2_fwd_ex: split storage needs the input's bound and the output's slot");
    RsP p = {}; p.src = x; p.dst = y; p.C = C; p.CtotS = CtotX; p.CtotD = CtotY; p.N = N; p.x_amax = x_amax; p.y_amax = y_amax; p.pack = out_packed; p.flags = uz::dev_flags_ptr();
    p.Ho = H; p.Wo = W; p.H = (H + 1) / 2; p.W = (W + 1) / 2;
    if (H % 2 == 0 && W % 4 == 0 && (reinterpret_cast<uintptr_t>(x) & 15) == 0 && (reinterpret_cast<uintptr_t>(y) & 7) == 0) {
        // (the grid's x extent stays ceil(H W / 4 / PCH) workgroups per plane: PCH / 2 float2 outputs each)
        hipLaunchKernelGGL(avgpool_fwd_v4, dim3(uz::ceil_div(p.H * p.W, PCH), C, N), dim3(256), 0, uz::S(stream), p);
        return uz::check_launch("avgpool_fwd_v4");
    }
    RS_LAUNCH(avgpool_fwd_k, p.H * p.W);
}
static int avgpool2_bwd_impl(const float* dy, int C, int CtotDy, float* dx, int CtotDx, int N, int H, int W, int accumulate,
                             const float* a, int CtotA, double* part, float* dx_amax, void* stream) {
    if (int rc = check_dims("avgpool2_bwd", C, N, H, W)) return rc;
    RsP p = {}; p.src = dy; p.dst = dx; p.C = C; p.CtotS = CtotDy; p.CtotD = CtotDx; p.N = N;
    p.Ho = H; p.Wo = W; p.H = (H + 1) / 2; p.W = (W + 1) / 2; p.accumulate = accumulate;
    p.mask = a; p.CtotM = CtotA; p.part = part; p.m_amax = dx_amax;
    if (H % 2 == 0 && W % 4 == 0 && (reinterpret_cast<uintptr_t>(dx) & 15) == 0 && (reinterpret_cast<uintptr_t>(dy) & 7) == 0 && (reinterpret_cast<uintptr_t>(a) & 15) == 0) {
        // same grid as the scalar kernel (uz_resample_bwd_relu_rows counts its x extent): PCH / 4 float4 of dx per workgroup
        hipLaunchKernelGGL(avgpool_bwd_v4, dim3(uz::ceil_div(H * W, PCH), C, N), dim3(256), 0, uz::S(stream), p);
        return uz::check_launch("avgpool_bwd_v4");
    }
    RS_LAUNCH(avgpool_bwd_k, H * W);
}
extern "C" int uz_avgpool2_bwd(const float* dy, int C, int CtotDy, float* dx, int CtotDx, int N, int H, int W, int accumulate, void* stream) {
    return avgpool2_bwd_impl(dy, C, CtotDy, dx, CtotDx, N, H, W, accumulate, nullptr, 0, nullptr, nullptr, stream);
}
// the grid's x extent of the backward kernels = partial rows per image of the *_bwd_relu entry points (kind 0: avgpool2, H x W = the
// HIGH-resolution plane; kind 1: bilinear2x, H x W = the LOW-resolution plane)
static int resample_bwd_gx(int kind, int C, int N, int H, int W) {
    if (kind == 0) return uz::ceil_div(H * W, PCH);
    if (2 * W <= BWMAX && H >= 4 && 256 % W == 0) return (long long)C * N >= 2048 ? 1 : uz::ceil_div(H, LB);
    return uz::ceil_div(H * W, PCH);
}
extern "C" int uz_resample_bwd_relu_rows(int kind, int C, int N, int H, int W) { return N * resample_bwd_gx(kind, C, N, H, W); }
extern "C" int uz_avgpool2_bwd_relu(const float* dy, int C, int CtotDy, float* dx, int CtotDx, int N, int H, int W, int accumulate,
                                    const float* a, int CtotA, double* partials, float* dx_amax, void* stream) {
    UZ_REQUIRE(a && partials, "avgpool2_bwd_relu: needs the activation and the partial-sum rows");
    return avgpool2_bwd_impl(dy, C, CtotDy, dx, CtotDx, N, H, W, accumulate, a, CtotA, partials, dx_amax, stream);
}
static void bil_scales(RsP& p) {
    if (p.ac) { p.sh = p.Ho > 1 ? (float)(p.H - 1) / (float)(p.Ho - 1) : 0.f; p.sw = p.Wo > 1 ? (float)(p.W - 1) / (float)(p.Wo - 1) : 0.f; }
    else { p.sh = 0.5f; p.sw = 0.5f; }     // scale_factor=2 given: scale = 1/scale_factor
}
extern "C" int uz_bilinear2x_fwd(const float* x, int C, int CtotX, float* y, int CtotY, int N, int H, int W, int align_corners,
                                 const float* x_amax, float* y_amax, void* stream) {
    return uz_bilinear2x_fwd_ex(x, C, CtotX, y, CtotY, N, H, W, align_corners, x_amax, y_amax, 0, stream);
}
extern "C" int uz_bilinear2x_fwd_ex(const float* x, int C, int CtotX, float* y, int CtotY, int N, int H, int W, int align_corners,
                                    const float* x_amax, float* y_amax, int out_packed, void* stream) {
    if (int rc = check_dims("bilinear2x_fwd", C, N, H, W)) return rc;
    UZ_REQUIRE(!out_packed || (x_amax && y_amax), "bilinear2x_fwd_ex: split storage needs the input's bound and the output's slot");
    RsP p = {}; p.src = x; p.dst = y; p.C = C; p.CtotS = CtotX; p.CtotD = CtotY; p.N = N; p.x_amax = x_amax; p.y_amax = y_amax; p.pack = out_packed; p.flags = uz::dev_flags_ptr();
    p.H = H; p.W = W; p.Ho = 2 * H; p.Wo = 2 * W; p.ac = align_corners; bil_scales(p);
    auto al16 = [](const void* q) { return (reinterpret_cast<uintptr_t>(q) & 15) == 0; };
    if (W % 4 == 0 && W <= FWMAX && H >= 4 && al16(x) && al16(y)) {     // W % 4 == 0 keeps every (image, channel) plane of both sides float4-aligned
        hipLaunchKernelGGL(bilinear_fwd_band_k, dim3(uz::ceil_div(p.Ho, OB), C, N), dim3(256), 0, uz::S(stream), p);
        return uz::check_launch("bilinear_fwd_band_k");
    }
    RS_LAUNCH(bilinear_fwd_k, p.Ho * p.Wo);
}
static int bilinear2x_bwd_impl(const float* dy, int C, int CtotDy, float* dx, int CtotDx, int N, int H, int W, int align_corners, int accumulate,
                               const float* a, int CtotA, double* part, float* dx_amax, void* stream);
extern "C" int uz_bilinear2x_bwd(const float* dy, int C, int CtotDy, float* dx, int CtotDx, int N, int H, int W, int align_corners, int accumulate, void* stream) {
    return bilinear2x_bwd_impl(dy, C, CtotDy, dx, CtotDx, N, H, W, align_corners, accumulate, nullptr, 0, nullptr, nullptr, stream);
}
extern "C" int uz_bilinear2x_bwd_relu(const float* dy, int C, int CtotDy, float* dx, int CtotDx, int N, int H, int W, int align_corners, int accumulate,
                                      const float* a, int CtotA, double* partials, float* dx_amax, void* stream) {
    UZ_REQUIRE(a && partials, "bilinear2x_bwd_relu: needs the activation and the partial-sum rows");
    UZ_REQUIRE((reinterpret_cast<uintptr_t>(dy) & 7) == 0, "bilinear2x_bwd_relu: dy must be 8-byte aligned");      // (keeps the kernel choice = uz_resample_bwd_relu_rows)
    return bilinear2x_bwd_impl(dy, C, CtotDy, dx, CtotDx, N, H, W, align_corners, accumulate, a, CtotA, partials, dx_amax, stream);
}
static int bilinear2x_bwd_impl(const float* dy, int C, int CtotDy, float* dx, int CtotDx, int N, int H, int W, int align_corners, int accumulate,
                               const float* a, int CtotA, double* part, float* dx_amax, void* stream) {
    if (int rc = check_dims("bilinear2x_bwd", C, N, H, W)) return rc;
    RsP p = {}; p.src = dy; p.dst = dx; p.C = C; p.CtotS = CtotDy; p.CtotD = CtotDx; p.N = N;
    p.H = H; p.W = W; p.Ho = 2 * H; p.Wo = 2 * W; p.ac = align_corners; p.accumulate = accumulate; bil_scales(p);
    p.mask = a; p.CtotM = CtotA; p.part = part; p.m_amax = dx_amax;
    if (p.Wo <= BWMAX && p.H >= 4 && 256 % p.W == 0 && (reinterpret_cast<uintptr_t>(dy) & 7) == 0) {
        // Wo is even: an 8-byte aligned view keeps every float2 of every row aligned
        // enough planes to fill the chip: one workgroup per plane walks its bands (2.8 -> 3.3 TB/s on 192 ch 64^2 -> 128^2, 3.1 -> 3.75 on 32^2)
        const dim3 grid((long long)C * N >= 2048 ? 1 : uz::ceil_div(H, LB), C, N);
        static const bool quad = !getenv("UZ_BILINEAR_BWD_PAIR");
        if (quad && W >= 32 && 64 % (W / 2) == 0 &&       // 16 x 16 planes: one or two band rows per thread, the pair kernel is 3 us quicker
            (reinterpret_cast<uintptr_t>(dy) & 15) == 0 && (reinterpret_cast<uintptr_t>(dx) & 7) == 0) {
            hipLaunchKernelGGL(bilinear_bwd_sep4_k, grid, dim3(256), 0, uz::S(stream), p);
            return uz::check_launch("bilinear_bwd_sep4_k");
        }
        hipLaunchKernelGGL(bilinear_bwd_sep_k, grid, dim3(256), 0, uz::S(stream), p);
        return uz::check_launch("bilinear_bwd_sep_k");
    }
    RS_LAUNCH(bilinear_bwd_k, H * W);
}
// bf16 STORAGE of the HIGH-resolution side (include/uz_api.h, "bf16 storage"): the in-plane stage of a volume's trilinear
// interpolation writes its output (forward) / reads the incoming gradient (backward) as 2-byte bf16 elements; the low-resolution
// side stays fp32.  Band kernels only: W % 4 == 0, W <= 128 / 2 W <= 128, H >= 4, 256 % W == 0 (backward), 16-byte aligned views.
extern "C" int uz_bilinear2x_fwd_b16(const float* x, int C, int CtotX, void* y, int CtotY, int N, int H, int W, int align_corners, int y_b16, void* stream) {
    if (int rc = check_dims("bilinear2x_fwd_b16", C, N, H, W)) return rc;
    auto al16 = [](const void* q) { return (reinterpret_cast<uintptr_t>(q) & 15) == 0; };
    UZ_REQUIRE(W % 4 == 0 && W <= FWMAX && H >= 4 && al16(x) && al16(y), "bilinear2x_fwd_b16: shape not served by the band kernel (W %% 4 == 0, W <= 128, H >= 4, 16-byte aligned views)");
    RsP p = {}; p.src = x; p.dst = static_cast<float*>(y); p.C = C; p.CtotS = CtotX; p.CtotD = CtotY; p.N = N; p.hb16 = y_b16 != 0;
    p.H = H; p.W = W; p.Ho = 2 * H; p.Wo = 2 * W; p.ac = align_corners; bil_scales(p);
    hipLaunchKernelGGL(bilinear_fwd_band_k, dim3(uz::ceil_div(p.Ho, OB), C, N), dim3(256), 0, uz::S(stream), p);
    return uz::check_launch("bilinear_fwd_band_k");
}
extern "C" int uz_bilinear2x_bwd_b16(const void* dy, int C, int CtotDy, float* dx, int CtotDx, int N, int H, int W, int align_corners, int accumulate, int dy_b16, void* stream) {
    if (int rc = check_dims("bilinear2x_bwd_b16", C, N, H, W)) return rc;
    UZ_REQUIRE(2 * W <= BWMAX && H >= 4 && 256 % W == 0 && (reinterpret_cast<uintptr_t>(dy) & 7) == 0, "bilinear2x_bwd_b16: shape not served by the band kernel (2 W <= 128, H >= 4, 256 %% W == 0, 8-byte aligned dy)");
    RsP p = {}; p.src = static_cast<const float*>(dy); p.dst = dx; p.C = C; p.CtotS = CtotDy; p.CtotD = CtotDx; p.N = N; p.hb16 = dy_b16 != 0;
    p.H = H; p.W = W; p.Ho = 2 * H; p.Wo = 2 * W; p.ac = align_corners; p.accumulate = accumulate; bil_scales(p);
    const dim3 grid((long long)C * N >= 2048 ? 1 : uz::ceil_div(H, LB), C, N);
    if (W >= 32 && 64 % (W / 2) == 0 && (reinterpret_cast<uintptr_t>(dx) & 7) == 0) {        // (8-byte aligned dy: four bf16 columns or, in fp32, checked below)
        if (dy_b16 || (reinterpret_cast<uintptr_t>(dy) & 15) == 0) {
            hipLaunchKernelGGL(bilinear_bwd_sep4_k, grid, dim3(256), 0, uz::S(stream), p);
            return uz::check_launch("bilinear_bwd_sep4_k");
        }
    }
    hipLaunchKernelGGL(bilinear_bwd_sep_k, grid, dim3(256), 0, uz::S(stream), p);
    return uz::check_launch("bilinear_bwd_sep_k");
}
extern "C" int uz_nearest_fwd(const float* x, int C, int CtotX, float* y, int CtotY, int N, int H, int W, int factor, void* stream) {
    if (int rc = check_dims("nearest_fwd", C, N, H, W)) return rc;
    UZ_REQUIRE(factor >= 1, "nearest_fwd: factor must be >= 1");
    RsP p = {}; p.src = x; p.dst = y; p.C = C; p.CtotS = CtotX; p.CtotD = CtotY; p.N = N;
    p.H = H; p.W = W; p.Ho = H * factor; p.Wo = W * factor; p.factor = factor;
    RS_LAUNCH(nearest_fwd_k, p.Ho * p.Wo);
}
namespace {
// Large factors (the deep levels' logits resized to full resolution, phiseg.py:321: 8 x 8 and 16 x 16 children per element): one WAVE per
// low-resolution element - its lanes stride over the children (x fastest: coalesced runs of `factor` floats), butterfly sum, lane 0
// writes - instead of one thread walking 64 - 256 strided values (25 us a launch at the head of the backward tape).  The children are
// added in a different order than nearest_bwd_k adds them: rounding-level difference (sum of f^2 fp32 values).
__global__ __launch_bounds__(256) void nearest_bwd_wave_k(const RsP p) {
    const int c = blockIdx.y, b = blockIdx.z, n = p.H * p.W;
    const int q = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (q >= n) return;
    const float* s = p.src + ((size_t)b * p.CtotS + c) * p.Ho * p.Wo;
    const int iy = q / p.W, ix = q - iy * p.W, f = p.factor, m = f * f;
    float acc = 0.f;
    for (int e = lane; e < m; e += 64) {
        const int yy = e / f, xx = e - yy * f;
        acc += s[(size_t)(iy * f + yy) * p.Wo + ix * f + xx];
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o, 64);
    if (lane == 0) {
        float* d = p.dst + ((size_t)b * p.CtotD + c) * n + q;
        *d = p.accumulate ? *d + acc : acc;
    }
}
}  // namespace
extern "C" int uz_nearest_bwd(const float* dy, int C, int CtotDy, float* dx, int CtotDx, int N, int H, int W, int factor, int accumulate, void* stream) {
    if (int rc = check_dims("nearest_bwd", C, N, H, W)) return rc;
    UZ_REQUIRE(factor >= 1, "nearest_bwd: factor must be >= 1");
    RsP p = {}; p.src = dy; p.dst = dx; p.C = C; p.CtotS = CtotDy; p.CtotD = CtotDx; p.N = N;
    p.H = H; p.W = W; p.Ho = H * factor; p.Wo = W * factor; p.factor = factor; p.accumulate = accumulate;
    if (factor * factor >= 64 && (H * W + 3) / 4 <= 65535) {
        hipLaunchKernelGGL(nearest_bwd_wave_k, dim3((H * W + 3) / 4, C, N), dim3(256), 0, uz::S(stream), p);
        return uz::check_launch("nearest_bwd_wave_k");
    }
    RS_LAUNCH(nearest_bwd_k, H * W);
}
extern "C" int uz_spatial_mean_fwd(const float* x, int C, int CtotX, float* y, int N, int H, int W, void* stream) {
    if (int rc = check_dims("spatial_mean_fwd", C, N, H, W)) return rc;
    hipLaunchKernelGGL(spatial_mean_fwd_k, dim3(C, N), dim3(256), 0, uz::S(stream), x, C, CtotX, y, H, W);
    return uz::check_launch("spatial_mean_fwd_k");
}
extern "C" int uz_spatial_mean_bwd(const float* dy, int C, float* dx, int CtotDx, int N, int H, int W, int accumulate, void* stream) {
    if (int rc = check_dims("spatial_mean_bwd", C, N, H, W)) return rc;
    hipLaunchKernelGGL(spatial_mean_bwd_k, dim3(C, N), dim3(256), 0, uz::S(stream), dy, C, dx, CtotDx, H, W, accumulate);
    return uz::check_launch("spatial_mean_bwd_k");
}
extern "C" int uz_bcast_channels_fwd(const float* z, int L, float* y, int CtotY, int N, int H, int W, void* stream) {
    if (int rc = check_dims("bcast_channels_fwd", L, N, H, W)) return rc;
    hipLaunchKernelGGL(bcast_fwd_k, dim3(uz::ceil_div(H * W, 1024), L, N), dim3(256), 0, uz::S(stream), z, L, y, CtotY, H * W);
    return uz::check_launch("bcast_fwd_k");
}
extern "C" int uz_bcast_channels_bwd(const float* dy, int CtotDy, int L, float* dz, int N, int H, int W, void* stream) {
    if (int rc = check_dims("bcast_channels_bwd", L, N, H, W)) return rc;
    hipLaunchKernelGGL(bcast_bwd_k, dim3(L, N), dim3(256), 0, uz::S(stream), dy, CtotDy, L, dz, H * W);
    return uz::check_launch("bcast_bwd_k");
}
