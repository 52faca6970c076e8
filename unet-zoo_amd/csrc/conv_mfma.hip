// fp32 implicit-GEMM convolution on the CDNA4 matrix cores (v_mfma_f32_32x32x2_f32), im2col-free.
//
// Replaces nn.Conv2d(k=3,s=1,p=1) / nn.Conv2d(k=1) forward (reference torchlayers.py:18, unet.py:25-29,
// phiseg.py:95-96,281-284) and the input-gradient half of aten::convolution_backward.
//
// GEMM view per 3x3 tap:  D[co][pixel] += W_tap[co][ci] * X[ci][pixel + tap offset]
//   M = output channels  -> MFMA A operand, lane (l&31) = co,    k = l>>5
//   N = output pixels    -> MFMA B operand, lane (l&31) = pixel, k = l>>5
//   D: col = lane&31 = pixel (consecutive x -> coalesced 128-B row stores into NCHW),
//      row = (r&3) + 8*(r>>2) + 4*(lane>>5) = co.
// A workgroup (4 waves) owns a tile of 32*MSUB output channels x 256 output pixels (a TB x TH x TW
// block of images/rows/cols, each wave 64 pixels).  Per 8-input-channel chunk it stages, into LDS,
//   - the haloed input patch  [8][TB][(TH+2)][(TW+2)]  (zero filled outside the image = padding),
//   - the weight panel        [tap][8][32*MSUB (+1 pad)], read straight from the PyTorch
//     [Cout][Cin][3][3] parameter (forward) or transposed + tap-flipped (input gradient),
// double buffered: global loads for chunk c+1 are issued before the 36 k-steps of chunk c and
// written to the other buffer afterwards (one barrier per chunk).  All 9 taps reuse the same
// patch through a constant LDS offset, so HBM/L2 sees each input element once per channel tile.
#include <stdlib.h>
#include "uz_common.h"
#include "split_f16.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
constexpr int CK = 8;      // input channels per LDS chunk
constexpr int NSUB = 2;    // 32-pixel sub-tiles per wave

struct ConvP {
    const float* x; const float* w; const float* bias; float* y;
    int N, H, W, HW;
    int Cin, CinTot, Cout, CoutTot;   // GEMM-K channels (input view), GEMM-M channels (output view)
    int wCi, wCo;                     // dims 1 and 0 of the weight tensor (the layer's true Cin / Cout)
    int TW, TH, TB, lgTW, lgTH;
    int tilesX, tilesY, nCoTiles;
    int PW, PSI, PS;                  // patch row stride, patch floats per image, per channel
    int relu, accumulate;
    int ksplit, cps;                  // split-K over input-channel chunks: number of splits, chunks per split
    float* slab;                      // [ksplit][N][Cout][H][W] partial sums when ksplit > 1
    float* y_amax;                    // nullable: atomic max of |y| (bound for a following split-fp16 convolution)
};

template <int KS, int MSUB, int JMAX, bool DGRAD>
#ifndef UZ_SMALL_OCC
#define UZ_SMALL_OCC 0          // experiment builds: minimum waves per SIMD of the small-plane kernels (caps their VGPRs: 4 -> 128, 5 -> 96)
#endif
__global__ __launch_bounds__(256, UZ_SMALL_OCC ? UZ_SMALL_OCC : (JMAX == 2 ? 2 : 1)) void conv_mfma_kernel(const ConvP p) {
    constexpr int KK = KS * KS, HALO = KS / 2;
    // (round 4, measured and dropped: LDS images as [k half][row][4 k] so that one ds_read_b128 per operand feeds four MFMA steps -
    //  a quarter of the LDS instructions - ran 224 -> 128 @ 128 x 128 at 2.42 ms against 2.16 ms and the step at 879 against 893
    //  images/s in fp32 mode: this kernel is not bound by its LDS reads)
    constexpr int COT = 32 * MSUB, COTP = COT + 1;
    constexpr int WSZ = KK * CK * COTP;
    constexpr int WELEMS = COT * CK * KK;
    constexpr int WREGS = (WELEMS + 255) / 256;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    static_assert(WELEMS % 256 == 0, "weight panel must divide evenly over the workgroup");
    constexpr int PSR = JMAX * 256;          // LDS patch row stride (>= PS): every thread stores, no predication
    const int BUF = WSZ + CK * PSR;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, h = lane >> 5;
    const int wid0 = uz::xcd_remap(blockIdx.x, gridDim.x);
    const int ksp = wid0 % p.ksplit, wid = wid0 / p.ksplit;
    const int coT = wid % p.nCoTiles, pixT = wid / p.nCoTiles;
    const int txi = pixT % p.tilesX, t2 = pixT / p.tilesX;
    const int tyi = t2 % p.tilesY, tbi = t2 / p.tilesY;
    const int x0 = txi * p.TW, y0 = tyi * p.TH, b0 = tbi * p.TB;
    const int co0 = coT * COT;

    // ---- staging maps, all chunk-invariant and held in registers.  Global loads are raw buffer loads
    // (uniform resource + per-lane byte offset): the hardware range check returns 0 for padding pixels,
    // for output-channel tiles that overhang Cout and for the K tail, so staging needs no predication,
    // no 64-bit pointer arithmetic and no per-chunk index arithmetic.
    const int nImg = min(p.TB, p.N - b0);
    const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(p.x + (size_t)b0 * p.CinTot * p.HW), 0,
        (unsigned)(((size_t)(nImg - 1) * p.CinTot + p.Cin) * p.HW * sizeof(float)), 0x00020000);
    const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(p.w), 0, (unsigned)((size_t)p.wCo * p.wCi * KK * sizeof(float)), 0x00020000);
    unsigned goff[JMAX], gmask[JMAX];   // byte offset of this thread's patch words inside channel 0; all-ones mask = padding
#pragma unroll
    for (int j = 0; j < JMAX; ++j) {
        const int r = tid + j * 256;
        unsigned g = 0, gm = 0xFFFFFFFFu;
        if (r < p.PS) {
            const int tb = r / p.PSI, rr = r - tb * p.PSI;
            const int py = rr / p.PW, px = rr - py * p.PW;
            const int b = b0 + tb, yy = y0 + py - HALO, xx = x0 + px - HALO;
            if (b < p.N && yy >= 0 && yy < p.H && xx >= 0 && xx < p.W) { g = 4u * (unsigned)(tb * p.CinTot * p.HW + yy * p.W + xx); gm = 0; }
        }
        goff[j] = g; gmask[j] = gm;
    }
    unsigned woff[WREGS];       // byte offset of this thread's weight words for chunk 0 (all ones: channel tile overhang)
    int wlds[WREGS];            // their LDS word index inside the weight panel
#pragma unroll
    for (int i = 0; i < WREGS; ++i) {
        const int e = tid + i * 256;
        unsigned wo = 0xFFFFFFFFu;
        int wl;
        if (!DGRAD) {
            const int co = e / (CK * KK), rem = e - co * (CK * KK), ci = rem / KK, tap = rem - ci * KK;
            wl = (tap * CK + ci) * COTP + co;
            if (co0 + co < p.Cout) wo = 4u * (unsigned)((co0 + co) * p.wCi * KK + rem);
        } else {    // GEMM-K index k walks the weight's dim 0, GEMM-M index m its dim 1; taps flipped
            const int k = e / (COT * KK), rem = e - k * (COT * KK), m = rem / KK, t = rem - m * KK;
            wl = ((KK - 1 - t) * CK + k) * COTP + m;
            if (co0 + m < p.Cout) wo = 4u * (unsigned)((k * p.wCi + co0) * KK + rem);
        }
        woff[i] = wo; wlds[i] = wl;
    }
    const unsigned wstep = 4u * (unsigned)(DGRAD ? CK * p.wCi * KK : CK * KK);      // weight byte step per chunk
    const unsigned xstep = 4u * (unsigned)p.HW;                                      // input byte step per channel

    // ---- per-lane output pixels (B operand columns)
    int poff[NSUB], oidx[NSUB];
    const int npix = p.TB << (p.lgTW + p.lgTH);
#pragma unroll
    for (int n = 0; n < NSUB; ++n) {
        const int pp = wave * (32 * NSUB) + n * 32 + l31;
        const int tx = pp & (p.TW - 1), ty = (pp >> p.lgTW) & (p.TH - 1), tb = pp >> (p.lgTW + p.lgTH);
        const bool v = pp < npix && (b0 + tb) < p.N && (y0 + ty) < p.H && (x0 + tx) < p.W;
        poff[n] = v ? (tb * p.PSI + ty * p.PW + tx + h * PSR) : (h * PSR);
        oidx[n] = v ? ((b0 + tb) * (p.ksplit > 1 ? p.Cout : p.CoutTot) * p.HW + (y0 + ty) * p.W + (x0 + tx)) : -1;
    }

    f32x16 acc[MSUB][NSUB];
#pragma unroll
    for (int m = 0; m < MSUB; ++m)
#pragma unroll
        for (int n = 0; n < NSUB; ++n)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[m][n][r] = 0.f;

    float pr[CK * JMAX], wr[WREGS];

    // Branch-free staging.  K tail (channels >= Cin in the last chunk): the input rows are forced to zero
    // by an all-ones offset; the matching weight words are either in-bounds neighbours (finite, multiplied
    // by zero) or past the end of the tensor (range check returns 0), so they need no mask.
    auto gload = [&](int c) {
        const int ci0 = c * CK;
#pragma unroll
        for (int ci = 0; ci < CK; ++ci) {
            const unsigned cvm = (ci0 + ci) < p.Cin ? 0u : 0xFFFFFFFFu;        // uniform
            const unsigned cbase = (unsigned)(ci0 + ci) * xstep;
#pragma unroll
            for (int j = 0; j < JMAX; ++j)
                pr[ci * JMAX + j] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rx, (goff[j] + cbase) | gmask[j] | cvm, 0, 0));
        }
        const unsigned wbase = (unsigned)c * wstep;
#pragma unroll
        for (int i = 0; i < WREGS; ++i) {
            const unsigned off = woff[i] == 0xFFFFFFFFu ? woff[i] : woff[i] + wbase;
            wr[i] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rw, off, 0, 0));
        }
    };
    auto lstore = [&](int buf) {
        float* Wl = lds + buf * BUF;
        float* Pl = Wl + WSZ;
#pragma unroll
        for (int ci = 0; ci < CK; ++ci)
#pragma unroll
            for (int j = 0; j < JMAX; ++j) Pl[ci * PSR + tid + j * 256] = pr[ci * JMAX + j];
#pragma unroll
        for (int i = 0; i < WREGS; ++i) Wl[wlds[i]] = wr[i];
    };

    // a 64-channel tile whose upper half lies beyond Cout (224 = 3 x 64 + 32: the data gradient of PHiSeg's heaviest layer) skips that
    // half's MFMAs - an eighth of that launch's matrix work was zeros (the split-fp16 kernel has done this since round 3)
    const bool half_tile = MSUB == 2 && p.Cout - co0 <= 32;
    const int nChunksAll = (p.Cin + CK - 1) / CK;
    const int cBeg = ksp * p.cps;
    const int nChunks = min(nChunksAll, cBeg + p.cps);
    gload(cBeg);
    lstore(cBeg & 1);
    __syncthreads();
    for (int c = cBeg; c < nChunks; ++c) {
        const bool more = (c + 1) < nChunks;
        if (more) gload(c + 1);
        const float* Wl = lds + (c & 1) * BUF;
        const float* Pl = Wl + WSZ;
        const float* Al = Wl + h * COTP + l31;
#pragma unroll
        for (int tap = 0; tap < KK; ++tap) {
            const int tapoff = (tap / KS) * p.PW + (tap % KS);
#pragma unroll
            for (int kk = 0; kk < CK / 2; ++kk) {
                float a[MSUB], b[NSUB];
#pragma unroll
                for (int m = 0; m < MSUB; ++m) a[m] = Al[(tap * CK + 2 * kk) * COTP + m * 32];
#pragma unroll
                for (int n = 0; n < NSUB; ++n) b[n] = Pl[poff[n] + 2 * kk * PSR + tapoff];
#pragma unroll
                for (int m = 0; m < MSUB; ++m) {
                    if (MSUB == 2 && m == 1 && half_tile) continue;      // workgroup-uniform: the upper 32 channels of this tile lie beyond Cout
#pragma unroll
                    for (int n = 0; n < NSUB; ++n)
                        acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[m], b[n], acc[m][n], 0, 0, 0);
                }
            }
        }
        if (more) lstore((c + 1) & 1);
        __syncthreads();
    }

    // ---- split-K: raw partial sums to this split's slab; bias / accumulate / ReLU happen in the reduce
    if (p.ksplit > 1) {
        float* base = p.slab + (size_t)ksp * p.N * p.Cout * p.HW;
#pragma unroll
        for (int m = 0; m < MSUB; ++m)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int co = co0 + m * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
                if (co < p.Cout) {
#pragma unroll
                    for (int n = 0; n < NSUB; ++n)
                        if (oidx[n] >= 0) base[(size_t)oidx[n] + (size_t)co * p.HW] = acc[m][n][r];
                }
            }
        return;
    }
    // ---- epilogue: bias, optional accumulate / ReLU, coalesced NCHW stores
    float vmax = 0.f;
#pragma unroll
    for (int m = 0; m < MSUB; ++m) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int co = co0 + m * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
            if (co < p.Cout) {
                const float bv = p.bias ? p.bias[co] : 0.f;
#pragma unroll
                for (int n = 0; n < NSUB; ++n) {
                    if (oidx[n] >= 0) {
                        float* dst = p.y + (size_t)oidx[n] + (size_t)co * p.HW;
                        float v = acc[m][n][r] + bv;
                        if (p.accumulate) v += *dst;
                        if (p.relu) v = fmaxf(v, 0.f);
                        *dst = v;
                        vmax = fmaxf(vmax, fabsf(v));
                    }
                }
            }
        }
    }
    if (p.y_amax) uz::amax_publish(vmax, p.y_amax);
}

// ---------------------------------------------------------------- LDS-free form for the small planes (round 6)
// ONE WAVE per workgroup, no LDS, ~110 VGPRs: such a workgroup needs one wave slot and its registers on ONE SIMD, so it starts beside the
// double-buffered split convolution (2 x 184 VGPRs per SIMD, 148.5 of 160 KB of LDS on every CU) or a BatchNorm sweep, where a
// workgroup of the LDS-staged kernel above (four waves, 35 - 52 KB of LDS) waits until a whole CU drains: in the step the 2 x 2 ... 8 x 8
// levels' convolutions took 3 - 4 x their isolated time (profiles/NOTES_r5.md section 4).  Same arithmetic (v_mfma_f32_32x32x2_f32),
// same split of the channel loop and the same slabs as the kernel above - it replaces only the main launch of a split-K call.
// Wave tile: 32 output channels x 32 flattened pixels (n, y, x).  K walks the input channels two at a time (the MFMA's K = 2: lane
// half h takes channel c + h), nine taps each: the nine weights of a (co, ci) pair are 36 contiguous bytes (two 16-byte loads + one
// dword), the input comes as one dword per tap and lane (consecutive pixels: coalesced rows).  Loads run one group of two channel
// pairs (36 registers) ahead of the MFMAs.
template <bool DGRAD>
__global__ __launch_bounds__(64) void conv_free_kernel(const ConvP p) {
    const int lane = threadIdx.x, l31 = lane & 31, h = lane >> 5;
    const int P = p.N * p.HW, PB = (P + 31) / 32;
    int wid = blockIdx.x;
    const int ksp = wid % p.ksplit; wid /= p.ksplit;
    const int pb = wid % PB, co0 = (wid / PB) * 32;
    const int pidx = pb * 32 + l31;
    const bool pvalid = pidx < P;
    const int pc = pvalid ? pidx : P - 1;
    const int n = pc / p.HW, hw = pc - n * p.HW, yy = hw / p.W, xx = hw - yy * p.W;
    const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(p.x), 0, (unsigned)(((size_t)(p.N - 1) * p.CinTot + p.Cin) * p.HW * sizeof(float)), 0x00020000);
    const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(p.w), 0, (unsigned)((size_t)p.wCo * p.wCi * 9 * sizeof(float)), 0x00020000);
    // byte offsets of the nine taps inside channel 0 of this lane's image; 0x80000000 (+ any channel offset: still past the end of
    // the buffer, no wrap) = padding, the range check returns 0
    unsigned boff[9];
#pragma unroll
    for (int t = 0; t < 9; ++t) {
        const int sy = yy + t / 3 - 1, sx = xx + t % 3 - 1;
        boff[t] = (pvalid && sy >= 0 && sy < p.H && sx >= 0 && sx < p.W) ? 4u * (unsigned)(n * p.CinTot * p.HW + sy * p.W + sx) : 0x80000000u;
    }
    // weights: forward A[co][ci][tap] = w[co][ci][tap]; data gradient A[m][k][tap] = w[k][co0 + m][8 - tap] (k walks dim 0)
    const bool mvalid = (co0 + l31) < p.Cout;
    const unsigned wlane = 36u * (unsigned)(DGRAD ? (co0 + l31) : (co0 + l31) * p.wCi);
    const unsigned wkstep = 36u * (unsigned)(DGRAD ? p.wCi : 1);          // bytes per step of the K channel
    const unsigned xstep = 4u * (unsigned)p.HW;

    struct Grp { float a[2][9]; float b[2][9]; };
    auto gload = [&](Grp& g, int c) __attribute__((always_inline)) {      // channels c .. c + 3 (two pairs; lane half h: c + h, c + 2 + h)
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const int ci = c + 2 * q + h;
            const unsigned cvm = ci < p.Cin ? 0u : 0xFFFFFFFFu;
            const unsigned wo = (mvalid && ci < p.Cin) ? wlane + (unsigned)ci * wkstep : 0x80000000u;     // (+ 32 bytes of immediate offset: no wrap)
            const uz::u32x4 w0 = __builtin_amdgcn_raw_buffer_load_b128(rw, wo, 0, 0);
            const uz::u32x4 w1 = __builtin_amdgcn_raw_buffer_load_b128(rw, wo, 16, 0);
            const unsigned w2 = __builtin_amdgcn_raw_buffer_load_b32(rw, wo, 32, 0);
            const unsigned raw[9] = {w0.x, w0.y, w0.z, w0.w, w1.x, w1.y, w1.z, w1.w, w2};
#pragma unroll
            for (int t = 0; t < 9; ++t) g.a[q][t] = __builtin_bit_cast(float, raw[DGRAD ? 8 - t : t]);
            const unsigned cbase = (unsigned)ci * xstep;
#pragma unroll
            for (int t = 0; t < 9; ++t) g.b[q][t] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rx, (boff[t] + cbase) | cvm, 0, 0));
        }
    };
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    auto mma = [&](const Grp& g) __attribute__((always_inline)) {
#pragma unroll
        for (int q = 0; q < 2; ++q)
#pragma unroll
            for (int t = 0; t < 9; ++t) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(g.a[q][t], g.b[q][t], acc, 0, 0, 0);
    };
    const int nChunksAll = (p.Cin + CK - 1) / CK;
    const int cBeg = ksp * p.cps * CK, cEnd = min(nChunksAll, ksp * p.cps + p.cps) * CK;       // channel range of this split (multiples of 8)
    Grp g0, g1;
    gload(g0, cBeg);
    for (int c = cBeg; c < cEnd; c += 8) {       // two groups per trip: the registers of a group are named at compile time
        gload(g1, c + 4);                        // (c + 4 < cEnd always: the range is a multiple of 8 channels)
        mma(g0);
        gload(g0, c + 8);                        // past the range: channels >= cEnd are either masked (>= Cin) or finite and unused
        mma(g1);
    }
    if (!pvalid) return;
    float* base = p.slab + ((size_t)ksp * p.N + n) * p.Cout * p.HW + hw;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int co = co0 + (r & 3) + 8 * (r >> 2) + 4 * h;
        if (co < p.Cout) base[(size_t)co * p.HW] = acc[r];
    }
}

// y[b,co,p] = (accumulate ? y : 0) + bias[co] + sum_s slab[s][b,co,p]  (fixed order), optional ReLU.
// V = 4: float4 sweep (HW % 4 == 0, 16-byte aligned output view); one 32-bit division pair per vector.
template <int V>
__global__ __launch_bounds__(256) void conv_splitk_reduce(const float* __restrict__ slab, int ksplit, const float* __restrict__ bias,
                                                          float* __restrict__ y, int Cout, int CoutTot, int N, int HW, int relu, int accumulate,
                                                          float* __restrict__ y_amax) {
    const unsigned n = (unsigned)N * Cout * HW;              // < 2^31 floats per slab (checked by the host)
    const unsigned chw = (unsigned)Cout * HW;
    float vmax = 0.f;
    for (unsigned i = (blockIdx.x * 256u + threadIdx.x) * V; i < n; i += gridDim.x * 256u * V) {
        const unsigned b = i / chw, r = i - b * chw, co = r / (unsigned)HW;
        float* dst = y + (size_t)b * CoutTot * HW + r;
        const float bv = bias ? bias[co] : 0.f;
        if (V == 4) {
            float4 s = make_float4(bv, bv, bv, bv);
            for (int k = 0; k < ksplit; ++k) {
                const float4 v = *reinterpret_cast<const float4*>(slab + (size_t)k * n + i);
                s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
            }
            if (accumulate) { const float4 o = *reinterpret_cast<const float4*>(dst); s.x += o.x; s.y += o.y; s.z += o.z; s.w += o.w; }
            if (relu) { s.x = fmaxf(s.x, 0.f); s.y = fmaxf(s.y, 0.f); s.z = fmaxf(s.z, 0.f); s.w = fmaxf(s.w, 0.f); }
            *reinterpret_cast<float4*>(dst) = s;
            vmax = fmaxf(fmaxf(vmax, fmaxf(fabsf(s.x), fabsf(s.y))), fmaxf(fabsf(s.z), fabsf(s.w)));
        } else {
            float s = bv;
            for (int k = 0; k < ksplit; ++k) s += slab[(size_t)k * n + i];
            if (accumulate) s += *dst;
            if (relu) s = fmaxf(s, 0.f);
            *dst = s;
            vmax = fmaxf(vmax, fabsf(s));
        }
    }
    if (y_amax) uz::amax_publish(vmax, y_amax);
}


// ---------------------------------------------------------------- 3x3 forward of the 1..4-channel input layers
// The first convolution of every encoder (image / image + one-hot mask -> 32 channels at full resolution) writes 32x more than
// it reads: a streaming kernel - one thread per output pixel keeps its 9 Cin inputs in registers and walks the output channels,
// weights broadcast from LDS, every store a coalesced row of pixels - instead of an MFMA tile that would be 3 % full.
template <int CIN>
__global__ __launch_bounds__(256) void conv_thin_fwd_kernel(const float* __restrict__ x, int CinTot, const float* __restrict__ w, const float* __restrict__ bias,
                                                             float* __restrict__ y, int Cout, int CoutTot, int H, int W, int relu, float* __restrict__ y_amax) {
    extern __shared__ float wl[];                           // [Cout][CIN * 9] + [Cout] bias
    const int HW = H * W, b = blockIdx.y;
    for (int e = threadIdx.x; e < Cout * CIN * 9; e += 256) wl[e] = w[e];
    for (int e = threadIdx.x; e < Cout; e += 256) wl[Cout * CIN * 9 + e] = bias ? bias[e] : 0.f;
    __syncthreads();
    const int q = blockIdx.x * 256 + threadIdx.x;
    float vmax = 0.f;
    if (q < HW) {
        const int yy = q / W, xx = q - yy * W;
        float xin[CIN * 9];
#pragma unroll
        for (int c = 0; c < CIN; ++c)
#pragma unroll
            for (int t = 0; t < 9; ++t) {
                const int sy = yy + t / 3 - 1, sx = xx + t % 3 - 1;
                xin[c * 9 + t] = (sy >= 0 && sy < H && sx >= 0 && sx < W) ? x[((size_t)b * CinTot + c) * HW + (size_t)sy * W + sx] : 0.f;
            }
        float* yo = y + (size_t)b * CoutTot * HW + q;
        for (int co = 0; co < Cout; ++co) {
            const float* wr = wl + co * CIN * 9;
            float acc = wl[Cout * CIN * 9 + co];
#pragma unroll
            for (int k = 0; k < CIN * 9; ++k) acc += wr[k] * xin[k];
            if (relu) acc = fmaxf(acc, 0.f);
            yo[(size_t)co * HW] = acc;
            vmax = fmaxf(vmax, fabsf(acc));
        }
    }
    if (y_amax) uz::amax_publish(vmax, y_amax);
}
inline bool conv_thin_ok(int Cin, int Cout, int N, int H, int W, int ks) {
    return ks == 3 && Cin <= 4 && Cout <= 256 && (long long)N * H * W >= 64 * 1024 && N <= 65535;
}

struct Geom { int TW, TH, TB, PW, PSI, PS, tilesX, tilesY, tilesB; };

// Split the input-channel loop over several workgroups when the output grid alone cannot fill the
// 256 CUs (deep, small-resolution levels: 2x2 ... 16x16 pixels).  Small cost model: MFMA time of the
// busiest CU (whole rounds of 256 workgroups x chunks per workgroup) + the slab round trip through
// HBM + the reduce launch; every split count is tried because base_grid * ksplit should land just
// under a whole number of rounds.
void pick_split(long long base_grid, int nChunks, int msub, int kk, double out_bytes, int& ksplit, int& cps) {
    ksplit = 1; cps = nChunks;
    if (base_grid >= 512 || nChunks <= 1) return;          // 2 workgroups per CU already: no split
    const double t_chunk = kk * (CK / 2) * msub * NSUB * 64.0 / 2.4e9 / 0.85;     // s per chunk per workgroup, MFMA-bound
    double best = 1e30;
    for (int want = 1; want <= nChunks; ++want) {
        const int c = (nChunks + want - 1) / want, k = (nChunks + c - 1) / c;
        if (k != want) continue;                            // same split as a smaller `want`
        const double rounds = (double)((base_grid * k + 255) / 256);
        const double lonely = base_grid * k <= 256 ? 1.2 : 1.0;     // one workgroup per CU: nothing overlaps its staging and barriers
        const double t = rounds * (c * t_chunk * lonely + 1.5e-6) + (k > 1 ? (k + 2.0) * out_bytes / 3.0e12 + 4e-6 : 0.0);
        if (t < best) { best = t; ksplit = k; cps = c; }
    }
}

// 32- or 64-channel tiles.  64 halves the operand reads per MFMA, but on the 4 x 4 and 2 x 2 planes even one chunk per workgroup
// leaves most CUs idle (6 tiles x 12 chunks = 72 workgroups): there 32-channel tiles double the workgroups and halve the chain.
int pick_msub(int Mc, long long pixel_tiles, int nChunks) {
    if (Mc <= 32) return 1;
    static const bool off = getenv("UZ_MSUB_SMALL") && atoi(getenv("UZ_MSUB_SMALL")) == 0;
    if (off) return 2;
    return pixel_tiles * uz::ceil_div(Mc, 64) * nChunks < 256 ? 1 : 2;
}

Geom pick_geom(int N, int H, int W, int halo) {
    Geom g;
    g.TW = W >= 32 ? 32 : uz::pow2_ceil(W);
    g.TH = uz::pow2_ceil(H);
    if (g.TH > 256 / g.TW) g.TH = 256 / g.TW;
    g.TB = 256 / (g.TW * g.TH);
    if (g.TB > uz::pow2_ceil(N)) g.TB = uz::pow2_ceil(N);
    g.PW = g.TW + 2 * halo;
    g.PSI = (g.TH + 2 * halo) * g.PW;
    while (g.TB > 1 && g.TB * g.PSI > 1024) g.TB >>= 1;
    g.PS = g.TB * g.PSI;
    g.tilesX = uz::ceil_div(W, g.TW);
    g.tilesY = uz::ceil_div(H, g.TH);
    g.tilesB = uz::ceil_div(N, g.TB);
    return g;
}

template <int KS, int MSUB, int JMAX, bool DG>
int launch_one(const ConvP& p, int grid, size_t smem, hipStream_t st) {
    static bool attr_done = false;
    auto kern = conv_mfma_kernel<KS, MSUB, JMAX, DG>;
    if (!attr_done) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess)
            return uz::fail("conv_mfma: cannot raise dynamic LDS limit");
        attr_done = true;
    }
    hipLaunchKernelGGL(kern, dim3(grid), dim3(256), smem, st, p);
    return uz::check_launch("conv_mfma_kernel");
}

template <int KS, bool DG>
int launch_ks(const ConvP& p, int msub, int jmax, int grid, size_t smem, hipStream_t st) {
    if (msub == 1) return jmax == 2 ? launch_one<KS, 1, 2, DG>(p, grid, smem, st) : launch_one<KS, 1, 4, DG>(p, grid, smem, st);
    return jmax == 2 ? launch_one<KS, 2, 2, DG>(p, grid, smem, st) : launch_one<KS, 2, 4, DG>(p, grid, smem, st);
}

}  // namespace

namespace uz {

// ordered sum of the split-K slabs [ksplit][N][Mc][HW] into the output view (+ bias, accumulate, ReLU, magnitude bound)
int splitk_reduce(const float* slab, int ksplit, const float* bias, float* y, int Mc, int McTot, int N, int HW, int relu, int accumulate,
                  float* y_amax, hipStream_t st) {
    const size_t n = (size_t)N * Mc * HW;
    UZ_REQUIRE(n < (1ull << 31), "conv: split-K slab too large");
    const bool v4 = HW % 4 == 0 && (reinterpret_cast<uintptr_t>(y) & 15) == 0 && (reinterpret_cast<uintptr_t>(slab) & 15) == 0;
    int rgrid = (int)((n / (v4 ? 4 : 1) + 255) / 256);
    if (rgrid > 2048) rgrid = 2048;
    if (v4) hipLaunchKernelGGL(conv_splitk_reduce<4>, dim3(rgrid), dim3(256), 0, st, slab, ksplit, bias, y, Mc, McTot, N, HW, relu, accumulate, y_amax);
    else hipLaunchKernelGGL(conv_splitk_reduce<1>, dim3(rgrid), dim3(256), 0, st, slab, ksplit, bias, y, Mc, McTot, N, HW, relu, accumulate, y_amax);
    return check_launch("conv_splitk_reduce");
}

// x: input view (Kc channels), y: output view (Mc channels); w = PyTorch [Cout][Cin][ks][ks] parameter.
size_t conv_workspace(int Kc, int Mc, int N, int H, int W, int ks, int dgrad) {
    if (conv_split_ok(Kc, Mc, N, H, W, ks, dgrad)) return conv_split_workspace(Kc, Mc, N, H, W);
    const Geom g = pick_geom(N, H, W, ks / 2);
    const int cot = 32 * pick_msub(Mc, (long long)g.tilesX * g.tilesY * g.tilesB, ceil_div(Kc, CK));
    int ksplit, cps;
    pick_split((long long)g.tilesX * g.tilesY * g.tilesB * ceil_div(Mc, cot), ceil_div(Kc, CK), cot / 32, ks * ks,
               (double)N * Mc * H * W * sizeof(float), ksplit, cps);
    return ksplit > 1 ? (size_t)ksplit * N * Mc * H * W * sizeof(float) : 0;
}

// slabs_only: stop after the split-K main kernel and leave the partial sums [ksplit][N][Mc][HW] at the start of the workspace
// (the caller's BatchNorm adds them, uz_bn_relu_fwd_slabs); only legal where conv_splitk_parts() > 1.
static int conv_mfma_impl(const float* x, int Kc, int KcTot, const float* w, int wCi, const float* bias,
              float* y, int Mc, int McTot, int N, int H, int W, int ks, int dgrad, int relu, int accumulate,
              const float* x_amax, const float* w_amax, float* y_amax,
              void* workspace, size_t workspace_bytes, const void* packed_w, float* bn_partials, hipStream_t st, bool slabs_only,
              float* slabs_out = nullptr);
int conv_mfma(const float* x, int Kc, int KcTot, const float* w, int wCi, const float* bias,
              float* y, int Mc, int McTot, int N, int H, int W, int ks, int dgrad, int relu, int accumulate,
              const float* x_amax, const float* w_amax, float* y_amax,
              void* workspace, size_t workspace_bytes, const void* packed_w, float* bn_partials, hipStream_t st) {
    return conv_mfma_impl(x, Kc, KcTot, w, wCi, bias, y, Mc, McTot, N, H, W, ks, dgrad, relu, accumulate, x_amax, w_amax, y_amax,
                          workspace, workspace_bytes, packed_w, bn_partials, st, false);
}
// split count of the fp32 forward kernel for this shape when it is given its workspace (1: unsplit, or another kernel family)
int conv_splitk_parts(int Kc, int Mc, int N, int H, int W, int ks) {
    if (conv_split_ok(Kc, Mc, N, H, W, ks, 0)) return 1;
    const Geom g = pick_geom(N, H, W, ks / 2);
    const long long tiles = (long long)g.tilesX * g.tilesY * g.tilesB;
    const int msub = pick_msub(Mc, tiles, ceil_div(Kc, CK));
    int ksplit, cps;
    pick_split(tiles * ceil_div(Mc, 32 * msub), ceil_div(Kc, CK), msub, ks * ks, (double)N * Mc * H * W * sizeof(float), ksplit, cps);
    return ksplit;
}
static int conv_mfma_impl(const float* x, int Kc, int KcTot, const float* w, int wCi, const float* bias,
              float* y, int Mc, int McTot, int N, int H, int W, int ks, int dgrad, int relu, int accumulate,
              const float* x_amax, const float* w_amax, float* y_amax,
              void* workspace, size_t workspace_bytes, const void* packed_w, float* bn_partials, hipStream_t st, bool slabs_only,
              float* slabs_out) {
    UZ_REQUIRE(ks == 1 || ks == 3, "conv: kernel size %d unsupported (1 or 3)", ks);
    UZ_REQUIRE(N > 0 && H > 0 && W > 0 && Kc > 0 && Mc > 0, "conv: empty tensor");
    UZ_REQUIRE(H <= 4096 && W <= 4096, "conv: spatial size too large");
    // large 3x3 layers: split-bf16 matrix pipe (conv_split.hip); its packed weight image lives in the workspace
    UZ_REQUIRE(!slabs_only || !conv_split_ok(Kc, Mc, N, H, W, ks, dgrad), "conv_*_slabs: this shape does not run the split-K fp32 kernel");
    if (conv_split_ok(Kc, Mc, N, H, W, ks, dgrad) && workspace && workspace_bytes >= conv_split_workspace(Kc, Mc, N, H, W))
        return conv_split(x, Kc, KcTot, w, wCi, bias, y, Mc, McTot, N, H, W, dgrad, relu, accumulate, x_amax, w_amax, y_amax, workspace, packed_w, bn_partials, st);
    UZ_REQUIRE(!packed_w, "conv: a pre-packed weight image was supplied for a layer that does not take the split path");
    UZ_REQUIRE(!bn_partials, "conv: fused BatchNorm statistics were requested for a layer that does not take the split path (uz_conv_bn_partials() == 0)");
    const Geom g = pick_geom(N, H, W, ks / 2);
    ConvP p;
    p.x = x; p.w = w; p.bias = bias; p.y = y;
    p.N = N; p.H = H; p.W = W; p.HW = H * W;
    p.Cin = Kc; p.CinTot = KcTot; p.Cout = Mc; p.CoutTot = McTot; p.wCi = wCi; p.wCo = dgrad ? Kc : Mc;
    p.TW = g.TW; p.TH = g.TH; p.TB = g.TB; p.lgTW = ilog2(g.TW); p.lgTH = ilog2(g.TH);
    p.tilesX = g.tilesX; p.tilesY = g.tilesY;
    p.PW = g.PW; p.PSI = g.PSI; p.PS = g.PS;
    p.relu = relu; p.accumulate = accumulate; p.y_amax = y_amax;
    const int msub = pick_msub(Mc, (long long)g.tilesX * g.tilesY * g.tilesB, ceil_div(Kc, CK));
    const int cot = 32 * msub;
    p.nCoTiles = ceil_div(Mc, cot);
    const int jmax = g.PS <= 512 ? 2 : 4;
    const int kk = ks * ks;
    const size_t smem = 2 * (size_t)(kk * CK * (cot + 1) + CK * jmax * 256) * sizeof(float);
    const long long base_grid = (long long)g.tilesX * g.tilesY * g.tilesB * p.nCoTiles;
    pick_split(base_grid, ceil_div(Kc, CK), msub, kk, (double)N * Mc * H * W * sizeof(float), p.ksplit, p.cps);
    const size_t need = p.ksplit > 1 ? (size_t)p.ksplit * N * Mc * H * W * sizeof(float) : 0;
    if (p.ksplit > 1 && !slabs_out && (!workspace || workspace_bytes < need)) { p.ksplit = 1; p.cps = ceil_div(Kc, CK); }   // no workspace: stay unsplit
    UZ_REQUIRE(!slabs_only || p.ksplit > 1, "conv_fwd_slabs: the chunk loop of this shape is not split (uz_conv_splitk_parts() == 1) or the workspace is too small");
    p.slab = slabs_out ? slabs_out : static_cast<float*>(workspace);
    if (p.ksplit > 1) p.y_amax = nullptr;               // split-K: the reduce kernel sees the final values
    const long long grid = base_grid * p.ksplit;
    UZ_REQUIRE(grid < (1ll << 31), "conv: grid too large");
    int rc;
    // small planes, split chunk loop: the LDS-free one-wave form (conv_free_kernel; UZ_CONV_FREE_PX = largest N * H * W it takes).  OFF by
    // default: beside the headline convolution a 192 -> 192 @ 4 x 4 call drops from 83 to 53 us (alone 19 -> 17), but the PHiSeg STEP loses
    // 1.4 % at 2048 and 0.4 % at 512 (two alternations; profiles/NOTES_r6.md section 8): 32-pixel wave tiles re-read the weights once per
    // tile (64 x on an 8 x 8 plane) and its fp32 MFMAs now sit on every SIMD of the chip beside the split kernels' instead of on a few CUs
    static const long long free_px = getenv("UZ_CONV_FREE_PX") ? atoll(getenv("UZ_CONV_FREE_PX")) : 0;
    if (ks == 3 && p.ksplit > 1 && (long long)N * H * W <= free_px && (size_t)N * KcTot * H * W < (1u << 28) && (size_t)p.wCo * wCi * 9 < (1u << 28)) {
        const long long fgrid = (long long)ceil_div(N * H * W, 32) * ceil_div(Mc, 32) * p.ksplit;
        if (dgrad) hipLaunchKernelGGL(conv_free_kernel<true>, dim3((unsigned)fgrid), dim3(64), 0, st, p);
        else hipLaunchKernelGGL(conv_free_kernel<false>, dim3((unsigned)fgrid), dim3(64), 0, st, p);
        rc = check_launch("conv_free_kernel");
    } else
    if (ks == 3) rc = dgrad ? launch_ks<3, true>(p, msub, jmax, (int)grid, smem, st) : launch_ks<3, false>(p, msub, jmax, (int)grid, smem, st);
    else rc = dgrad ? launch_ks<1, true>(p, msub, jmax, (int)grid, smem, st) : launch_ks<1, false>(p, msub, jmax, (int)grid, smem, st);
    if (rc || p.ksplit == 1 || slabs_only) return rc;
    return splitk_reduce(p.slab, p.ksplit, bias, y, Mc, McTot, N, H * W, relu, accumulate, y_amax, st);
}

}  // namespace uz

extern "C" int uz_conv_splitk_parts(int Cin, int Cout, int N, int H, int W, int ks) {
    if ((ks != 1 && ks != 3) || Cin <= 0 || Cout <= 0 || N <= 0 || H <= 0 || W <= 0) return 1;
    if (ks == 1 && uz::conv1x1_small_ok(Cin, Cout)) return 1;
    if (conv_thin_ok(Cin, Cout, N, H, W, ks)) return 1;
    return uz::conv_splitk_parts(Cin, Cout, N, H, W, ks);
}
extern "C" int uz_conv_fwd_slabs(const float* x, int Cin, int CinTot, const float* w, int Cout, int N, int H, int W, int ks,
                                 void* workspace, size_t workspace_bytes, void* stream) {
    UZ_REQUIRE(uz_conv_splitk_parts(Cin, Cout, N, H, W, ks) > 1, "conv_fwd_slabs: uz_conv_splitk_parts() == 1 for this shape");
    return uz::conv_mfma_impl(x, Cin, CinTot, w, Cin, nullptr, nullptr, Cout, Cout, N, H, W, ks, 0, 0, 0, nullptr, nullptr, nullptr,
                              workspace, workspace_bytes, nullptr, nullptr, uz::S(stream), true);
}

// The same for the data gradient of the small planes (round 4): where uz_conv_bwd_splitk_parts() > 1 the call may stop behind its
// main kernel and leave the partial sums [parts][N][Cin][H*W] in slabs_out; the BatchNorm backward of the unit that produced the
// convolution's input adds them itself (uz_bn_relu_bwd_ex, da_slabs) instead of reading dA - one reduction launch less on the
// backward chains of the 8 x 8 ... 2 x 2 levels, same values (slab order).
extern "C" int uz_conv_bwd_splitk_parts(int Cin, int Cout, int N, int H, int W, int ks) {
    if ((ks != 1 && ks != 3) || Cin <= 0 || Cout <= 0 || N <= 0 || H <= 0 || W <= 0) return 1;
    if (ks == 1 && uz::conv1x1_small_ok(Cin, Cout)) return 1;
    if (uz::conv_split_ok(Cout, Cin, N, H, W, ks, 1)) return 1;
    const Geom g = pick_geom(N, H, W, ks / 2);
    const long long tiles = (long long)g.tilesX * g.tilesY * g.tilesB;
    const int msub = pick_msub(Cin, tiles, uz::ceil_div(Cout, CK));
    int ksplit, cps;
    pick_split(tiles * uz::ceil_div(Cin, 32 * msub), uz::ceil_div(Cout, CK), msub, ks * ks, (double)N * Cin * H * W * sizeof(float), ksplit, cps);
    return ksplit;
}
extern "C" int uz_conv_bwd_data_slabs(const float* dy, int Cout, int CoutTot, const float* w, int Cin, int N, int H, int W, int ks,
                                      float* slabs_out, void* stream) {
    UZ_REQUIRE(slabs_out && uz_conv_bwd_splitk_parts(Cin, Cout, N, H, W, ks) > 1, "conv_bwd_data_slabs: uz_conv_bwd_splitk_parts() == 1 for this shape");
    return uz::conv_mfma_impl(dy, Cout, CoutTot, w, Cin, nullptr, nullptr, Cin, Cin, N, H, W, ks, 1, 0, 0, nullptr, nullptr, nullptr,
                              nullptr, 0, nullptr, nullptr, uz::S(stream), true, slabs_out);
}

extern "C" size_t uz_conv_workspace(int Cin, int Cout, int N, int H, int W, int ks) {
    const size_t a = uz::conv_workspace(Cin, Cout, N, H, W, ks, 0), b = uz::conv_workspace(Cout, Cin, N, H, W, ks, 1);
    return a > b ? a : b;
}

extern "C" int uz_conv_bn_partials(int Cin, int Cout, int N, int H, int W, int ks) {
    if (ks != 3 || conv_thin_ok(Cin, Cout, N, H, W, ks)) return 0;
    return uz::conv_split_bn_partials(Cin, Cout, N, H, W);
}
extern "C" int uz_conv_fwd_bnstats(const float* x, int Cin, int CinTot, const float* w, const float* bias,
                                   float* y, int Cout, int CoutTot, int N, int H, int W, int ks, int relu,
                                   const float* x_amax, const float* w_amax, float* y_amax,
                                   void* workspace, size_t workspace_bytes, const void* packed_w, float* bn_partials, void* stream) {
    UZ_REQUIRE(!bn_partials || uz_conv_bn_partials(Cin, Cout, N, H, W, ks) > 0, "conv_fwd_bnstats: this shape writes no fused statistics (uz_conv_bn_partials() == 0)");
    if (ks == 1 && !relu && uz::conv1x1_small_ok(Cin, Cout)) {        // 2..8-output heads: streaming VALU kernel
        const int rc = uz::conv1x1_small_fwd(x, Cin, CinTot, w, bias, y, Cout, CoutTot, N, H, W, uz::S(stream));
        if (rc != -2) return rc;
    }
    if (conv_thin_ok(Cin, Cout, N, H, W, ks) && !packed_w) {           // 1..4 input channels at full resolution: streaming kernel
        const dim3 grid(uz::ceil_div(H * W, 256), N);
        const size_t smem = (size_t)(Cout * Cin * 9 + Cout) * sizeof(float);
#define UZ_THINF(C_) hipLaunchKernelGGL(conv_thin_fwd_kernel<C_>, grid, dim3(256), smem, uz::S(stream), x, CinTot, w, bias, y, Cout, CoutTot, H, W, relu, y_amax)
        if (Cin == 1) UZ_THINF(1); else if (Cin == 2) UZ_THINF(2); else if (Cin == 3) UZ_THINF(3); else UZ_THINF(4);
#undef UZ_THINF
        return uz::check_launch("conv_thin_fwd_kernel");
    }
    return uz::conv_mfma(x, Cin, CinTot, w, Cin, bias, y, Cout, CoutTot, N, H, W, ks, 0, relu, 0, x_amax, w_amax, y_amax, workspace, workspace_bytes, packed_w, bn_partials, uz::S(stream));
}
extern "C" int uz_conv_fwd_packed(const float* x, int Cin, int CinTot, const float* w, const float* bias,
                                  float* y, int Cout, int CoutTot, int N, int H, int W, int ks, int relu,
                                  const float* x_amax, const float* w_amax, float* y_amax,
                                  void* workspace, size_t workspace_bytes, const void* packed_w, void* stream) {
    return uz_conv_fwd_bnstats(x, Cin, CinTot, w, bias, y, Cout, CoutTot, N, H, W, ks, relu, x_amax, w_amax, y_amax, workspace, workspace_bytes, packed_w, nullptr, stream);
}
extern "C" int uz_conv_fwd(const float* x, int Cin, int CinTot, const float* w, const float* bias,
                           float* y, int Cout, int CoutTot, int N, int H, int W, int ks, int relu,
                           const float* x_amax, const float* w_amax, float* y_amax,
                           void* workspace, size_t workspace_bytes, void* stream) {
    return uz_conv_fwd_packed(x, Cin, CinTot, w, bias, y, Cout, CoutTot, N, H, W, ks, relu, x_amax, w_amax, y_amax, workspace, workspace_bytes, nullptr, stream);
}

// Round 4 (include/uz_api.h, "split storage + folded BatchNorm backward").  Without options these are uz_conv_fwd_bnstats /
// uz_conv_bwd_data_packed; with them the call must take the split-fp16 path (uz_conv_route() == 1) - nothing else reads split storage.
extern "C" int uz_conv_fwd_ex(const float* x, int Cin, int CinTot, const float* w, const float* bias,
                              float* y, int Cout, int CoutTot, int N, int H, int W, int ks, int relu,
                              const float* x_amax, const float* w_amax, float* y_amax,
                              void* workspace, size_t workspace_bytes, const void* packed_w, float* bn_partials,
                              int x_packed, const float* x_amax2, int seg_channels, void* stream) {
    if (!x_packed)
        return uz_conv_fwd_bnstats(x, Cin, CinTot, w, bias, y, Cout, CoutTot, N, H, W, ks, relu, x_amax, w_amax, y_amax, workspace, workspace_bytes, packed_w, bn_partials, stream);
    UZ_REQUIRE(ks == 3 && uz_conv_route(0, Cin, Cout, N, H, W, ks) == 1 && uz::conv_np() == 2, "conv_fwd_ex: input in split storage, but this shape / math mode does not take the split-fp16 path");
    UZ_REQUIRE(workspace && workspace_bytes >= uz::conv_split_workspace(Cin, Cout, N, H, W), "conv_fwd_ex: workspace too small");
    UZ_REQUIRE(!bn_partials || uz_conv_bn_partials(Cin, Cout, N, H, W, ks) > 0, "conv_fwd_ex: this shape writes no fused statistics (uz_conv_bn_partials() == 0)");
    uz::SplitOpts o;
    o.x_packed = 1; o.x_amax2 = x_amax2; o.seg_channels = seg_channels;
    return uz::conv_split_ex(x, Cin, CinTot, w, Cin, bias, y, Cout, CoutTot, N, H, W, 0, relu, 0, x_amax, w_amax, y_amax, workspace, packed_w, bn_partials, uz::S(stream), o);
}
// uz_conv_fwd_ex for a layer that follows a Conv -> BatchNorm -> ReLU unit (torchlayers.py:18-21), reading that unit's PRE-normalisation
// output y_prev and its statistics table bn_save ([4][Cin]: mean, rstd, alpha = gamma rstd, beta' = beta - mean alpha, as uz_bn_relu_fwd_ex
// writes it): the staging computes a = max(alpha y_prev + beta', 0) (bn_relu; zero padding stays zero) and splits it with the scale of
// a_amax, the bound of the APPLIED activation - the same values the unit's apply pass would have stored, without waiting for that pass.
// Split-fp16 path only (uz_conv_route() == 1, two-piece mode).
extern "C" int uz_conv_fwd_bn_ex(const float* y_prev, int Cin, int CinTot, const float* bn_save, int bn_relu,
                                 const float* w, const float* bias, float* y, int Cout, int CoutTot, int N, int H, int W, int ks,
                                 const float* a_amax, const float* w_amax, float* y_amax,
                                 void* workspace, size_t workspace_bytes, const void* packed_w, float* bn_partials, void* stream) {
    UZ_REQUIRE(y_prev && bn_save && a_amax, "conv_fwd_bn_ex: needs the producer's output, its statistics table and the activation's bound");
    UZ_REQUIRE(ks == 3 && uz_conv_route(0, Cin, Cout, N, H, W, ks) == 1 && uz::conv_np() == 2, "conv_fwd_bn_ex: this shape / math mode does not take the split-fp16 path");
    UZ_REQUIRE(workspace && workspace_bytes >= uz::conv_split_workspace(Cin, Cout, N, H, W), "conv_fwd_bn_ex: workspace too small");
    UZ_REQUIRE(!bn_partials || uz_conv_bn_partials(Cin, Cout, N, H, W, ks) > 0, "conv_fwd_bn_ex: this shape writes no fused statistics (uz_conv_bn_partials() == 0)");
    uz::SplitOpts o;
    o.aff = bn_save; o.aff_relu = bn_relu;
    return uz::conv_split_ex(y_prev, Cin, CinTot, w, Cin, bias, y, Cout, CoutTot, N, H, W, 0, 0, 0, a_amax, w_amax, y_amax, workspace, packed_w, bn_partials, uz::S(stream), o);
}
extern "C" int uz_conv_bwd_data_ex(const float* dy, int Cout, int CoutTot, const float* w, float* dx, int Cin, int CinTot,
                                   int N, int H, int W, int ks, int accumulate, const float* dy_amax, const float* w_amax,
                                   void* workspace, size_t workspace_bytes, const void* packed_w, int dy_packed,
                                   const float* bn_y, int bn_yCtot, const float* bn_save, int bn_relu, float* bn_partials, void* stream) {
    if (!dy_packed && !bn_y)
        return uz_conv_bwd_data_packed(dy, Cout, CoutTot, w, dx, Cin, CinTot, N, H, W, ks, accumulate, dy_amax, w_amax, workspace, workspace_bytes, packed_w, stream);
    UZ_REQUIRE(ks == 3 && uz_conv_route(1, Cin, Cout, N, H, W, ks) == 1 && uz::conv_np() == 2, "conv_bwd_data_ex: split storage / folded reduction, but this shape / math mode does not take the split-fp16 path");
    UZ_REQUIRE(workspace && workspace_bytes >= uz::conv_split_workspace(Cout, Cin, N, H, W), "conv_bwd_data_ex: workspace too small");
    uz::SplitOpts o;
    o.x_packed = dy_packed;
    if (bn_y) {
        UZ_REQUIRE(bn_save && bn_partials && !accumulate && uz_conv_bwd_relu_partials(Cin, Cout, N, H, W, ks) > 0,
                   "conv_bwd_data_ex: the folded BatchNorm reduction needs the statistics table, the partial rows, an overwriting unsplit launch");
        o.mk = 2; o.mask = bn_y; o.maskCtot = bn_yCtot; o.mk_save = bn_save; o.mk_relu = bn_relu;
    }
    return uz::conv_split_ex(dy, Cout, CoutTot, w, Cin, nullptr, dx, Cin, CinTot, N, H, W, 1, 0, accumulate, dy_amax, w_amax, nullptr, workspace, packed_w,
                             bn_y ? bn_partials : nullptr, uz::S(stream), o);
}
// bf16 STORAGE (include/uz_api.h, "bf16 storage"; BASELINE config 5): the input and / or the output tensor hold 2-byte bf16 elements
// (same NCHW shape).  Single-piece bf16 mode (uz_set_conv_math(3)), 3x3 shapes on the matrix-pipe path with planes wider than 32;
// fp32 accumulation, values rounded to nearest even when written; the fused BatchNorm statistics are those of the STORED values.
extern "C" int uz_conv_fwd_b16(const void* x, int Cin, int CinTot, const float* w, const float* bias,
                               void* y, int Cout, int CoutTot, int N, int H, int W, int ks,
                               void* workspace, size_t workspace_bytes, const void* packed_w, float* bn_partials,
                               int x_b16, int y_b16, void* stream) {
    UZ_REQUIRE(ks == 3 && uz::conv_np() == 1 && uz_conv_route(0, Cin, Cout, N, H, W, ks) == 1,
               "conv_fwd_b16: bf16 storage needs uz_set_conv_math(3) and a 3x3 shape on the matrix-pipe path (unsplit chunk loop)");
    UZ_REQUIRE(workspace && workspace_bytes >= uz::conv_split_workspace(Cin, Cout, N, H, W), "conv_fwd_b16: workspace too small");
    UZ_REQUIRE(!bn_partials || uz_conv_bn_partials(Cin, Cout, N, H, W, ks) > 0, "conv_fwd_b16: this shape writes no fused statistics (uz_conv_bn_partials() == 0)");
    uz::SplitOpts o;
    o.x_b16 = x_b16 != 0; o.y_b16 = y_b16 != 0;
    return uz::conv_split_ex(static_cast<const float*>(x), Cin, CinTot, w, Cin, bias, static_cast<float*>(y), Cout, CoutTot, N, H, W, 0, 0, 0,
                             nullptr, nullptr, nullptr, workspace, packed_w, bn_partials, uz::S(stream), o);
}
extern "C" int uz_conv_bwd_data_b16(const void* dy, int Cout, int CoutTot, const float* w, void* dx, int Cin, int CinTot,
                                    int N, int H, int W, int ks, int accumulate,
                                    void* workspace, size_t workspace_bytes, const void* packed_w, int dy_b16, int dx_b16, void* stream) {
    UZ_REQUIRE(ks == 3 && uz::conv_np() == 1 && uz_conv_route(1, Cin, Cout, N, H, W, ks) == 1,
               "conv_bwd_data_b16: bf16 storage needs uz_set_conv_math(3) and a 3x3 shape on the matrix-pipe path (unsplit chunk loop)");
    UZ_REQUIRE(workspace && workspace_bytes >= uz::conv_split_workspace(Cout, Cin, N, H, W), "conv_bwd_data_b16: workspace too small");
    uz::SplitOpts o;
    o.x_b16 = dy_b16 != 0; o.y_b16 = dx_b16 != 0;
    return uz::conv_split_ex(static_cast<const float*>(dy), Cout, CoutTot, w, Cin, nullptr, static_cast<float*>(dx), Cin, CinTot, N, H, W, 1, 0, accumulate,
                             nullptr, nullptr, nullptr, workspace, packed_w, nullptr, uz::S(stream), o);
}
extern "C" int uz_conv_bwd_data_packed(const float* dy, int Cout, int CoutTot, const float* w,
                                       float* dx, int Cin, int CinTot, int N, int H, int W, int ks, int accumulate,
                                       const float* dy_amax, const float* w_amax,
                                       void* workspace, size_t workspace_bytes, const void* packed_w, void* stream) {
    if (ks == 1 && uz::conv1x1_small_ok(Cin, Cout)) {
        const int rc = uz::conv1x1_small_bwd_data(dy, Cout, CoutTot, w, dx, Cin, CinTot, N, H, W, accumulate, uz::S(stream));
        if (rc != -2) return rc;
    }
    return uz::conv_mfma(dy, Cout, CoutTot, w, Cin, nullptr, dx, Cin, CinTot, N, H, W, ks, 1, 0, accumulate, dy_amax, w_amax, nullptr, workspace, workspace_bytes, packed_w, nullptr, uz::S(stream));
}
// ---- ReLU backward folded into the data gradient that produces dA (vanilla U-Net blocks: Conv -> ReLU -> Conv, unet.py:25-30)
namespace {
// dbias[c] = sum over the partial rows of partials[row][c].x (ordered, fp64), one wave per channel
__global__ __launch_bounds__(256) void chan_sum_partials_k(const float* __restrict__ part, int nrows, int C, float* __restrict__ out) {
    const int c = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (c >= C) return;
    double s = 0.0;
    for (int i = lane; i < nrows; i += 64) s += (double)part[((size_t)i * C + c) * 4];
    s = uz::wave_sum_d(s);
    if (lane == 0) out[c] = (float)s;
}
}  // namespace
namespace {
// all bias gradients of a tape in ONE launch: table rows {partials, out, n_rows, C, is_double}; blockIdx.y = row, one wave per channel
__global__ __launch_bounds__(256) void chan_sum_table_k(const long long* __restrict__ table) {
    const long long* e = table + 5 * blockIdx.y;
    const int nrows = (int)e[2], C = (int)e[3];
    const int c = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (c >= C) return;
    double s = 0.0;
    if (e[4]) {
        const double* part = reinterpret_cast<const double*>(e[0]);
        for (int i = lane; i < nrows; i += 64) s += part[(size_t)i * C + c];
    } else {
        const float* part = reinterpret_cast<const float*>(e[0]);
        for (int i = lane; i < nrows; i += 64) s += (double)part[((size_t)i * C + c) * 4];
    }
    s = uz::wave_sum_d(s);
    if (lane == 0) reinterpret_cast<float*>(e[1])[c] = (float)s;
}
}  // namespace
extern "C" int uz_chan_sum_table(const int64_t* table, int n_entries, int max_channels, void* stream) {
    UZ_REQUIRE(table && n_entries > 0 && n_entries <= 65535 && max_channels > 0, "chan_sum_table: bad arguments");
    hipLaunchKernelGGL(chan_sum_table_k, dim3(uz::ceil_div(max_channels, 4), n_entries), dim3(256), 0, uz::S(stream), reinterpret_cast<const long long*>(table));
    return uz::check_launch("chan_sum_table_k");
}
extern "C" int uz_conv_bwd_relu_partials(int Cin, int Cout, int N, int H, int W, int ks) {
    if (ks != 3 || !uz::conv_split_ok(Cout, Cin, N, H, W, ks, 1)) return 0;
    return uz::conv_split_bn_partials(Cout, Cin, N, H, W);          // 0 when the chunk loop is split over workgroups
}
extern "C" int uz_conv_bwd_data_relu(const float* dy, int Cout, int CoutTot, const float* w, float* dx, int Cin, int CinTot,
                                     int N, int H, int W, int ks, int accumulate, const float* dy_amax, const float* w_amax,
                                     void* workspace, size_t workspace_bytes, const void* packed_w,
                                     const float* a, int aCtot, float* partials, float* dx_amax, void* stream) {
    UZ_REQUIRE(uz_conv_bwd_relu_partials(Cin, Cout, N, H, W, ks) > 0, "conv_bwd_data_relu: this shape does not support the folded ReLU backward (uz_conv_bwd_relu_partials() == 0)");
    UZ_REQUIRE(workspace && workspace_bytes >= uz::conv_split_workspace(Cout, Cin, N, H, W), "conv_bwd_data_relu: workspace too small");
    return uz::conv_split_dgrad_relu(dy, Cout, CoutTot, w, Cin, dx, Cin, CinTot, N, H, W, accumulate, dy_amax, w_amax, dx_amax, workspace, packed_w,
                                     a, aCtot, partials, uz::S(stream));
}
extern "C" int uz_chan_sum_partials(const float* partials, int n_rows, int C, float* out, void* stream) {
    UZ_REQUIRE(partials && out && n_rows > 0 && C > 0, "chan_sum_partials: bad arguments");
    hipLaunchKernelGGL(chan_sum_partials_k, dim3(uz::ceil_div(C, 4)), dim3(256), 0, uz::S(stream), partials, n_rows, C, out);
    return uz::check_launch("chan_sum_partials_k");
}
extern "C" int uz_conv_bwd_data(const float* dy, int Cout, int CoutTot, const float* w,
                                float* dx, int Cin, int CinTot, int N, int H, int W, int ks, int accumulate,
                                const float* dy_amax, const float* w_amax,
                                void* workspace, size_t workspace_bytes, void* stream) {
    return uz_conv_bwd_data_packed(dy, Cout, CoutTot, w, dx, Cin, CinTot, N, H, W, ks, accumulate, dy_amax, w_amax, workspace, workspace_bytes, nullptr, stream);
}
