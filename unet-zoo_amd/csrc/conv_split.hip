// 3x3 convolution (forward and data gradient) on the fp16 matrix pipe with fp32-accurate results.
//
// MI355X runs v_mfma_f32_32x32x16_f16 at 16x the FLOP rate of v_mfma_f32_32x32x2_f32.  Every fp32 operand is scaled by a
// power of two (from an upper bound of its tensor's magnitudes, so that nothing overflows fp16) and split into TWO fp16
// pieces, a*s = a1 + a2 (11 + 11 significand bits, remainder <= 2^-22 |a|); the product a*b is accumulated in fp32 from
// the three piece products a2b1, a1b2, a1b1 (fp16 x fp16 is exact in fp32; the dropped a2b2 is <= 2^-22 |ab|), and the
// epilogue undoes the scales exactly.  It is a 22-bit emulation, not fp32: measured against an fp64 evaluation one layer's error
// is 0.5 - 0.8x that of the fp32-MFMA kernel (three partial sums per 16-deep k-step round less often than the sixteen of a
// k-ordered fmaf chain; gate: <= 2x, tests/test_full_configs_gpu.py), and end to end at batch 32 the gradients' median error is
// 1.6x the fp32 reference's own (tests/test_phiseg_gpu.py; with UZ_CONV_MATH=f32: equal) - logits within 1e-4, argmax bit-equal.
// Three fp16 MFMAs per 16-deep k-step cost 96 cycles against 512 for the eight fp32 MFMAs they replace.
//
// Tile: 64 (32) output channels x 512 pixels (16 rows x 32 columns of one image) per 512-thread
// workgroup.  Per 16-channel chunk the haloed 18 x 34 patch and the 9-tap weight panel are fetched to registers (raw
// buffer loads, hardware range check = padding), split into the two fp16 planes and written to LDS in MFMA fragment order
// ([plane][tap][co][16 ci] and [plane][pixel][16 ci]: every fragment is one ds_read_b128); the next
// chunk's global loads are in flight during the 108 (54) MFMAs of the current one.
// dgrad is the same kernel with the weight panel gathered transposed and tap-flipped.
#include <stdlib.h>
#include <string.h>
#include "uz_common.h"
#include <type_traits>
#include "split_f16.h"

namespace {

using uz::f32x16; using uz::f32x4; using uz::f16x8; using uz::u32x4; using uz::split2;


constexpr int CK = 16, NSUB = 2, KK = 9, TH = 16;
#ifndef UZ_EXP_PRODUCTS
#define UZ_EXP_PRODUCTS 3
#endif
// NP = planes per operand: 2 = fp32-accurate split (two fp16 pieces, three products), 1 = ONE bf16 piece and one product per
// (a, b) pair - bf16 arithmetic with fp32 accumulation (UZ_CONV_MATH=bf16: BASELINE config 5, PHiSeg3D "bf16"); bf16 keeps fp32's
// exponent range, so that mode needs neither operand scales nor magnitude bounds.
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
template <int NP> __device__ __forceinline__ void pieces(float v0, float v1, unsigned (&out)[NP]) {
    if constexpr (NP == 2) split2(v0, v1, out[0], out[1]);
    else out[0] = __builtin_bit_cast(unsigned, __builtin_convertvector(uz::f32x2{v0, v1}, bf16x2));      // v_cvt_pk_bf16_f32, round to nearest even
}
template <int NP> __device__ __forceinline__ f32x16 mma(f32x16 t, const u32x4 (&a)[NP], const u32x4 (&b)[NP]) {
    if constexpr (NP == 2) {                       // smallest products first
        // (UZ_EXP_PRODUCTS = 2 / 1: timing-only experiment builds that drop one / two of the three piece products - what the step would
        //  cost with fewer matrix products per output, everything else unchanged; results lose accuracy, never used in a product build)
        if constexpr (UZ_EXP_PRODUCTS >= 3) t = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a[1]), __builtin_bit_cast(f16x8, b[0]), t, 0, 0, 0);
        if constexpr (UZ_EXP_PRODUCTS >= 2) t = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a[0]), __builtin_bit_cast(f16x8, b[1]), t, 0, 0, 0);
        t = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a[0]), __builtin_bit_cast(f16x8, b[0]), t, 0, 0, 0);
    } else {
        t = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a[0]), __builtin_bit_cast(bf16x8, b[0]), t, 0, 0, 0);
    }
    return t;
}
// Two tile geometries share the kernel: 512 threads on 16 x 32 pixels (planes at least 32 wide) and 256 threads
// on 16 x 16 pixels (one 16 x 16 plane per workgroup, 32 output channels, 60 KB of LDS -> two workgroups per CU).
template <int NTv, int TWv> struct Geo {
    static constexpr int NT = NTv, TW = TWv, PW = TWv + 2, PSI = (TH + 2) * PW;      // 612 / 324 patch pixels
    static constexpr int EXTRA = PSI - NT;                                            // rows beyond one per thread: 100 / 68
    static constexpr int G = EXTRA * 4 <= NT ? 4 : 2;                                 // threads sharing one extra row
    static constexpr int CE = CK / G;                                                 // channels of an extra row per thread
    static constexpr int PSR = (NT + NT / G + 15) / 16 * 16;                          // padded patch rows: 640 / 384
    static_assert(EXTRA > 0 && EXTRA <= NT / G, "extra patch rows must fit the shared-row scheme");
    static_assert(NT / 64 * 32 * NSUB == TH * TW, "one wave covers 64 pixels");
};

struct SP {
    const float* x; const char* wp; const float* bias; float* y;     // wp: packed split weights (pack_weights_kernel)
    int N, H, W, HW;
    int Cin, CinTot, Cout, CoutTot;   // GEMM-K channels (input view), GEMM-M channels (output view)
    int tilesX, tilesY, nCoTiles, nChunks;
    int relu, accumulate;
    const float* x_amax; const float* w_amax;     // device scalars: upper bounds of |x| and |w| (never null here)
    float* y_amax;                                // nullable: atomic max of |y| (bound for the next layer's split)
    int kSplit, cps;                              // split-K: the chunk loop is shared out over kSplit workgroups, cps chunks each
    const float* mask; int maskCtot;              // data gradient with the producer's ReLU mask folded in: y = (mask > 0) ? y : 0 (view of Cout channels)
    // MK == 2: the producer is a Conv -> BatchNorm -> ReLU unit (torchlayers.py:18-21); mask = its pre-normalisation output, mk_save =
    // its [4][Cout] {mean, rstd, alpha, beta'} table: the epilogue masks with alpha y + beta' > 0 and leaves the unit's backward
    // reduction {sum dz, sum dz x_hat, max |dz|, max |x_hat|} per (tile, channel) in bnpart
    const float* mk_save; int mk_relu;
    // XPK: the first segc chunks of the input were scaled from x_amax, the rest from x_amax2 (a concat buffer's two producers);
    // segc <= 0 or >= nChunks: one bound
    const float* x_amax2; int segc;
    float* slab;                                  // [kSplit][N][Cout][HW] partial sums (kSplit > 1), summed in order by splitk_reduce
    long long* stamps;                            // diagnostics (uz_debug_stamps): 8 cycle stamps per workgroup, normally null
    int* flags; int flag_bit;                     // device flag word (bound violations), nullable; the bit this launch raises: activation (forward) or gradient (data gradient)
    float* bnpart;                                // nullable: per-(pixel tile, row half, channel) {sum, sum of squares, max, max of negated} of y
    int yb16;                                     // y (and what it accumulates onto) is stored as bf16 (single-piece mode, kSplit == 1)
    // XF == 3: x is the PRE-normalisation output y' of the Conv -> BatchNorm -> ReLU unit in front of this layer and aff its [4][Cin] {mean, rstd,
    // alpha, beta'} table: the staging applies a = max(alpha y' + beta', 0) (aff_relu; padding stays 0) before the operand split - the unit's apply pass
    // leaves the chain of dependent launches (it still writes `a` for the weight gradient, beside this layer instead of in front of it)
    const float* aff; int aff_relu;
};


// LDS images are [plane][k half][row][8 k] fp16: a row's 16 k-values live in two 16-byte pieces, one per MFMA lane half
// (lanes 0..31 hold k 0..7, lanes 32..63 k 8..15 of a 32x32x16 fragment).  Consecutive rows are then 16 bytes apart, so the
// 64 lanes of a fragment read (one ds_read_b128) cover two contiguous 512-byte runs - conflict free; with whole 32-byte
// rows every second 16-lane group of the read collided on its banks (2-way).
// 16 fp32 values (one row of 16 channels), scaled -> fp16 pieces at dst + plane * plane_stride + half * half_stride (bytes)
template <int NP>
__device__ __forceinline__ void split_store16(const float (&v)[CK], float scale, char* dst, int plane_stride, int half_stride, int* flags = nullptr) {
    unsigned pc[8][NP];
    bool bad = false;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        if (NP == 2) bad |= uz::bound_violated(v[2 * i] * scale, v[2 * i + 1] * scale);
        pieces<NP>(v[2 * i] * scale, v[2 * i + 1] * scale, pc[i]);
    }
    if (bad && flags) atomicOr(flags, uz::FLAG_W_BOUND);
#pragma unroll
    for (int q = 0; q < NP; ++q) {
        *reinterpret_cast<u32x4*>(dst + q * plane_stride) = u32x4{pc[0][q], pc[1][q], pc[2][q], pc[3][q]};
        *reinterpret_cast<u32x4*>(dst + q * plane_stride + half_stride) = u32x4{pc[4][q], pc[5][q], pc[6][q], pc[7][q]};
    }
}

// Weight panel of one layer and direction, split once per call into the kernel's LDS image:
// packed[chunk][coTile][plane 2][k half 2][tap 9][co COT][8 k] fp16, scaled by split_scale(*w_amax).  One thread per (chunk, coTile, tap, co) row.
template <bool DGRAD, int NP>
__global__ __launch_bounds__(256) void pack_weights_kernel(const float* __restrict__ w, char* __restrict__ packed, const float* __restrict__ w_amax,
                                                           int Mc, int Kc, int wCi, int nChunks, int nCoTiles, int COT, int* flags) {
    const int rows = nChunks * nCoTiles * KK * COT;
    const int e = blockIdx.x * 256 + threadIdx.x;
    if (e >= rows) return;
    const int m = e % COT, t1 = e / COT, tap = t1 % KK, t2 = t1 / KK, coT = t2 % nCoTiles, c = t2 / nCoTiles;
    const int mo = coT * COT + m;
    float v[CK];
#pragma unroll
    for (int k = 0; k < CK; ++k) {
        const int kk = c * CK + k;
        float x = 0.f;
        if (mo < Mc && kk < Kc) x = DGRAD ? w[((size_t)kk * wCi + mo) * KK + tap] : w[((size_t)mo * wCi + kk) * KK + tap];
        v[k] = x;
    }
    const int tapL = DGRAD ? KK - 1 - tap : tap;
    const int wplane = KK * COT * CK * 2;
    split_store16<NP>(v, NP == 2 ? uz::split_scale(uz::amax_read(w_amax)) : 1.f, packed + (size_t)(c * nCoTiles + coT) * NP * wplane + (tapL * COT + m) * 16, wplane, wplane / 2, flags);
}

// dynamic LDS of an instance: the staging images of the main loop, or (single-plane mode) the 16 channel rows x all pixels the
// epilogue passes through it, whichever is larger
template <int MSUB, int NTv, int TWv, int NP, int DB = 0>
constexpr int lds_bytes() {
    const int main_loop = (DB ? 2 : 1) * (NP * (KK * 32 * MSUB * CK * 2) + NP * (Geo<NTv, TWv>::PSR * CK * 2));
    const int epilogue = 16 * (NTv + 4) * 4;
    return main_loop > epilogue ? main_loop : epilogue;
}
// MK: 1 = the folded ReLU-backward mask (kernels of their own: +12 VGPRs), 2 = the folded BatchNorm-backward reduction;
// XF = input format: 0 fp32 values, 1 split storage (two fp16 pieces per word), 2 bf16 storage (2-byte elements, single-piece mode)
// DB (round 5): TWO staging images.  The chunk loop runs the MFMAs of chunk c on one image while chunk c + 1 is staged into the other:
// the packed weight block goes global -> LDS by LDS-DMA (buffer_load ... lds: it is stored in the LDS image's order, so no register,
// no VALU and no ds_write touches it), the patch values are loaded early in the chunk and written to the other image behind its last
// tap.  One barrier per chunk instead of two, and no phase in which every wave stages while the matrix pipe idles.
template <int MSUB, int NTv, int TWv, int NP, int MK = 0, int XF = 0, int DB = 0>
__device__ __forceinline__ void conv_split_body(const SP& p, const int tile_id, const int n_tiles) {
    constexpr bool XPK = XF == 1, XB = XF == 2, XA = XF == 3;
    static_assert(!XA || NP == 2, "the folded BatchNorm apply feeds the two-piece mode");
    constexpr unsigned ESZ = XB ? 2u : 4u;               // bytes per input element
    static_assert(!XPK || NP == 2, "split storage is the two-piece fp16 format");
    static_assert(!XB || NP == 1, "bf16 storage feeds the single-piece bf16 mode");
    using uz::u32x4;
    using GEO = Geo<NTv, TWv>;
    constexpr int NT = GEO::NT, TW = GEO::TW, PW = GEO::PW, PSI = GEO::PSI, PSR = GEO::PSR, G = GEO::G, CE = GEO::CE;
    constexpr int COT = 32 * MSUB;
    constexpr int WPLANE = KK * COT * CK * 2;            // bytes per weight plane
    constexpr int PPLANE = PSR * CK * 2;                 // bytes per patch plane
    constexpr int WVEC = NP * WPLANE / 16;               // 16-byte vectors of one packed weight block (2304 / 1152)
    constexpr int WREGS = DB ? 1 : (WVEC + NT - 1) / NT; // per thread (DB: the weights never pass through registers)
    constexpr int IMG = NP * WPLANE + NP * PPLANE;       // bytes of one staging image (weights + patch)
    constexpr int WDMA = WVEC / 64;                      // DB: 1 KiB LDS-DMA pieces of one weight block (36 / 18 / 9)
    static_assert(!DB || (WVEC % 64 == 0 && TWv == 32), "LDS-DMA staging: whole 1 KiB pieces, the 32-pixel-wide geometry");
    // (bf16 single-product mode on the 32-wide geometry, round 4: DEEP 1 030 -> 1 208 us and DEEP + PREF 1 085 us on 288 -> 96 @ 128 x
    //  (128 x 64) - that kernel is not waiting for its loads; timing-only builds with half the patch load instructions -13 %, with a
    //  third of the B-fragment reads -2 %: neither the LDS reads nor the vector-memory issue rate alone is the bound)
    constexpr bool DEEP = TWv == 16 && !DB;              // patch loads two chunks ahead (see stage_deep)
#ifndef UZ_EXP_PREF_ALL
#define UZ_EXP_PREF_ALL 0
#endif
    constexpr bool PREF = TWv == 16 || (UZ_EXP_PREF_ALL && MSUB == 2);  // fragment reads one tap ahead of the MFMAs (UZ_EXP_PREF_ALL: experiment builds, every geometry)
#ifndef UZ_UNCOND_TW
#define UZ_UNCOND_TW 16
#endif
    constexpr bool UNCOND = TWv <= UZ_UNCOND_TW;         // stage the (non-existent) chunk after the last one too
    extern __shared__ __attribute__((aligned(16))) char lds[];
    char* Wl = lds;                                      // DB: the image being READ; the staging lambdas write to Wl / Pl + sto
    char* Pl = lds + NP * WPLANE;
    int sto = 0;                                         // DB: byte offset of the image being WRITTEN relative to the one being read (+-IMG)

    const int tid = threadIdx.x, lane = tid & 63, l31 = lane & 31, h = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wid0 = uz::xcd_remap(tile_id, n_tiles);
    const int part = wid0 % p.kSplit, wid = wid0 / p.kSplit;     // the parts of one tile are neighbours: they share the patch in L2
    const int cbeg = part * p.cps, cend = min(p.nChunks, cbeg + p.cps);
    const int coT = wid % p.nCoTiles, pixT = wid / p.nCoTiles;
    const int txi = pixT % p.tilesX, t2 = pixT / p.tilesX;
    const int tyi = t2 % p.tilesY, b0 = t2 / p.tilesY;
    const int x0 = txi * TW, y0 = tyi * TH;
    const int co0 = coT * COT;

    // ---- staging maps (chunk invariant).  An all-ones mask OR'd into an offset fails the buffer range check: the load returns 0.
    const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<char*>(reinterpret_cast<const char*>(p.x) + (size_t)b0 * p.CinTot * p.HW * ESZ), 0, (unsigned)((size_t)p.Cin * p.HW * ESZ), 0x00020000);
    const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<char*>(p.wp), 0, (unsigned)((size_t)p.nChunks * p.nCoTiles * NP * WPLANE), 0x00020000);
    // patch rows: row tid (all 16 channels) for every thread; the rows beyond NT are shared out G threads per
    // row, CE channels each (so no thread carries a second full row in registers)
    unsigned goff[2], gmask[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int r = j == 0 ? tid : NT + tid / G;
        unsigned g = 0, gm = 0xFFFFFFFFu;
        if (r < PSI) {
            const int py = r / PW, px = r - py * PW;
            const int yy = y0 + py - 1, xx = x0 + px - 1;
            if (yy >= 0 && yy < p.H && xx >= 0 && xx < p.W) { g = ESZ * (unsigned)(yy * p.W + xx); gm = 0; }
        }
        goff[j] = g; gmask[j] = gm;
    }
    // XB (bf16 storage): the own rows are staged in PAIRS - threads 2 j and 2 j + 1 hold the patch pixels 2 j and 2 j + 1 (same patch row:
    // PW is even); each loads ONE dword = both pixels of a channel, thread 2 j for channels 0 - 7, thread 2 j + 1 for channels 8 - 15, and
    // the two exchange the halves that belong to the other's pixel through a DPP move.  8 load instructions per thread and chunk instead of
    // 16 (a timing-only build with half the patch load instructions ran the single-product kernels 10 - 13 % faster: with one MFMA per
    // operand pair the vector-memory ISSUE rate counts, not the bytes).  The pair's dword starts at an odd column (2-byte aligned: fine
    // for buffer loads on gfx950, tools/micro/unaligned_buf.hip); at the image borders it is fetched one column further in and the
    // field extraction below picks / zeroes the right half: my* / ot* = (bit offset, width) of this thread's own pixel and of its
    // partner's pixel inside the loaded dword (width 0 = padding).
    // (The same scheme for 4-byte elements - 8-byte pair loads of fp32 values / split-storage words - was built and measured: isolated
    //  kernels +-1 % (768 -> 778 us on 224 -> 128 @ 128 x 128), PHiSeg step +0.5 % inside the noise: with three MFMAs per operand pair the
    //  load issue rate is not what limits the two-piece kernels.  Not kept.)
    const int h2 = tid & 1;
    unsigned gpair = 0, gpmask = 0xFFFFFFFFu, my_off = 0, my_w = 0, ot_off = 0, ot_w = 0;
    if constexpr (XB) {
        const int rp = tid & ~1, py = rp / PW, px = rp - py * PW;
        const int yy = y0 + py - 1, c0 = x0 + px - 1;            // image columns c0 (even pixel of the pair), c0 + 1 (odd pixel)
        if (yy >= 0 && yy < p.H && c0 < p.W && p.W >= 2) {
            int col = c0, e_off = 0, e_w = 16, o_off = 16, o_w = 16;        // fields of the even / odd pixel in the dword loaded at `col`
            if (c0 < 0) { col = 0; e_w = 0; o_off = 0; }                    // left image border: load (0, 1), the odd pixel is its low half
            else if (c0 + 1 >= p.W) { col = p.W - 2; e_off = 16; o_w = 0; } // right border: load (W - 2, W - 1), the even pixel is its high half
            gpair = ESZ * (unsigned)(yy * p.W + col); gpmask = 0;
            my_off = h2 ? o_off : e_off; my_w = h2 ? o_w : e_w;
            ot_off = h2 ? e_off : o_off; ot_w = h2 ? e_w : o_w;
        }
    }
    const int prow1 = NT + tid / G, q4 = tid & (G - 1);
    const unsigned xstep = ESZ * (unsigned)p.HW;
    const unsigned wblock = (unsigned)NP * WPLANE;
    const float xs = (NP == 2 && !XPK) ? uz::split_scale(uz::amax_read(p.x_amax)) : 1.f;

    // ---- per-lane output pixels (B operand columns)
    int poff[NSUB];
#pragma unroll
    for (int n = 0; n < NSUB; ++n) {
        const int pp = wave * (32 * NSUB) + n * 32 + l31;          // pixel index inside the tile
        const int tx = pp & (TW - 1), ty = pp / TW;
        poff[n] = (ty * PW + tx) * 16 + h * (PPLANE / 2);           // byte offset of this lane's fragment for tap (0, 0)
    }

    f32x16 acc[MSUB][NSUB];
#pragma unroll
    for (int m = 0; m < MSUB; ++m)
#pragma unroll
        for (int n = 0; n < NSUB; ++n)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[m][n][r] = 0.f;

    unsigned pw[XB ? 8 : 1], mypk[XB ? 4 : 1], rcpk[XB ? 4 : 1];      // XB: the pair dwords of this thread's 8 channels; packed channel pairs of its own pixel (own / received)
    float pr[CK], pr1[CE];            // raw patch values of the next chunk: own row, share of an extra row
    unsigned pk[NP][CK / 2], pk1[NP][CE / 2]; // ... and their two fp16 planes, packed pairwise as they get split
    u32x4 wq[WREGS];
    // Staging of the next chunk is spread over the nine taps of the MFMA loop so that neither the memory
    // pipeline's queue nor the VALU work of the operand split ever stands between two MFMAs for long:
    //   taps 0..3  issue the patch loads (four k-values each; tap 0 also the shared-row quarter),
    //   taps 4..8  issue the packed-weight loads,
    //   taps 4..7  split the patch values that arrived (two parts each) - pure register work.
    // After the barrier only LDS writes remain.
    // one input element (XB: the 2-byte element sits zero-extended in the low half of the register)
    auto xload = [&](unsigned off) -> float {
        if constexpr (XB) return __builtin_bit_cast(float, (unsigned)(unsigned short)__builtin_amdgcn_raw_buffer_load_b16(rx, off, 0, 0));
        else return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rx, off, 0, 0));
    };
    auto patch_loads = [&](int c, int part) {
        const int k0 = c * CK;
        if constexpr (XB) {           // one dword (both pixels of the pair) of channel 8 h2 + part
            const int k = 8 * h2 + part;
            const unsigned kvm = (k0 + k) < p.Cin ? 0u : 0xFFFFFFFFu;
            pw[part] = (unsigned)__builtin_amdgcn_raw_buffer_load_b32(rx, (gpair + (unsigned)(k0 + k) * xstep) | gpmask | kvm, 0, 0);
            return;
        }
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            const int k = 2 * part + kk;
            const unsigned kvm = (k0 + k) < p.Cin ? 0u : 0xFFFFFFFFu;       // K tail
            pr[k] = xload((goff[0] + (unsigned)(k0 + k) * xstep) | gmask[0] | kvm);
        }
    };
    auto shared_row_loads = [&](int c) {
        const int k0 = c * CK + CE * q4;
#pragma unroll
        for (int kk = 0; kk < CE; ++kk) {
            const unsigned kvm = (k0 + kk) < p.Cin ? 0u : 0xFFFFFFFFu;
            pr1[kk] = xload((goff[1] + (unsigned)(k0 + kk) * xstep) | gmask[1] | kvm);
        }
    };
    // DB: this wave's share of the weight block of chunk c, 1 KiB per instruction straight into the image being written
    auto weight_dma = [&](int c) {
        const unsigned blk = (unsigned)(c * p.nCoTiles + coT) * wblock + 16u * (unsigned)lane;
#pragma unroll
        for (int i = 0; i < (WDMA + NT / 64 - 1) / (NT / 64); ++i) {
            const int j = i * (NT / 64) + wave;
            if (j < WDMA)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rw, (__attribute__((address_space(3))) void*)(Wl + sto + j * 1024), 16, blk + 1024u * (unsigned)j, 0, 0, 0);
        }
    };
#ifdef UZ_EXP_PATCH_DMA
    // TIMING-ONLY experiment build (tools/exp_variants.sh): the patch image of the DB kernels fetched by LDS-DMA too, as if the storage
    // were plane-separated 16-byte pieces (4 pixels of one channel row per lane) - no registers, no permutes, no ds_write.  The values
    // land in the wrong layout: the build prices the staging, it does not compute.
    auto patch_dma = [&](int c) {
        constexpr int PD = (NP * PPLANE + 1023) / 1024;
#pragma unroll
        for (int i = 0; i < (PD + NT / 64 - 1) / (NT / 64); ++i) {
            const int j = i * (NT / 64) + wave;
            if (j < PD) {
                const int e = j * 64 + lane, k = e & 15, q = e >> 4, row = q / 9, g = q - row * 9;
                const int yy = y0 + row - 1, xx = x0 + 4 * g - 1;
                const unsigned off = (yy >= 0 && yy < p.H && xx >= 0 && xx + 3 < p.W && c * CK + k < p.Cin)
                                         ? ESZ * (unsigned)(yy * p.W + xx) + (unsigned)(c * CK + k) * xstep : 0xFFFFFFFFu;
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rx, (__attribute__((address_space(3))) void*)(Pl + sto + j * 1024), 16, off, 0, 0, 0);
            }
        }
    };
#endif
    auto weight_load = [&](int c, int i) {
        const int v = tid + i * NT;
        const unsigned off = v < WVEC ? (unsigned)(c * p.nCoTiles + coT) * wblock + 16u * (unsigned)v : 0xFFFFFFFFu;
        wq[i] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rw, off, 0, 0));
    };
    // XPK: the input arrives as split storage (split_f16.h) - every word already holds the two fp16 pieces of its scaled value,
    // staging is two byte permutes per pair instead of scale / clamp / convert / subtract / convert
    // fp32 inputs of the two-piece mode: every staged value is tested against the fp16 range (one compare beside split2's clamp),
    // the predicate is accumulated over ALL chunks and raises the device flag word once at the end of the loop (round 5; until
    // round 4 only the first chunk was tested).  Split storage needs no test here: its producer clamped and flagged (pack_split).
    bool xbad = false;
    auto pair_pieces = [&](float v0, float v1, unsigned (&out)[NP]) {
        if constexpr (XPK) uz::packed_pair(__builtin_bit_cast(unsigned, v0), __builtin_bit_cast(unsigned, v1), out[0], out[1]);
        else if constexpr (XB) out[0] = __builtin_amdgcn_perm(__builtin_bit_cast(unsigned, v1), __builtin_bit_cast(unsigned, v0), 0x05040100u);   // the two stored bf16 values, as they are
        else {
            if constexpr (NP == 2) xbad |= uz::bound_violated(v0 * xs, v1 * xs);
            pieces<NP>(v0 * xs, v1 * xs, out);
        }
    };
    int cvt_c = 0;                           // XA: the chunk whose values convert() is working on (set by stage / stage_deep)
    const float aff_floor = (XA && !p.aff_relu) ? -INFINITY : 0.f;
    auto aff_own = [&](int k) {              // own row, channel k of the chunk: alpha / beta' are wave-uniform (scalar loads)
        const int ch = cvt_c * CK + k, chc = min(ch, p.Cin - 1);
        const float t = fmaxf(fmaf(pr[k], p.aff[2 * p.Cin + chc], p.aff[3 * p.Cin + chc]), aff_floor);
        pr[k] = (gmask[0] == 0u && ch < p.Cin) ? t : 0.f;           // padding pixels and the K tail stay zero
    };
    auto aff_shared = [&](int i) {           // shared rows: this lane's channels CE q4 .. CE q4 + CE - 1
        const int ch = cvt_c * CK + CE * q4 + i, chc = min(ch, p.Cin - 1);
        const float t = fmaxf(fmaf(pr1[i], p.aff[2 * p.Cin + chc], p.aff[3 * p.Cin + chc]), aff_floor);
        pr1[i] = (gmask[1] == 0u && ch < p.Cin) ? t : 0.f;
    };
    auto convert = [&](int j) {              // split / round the values of k 4j .. 4j + 3 (j = 0 also the shared-row share)
        const int i0 = 2 * j;
        if constexpr (XA) {
#pragma unroll
            for (int k = 0; k < 4; ++k) aff_own(4 * j + k);
            if (j == 0) {
#pragma unroll
                for (int i = 0; i < CE; ++i) aff_shared(i);
            }
        }
        if constexpr (XB) {
            // pair dwords 2 j, 2 j + 1 (channels 8 h2 + 2 j, + 1): own pixel's halves stay, the partner pixel's halves go to the partner
            const unsigned m0 = __builtin_amdgcn_ubfe(pw[2 * j], my_off, my_w), m1 = __builtin_amdgcn_ubfe(pw[2 * j + 1], my_off, my_w);
            const unsigned o0 = __builtin_amdgcn_ubfe(pw[2 * j], ot_off, ot_w), o1 = __builtin_amdgcn_ubfe(pw[2 * j + 1], ot_off, ot_w);
            mypk[j] = m0 | (m1 << 16);
            rcpk[j] = (unsigned)__builtin_amdgcn_mov_dpp((int)(o0 | (o1 << 16)), 0xB1, 0xF, 0xF, true);      // quad_perm [1, 0, 3, 2]: lane ^ 1
        } else {
        unsigned t0[NP], t1[NP];
        pair_pieces(pr[2 * i0], pr[2 * i0 + 1], t0);
        pair_pieces(pr[2 * i0 + 2], pr[2 * i0 + 3], t1);
#pragma unroll
        for (int q = 0; q < NP; ++q) { pk[q][i0] = t0[q]; pk[q][i0 + 1] = t1[q]; }
        }
        if (j == 0) {
#pragma unroll
            for (int i = 0; i < CE / 2; ++i) {
                unsigned t2[NP];
                pair_pieces(pr1[2 * i], pr1[2 * i + 1], t2);
#pragma unroll
                for (int q = 0; q < NP; ++q) pk1[q][i] = t2[q];
            }
        }
    };
    auto stage = [&](int c, int tap) {
        cvt_c = c;
        if constexpr (DB) {             // weights first (DMA), patch loads over taps 1..4, conversions over taps 5..8
#ifdef UZ_EXP_PATCH_DMA
            if (tap == 0) { weight_dma(c); patch_dma(c); }
            return;
#endif
            if (tap == 0) weight_dma(c);
            else if (tap < 5) {
                patch_loads(c, 2 * (tap - 1)); patch_loads(c, 2 * (tap - 1) + 1);
                if (tap == 1) shared_row_loads(c);
            } else convert(tap - 5);
            return;
        }
        if (tap < 4) {
            patch_loads(c, 2 * tap); patch_loads(c, 2 * tap + 1);
            if (tap == 0) shared_row_loads(c);
        } else {
            if (tap - 4 < WREGS) weight_load(c, tap - 4);
            if (tap == 8) { for (int i = 5; i < WREGS; ++i) weight_load(c, i); }
            if (tap < 8) convert(tap - 4);
        }
    };
    // DEEP (the 16 x 16-pixel geometry: few workgroups, one to three waves per SIMD, so nothing else covers a load's latency):
    // the patch values are fetched TWO chunks ahead - taps 5..8 of chunk c issue the loads of chunk c + 2 into the raw registers
    // that taps 0..3 have just converted (chunk c + 1); the packed weights (L2 resident, short latency) stay one chunk ahead and
    // are issued first (tap 4) so that the in-order wait before the LDS writes does not include the patch loads.
    auto stage_deep = [&](int c, int tap, bool more, bool more2) {
        cvt_c = c + 1;
        if (tap < 4) { if (more) convert(tap); }
        else if (tap == 4) {
            if (more) {
#pragma unroll
                for (int i = 0; i < WREGS; ++i) weight_load(c + 1, i);
            }
        } else if (more2) {
            patch_loads(c + 2, 2 * (tap - 5)); patch_loads(c + 2, 2 * (tap - 5) + 1);
            if (tap == 5) shared_row_loads(c + 2);
        }
    };
    auto lstore = [&]() {
#ifdef UZ_EXP_PATCH_DMA
        if constexpr (DB) return;
#endif
        char* dst = Pl + sto + tid * 16;
        if constexpr (XB) {           // this thread loaded channels 8 h2 .. 8 h2 + 7 itself and received the other eight
            *reinterpret_cast<u32x4*>(dst + (h2 ? PPLANE / 2 : 0)) = u32x4{mypk[0], mypk[1], mypk[2], mypk[3]};
            *reinterpret_cast<u32x4*>(dst + (h2 ? 0 : PPLANE / 2)) = u32x4{rcpk[0], rcpk[1], rcpk[2], rcpk[3]};
        } else {
#pragma unroll
        for (int q = 0; q < NP; ++q) {
            *reinterpret_cast<u32x4*>(dst + q * PPLANE) = u32x4{pk[q][0], pk[q][1], pk[q][2], pk[q][3]};                     // k 0..7
            *reinterpret_cast<u32x4*>(dst + q * PPLANE + PPLANE / 2) = u32x4{pk[q][4], pk[q][5], pk[q][6], pk[q][7]};       // k 8..15
        }
        }
        // shared rows: this thread holds channels [CE * q4, CE * q4 + CE) of row prow1 (CE = 4: 8 bytes, CE = 8: one 16-byte piece)
        char* dst1 = Pl + sto + prow1 * 16 + ((CE * q4) >> 3) * (PPLANE / 2) + ((CE * q4) & 7) * 2;
#pragma unroll
        for (int q = 0; q < NP; ++q) {
            if (CE == 4) *reinterpret_cast<uint2*>(dst1 + q * PPLANE) = make_uint2(pk1[q][0], pk1[q][1]);
            else *reinterpret_cast<u32x4*>(dst1 + q * PPLANE) = u32x4{pk1[q][0], pk1[q][1], pk1[q][CE / 2 - 2], pk1[q][CE / 2 - 1]};
        }
        if constexpr (!DB) {
#pragma unroll
        for (int i = 0; i < WREGS; ++i)
            if (tid + i * NT < WVEC) *reinterpret_cast<u32x4*>(Wl + 16 * (tid + i * NT)) = wq[i];
        }
    };

    const int nChunks = cend;
    long long st0 = 0, st1 = 0, stl = 0, rt0 = 0;
    if (p.stamps) { st0 = __builtin_amdgcn_s_memtime(); rt0 = __builtin_amdgcn_s_memrealtime(); }
#pragma unroll
    for (int tap = 0; tap < KK; ++tap) stage(cbeg, tap);
    // A 64-channel tile whose upper 32 channels lie beyond Cout (Cout = 224 = 3 x 64 + 32: the data gradient of the heaviest
    // layer) skips that half's MFMAs, fragment reads and epilogue pass: 12.5 % of that launch's matrix work were zeros.
    // 32-channel sub-tiles of this tile that hold real output channels (MSUB == 2: the upper half is empty or not)
    const int mact = MSUB == 4 ? min(MSUB, (p.Cout - co0 + 31) / 32) : ((MSUB == 2 && p.Cout - co0 <= 32) ? 1 : MSUB);
    const bool half_tile = mact < MSUB;
    constexpr int MS = MSUB;
    if (DEEP && cbeg + 1 < cend) {
#pragma unroll
        for (int tap = 0; tap < 4; ++tap) stage(cbeg + 1, tap);       // the raw registers are free again: chunk cbeg is converted
    }
    if constexpr (DB) {                        // the first chunk goes into the image the loop reads first
        lstore();
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // this wave's LDS-DMA pieces have landed ...
        __syncthreads();                                      // ... and so have everybody else's
        sto = IMG;
    }
    for (int c = cbeg; c < nChunks; ++c) {
        long long ta = 0;
        if (p.stamps) ta = __builtin_amdgcn_s_memtime();
        if (XPK && c == p.segc && c > cbeg) {
            // the input's second segment was scaled from another bound: bring the sums so far into its units (a power of two: exact)
            const float ratio = uz::split_scale(uz::amax_read(p.x_amax2)) * uz::split_inv_scale(uz::amax_read(p.x_amax));
#pragma unroll
            for (int m = 0; m < MSUB; ++m)
#pragma unroll
                for (int n = 0; n < NSUB; ++n) acc[m][n] *= ratio;
        }
        if constexpr (!DB) {
        __syncthreads();                       // every wave has finished the MFMAs of the previous chunk
        lstore();
        __syncthreads();
        }
        if (p.stamps) { const long long tb = __builtin_amdgcn_s_memtime(); if (c == cbeg) st1 = tb; else stl += tb - ta; }
        const bool more = c + 1 < nChunks;
        const char* Al = Wl + l31 * 16 + h * (WPLANE / 2);
        // Fragment reads run one tap ahead of the MFMAs (PREF: the 16 x 16-pixel geometry, where one to three waves per SIMD do not
        // cover a ds_read's latency between themselves); the first tap of a chunk has to wait for the barrier above.
        u32x4 a[PREF ? 2 : 1][MS][NP], b[PREF ? 2 : 1][NSUB][NP];
        auto frag_loads = [&](int tap, int buf) {
            const int tapoff = ((tap / 3) * PW + (tap % 3)) * 16;
#pragma unroll
            for (int m = 0; m < MS; ++m)
#pragma unroll
                for (int q = 0; q < NP; ++q)
                    a[buf][m][q] = *reinterpret_cast<const u32x4*>(Al + q * WPLANE + (tap * COT + m * 32) * 16);
#pragma unroll
            for (int n = 0; n < NSUB; ++n)
#pragma unroll
                for (int q = 0; q < NP; ++q)
                    b[buf][n][q] = *reinterpret_cast<const u32x4*>(Pl + q * PPLANE + poff[n] + tapoff);
        };
        if (PREF) frag_loads(0, 0);
#pragma unroll
        for (int tap = 0; tap < KK; ++tap) {
            const int cur = PREF ? (tap & 1) : 0;
            if (!PREF) frag_loads(tap, 0);
            else if (tap + 1 < KK) frag_loads(tap + 1, cur ^ 1);
            // Unconditional (a branch here would fence the instruction scheduler: the staging VALU block ran with the matrix pipe
            // idle): past the last chunk the K-tail masks of the loads turn them into range-checked no-ops returning 0.
            if (DEEP) stage_deep(c, tap, UNCOND || more, UNCOND || c + 2 < nChunks);
            else if (UNCOND || more) stage(c + 1, tap);
            // smallest products first
#pragma unroll
            for (int m = 0; m < MS; ++m) {
                if (half_tile && m >= mact) continue;        // workgroup-uniform: a scalar branch around six MFMAs
#pragma unroll
                for (int n = 0; n < NSUB; ++n) acc[m][n] = mma<NP>(acc[m][n], a[cur][m], b[cur][n]);
            }
            if (UNCOND) {
                // one wave per SIMD: nobody else fills the gaps, so the staging VALU / memory instructions of this tap are
                // asked to sit BETWEEN its MFMAs (in program order behind them they would run with the matrix pipe idle)
#pragma unroll
                for (int i = 0; i < MS * NSUB * (NP == 2 ? 3 : 1); ++i) {
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                    __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                    __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
                    __builtin_amdgcn_sched_group_barrier(0x002, 10, 0);
                }
                __builtin_amdgcn_sched_barrier(0);           // ... and the next tap's work stays behind this tap's MFMAs
            }
        }
        if constexpr (DB) {
            // hand-over: the patch of chunk c + 1 goes into the other image (nobody reads it: every wave passed the previous
            // hand-over), this wave's DMA pieces are waited for, and ONE barrier closes the chunk; then the images swap roles
            if (more) lstore();
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            Wl += sto; Pl += sto; sto = -sto;
        }
    }

    if (NP == 2 && !XPK) uz::raise_flag(p.flags, xbad, p.flag_bit);
    long long st2 = 0;
    if (p.stamps) st2 = __builtin_amdgcn_s_memtime();
    // ---- epilogue: undo the operand scales (exact), bias, optional accumulate / ReLU, NCHW stores.
    // An accumulator register holds ONE pixel of a channel per lane, so storing from registers costs a 4-byte store instruction per
    // value (measured with the cycle stamps: 24 k of a workgroup's 105 - 160 k cycles, every CU's store queue full while the
    // matrix pipe idles).  Instead the tile goes through the (now dead) staging LDS, 8 * GP channels x all pixels per pass, and
    // comes back as float4 rows: a quarter of the store instructions, each wave writing whole 128-byte lines.
    const float* const xa_last = (XPK && p.segc > 0 && nChunks > p.segc) ? p.x_amax2 : p.x_amax;      // the bound the LAST chunk of this part was scaled from
    const float inv = NP == 2 ? uz::split_inv_scale(uz::amax_read(xa_last)) * uz::split_inv_scale(uz::amax_read(p.w_amax)) : 1.f;
    constexpr int LDS_BYTES = DB ? IMG : lds_bytes<MSUB, NTv, TWv, NP>();      // DB: the epilogue passes fit one image
    constexpr int GP = (32 * NT * 4 <= LDS_BYTES) ? 4 : ((16 * NT * 4 <= LDS_BYTES) ? 2 : 1);    // channel groups of 8 per pass
    constexpr int ROWF = NT + 4;                         // floats per channel row (one pixel per thread, + 16 B: rows start on different banks)
    static_assert(8 * GP * ROWF * 4 <= LDS_BYTES, "epilogue staging must fit the main loop's LDS");
    constexpr int Q = NT / 4;                            // float4 per channel row
    float* El = reinterpret_cast<float*>(lds);
    const bool split = p.kSplit > 1;                     // partial sums only: bias / accumulate / ReLU / bound happen in the reduce
    const int yb = (NP == 1 && !split) ? p.yb16 : 0;     // the output tensor holds bf16 elements (offsets below are in ELEMENTS either way)
    float* const obase = split ? p.slab + (size_t)part * p.N * p.Cout * p.HW + (size_t)b0 * p.Cout * p.HW
                       : yb ? reinterpret_cast<float*>(reinterpret_cast<unsigned short*>(p.y) + (size_t)b0 * p.CoutTot * p.HW)
                            : p.y + (size_t)b0 * p.CoutTot * p.HW;
    const float* const mbase = (MK != 0 && p.mask) ? p.mask + (size_t)b0 * p.maskCtot * p.HW : nullptr;
    const bool vec = (p.W & 3) == 0 && (reinterpret_cast<uintptr_t>(obase) & (yb ? 7 : 15)) == 0 && (reinterpret_cast<uintptr_t>(mbase) & 15) == 0;
    const int c4 = tid % Q, rsub = tid / Q;              // this thread's float4 column and row phase (rows rsub, rsub + 4, ...)
    const int px = 4 * c4, ty = px / TW, tx = px % TW;
    const int oy = y0 + ty, ox = x0 + tx;
    const bool rowok = oy < p.H && ox < p.W;
    float vmax = 0.f;
#pragma unroll
    for (int m = 0; m < MSUB; ++m) {
        if (half_tile && m >= mact) break;               // workgroup-uniform
#pragma unroll
        for (int g0 = 0; g0 < 4; g0 += GP) {
            // What the rows of this pass read from memory - the tensor they accumulate onto, the producing unit's activation (MK) -
            // is requested BEFORE the tile goes through LDS: fetched row by row inside the loop below, every row waited for its
            // own load with one workgroup per CU to hide it (folded data gradient of 128 -> 128 @ 128 x 128: 451 -> 630 us)
            f32x4 pre_acc[2 * GP], pre_mk[2 * GP];
            const bool pre_a = vec && !split && p.accumulate, pre_m = vec && !split && MK != 0 && mbase;
            if (pre_a || pre_m) {
#pragma unroll
                for (int j = 0; j < 2 * GP; ++j) {
                    const int co = co0 + m * 32 + g0 * 8 + 4 * j + rsub;
                    if (co < p.Cout && rowok) {
                        const size_t off = (size_t)co * p.HW + (size_t)oy * p.W + ox;
                        if (pre_a) pre_acc[j] = uz::ld_elem4(obase, off, yb);
                        if (pre_m) pre_mk[j] = *reinterpret_cast<const f32x4*>(mbase + off);
                    }
                }
            }
            __syncthreads();                             // main loop / previous pass done with this LDS
#pragma unroll
            for (int g = 0; g < GP; ++g)
#pragma unroll
                for (int rr = 0; rr < 4; ++rr)
#pragma unroll
                    for (int n = 0; n < NSUB; ++n)
                        El[(g * 8 + rr + 4 * h) * ROWF + wave * (32 * NSUB) + n * 32 + l31] = acc[m][n][(g0 + g) * 4 + rr];
            __syncthreads();
#pragma unroll
            for (int j = 0; j < 2 * GP; ++j) {
                const int row = 4 * j + rsub;            // 0 .. 8 GP - 1
                const int co = co0 + m * 32 + g0 * 8 + row;
                f32x4 st4 = {0.f, 0.f, -INFINITY, -INFINITY};     // BatchNorm partials of this thread's four pixels: sum, sum of squares, max, max(-y)
                if (co < p.Cout && rowok) {
                    f32x4 v = *reinterpret_cast<const f32x4*>(El + row * ROWF + px);
                    const size_t doff = (size_t)co * p.HW + (size_t)oy * p.W + ox;
                    if (split) {
                        v *= inv;
                    } else {
                        const float bv = p.bias ? p.bias[co] : 0.f;
                        v = v * inv + bv;
                    }
                    if (vec) {
                        f32x4 xh = {0.f, 0.f, 0.f, 0.f};
                        if (!split) {
                            if (p.accumulate) v += pre_acc[j];
                            if (MK == 1 && mbase) {          // ReLU backward of the producing unit (unet.py:25-30), behind the last accumulation
                                const f32x4 mk = pre_mk[j];
                                v.x = mk.x > 0.f ? v.x : 0.f; v.y = mk.y > 0.f ? v.y : 0.f; v.z = mk.z > 0.f ? v.z : 0.f; v.w = mk.w > 0.f ? v.w : 0.f;
                            }
                            if (MK == 2 && mbase) {          // ReLU mask of the producing Conv -> BatchNorm -> ReLU unit and its normalised activation
                                const f32x4 yy = pre_mk[j];
                                const float mean = p.mk_save[co], rstd = p.mk_save[p.Cout + co], al = p.mk_save[2 * p.Cout + co], be = p.mk_save[3 * p.Cout + co];
                                if (p.mk_relu) {
                                    v.x = fmaf(yy.x, al, be) > 0.f ? v.x : 0.f; v.y = fmaf(yy.y, al, be) > 0.f ? v.y : 0.f;
                                    v.z = fmaf(yy.z, al, be) > 0.f ? v.z : 0.f; v.w = fmaf(yy.w, al, be) > 0.f ? v.w : 0.f;
                                }
                                xh = f32x4{(yy.x - mean) * rstd, (yy.y - mean) * rstd, (yy.z - mean) * rstd, (yy.w - mean) * rstd};
                            }
                            if (p.relu) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
                            vmax = fmaxf(fmaxf(vmax, fmaxf(fabsf(v.x), fabsf(v.y))), fmaxf(fabsf(v.z), fabsf(v.w)));
                        }
                        if (yb) {                           // round once: the statistics below are those of the STORED values
                            const uint2 w2 = uz::bf16x4_pack(v);
                            *reinterpret_cast<uint2*>(reinterpret_cast<unsigned short*>(obase) + doff) = w2;
                            v = uz::bf16x4_widen(w2);
                        } else *reinterpret_cast<f32x4*>(obase + doff) = v;
                        if (MK == 2) st4 = f32x4{(v.x + v.y) + (v.z + v.w), (v.x * xh.x + v.y * xh.y) + (v.z * xh.z + v.w * xh.w),
                                                 fmaxf(fmaxf(fabsf(v.x), fabsf(v.y)), fmaxf(fabsf(v.z), fabsf(v.w))),
                                                 fmaxf(fmaxf(fabsf(xh.x), fabsf(xh.y)), fmaxf(fabsf(xh.z), fabsf(xh.w)))};
                        else st4 = f32x4{(v.x + v.y) + (v.z + v.w), (v.x * v.x + v.y * v.y) + (v.z * v.z + v.w * v.w),
                                         fmaxf(fmaxf(v.x, v.y), fmaxf(v.z, v.w)), fmaxf(fmaxf(-v.x, -v.y), fmaxf(-v.z, -v.w))};
                    } else {
#pragma unroll
                        for (int e = 0; e < 4; ++e)
                            if (ox + e < p.W) {
                                float t = v[e];
                                if (!split) {
                                    if (p.accumulate) t += uz::ld_elem(obase, doff + e, yb);
                                    if (MK == 1 && mbase) t = mbase[doff + e] > 0.f ? t : 0.f;
                                    if (p.relu) t = fmaxf(t, 0.f);
                                    vmax = fmaxf(vmax, fabsf(t));
                                }
                                if (MK == 2 && mbase) {
                                    const float yy = mbase[doff + e];
                                    if (p.mk_relu) t = fmaf(yy, p.mk_save[2 * p.Cout + co], p.mk_save[3 * p.Cout + co]) > 0.f ? t : 0.f;
                                    const float xh1 = (yy - p.mk_save[co]) * p.mk_save[p.Cout + co];
                                    obase[doff + e] = t;
                                    st4 = f32x4{st4.x + t, st4.y + t * xh1, fmaxf(st4.z, fabsf(t)), fmaxf(st4.w, fabsf(xh1))};
                                    continue;
                                }
                                if (yb) t = uz::bf16_round(t);
                                uz::st_elem(obase, doff + e, t, yb);
                                st4 = f32x4{st4.x + t, st4.y + t * t, fmaxf(st4.z, t), fmaxf(st4.w, -t)};
                            }
                    }
                }
                if (p.bnpart) {
                    // BatchNorm statistics of the tile while it is in registers (torchlayers.py:18-21: every Conv2d feeds a BatchNorm2d):
                    // the 64 lanes of a wave hold 256 pixels of ONE channel row; wave reduce, lane 0 writes the partial.
#pragma unroll
                    for (int o = 32; o > 0; o >>= 1) {
                        st4.x += __shfl_xor(st4.x, o, 64); st4.y += __shfl_xor(st4.y, o, 64);
                        st4.z = fmaxf(st4.z, __shfl_xor(st4.z, o, 64)); st4.w = fmaxf(st4.w, __shfl_xor(st4.w, o, 64));
                    }
                    if (lane == 0 && co < p.Cout)
                        *reinterpret_cast<f32x4*>(p.bnpart + ((size_t)(pixT * (Q / 64) + c4 / 64) * p.Cout + co) * 4) = st4;
                }
            }
        }
    }
    if (!split && p.y_amax) uz::amax_publish(vmax, p.y_amax);
    if (p.stamps) {
        __builtin_amdgcn_s_waitcnt(0);          // the stores have left the wave
        const long long st3 = __builtin_amdgcn_s_memtime(), rt1 = __builtin_amdgcn_s_memrealtime();
        if (tid == 0 && blockIdx.x < 4096) {
            long long* o = p.stamps + 8 * blockIdx.x;
            o[0] = st0; o[1] = st1; o[2] = stl; o[3] = st2; o[4] = st3; o[5] = rt0; o[6] = rt1; o[7] = nChunks - cbeg;
        }
    }
}

// The instances as kernels of their own (attributes take literal constants only; names show up in profiles): three tile
// geometries x {split-fp16 (fp32-accurate), single-piece bf16} x epilogue {plain, folded ReLU backward, folded BatchNorm-backward
// reduction} x input {fp32, split storage}.
template <int MSUB, int NTv, int TWv, int NP, int MK, int XF, int DB = 0> struct SplitKernel;
#define UZ_SPLIT_KERNEL_(name, MSUB_, NT_, TW_, NP_, MK_, XF_, OCC_, DB_)                                                     \
    __global__ __launch_bounds__(NT_, OCC_) void name(const SP p) { conv_split_body<MSUB_, NT_, TW_, NP_, MK_, XF_, DB_>(p, blockIdx.x, gridDim.x); } \
    template <> struct SplitKernel<MSUB_, NT_, TW_, NP_, MK_, XF_, DB_> { static constexpr auto fn = name; };
#define UZ_SPLIT_KERNEL(name, MSUB_, NT_, TW_, NP_, MK_, XF_, OCC_) UZ_SPLIT_KERNEL_(name, MSUB_, NT_, TW_, NP_, MK_, XF_, OCC_, 0)
// (round 4: the 32-channel-tile kernels keep (512, 4) although that bound costs them ~30 spilled registers - with (512, 3), no spills
//  but one workgroup per CU, 32 -> 32 @ 128 x 128 ran 55 -> 61 us forward and the PHiSeg step lost 0.8 %)
// (round 4, single-product mode: (512, 4) = two workgroups per CU on the 64-channel-tile bf16 kernels needs 128 VGPRs - 168 in use, 76 - 80
//  spilled into the chunk loop: 288 -> 96 @ 128 x (128 x 64) 1 030 -> 2 253 us, PHiSeg3D 34.9 -> 43.6 ms.  One workgroup per CU it stays.)
// UZ_OCC16 / UZ_OCC16P: minimum workgroups per SIMD of the 16 x 16-pixel instances (fp32 / split-storage input); experiment builds raise
// them to 4 (at most 128 VGPRs: such a workgroup fits beside a 64-channel-tile convolution's two waves per SIMD, profiles/NOTES_r5.md section 4)
#ifndef UZ_OCC16
#define UZ_OCC16 2
#endif
#ifndef UZ_OCC16P
#define UZ_OCC16P 3
#endif
UZ_SPLIT_KERNEL(conv_split_kernel_2_512_32, 2, 512, 32, 2, 0, 0, 1)
UZ_SPLIT_KERNEL(conv_split_kernel_1_512_32, 1, 512, 32, 2, 0, 0, 4)
UZ_SPLIT_KERNEL(conv_split_kernel_1_256_16, 1, 256, 16, 2, 0, 0, UZ_OCC16)
UZ_SPLIT_KERNEL(conv_bf16_kernel_2_512_32, 2, 512, 32, 1, 0, 0, 1)
UZ_SPLIT_KERNEL(conv_bf16_kernel_1_512_32, 1, 512, 32, 1, 0, 0, 4)
UZ_SPLIT_KERNEL(conv_bf16_kernel_1_256_16, 1, 256, 16, 1, 0, 0, 3)
UZ_SPLIT_KERNEL(conv_split_relu_kernel_2_512_32, 2, 512, 32, 2, 1, 0, 1)
UZ_SPLIT_KERNEL(conv_split_relu_kernel_1_512_32, 1, 512, 32, 2, 1, 0, 4)
UZ_SPLIT_KERNEL(conv_split_relu_kernel_1_256_16, 1, 256, 16, 2, 1, 0, UZ_OCC16)
UZ_SPLIT_KERNEL(conv_bf16_relu_kernel_2_512_32, 2, 512, 32, 1, 1, 0, 1)
UZ_SPLIT_KERNEL(conv_bf16_relu_kernel_1_512_32, 1, 512, 32, 1, 1, 0, 4)
UZ_SPLIT_KERNEL(conv_bf16_relu_kernel_1_256_16, 1, 256, 16, 1, 1, 0, 3)
// round 4: input in split storage (conv_splitp_*), BatchNorm-backward reduction in the data gradient's epilogue (*_bn_*)
UZ_SPLIT_KERNEL(conv_splitp_kernel_2_512_32, 2, 512, 32, 2, 0, 1, 1)
UZ_SPLIT_KERNEL(conv_splitp_kernel_1_512_32, 1, 512, 32, 2, 0, 1, 4)
UZ_SPLIT_KERNEL(conv_splitp_kernel_1_256_16, 1, 256, 16, 2, 0, 1, UZ_OCC16P)
UZ_SPLIT_KERNEL(conv_split_bn_kernel_2_512_32, 2, 512, 32, 2, 2, 0, 1)
UZ_SPLIT_KERNEL(conv_split_bn_kernel_1_512_32, 1, 512, 32, 2, 2, 0, 4)
UZ_SPLIT_KERNEL(conv_split_bn_kernel_1_256_16, 1, 256, 16, 2, 2, 0, UZ_OCC16)
UZ_SPLIT_KERNEL(conv_splitp_bn_kernel_2_512_32, 2, 512, 32, 2, 2, 1, 1)
UZ_SPLIT_KERNEL(conv_splitp_bn_kernel_1_512_32, 1, 512, 32, 2, 2, 1, 4)
UZ_SPLIT_KERNEL(conv_splitp_bn_kernel_1_256_16, 1, 256, 16, 2, 2, 1, UZ_OCC16P)
// round 6: the producing unit's BatchNorm apply folded into the staging (conv_splita_*: x = its pre-normalisation output + its statistics table)
UZ_SPLIT_KERNEL(conv_splita_kernel_2_512_32, 2, 512, 32, 2, 0, 3, 1)
UZ_SPLIT_KERNEL(conv_splita_kernel_1_512_32, 1, 512, 32, 2, 0, 3, 4)
UZ_SPLIT_KERNEL(conv_splita_kernel_1_256_16, 1, 256, 16, 2, 0, 3, UZ_OCC16)
// bf16 STORAGE of the input (the volume path, planes wider than 32): 2-byte patch loads, no conversion while staging
UZ_SPLIT_KERNEL(conv_b16_kernel_2_512_32, 2, 512, 32, 1, 0, 2, 1)
UZ_SPLIT_KERNEL(conv_b16_kernel_1_512_32, 1, 512, 32, 1, 0, 2, 4)
UZ_SPLIT_KERNEL(conv_b16_kernel_1_256_16, 1, 256, 16, 1, 0, 2, 3)
// round 5: the 64-channel tile with two staging images and LDS-DMA weight staging (conv_*db_*), one workgroup per CU (152 KB of LDS)
UZ_SPLIT_KERNEL_(conv_split_db_kernel_2_512_32, 2, 512, 32, 2, 0, 0, 1, 1)
UZ_SPLIT_KERNEL_(conv_split_relu_db_kernel_2_512_32, 2, 512, 32, 2, 1, 0, 1, 1)
UZ_SPLIT_KERNEL_(conv_splitp_db_kernel_2_512_32, 2, 512, 32, 2, 0, 1, 1, 1)
UZ_SPLIT_KERNEL_(conv_splita_db_kernel_2_512_32, 2, 512, 32, 2, 0, 3, 1, 1)
UZ_SPLIT_KERNEL_(conv_split_bn_db_kernel_2_512_32, 2, 512, 32, 2, 2, 0, 1, 1)
UZ_SPLIT_KERNEL_(conv_splitp_bn_db_kernel_2_512_32, 2, 512, 32, 2, 2, 1, 1, 1)
UZ_SPLIT_KERNEL_(conv_bf16_db_kernel_2_512_32, 2, 512, 32, 1, 0, 0, 1, 1)
UZ_SPLIT_KERNEL_(conv_bf16_relu_db_kernel_2_512_32, 2, 512, 32, 1, 1, 0, 1, 1)
UZ_SPLIT_KERNEL_(conv_b16_db_kernel_2_512_32, 2, 512, 32, 1, 0, 2, 1, 1)
// single-piece bf16 mode, 128-channel tile (always the two-image form: 115 KB of LDS)
UZ_SPLIT_KERNEL_(conv_bf16_db_kernel_4_512_32, 4, 512, 32, 1, 0, 0, 1, 1)
UZ_SPLIT_KERNEL_(conv_bf16_relu_db_kernel_4_512_32, 4, 512, 32, 1, 1, 0, 1, 1)
UZ_SPLIT_KERNEL_(conv_b16_db_kernel_4_512_32, 4, 512, 32, 1, 0, 2, 1, 1)
#undef UZ_SPLIT_KERNEL
#undef UZ_SPLIT_KERNEL_
// UZ_CONV_DB=0: the single-image kernels everywhere (A/B runs)
inline bool db_enabled() { static const bool on = !(getenv("UZ_CONV_DB") && atoi(getenv("UZ_CONV_DB")) == 0); return on; }

template <int MSUB, int NTv, int TWv, int NP, int MK, int XF, int DB>
int launch_db(const SP& p, int grid, hipStream_t st) {
    constexpr size_t smem = lds_bytes<MSUB, NTv, TWv, NP, DB>();
    static bool attr_done = false;
    auto kern = SplitKernel<MSUB, NTv, TWv, NP, MK, XF, DB>::fn;
    if (!attr_done) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess)
            return uz::fail("conv_split: cannot raise dynamic LDS limit");
        attr_done = true;
    }
    // (round 5: a persistent form - fewer workgroups than tiles, each walking its tiles, so that the grid leaves CUs free for the other
    //  lanes' small launches - was built and measured at commit d333d48: 224 workgroups unblock a chain of small launches in the
    //  micro-benchmark, the step lost 2.5 %, and the tile loop around the body cost every instance registers: 72 -> 184 bytes of
    //  scratch in the 32-channel-tile kernel, 144 -> 177 VGPRs in the 16 x 16-pixel one, SGPRs at their cap.  Removed.)
    hipLaunchKernelGGL(kern, dim3(grid), dim3(NTv), smem, st, p);
    return uz::check_launch("conv_split_kernel");
}
template <int MSUB, int NTv, int TWv, int NP, int MK, int XF>
int launch_one(const SP& p, int grid, hipStream_t st) {
    if constexpr (MSUB == 4) return launch_db<MSUB, NTv, TWv, NP, MK, XF, 1>(p, grid, st);
    else {
        if constexpr (MSUB == 2 && NTv == 512 && TWv == 32) {
            if (db_enabled()) return launch_db<MSUB, NTv, TWv, NP, MK, XF, 1>(p, grid, st);
        }
        return launch_db<MSUB, NTv, TWv, NP, MK, XF, 0>(p, grid, st);
    }
}
// mk: 0 plain, 1 folded ReLU backward, 2 folded BatchNorm-backward reduction; xpk: input in split storage (two-piece mode only)
template <int MSUB, int NTv, int TWv, int NP>
int launch(const SP& p, int grid, hipStream_t st, int mk, int xpk, int xb16) {
    if constexpr (NP == 2) {
        if (xb16 || p.yb16) return uz::fail("conv_split: bf16 storage needs the single-piece bf16 mode (uz_set_conv_math(3))");
        if (p.aff) {
            if (mk != 0 || xpk) return uz::fail("conv_split: the folded BatchNorm apply takes fp32 input and a plain epilogue");
            return launch_one<MSUB, NTv, TWv, 2, 0, 3>(p, grid, st);
        }
        if (xpk) {
            if (mk == 1) return uz::fail("conv_split: the folded ReLU backward takes an fp32 gradient");
            return mk == 2 ? launch_one<MSUB, NTv, TWv, 2, 2, 1>(p, grid, st) : launch_one<MSUB, NTv, TWv, 2, 0, 1>(p, grid, st);
        }
        if (mk == 2) return launch_one<MSUB, NTv, TWv, 2, 2, 0>(p, grid, st);
    } else {
        if (xpk || mk == 2) return uz::fail("conv_split: split storage / the folded BatchNorm reduction need the two-piece fp16 mode");
        if (xb16) {
            if (mk != 0) return uz::fail("conv_split: bf16 storage of the input has no folded-ReLU instance");
            return launch_one<MSUB, NTv, TWv, 1, 0, 2>(p, grid, st);
        }
    }
    return mk == 1 ? launch_one<MSUB, NTv, TWv, NP, 1, 0>(p, grid, st) : launch_one<MSUB, NTv, TWv, NP, 0, 0>(p, grid, st);
}

// tile geometry of a layer: 16 x 32 tiles (512 threads, 64- or 32-channel tiles) when the plane is wider than 32,
// else 16 x 16 tiles (256 threads, 32-channel tiles)
inline bool small_geo(int W) { return W <= 32; }       // measured: 16 x 16 tiles win or tie on 32 x 32 planes (128 -> 128: 91 -> 67 us), lose on 64 x 64
inline int tile_w(int W) { return small_geo(W) ? 16 : 32; }
// 16 x 16 tiles carry 32 output channels (64 was measured 8 - 30 % slower: the layers that use this geometry want workgroups)
// ... and on the large planes when the contraction is short (Kc <= 32: two chunks, the workgroup is mostly prologue and
// epilogue): the 32-channel kernel keeps two workgroups per CU, the 64-channel one a single one (32 -> 96 @ 128 x 128: 202 -> 185 us).
// ... and in the single-piece bf16 mode 128 where more than 64 output channels exist (round 5, conv_*_db_kernel_4_512_32): one product per
// operand pair makes that mode LDS- and load-issue-bound in the 64-channel tile (one fragment read per MFMA); a wave tile of 128 channels x
// 64 pixels reads 6 fragments per 8 MFMAs and stages the patch once per 128 output channels.  UZ_COT128=0 switches it off.
int np_mode();
inline bool cot128_enabled() { static const bool on = !(getenv("UZ_COT128") && atoi(getenv("UZ_COT128")) == 0); return on; }
inline int tile_cot(int Kc, int Mc, int W) {
    if (small_geo(W) || Mc <= 32 || Kc <= 32) return 32;
    return (np_mode() == 1 && Mc > 64 && cot128_enabled()) ? 128 : 64;
}

}  // namespace

namespace { int np_mode() { return uz::conv_math_mode() == 3 ? 1 : 2; } }
namespace uz {

// planes per operand under the current math mode: 1 = bf16 (UZ_CONV_MATH=bf16), 2 = fp32-accurate fp16 split
int conv_np() { return conv_math_mode() == 3 ? 1 : 2; }
long long* debug_stamps = nullptr;
extern "C" void uz_debug_stamps(void* buf) { debug_stamps = static_cast<long long*>(buf); }

// Device flag word: bit FLAG_X_BOUND / FLAG_DY_BOUND = an activation / gradient tensor exceeded the magnitude bound a split-fp16
// kernel was given by more than 4x, bit FLAG_W_BOUND = a weight did (the values were clamped: results are wrong but finite).
int* dev_flags_ptr() {
    static int* ptr = nullptr;
    static bool tried = false;
    if (!tried) {
        tried = true;
        if (hipMalloc(&ptr, sizeof(int)) != hipSuccess || hipMemset(ptr, 0, sizeof(int)) != hipSuccess) ptr = nullptr;
    }
    return ptr;
}
extern "C" int uz_device_flags(int* out, int clear, void* stream) {
    UZ_REQUIRE(out, "device_flags: null output");
    int* d = dev_flags_ptr();
    *out = 0;
    if (!d) return 0;
    if (hipStreamSynchronize(S(stream)) != hipSuccess) return fail("device_flags: stream synchronize failed");
    if (hipMemcpy(out, d, sizeof(int), hipMemcpyDeviceToHost) != hipSuccess) return fail("device_flags: copy failed");
    if (clear && *out && hipMemset(d, 0, sizeof(int)) != hipSuccess) return fail("device_flags: clear failed");
    return 0;
}

// Which layers take the split-fp16 path: 3x3, enough channels for a dense contraction and enough tiles to occupy
// the chip.  Planes wider than 32 use 16 x 32 tiles; 32 x 32 and 16 x 16 planes use 16 x 16 tiles.
bool conv_split_ok(int Kc, int Mc, int N, int H, int W, int ks, int /*dgrad: same policy for both directions (tiny planes forward-only was measured slower)*/) {
    // UZ_CONV_MATH: "f32" = fp32 MFMA only; "split" = split-fp16 on every 3x3 shape (tests); default = where it pays
    const int mode = conv_math_mode();
    if (!mode || ks != 3) return false;
    if (mode == 2) return true;
    constexpr int min_grid = 64;        // measured: 128 -> 128 @ 32 x 32 (128 tiles) already gains 20 % over the fp32 kernel
    if (!small_geo(W)) {
        // fewer than 32 output channels only on big tensors (the 2-channel latent gradients of a volume: a 32-wide tile that is 6 % full
        // still beats the fp32 pipe 3x there: 0.96 -> 0.3 ms at 192 -> 2 @ 128 x 128 x 64)
        // fewer than 16 input channels (one zero-padded chunk) likewise on big tensors only: 12 -> 32 @ 128 x 128 x 64 forward 132 -> 44 us
        static const int kcmin = getenv("UZ_SPLIT_KCMIN") ? atoi(getenv("UZ_SPLIT_KCMIN")) : 5;
        if (H < 16 || (Kc < 16 && (Kc < kcmin || (long long)N * H * W < 262144)) || (Mc < 32 && (long long)N * H * W < 262144)) return false;
        const long long grid = (long long)N * ((H + TH - 1) / TH) * ((W + 31) / 32) * ((Mc + 63) / 64);
        return grid >= min_grid;
    }
    if (W >= 16 && H >= 16 && Kc >= 32 && Mc >= 32) {       // 16 x 16 tiles, 32 output channels per workgroup, two workgroups per CU
        const long long grid = (long long)N * ((H + TH - 1) / TH) * ((W + 15) / 16) * ((Mc + 31) / 32);
        // 40 tiles (was 128): with the chunk loop shared out over up to four workgroups even 48 tiles beat the fp32 kernel 2x
        // (576 -> 192 @ 8 x 16 x 16: 62 -> 33 us); PHiSeg at 8 / 16 images per GPU +3 % / +1.4 %, batch 32 unchanged (192 tiles)
        static const int gmin16 = getenv("UZ_SPLIT16_GRID") ? atoi(getenv("UZ_SPLIT16_GRID")) : 40;
        return grid >= gmin16;
    }
    return false;
}

// workspace = [two fallback bound slots (x, w)] [packed weight image of one direction]
constexpr size_t WS_HEAD = 2 * AMAX_FLOATS * sizeof(float);
namespace {
// Split-K for 16 x 16-tile layers that leave most CUs with one workgroup (16 x 16 planes at batch 32: 192 / 256 tiles): the
// chunk loop is shared out over up to 4 workgroups (>= 3 chunks each, ~3 workgroups per CU) - the kernel is latency bound
// there, more resident waves are what it lacks - and the partial sums are added in order by splitk_reduce.
int split_parts(int Kc, int Mc, int N, int H, int W) {
    if (!small_geo(W)) return 1;
    const int cot = tile_cot(Kc, Mc, W), nChunks = ceil_div(Kc, CK);
    const long long g = (long long)N * ceil_div(H, TH) * ceil_div(W, 16) * ceil_div(Mc, cot);
    static const int gmin = getenv("UZ_SPLITK_GRID") ? atoi(getenv("UZ_SPLITK_GRID")) : 160;
    if (g >= gmin) return 1;                      // most CUs have a workgroup: a longer chunk loop beats slabs + a reduce launch
    int S = (int)((768 + g / 2) / g);
    static const int smax = getenv("UZ_SPLITK_MAX") ? atoi(getenv("UZ_SPLITK_MAX")) : 4;
    S = S > smax ? smax : S;
    if (S > nChunks / 3) S = nChunks / 3;
    return S < 1 ? 1 : S;
}
size_t image_bytes(int Kc, int Mc, int W) {
    const int cot = tile_cot(Kc, Mc, W);
    const size_t b = (size_t)ceil_div(Kc, CK) * ceil_div(Mc, cot) * 2 * (KK * cot * CK * 2);      // sized for two planes in either mode
    return (b + 255) / 256 * 256;
}
}  // namespace
// Number of per-channel statistics partials a forward launch with fused BatchNorm statistics writes (0: this shape does not
// support them - off the split path, or its chunk loop is split over workgroups and the partial sums are only added later).
int conv_split_bn_partials(int Kc, int Mc, int N, int H, int W) {
    if (!conv_split_ok(Kc, Mc, N, H, W, 3, 0) || split_parts(Kc, Mc, N, H, W) != 1) return 0;
    const int tw = tile_w(W), nt = tw == 16 ? 256 : 512;
    return N * ceil_div(H, TH) * ceil_div(W, tw) * (nt / 4 / 64);
}
size_t conv_split_workspace(int Kc, int Mc, int N, int H, int W) {
    const int S = split_parts(Kc, Mc, N, H, W);
    return WS_HEAD + image_bytes(Kc, Mc, W) + (S > 1 ? (size_t)S * N * Mc * H * W * sizeof(float) : 0);
}

namespace {
// absmax fallback (callers that supply no bound): grid-stride max over a channel-slice view -> atomicMax into a zeroed slot
__global__ __launch_bounds__(256) void absmax_view_kernel(const float* __restrict__ x, int C, int Ctot, int HW, size_t n, float* slot) {
    float m = 0.f;
    const size_t per = (size_t)C * HW;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        const size_t b = i / per, r = i - b * per;
        m = fmaxf(m, fabsf(x[b * (size_t)Ctot * HW + r]));
    }
    uz::amax_publish(m, slot);
}
}  // namespace

int absmax_view(const float* x, int C, int Ctot, int N, int HW, float* slot, hipStream_t st) {
    const size_t n = (size_t)N * C * HW;
    int grid = (int)((n + 256 * 16 - 1) / (256 * 16));
    grid = grid < 1 ? 1 : (grid > 2048 ? 2048 : grid);
    hipLaunchKernelGGL(absmax_view_kernel, dim3(grid), dim3(256), 0, st, x, C, Ctot, HW, n, slot);
    return check_launch("absmax_view_kernel");
}
int absmax_flat(const float* x, size_t n, float* slot, hipStream_t st) {
    int grid = (int)((n + 256 * 16 - 1) / (256 * 16));
    grid = grid < 1 ? 1 : (grid > 2048 ? 2048 : grid);
    hipLaunchKernelGGL(absmax_view_kernel, dim3(grid), dim3(256), 0, st, x, 1, 1, 1, n, slot);
    return check_launch("absmax_view_kernel");
}

// x_amax / w_amax: device scalars bounding |x| and |w| (any upper bound within ~2^10 of the true maximum keeps full
// accuracy); NULL = measure here (one extra pass over the tensor: the stand-alone C-ABI path, the model plans pass
// bounds that the producing kernels maintain).  y_amax (nullable): atomic max of |y| for the next consumer.
static int conv_split_impl(const float* x, int Kc, int KcTot, const float* w, int wCi, const float* bias,
               float* y, int Mc, int McTot, int N, int H, int W, int dgrad, int relu, int accumulate,
               const float* x_amax, const float* w_amax, float* y_amax, void* workspace, const void* packed_w, float* bn_partials, hipStream_t st,
               const SplitOpts& o);
int conv_split(const float* x, int Kc, int KcTot, const float* w, int wCi, const float* bias,
               float* y, int Mc, int McTot, int N, int H, int W, int dgrad, int relu, int accumulate,
               const float* x_amax, const float* w_amax, float* y_amax, void* workspace, const void* packed_w, float* bn_partials, hipStream_t st) {
    return conv_split_impl(x, Kc, KcTot, w, wCi, bias, y, Mc, McTot, N, H, W, dgrad, relu, accumulate, x_amax, w_amax, y_amax, workspace, packed_w,
                           bn_partials, st, SplitOpts());
}
// ... with the round-4 options (uz_common.h, SplitOpts): input in split storage with up to two scale segments, and / or the
// BatchNorm-backward reduction of the unit that produced the data gradient's output folded into the epilogue
int conv_split_ex(const float* x, int Kc, int KcTot, const float* w, int wCi, const float* bias,
                  float* y, int Mc, int McTot, int N, int H, int W, int dgrad, int relu, int accumulate,
                  const float* x_amax, const float* w_amax, float* y_amax, void* workspace, const void* packed_w, float* bn_partials, hipStream_t st,
                  const SplitOpts& o) {
    return conv_split_impl(x, Kc, KcTot, w, wCi, bias, y, Mc, McTot, N, H, W, dgrad, relu, accumulate, x_amax, w_amax, y_amax, workspace, packed_w,
                           bn_partials, st, o);
}
// Data gradient with the ReLU backward of the unit that PRODUCED its output's forward twin folded into the epilogue (vanilla U-Net
// blocks, unet.py:25-30: Conv -> ReLU with no normalisation in between): dx = (a > 0) ? conv_T(dy, w) [+ dx] : 0, where `a` is that
// unit's activation (a view of Mc channels); partials (nullable) receive the per-(tile, channel) sums of dx for the unit's bias
// gradient in the layout of conv_split_bn_partials(); dx_amax receives max |dx|.  Unsplit chunk loops only.
int conv_split_dgrad_relu(const float* dy, int Kc, int KcTot, const float* w, int wCi, float* dx, int Mc, int McTot, int N, int H, int W, int accumulate,
                          const float* dy_amax, const float* w_amax, float* dx_amax, void* workspace, const void* packed_w,
                          const float* a, int aCtot, float* partials, hipStream_t st) {
    UZ_REQUIRE(a, "conv_dgrad_relu: needs the producer's activation");
    SplitOpts o;
    o.mk = 1; o.mask = a; o.maskCtot = aCtot;
    return conv_split_impl(dy, Kc, KcTot, w, wCi, nullptr, dx, Mc, McTot, N, H, W, 1, 0, accumulate, dy_amax, w_amax, dx_amax, workspace, packed_w,
                           partials, st, o);
}
static int conv_split_impl(const float* x, int Kc, int KcTot, const float* w, int wCi, const float* bias,
               float* y, int Mc, int McTot, int N, int H, int W, int dgrad, int relu, int accumulate,
               const float* x_amax, const float* w_amax, float* y_amax, void* workspace, const void* packed_w, float* bn_partials, hipStream_t st,
               const SplitOpts& o) {
    const float* relu_mask = o.mask;
    SP p;
    p.mask = o.mask; p.maskCtot = o.maskCtot; p.mk_save = o.mk_save; p.mk_relu = o.mk_relu;
    p.aff = o.aff; p.aff_relu = o.aff_relu;
    float* slots = static_cast<float*>(workspace);
    char* image = static_cast<char*>(workspace) + WS_HEAD;
    const int np = conv_np();                            // 1: bf16 single-piece operands (no scales, no bounds)
    UZ_REQUIRE(np == 1 || !packed_w || w_amax, "conv_split: a pre-packed weight image needs the bound it was scaled with");
    UZ_REQUIRE(!o.x_packed || (np == 2 && x_amax), "conv_split: input in split storage needs the two-piece mode and the bound it was scaled from");
    UZ_REQUIRE(!o.aff || (np == 2 && x_amax && !dgrad && !o.x_packed && o.mk == 0), "conv_split: the folded BatchNorm apply serves a plain forward convolution in the two-piece mode and needs the activation's bound");
    UZ_REQUIRE(!o.x_packed || o.seg_channels == 0 || (o.x_amax2 && o.seg_channels % CK == 0 && o.seg_channels < Kc),
               "conv_split: the second scale segment must start on a multiple of 16 channels inside the view and carry its bound");
    UZ_REQUIRE(o.mk != 2 || (o.mask && o.mk_save && dgrad && bn_partials), "conv_split: the folded BatchNorm reduction needs y, the statistics table and the partial rows");
    p.x_amax2 = o.x_packed && o.seg_channels > 0 ? o.x_amax2 : nullptr;
    p.segc = p.x_amax2 ? o.seg_channels / CK : -1;
    if (np == 2 && (!x_amax || !w_amax)) {
        if (hipMemsetAsync(slots, 0, WS_HEAD, st) != hipSuccess) return fail("conv_split: memset failed");
        if (!x_amax) { if (int rc = absmax_view(x, Kc, KcTot, N, H * W, slots, st)) return rc; x_amax = slots; }
        if (!w_amax) {
            // the weight view may be a row / column slice of the parameter: bound over the enclosing rows is still a bound
            const size_t nw = dgrad ? (size_t)Kc * wCi * KK : (size_t)Mc * wCi * KK;
            if (int rc = absmax_flat(w, nw, slots + AMAX_FLOATS, st)) return rc;
            w_amax = slots + AMAX_FLOATS;
        }
    }
    p.x = x; p.wp = packed_w ? static_cast<const char*>(packed_w) : image; p.bias = bias; p.y = y;
    p.x_amax = x_amax; p.w_amax = w_amax; p.y_amax = y_amax;
    p.N = N; p.H = H; p.W = W; p.HW = H * W;
    p.Cin = Kc; p.CinTot = KcTot; p.Cout = Mc; p.CoutTot = McTot;
    const int tw = tile_w(W), cot = tile_cot(Kc, Mc, W);
    p.tilesX = ceil_div(W, tw); p.tilesY = ceil_div(H, TH);
    p.relu = relu; p.accumulate = accumulate;
    p.nCoTiles = ceil_div(Mc, cot);
    p.nChunks = ceil_div(Kc, CK);
    p.kSplit = split_parts(Kc, Mc, N, H, W);
    p.cps = ceil_div(p.nChunks, p.kSplit);
    p.kSplit = ceil_div(p.nChunks, p.cps);               // no empty parts
    UZ_REQUIRE(!bn_partials || (p.kSplit == 1 && !relu && ((!accumulate && !dgrad) || relu_mask)), "conv_split: fused BatchNorm statistics need an unsplit plain forward convolution");
    UZ_REQUIRE(!relu_mask || (p.kSplit == 1 && dgrad), "conv_split: the folded ReLU / BatchNorm backward needs an unsplit data gradient");
    p.slab = reinterpret_cast<float*>(image + image_bytes(Kc, Mc, W));
    p.stamps = uz::debug_stamps;
    p.flags = dev_flags_ptr(); p.flag_bit = dgrad ? FLAG_DY_BOUND : FLAG_X_BOUND;
    p.bnpart = bn_partials;
    p.yb16 = o.y_b16;
    UZ_REQUIRE(!(o.x_b16 || o.y_b16) || (np == 1 && p.kSplit == 1 && !relu_mask), "conv_split: bf16 storage needs the single-piece bf16 mode, an unsplit chunk loop and a plain epilogue");
    const long long grid = (long long)p.tilesX * p.tilesY * N * p.nCoTiles * p.kSplit;
    UZ_REQUIRE(grid < (1ll << 31), "conv_split: grid too large");
    // per-image buffer resources and 64-bit output addressing: only ONE image's input view has to fit 32-bit byte offsets
    UZ_REQUIRE((size_t)Kc * p.HW * 4 < (1ull << 32), "conv_split: one image's input view exceeds 4 GB");
    const int rows = p.nChunks * p.nCoTiles * KK * cot;
    if (packed_w) {}                                     // packed once per step by uz_conv_pack_weights
    else if (np == 2) {
        if (dgrad) hipLaunchKernelGGL((pack_weights_kernel<true, 2>), dim3(ceil_div(rows, 256)), dim3(256), 0, st, w, image, w_amax, Mc, Kc, wCi, p.nChunks, p.nCoTiles, cot, p.flags);
        else hipLaunchKernelGGL((pack_weights_kernel<false, 2>), dim3(ceil_div(rows, 256)), dim3(256), 0, st, w, image, w_amax, Mc, Kc, wCi, p.nChunks, p.nCoTiles, cot, p.flags);
    } else {
        if (dgrad) hipLaunchKernelGGL((pack_weights_kernel<true, 1>), dim3(ceil_div(rows, 256)), dim3(256), 0, st, w, image, w_amax, Mc, Kc, wCi, p.nChunks, p.nCoTiles, cot, p.flags);
        else hipLaunchKernelGGL((pack_weights_kernel<false, 1>), dim3(ceil_div(rows, 256)), dim3(256), 0, st, w, image, w_amax, Mc, Kc, wCi, p.nChunks, p.nCoTiles, cot, p.flags);
    }
    if (int rc = check_launch("pack_weights_kernel")) return rc;
    int rc;
    if (np == 2) {
        if (tw == 16) rc = launch<1, 256, 16, 2>(p, (int)grid, st, o.mk, o.x_packed, o.x_b16);
        else rc = cot == 32 ? launch<1, 512, 32, 2>(p, (int)grid, st, o.mk, o.x_packed, o.x_b16) : launch<2, 512, 32, 2>(p, (int)grid, st, o.mk, o.x_packed, o.x_b16);
    } else {
        if (tw == 16) rc = launch<1, 256, 16, 1>(p, (int)grid, st, o.mk, o.x_packed, o.x_b16);
        else rc = cot == 32 ? launch<1, 512, 32, 1>(p, (int)grid, st, o.mk, o.x_packed, o.x_b16)
                : cot == 128 ? launch<4, 512, 32, 1>(p, (int)grid, st, o.mk, o.x_packed, o.x_b16) : launch<2, 512, 32, 1>(p, (int)grid, st, o.mk, o.x_packed, o.x_b16);
    }
    if (rc || p.kSplit == 1) return rc;
    return splitk_reduce(p.slab, p.kSplit, bias, y, Mc, McTot, N, H * W, relu, accumulate, y_amax, st);
}

}  // namespace uz

namespace {
// All weight images of a tape in ONE launch.  table: n_layers rows of 8 int64 {w (address), packed (address), Mc, Kc, wCi,
// cot, flags, first row}; a workgroup finds its layer by bisection over the first-row column.  flags bit 0 = data gradient,
// bit 1 = Conv3d weight [Cout][Cin][3][3][3] read straight into the depth-window layout of csrc/vol.hip (forward: contraction
// index k = kd Cin + ci; data gradient: k = j Cout + co with kd = 2 - j), wCi = the 3-D Cin.
template <int NP>
__global__ __launch_bounds__(256) void pack_all_kernel(const long long* __restrict__ table, int n_layers, int total_rows, const float* __restrict__ w_amax, int* flags) {
    const int e = blockIdx.x * 256 + threadIdx.x;
    if (e >= total_rows) return;
    int lo = 0, hi = n_layers - 1;
    while (lo < hi) { const int mid = (lo + hi + 1) >> 1; if ((int)table[8 * mid + 7] <= e) lo = mid; else hi = mid - 1; }
    const long long* t = table + 8 * lo;
    const float* w = reinterpret_cast<const float*>(t[0]);
    char* packed = reinterpret_cast<char*>(t[1]);
    const int Mc = (int)t[2], Kc = (int)t[3], wCi = (int)t[4], COT = (int)t[5], dgrad = (int)t[6] & 1, vol = (int)t[6] >> 1, r = e - (int)t[7];
    const int nCoTiles = (Mc + COT - 1) / COT;
    const int m = r % COT, t1 = r / COT, tap = t1 % KK, t2 = t1 / KK, coT = t2 % nCoTiles, c = t2 / nCoTiles;
    const int mo = coT * COT + m;
    float v[CK];
#pragma unroll
    for (int k = 0; k < CK; ++k) {
        const int kk = c * CK + k;
        float x = 0.f;
        if (mo < Mc && kk < Kc) {
            if (!vol) x = dgrad ? w[((size_t)kk * wCi + mo) * KK + tap] : w[((size_t)mo * wCi + kk) * KK + tap];
            else if (!dgrad) { const int kd = kk / wCi, ci = kk - kd * wCi; x = w[(((size_t)mo * wCi + ci) * 3 + kd) * KK + tap]; }
            else { const int C3o = Kc / 3, j = kk / C3o, co = kk - j * C3o; x = w[(((size_t)co * wCi + mo) * 3 + (2 - j)) * KK + tap]; }
        }
        v[k] = x;
    }
    const int tapL = dgrad ? KK - 1 - tap : tap;
    const int wplane = KK * COT * CK * 2;
    split_store16<NP>(v, NP == 2 ? uz::split_scale(uz::amax_read(w_amax)) : 1.f, packed + (size_t)(c * nCoTiles + coT) * NP * wplane + (tapL * COT + m) * 16, wplane, wplane / 2, flags);
}
}  // namespace

// Number of workgroups the matrix-pipe kernel shares one tile's chunk loop out over (kind 0 forward, 1 data gradient); > 1: partial
// sums go through fp32 slabs and a reduce launch - such shapes have no bf16-storage form (uz_conv_fwd_b16 / uz_conv_bwd_data_b16)
extern "C" int uz_conv_split_parts(int kind, int Cin, int Cout, int N, int H, int W) {
    const int Kc = kind == 1 ? Cout : Cin, Mc = kind == 1 ? Cin : Cout;
    const int nChunks = uz::ceil_div(Kc, CK);
    const int S = uz::split_parts(Kc, Mc, N, H, W);
    return uz::ceil_div(nChunks, uz::ceil_div(nChunks, S));
}
extern "C" size_t uz_conv_packed_bytes(int Cin, int Cout, int W, int dgrad) {
    return dgrad ? uz::image_bytes(Cout, Cin, W) : uz::image_bytes(Cin, Cout, W);
}
extern "C" int uz_conv_pack_rows(int Cin, int Cout, int W, int dgrad) {
    const int Kc = dgrad ? Cout : Cin, Mc = dgrad ? Cin : Cout, cot = tile_cot(Kc, Mc, W);
    return uz::ceil_div(Kc, CK) * uz::ceil_div(Mc, cot) * KK * cot;
}
extern "C" int uz_conv_pack_cot(int Cin, int Cout, int W, int dgrad) { return tile_cot(dgrad ? Cout : Cin, dgrad ? Cin : Cout, W); }
extern "C" int uz_conv_pack_weights(const int64_t* table, int n_layers, int total_rows, const float* w_amax, void* stream) {
    UZ_REQUIRE(table && w_amax && n_layers >= 0 && total_rows >= 0, "conv_pack_weights: null argument");
    if (n_layers == 0 || total_rows == 0) return 0;
    if (uz::conv_np() == 2) hipLaunchKernelGGL(pack_all_kernel<2>, dim3(uz::ceil_div(total_rows, 256)), dim3(256), 0, uz::S(stream), reinterpret_cast<const long long*>(table), n_layers, total_rows, w_amax, uz::dev_flags_ptr());
    else hipLaunchKernelGGL(pack_all_kernel<1>, dim3(uz::ceil_div(total_rows, 256)), dim3(256), 0, uz::S(stream), reinterpret_cast<const long long*>(table), n_layers, total_rows, w_amax, uz::dev_flags_ptr());
    return uz::check_launch("pack_all_kernel");
}

extern "C" int uz_absmax(const float* x, size_t n, float* slot, void* stream) {
    UZ_REQUIRE(x && slot, "absmax: null argument");
    if (n == 0) return 0;
    return uz::absmax_flat(x, n, slot, uz::S(stream));
}
