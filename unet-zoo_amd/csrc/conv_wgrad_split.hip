// 3x3 weight gradient on the fp16 matrix pipe with fp32-accurate results (split operands, see conv_split.hip /
// split_f16.h for the arithmetic: a * s = a1 + a2 with two fp16 pieces, three piece products, fp32 accumulate).
//
//   dW[co][ci][tap] = sum_{b,y,x} dY[b,co,y,x] * X[b,ci,y+dy,x+dx]
// GEMM view: M = co (A = dY), N = ci (B = X shifted by the tap), K = pixels, 16 per MFMA.
// Workgroup (512 threads) = one 64 x 64 (co x ci) tile of all nine taps and one of S pixel splits; it
// walks its pixel tiles of 4 rows x 32 columns.  Waves = (co half) x (ci half) x (tap group 5 + 4).
// LDS holds the two fp16 planes of dY [co][128 px] and of the haloed X patch [ci][6 rows][40 px]
// (row stride 80 B so that every k-step starts 16-byte aligned).  A k-step is 16 consecutive pixels of
// one row; the B fragment of tap (dy, dx) starts dx pixels (2 bytes each) past an aligned address, so
// a lane reads five dwords (b128 + b32) per row and plane and forms the three dx variants in
// registers: dx = 0 -> dwords 0..3, dx = 2 -> dwords 1..4, dx = 1 -> v_alignbit of neighbours.
// The next tile's global loads are in flight during the MFMAs; partial sums go to one slab per
// workgroup and the ordered reduce kernels of conv_wgrad.hip add them (bitwise reproducible).
#include <stdlib.h>
#include <string.h>
#include <type_traits>
#include "uz_common.h"
#include "split_f16.h"

namespace {

using uz::f32x16; using uz::f32x4; using uz::f16x8; using uz::u32x4; using uz::split2;


constexpr int NT = 512, PT = 128;                        // 128 pixels per tile
// NP = planes per operand: 2 = fp32-accurate fp16 split (three products), 1 = one bf16 piece, one product (UZ_CONV_MATH=bf16)
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
template <int NP> __device__ __forceinline__ void pieces(float v0, float v1, unsigned (&out)[NP]) {
    if constexpr (NP == 2) split2(v0, v1, out[0], out[1]);
    else out[0] = __builtin_bit_cast(unsigned, __builtin_convertvector(uz::f32x2{v0, v1}, bf16x2));
}
template <int NP> __device__ __forceinline__ f32x16 mma(f32x16 t, const u32x4 (&a)[NP], const u32x4 (&b)[NP]) {
    if constexpr (NP == 2) {
#ifndef UZ_EXP_PRODUCTS
#define UZ_EXP_PRODUCTS 3          // (2 / 1: timing-only experiment builds, see conv_split.hip)
#endif
        if constexpr (UZ_EXP_PRODUCTS >= 3) t = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a[1]), __builtin_bit_cast(f16x8, b[0]), t, 0, 0, 0);
        if constexpr (UZ_EXP_PRODUCTS >= 2) t = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a[0]), __builtin_bit_cast(f16x8, b[1]), t, 0, 0, 0);
        t = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a[0]), __builtin_bit_cast(f16x8, b[0]), t, 0, 0, 0);
    } else {
        t = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a[0]), __builtin_bit_cast(bf16x8, b[0]), t, 0, 0, 0);
    }
    return t;
}
constexpr int DYROW = PT * 2 + 16;                       // bytes per co row of one dY plane (272: 16-byte aligned, skewed banks)
constexpr int XCH = 496;                                 // bytes per ci of one X plane: 6 rows x 80 B or 10 rows x 48 B, + 16
// tile geometry: 4 rows x 32 columns (planes whose width is a multiple of 32) or 8 rows x 16 columns (16-wide planes)
template <int TWv> struct WGeo {
    static constexpr int TW = TWv, TH = PT / TWv;
    static constexpr int XROW = (TWv + 8) * 2;           // bytes per patch row: TW + 2 pixels, padded to a 16-byte multiple (80 / 48)
    static constexpr int QROWX = TWv / 4;                // staging slots per patch row: TW / 4 quads (4 pixels) + one pair (the last two pixels)
    static constexpr int PROWS = TH + 2;
    static_assert((TH + 2) * XROW + 16 == XCH, "patch image per channel");
};

struct WS {
    const float* x; const float* dy; float* slab;
    int N, H, W, HW, Cin, CinTot, Cout, CoutTot;
    int tilesX, tilesY, T, S, nCoT, nCiT;
    const float* x_amax; const float* dy_amax;          // device scalars: upper bounds of |x| and |dy|
    const float* x_amax2; int seg_channels;             // XPK: input channels from seg_channels on were scaled from x_amax2 (a concat buffer's second producer); null: one bound
    long long* stamps;                                  // diagnostics (uz_debug_stamps), normally null
    int* flags;                                         // device flag word (bound violations), nullable
};


// CT = channel tile on both sides: 64 (waves = co half x ci half x tap group) or 32 (waves = pixel quarter x tap
// group; the four pixel quarters are folded through LDS at the end; 74 KB of LDS -> two workgroups per CU)
// XF / DF = storage format of X / dY: 0 fp32 values, 1 split storage (two fp16 pieces per word, NP == 2), 2 bf16 storage (2-byte
// elements, NP == 1: staged as they are - no conversion, half the bytes)
// M16: the matrix products on v_mfma_f32_16x16x32_f16 instead of 32x32x16 (64-channel tiles of the 32-wide geometry, two-piece mode): a
// wave's 32 x 32 block becomes 2 x 2 blocks of 16 x 16, one tile row (32 pixels) is one K step; same fragment bytes, same MFMA cycles,
// same accumulator registers - the chip holds a higher clock on this shape under load (MI355X_MICROARCH.md, DVFS item 7)
template <int TWv, int CT, int NP, int XF = 0, int DF = 0, int M16 = 0>
__global__ __launch_bounds__(NT) void wgrad_split_kernel(const WS p) {
    static_assert(!M16 || (CT == 64 && TWv == 32 && NP == 2), "16x16x32 form: 64-channel tiles, 32-wide rows, two-piece mode");
    constexpr bool XPK = XF == 1, DPK = DF == 1, XB = XF == 2, DB = DF == 2;
    constexpr unsigned XE = XB ? 2u : 4u, DE = DB ? 2u : 4u;      // bytes per element
    static_assert(NP == 2 || (!XPK && !DPK), "split storage is the two-piece fp16 format");
    static_assert(NP == 1 || (!XB && !DB), "bf16 storage feeds the single-piece bf16 mode");
    using GEO = WGeo<TWv>;
    constexpr int TW = GEO::TW, TH = GEO::TH, XROW = GEO::XROW, QROWX = GEO::QROWX, PROWS = GEO::PROWS;
    constexpr int QROW = TW / 4, SROW = TW / 16;          // float4 quads / 16-pixel k-steps per tile row
    constexpr int COT = CT, CIT = CT, WK = CT == 64 ? 1 : 4;
    constexpr int DYPLANE = COT * DYROW, XPLANE = CIT * XCH;
    // pixels per staging slot = one 16-byte load: 4 (fp32 values / split-storage words) or 8 (bf16 storage - half as many load
    // instructions per tile, which is what the single-product kernels are short of: a timing-only build with half the loads ran 5 - 12 % faster)
    constexpr int XPIX = XB ? 8 : 4, DPIX = DB ? 8 : 4;
    constexpr int DQ = PT / DPIX, DQROW = TW / DPIX;      // dY slots per co row of the tile, per tile row
    constexpr int XQ = TW / XPIX;                         // X slots per patch row (+ the pair slot of its last two pixels)
    constexpr int DYSLOTS = COT * DQ / NT;                // per thread: 4 / 2 (2 / 1 in bf16 storage)
    constexpr int XQSLOTS = (CIT * PROWS * XQ + NT - 1) / NT;        // per thread: 6 / 5 (64 channels), 3 (32 channels); half of that in bf16 storage
    constexpr int XPSLOTS = (CIT * PROWS + NT - 1) / NT;             // pairs per thread: 1 / 2
    extern __shared__ __attribute__((aligned(16))) char lds[];
    char* dYl = lds;
    char* Xl = lds + NP * DYPLANE;

    const int tid = threadIdx.x, lane = tid & 63, l31 = lane & 31, h = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    // waves w and w + 4 share a SIMD: give them different tap groups (5 + 4 taps) so every SIMD carries 9 taps' worth
    const int tg = (wave ^ (wave >> 2)) & 1;
    const int wn = CT == 64 ? (wave >> 1) & 1 : 0, wm = CT == 64 ? wave >> 2 : 0;
    const int wk = CT == 64 ? 0 : wave >> 1;             // pixel quarter: k-steps s with s % 4 == wk
    const int wid = uz::xcd_remap(blockIdx.x, gridDim.x);
    const int nTile = p.nCoT * p.nCiT;
    const int split = wid / nTile, tl = wid - split * nTile;
    const int co0 = (tl / p.nCiT) * COT, ci0 = (tl % p.nCiT) * CIT;

    // the pad words of the patch rows (columns 34..39) and row tails are never read by a valid fragment
    // except as dword 4 of the last k-step half: zero the whole image once so they are finite
    for (int i = tid * 16; i < NP * DYPLANE + NP * XPLANE; i += NT * 16) *reinterpret_cast<u32x4*>(lds + i) = u32x4{0u, 0u, 0u, 0u};

    const __amdgpu_buffer_rsrc_t rdy = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(p.dy), 0, (unsigned)(((size_t)(p.N - 1) * p.CoutTot + p.Cout) * p.HW * DE), 0x00020000);
    const __amdgpu_buffer_rsrc_t rxx = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(p.x), 0, (unsigned)(((size_t)(p.N - 1) * p.CinTot + p.Cin) * p.HW * XE), 0x00020000);

    // ---- staging maps
    // dY: float4 e = tid + i * 512 -> co = e / 32, quad q = e % 32 (row q / QROW, columns 4 (q % QROW) ..); the
    // per-slot offsets are recomputed per tile (a handful of integer ops) rather than held in registers
    f32x4 dreg[DYSLOTS];
    f32x4 xq[XQSLOTS];
    float xp[XPSLOTS][2];
    // The loads of the next tile are issued in NPARTS portions, one per k-step of the MFMA loop (part < 0: all at once, prologue):
    // issued as one block behind the barrier, their address arithmetic (a few hundred VALU instructions per wave) kept BOTH waves
    // of every SIMD off the matrix pipe at the start of each tile (cycle stamps: 15 k cycles per tile against 6.9 k of MFMAs).
    constexpr int NPARTS = PT / 16 / WK;
    auto gload = [&](int t, int part) __attribute__((always_inline)) {
        const int txi = t % p.tilesX, t2 = t / p.tilesX;
        const int x0 = txi * TW, y0 = (t2 % p.tilesY) * TH, b0 = t2 / p.tilesY;
        const unsigned dbase = DE * (unsigned)((b0 * p.CoutTot + co0) * p.HW + y0 * p.W + x0);
#pragma unroll
        for (int i = 0; i < DYSLOTS; ++i) {
            if (part >= 0 && i % NPARTS != part % NPARTS) continue;
            const int e = tid + i * NT, co = e / DQ, q = e % DQ, row = q / DQROW, c4 = (q % DQROW) * DPIX;
            const unsigned m = ((y0 + row) < p.H && (co0 + co) < p.Cout) ? 0u : 0xFFFFFFFFu;      // all-ones: the range check returns 0
            // (DB: eight bf16 pixels in the same 16 bytes, staged as they are)
            dreg[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rdy, (dbase + DE * (unsigned)(co * p.HW + row * p.W + c4)) | m, 0, 0));
        }
        // X patch: a quad slot is 4 consecutive patch pixels of one row (patch pixels 4 q .. 4 q + 3 = image columns x0 - 1 + 4 q ..):
        // one 16-byte load (dword aligned; a wave's lanes read one contiguous run) whose LDS image is one aligned 8-byte store per
        // plane; the last two pixels of every row are pair slots.  Quads and pairs have their own slot indices, so every lane of a
        // wave executes the same instructions.  A quarter of the load / store instructions and address arithmetic of pair-only staging.
        const int xbase = (b0 * p.CinTot + ci0) * p.HW + y0 * p.W + x0;
#pragma unroll
        for (int i = 0; i < XQSLOTS; ++i) {
            if (part >= 0 && (i + DYSLOTS) % NPARTS != part % NPARTS) continue;
            const int e = tid + i * NT;
            const int q = e % XQ, r = e / XQ, ci = r / PROWS, prow = r - ci * PROWS;
            const int yy = y0 + prow - 1;
            const bool rowok = e < CIT * PROWS * XQ && (ci0 + ci) < p.Cin && yy >= 0 && yy < p.H;
            const int sh = (q == 0 && x0 == 0) ? 1 : 0;           // image column -1: load columns 0.. and shift (never reads before the row)
            const int off = xbase + ci * p.HW + (prow - 1) * p.W + XPIX * q - 1 + sh;
            // (the shift itself happens in lstore: a use of the loaded value here would wait for the load inside the MFMA loop)
            // (XB: eight bf16 pixels, a 2-byte aligned 16-byte load - the slot starts at an odd column)
            xq[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rxx, rowok ? XE * (unsigned)off : 0xFFFFFFFFu, 0, 0));
        }
#pragma unroll
        for (int i = 0; i < XPSLOTS; ++i) {
            if (part >= 0 && (i + DYSLOTS + XQSLOTS) % NPARTS != part % NPARTS) continue;
            const int e = tid + i * NT;
            const int ci = e / PROWS, prow = e - ci * PROWS;
            const int yy = y0 + prow - 1;
            const bool rowok = e < CIT * PROWS && (ci0 + ci) < p.Cin && yy >= 0 && yy < p.H;
            const int off = xbase + ci * p.HW + (prow - 1) * p.W + TW - 1;        // image columns x0 + TW - 1 (inside the image) and x0 + TW
            if constexpr (XB) {
                xp[i][0] = __builtin_bit_cast(float, (unsigned)(unsigned short)__builtin_amdgcn_raw_buffer_load_b16(rxx, rowok ? XE * (unsigned)off : 0xFFFFFFFFu, 0, 0));
                xp[i][1] = __builtin_bit_cast(float, (unsigned)(unsigned short)__builtin_amdgcn_raw_buffer_load_b16(rxx, (rowok && x0 + TW < p.W) ? XE * (unsigned)(off + 1) : 0xFFFFFFFFu, 0, 0));
            } else {
                xp[i][0] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rxx, rowok ? 4u * (unsigned)off : 0xFFFFFFFFu, 0, 0));
                xp[i][1] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rxx, (rowok && x0 + TW < p.W) ? 4u * (unsigned)(off + 1) : 0xFFFFFFFFu, 0, 0));
            }
        }
    };
    const float sdy = NP == 2 ? uz::split_scale(uz::amax_read(p.dy_amax)) : 1.f, sx = NP == 2 ? uz::split_scale(uz::amax_read(p.x_amax)) : 1.f;
    // XPK / DPK: X / dY arrive as split storage (split_f16.h): every word already holds the two fp16 pieces of its scaled value
    // fp32 operands of the two-piece mode: every staged value is tested against the fp16 range, the predicates are accumulated over
    // ALL tiles and raise the device flag word once behind the tile loop (round 5; until round 4 only the first tile was tested)
    bool xbad = false, dbad = false;
    auto xpieces = [&](float v0, float v1, unsigned (&out)[NP]) __attribute__((always_inline)) {
        if constexpr (XPK) uz::packed_pair(__builtin_bit_cast(unsigned, v0), __builtin_bit_cast(unsigned, v1), out[0], out[1]);
        else {
            if constexpr (NP == 2) xbad |= uz::bound_violated(v0 * sx, v1 * sx);
            pieces<NP>(v0 * sx, v1 * sx, out);
        }
    };
    auto dpieces = [&](float v0, float v1, unsigned (&out)[NP]) __attribute__((always_inline)) {
        if constexpr (DPK) uz::packed_pair(__builtin_bit_cast(unsigned, v0), __builtin_bit_cast(unsigned, v1), out[0], out[1]);
        else {
            if constexpr (NP == 2) dbad |= uz::bound_violated(v0 * sdy, v1 * sdy);
            pieces<NP>(v0 * sdy, v1 * sdy, out);
        }
    };
    auto lstore = [&](int t) __attribute__((always_inline)) {
        const bool left_edge = t % p.tilesX == 0;               // tile column 0: the quads with q == 0 were loaded one column to the right
#pragma unroll
        for (int i = 0; i < DYSLOTS; ++i) {
            const int e = tid + i * NT, co = e / DQ, q = e % DQ;
            char* d = dYl + co * DYROW + ((q / DQROW) * TW + (q % DQROW) * DPIX) * 2;
            if constexpr (DB) *reinterpret_cast<u32x4*>(d) = __builtin_bit_cast(u32x4, dreg[i]);
            else {
                unsigned pa[NP], pb[NP];
                dpieces(dreg[i][0], dreg[i][1], pa);
                dpieces(dreg[i][2], dreg[i][3], pb);
#pragma unroll
                for (int pl = 0; pl < NP; ++pl) *reinterpret_cast<uint2*>(d + pl * DYPLANE) = make_uint2(pa[pl], pb[pl]);
            }
        }
#pragma unroll
        for (int i = 0; i < XQSLOTS; ++i) {
            const int e = tid + i * NT;
            if (e < CIT * PROWS * XQ) {
                const int q = e % XQ, r = e / XQ, ci = r / PROWS, prow = r - ci * PROWS;
                char* d = Xl + ci * XCH + prow * XROW + q * (2 * XPIX);
                if constexpr (XB) {
                    const u32x4 w = __builtin_bit_cast(u32x4, xq[i]);
                    // left image edge: columns 0..6 were loaded into pixels 0..6 - move them one pixel up, pixel 0 = padding
                    const u32x4 ws = u32x4{w.x << 16, __builtin_amdgcn_alignbit(w.y, w.x, 16), __builtin_amdgcn_alignbit(w.z, w.y, 16), __builtin_amdgcn_alignbit(w.w, w.z, 16)};
                    *reinterpret_cast<u32x4*>(d) = (left_edge && q == 0) ? ws : w;
                } else {
                    unsigned pa[NP], pb[NP];
                    const f32x4 v = (left_edge && q == 0) ? f32x4{0.f, xq[i][0], xq[i][1], xq[i][2]} : xq[i];
                    xpieces(v[0], v[1], pa);
                    xpieces(v[2], v[3], pb);
#pragma unroll
                    for (int pl = 0; pl < NP; ++pl) *reinterpret_cast<uint2*>(d + pl * XPLANE) = make_uint2(pa[pl], pb[pl]);
                }
            }
        }
#pragma unroll
        for (int i = 0; i < XPSLOTS; ++i) {
            const int e = tid + i * NT;
            if (e < CIT * PROWS) {
                const int ci = e / PROWS, prow = e - ci * PROWS;
                unsigned pa[NP];
                if constexpr (XB) pa[0] = __builtin_bit_cast(unsigned, xp[i][0]) | (__builtin_bit_cast(unsigned, xp[i][1]) << 16);
                else xpieces(xp[i][0], xp[i][1], pa);
                char* d = Xl + ci * XCH + prow * XROW + QROWX * 8;
#pragma unroll
                for (int pl = 0; pl < NP; ++pl) *reinterpret_cast<unsigned*>(d + pl * XPLANE) = pa[pl];
            }
        }
    };

    const char* Ab = dYl + (wm * 32 + l31) * DYROW + h * 16;
    const char* Bb = Xl + (wn * 32 + l31) * XCH + h * 16;

    auto run = [&](auto ntap_c, auto tap0_c) __attribute__((always_inline)) {
        constexpr int NTAP = decltype(ntap_c)::value, TAP0 = decltype(tap0_c)::value;
        constexpr int DY0 = TAP0 / 3;                        // first patch-row offset this tap group needs (0 or 1)
        f32x16 acc[M16 ? 1 : NTAP];
        f32x4 acc16[M16 ? NTAP : 1][2][2];
#pragma unroll
        for (int k = 0; k < (M16 ? 1 : NTAP); ++k)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[k][r] = 0.f;
#pragma unroll
        for (int k = 0; k < (M16 ? NTAP : 1); ++k)
#pragma unroll
            for (int r = 0; r < 4; ++r) { acc16[k][0][0][r] = 0.f; acc16[k][0][1][r] = 0.f; acc16[k][1][0][r] = 0.f; acc16[k][1][1][r] = 0.f; }
        const int l15 = lane & 15, kg = lane >> 4;
        const char* Ab16 = dYl + (wm * 32 + l15) * DYROW + kg * 16;
        const char* Bb16 = Xl + (wn * 32 + l15) * XCH + kg * 16;
        int t = split;
        long long st0 = 0, st1 = 0, stl = 0, rt0 = 0;
        if (p.stamps) { st0 = __builtin_amdgcn_s_memtime(); rt0 = __builtin_amdgcn_s_memrealtime(); }
        if (t < p.T) gload(t, -1);
        for (; t < p.T; t += p.S) {
            long long ta = 0;
            if (p.stamps) ta = __builtin_amdgcn_s_memtime();
            __syncthreads();                   // every wave finished the MFMAs of the previous tile
            lstore(t);
            __syncthreads();
            if (p.stamps) { const long long tb = __builtin_amdgcn_s_memtime(); if (t == split) st1 = tb; else stl += tb - ta; }
            if constexpr (M16) {
#pragma unroll
                for (int sr = 0; sr < TH; ++sr) {            // one tile row = 32 pixels = one K step
                    gload(t + p.S, 2 * sr);
                    gload(t + p.S, 2 * sr + 1);
                    u32x4 a[2][NP];
#pragma unroll
                    for (int sa = 0; sa < 2; ++sa)
#pragma unroll
                        for (int q = 0; q < NP; ++q) a[sa][q] = *reinterpret_cast<const u32x4*>(Ab16 + q * DYPLANE + sa * 16 * DYROW + sr * TW * 2);
#pragma unroll
                    for (int d = 0; d < 2; ++d)
#pragma unroll
                        for (int sb = 0; sb < 2; ++sb) {
                            u32x4 v[NP];
                            unsigned v4[NP];
#pragma unroll
                            for (int q = 0; q < NP; ++q) {
                                const char* src = Bb16 + q * XPLANE + sb * 16 * XCH + (sr + DY0 + d) * XROW;
                                v[q] = *reinterpret_cast<const u32x4*>(src);
                                v4[q] = *reinterpret_cast<const unsigned*>(src + 16);
                            }
#pragma unroll
                            for (int dx = 0; dx < 3; ++dx) {
                                const int tap = (DY0 + d) * 3 + dx, k = tap - TAP0;
                                if (k >= 0 && k < NTAP) {
                                    u32x4 b[NP];
#pragma unroll
                                    for (int q = 0; q < NP; ++q) {
                                        if (dx == 0) b[q] = v[q];
                                        else if (dx == 2) b[q] = u32x4{v[q].y, v[q].z, v[q].w, v4[q]};
                                        else b[q] = u32x4{__builtin_amdgcn_alignbit(v[q].y, v[q].x, 16), __builtin_amdgcn_alignbit(v[q].z, v[q].y, 16),
                                                          __builtin_amdgcn_alignbit(v[q].w, v[q].z, 16), __builtin_amdgcn_alignbit(v4[q], v[q].w, 16)};
                                    }
                                    const int kc = k < 0 ? 0 : (k >= NTAP ? NTAP - 1 : k);
#pragma unroll
                                    for (int sa = 0; sa < 2; ++sa) {
                                        f32x4 c = acc16[kc][sa][sb];
                                        c = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, a[sa][1]), __builtin_bit_cast(f16x8, b[0]), c, 0, 0, 0);
                                        c = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, a[sa][0]), __builtin_bit_cast(f16x8, b[1]), c, 0, 0, 0);
                                        c = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, a[sa][0]), __builtin_bit_cast(f16x8, b[0]), c, 0, 0, 0);
                                        acc16[kc][sa][sb] = c;
                                    }
                                }
                            }
                        }
                }
            } else
#pragma unroll
            for (int si = 0; si < PT / 16 / WK; ++si) {
                // A portion of the next tile's loads per k-step, in flight during the MFMAs.  Unconditional: behind the last tile the
                // image index is >= N, every offset fails the buffer range check and the loads return 0 without touching memory
                // (a branch here fenced the instruction scheduler: 950 -> 883 us on 224 -> 128 @ 128 x 128).
                gload(t + p.S, si);
                const int s = si * WK + wk;                 // WK == 1: compile-time; WK == 4: wave-uniform
                const int srow = s / SROW, scol = (s % SROW) * 16;
                u32x4 a[NP];
#pragma unroll
                for (int q = 0; q < NP; ++q) a[q] = *reinterpret_cast<const u32x4*>(Ab + q * DYPLANE + (srow * TW + scol) * 2);
                // B fragments: per patch row d and plane q five dwords (b128 + b32); the three dx variants of a row are
                // formed right before their MFMAs (dx = 0: dwords 0..3, dx = 2: dwords 1..4, dx = 1: v_alignbit of
                // neighbours) so that only one row's raw dwords and one shifted triple are live at a time
#pragma unroll
                for (int d = 0; d < 2; ++d) {
                    u32x4 v[NP];
                    unsigned v4[NP];
#pragma unroll
                    for (int q = 0; q < NP; ++q) {
                        const char* src = Bb + q * XPLANE + (srow + DY0 + d) * XROW + scol * 2;
                        v[q] = *reinterpret_cast<const u32x4*>(src);
                        v4[q] = *reinterpret_cast<const unsigned*>(src + 16);
                    }
#pragma unroll
                    for (int dx = 0; dx < 3; ++dx) {
                        constexpr int dummy = 0; (void)dummy;
                        const int tap = (DY0 + d) * 3 + dx, k = tap - TAP0;
                        if (k >= 0 && k < NTAP) {
                            u32x4 b[NP];
#pragma unroll
                            for (int q = 0; q < NP; ++q) {
                                if (dx == 0) b[q] = v[q];
                                else if (dx == 2) b[q] = u32x4{v[q].y, v[q].z, v[q].w, v4[q]};
                                else b[q] = u32x4{__builtin_amdgcn_alignbit(v[q].y, v[q].x, 16), __builtin_amdgcn_alignbit(v[q].z, v[q].y, 16),
                                                  __builtin_amdgcn_alignbit(v[q].w, v[q].z, 16), __builtin_amdgcn_alignbit(v4[q], v[q].w, 16)};
                            }
                            const int kc = k < 0 ? 0 : (k >= NTAP ? NTAP - 1 : k);      // (folds: the loops are fully unrolled)
                            acc[kc] = mma<NP>(acc[kc], a, b);
                        }
                    }
                }
            }
        }
        if (NP == 2) { uz::raise_flag(p.flags, xbad, uz::FLAG_X_BOUND); uz::raise_flag(p.flags, dbad, uz::FLAG_DY_BOUND); }
        long long st2 = 0;
        if (p.stamps) st2 = __builtin_amdgcn_s_memtime();
        if constexpr (WK > 1) {
            // fold the WK pixel-quarter partial sums pairwise through LDS (fixed order ((0+2)+(1+3)), three taps per
            // round so that the dead staging area suffices); group 0 writes the slab
            constexpr int RT = 3;
            float* red = reinterpret_cast<float*>(lds);
#pragma unroll
            for (int stride = WK / 2; stride >= 1; stride >>= 1) {
#pragma unroll
                for (int k0 = 0; k0 < 5; k0 += RT) {
                    __syncthreads();
                    if (wk >= stride && wk < 2 * stride) {
                        float* dstp = red + (size_t)(((wk - stride) * 2 + tg) * RT) * 16 * 64 + lane;
#pragma unroll
                        for (int kk = 0; kk < RT; ++kk)
                            if (k0 + kk < NTAP) {
#pragma unroll
                                for (int r = 0; r < 16; ++r) dstp[(kk * 16 + r) * 64] = acc[k0 + kk][r];
                            }
                    }
                    __syncthreads();
                    if (wk < stride) {
                        const float* srcp = red + (size_t)((wk * 2 + tg) * RT) * 16 * 64 + lane;
#pragma unroll
                        for (int kk = 0; kk < RT; ++kk)
                            if (k0 + kk < NTAP) {
#pragma unroll
                                for (int r = 0; r < 16; ++r) acc[k0 + kk][r] += srcp[(kk * 16 + r) * 64];
                            }
                    }
                }
            }
            if (wk != 0) return;
        }
        float* out = p.slab + (size_t)split * 9 * p.Cout * p.Cin;
        const float inv_dy = NP == 2 ? uz::split_inv_scale(uz::amax_read(p.dy_amax)) : 1.f;
        if constexpr (M16) {
#pragma unroll
            for (int sb = 0; sb < 2; ++sb) {
                const int ci = ci0 + wn * 32 + sb * 16 + l15;
                const float inv_x = uz::split_inv_scale(uz::amax_read((XPK && p.x_amax2 && ci >= p.seg_channels) ? p.x_amax2 : p.x_amax));
#pragma unroll
                for (int k = 0; k < NTAP; ++k)
#pragma unroll
                    for (int sa = 0; sa < 2; ++sa)
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            const int co = co0 + wm * 32 + sa * 16 + 4 * kg + r;
                            if (co < p.Cout && ci < p.Cin) out[((size_t)(TAP0 + k) * p.Cout + co) * p.Cin + ci] = acc16[k][sa][sb][r] * inv_dy * inv_x;
                        }
            }
        }
        const int ci = ci0 + wn * 32 + l31;
        // exact powers of two; with two-segment split storage the input scale is a property of the column (ci) - K is pixels here
        const float inv_x = NP == 2 ? uz::split_inv_scale(uz::amax_read((XPK && p.x_amax2 && ci >= p.seg_channels) ? p.x_amax2 : p.x_amax)) : 1.f;
        if constexpr (!M16) {
#pragma unroll
        for (int k = 0; k < NTAP; ++k)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int co = co0 + wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
                if (co < p.Cout && ci < p.Cin) out[((size_t)(TAP0 + k) * p.Cout + co) * p.Cin + ci] = acc[k][r] * inv_dy * inv_x;
            }
        }
        if (p.stamps) {
            __builtin_amdgcn_s_waitcnt(0);
            const long long st3 = __builtin_amdgcn_s_memtime(), rt1 = __builtin_amdgcn_s_memrealtime();
            if (tid == 0 && blockIdx.x < 4096) {
                long long* o = p.stamps + 8 * blockIdx.x;
                o[0] = st0; o[1] = st1; o[2] = stl; o[3] = st2; o[4] = st3; o[5] = rt0; o[6] = rt1; o[7] = (p.T - split + p.S - 1) / p.S;
            }
        }
    };
    if (tg == 0) run(std::integral_constant<int, 5>{}, std::integral_constant<int, 0>{});
    else run(std::integral_constant<int, 4>{}, std::integral_constant<int, 5>{});
}


// ---------------------------------------------------------------------------------------------------------------------------------
// Second form (round 6): 256 threads, one wave per SIMD and workgroup, TWO workgroups per CU.
// A workgroup still owns a 64 x 64 (co x ci) tile, but each of its four waves carries ALL NINE taps of a 32 x 32 block (144
// accumulator registers) and the pixel tile is 64 pixels (2 rows x 32 columns or 4 rows x 16):
//   * a k-step feeds 27 MFMAs from 2 dY fragment reads + 6 X fragment reads (the first form: 13.5 MFMAs from 2 + 4): 0.30 instead of
//     0.44 ds_read_b128 per MFMA, and the dY fragment is read once for nine taps instead of once per tap group;
//   * 61 KB of LDS and <= 256 registers per workgroup: two INDEPENDENT workgroups share a CU, so the staging phase of one (global ->
//     registers -> split -> LDS, behind its barriers) runs under the MFMA phase of the other - the first form's eight waves all stand
//     in the same phase; launched with one workgroup per CU it leaves half of every CU (registers, LDS, wave slots) to the other lanes'
//     kernels instead of nothing on half the chip.
// The price: the haloed X patch is 4 rows for 2 (2x the X bytes from L2 instead of 1.5x).  Same slab layout, same staging helpers.
constexpr int NT9 = 256, PT9 = 64;
constexpr int DYROW9 = PT9 * 2 + 16;                     // 144 B = 36 dwords per co row: sixteen lanes' b128 reads land on sixteen different 16-byte slots
template <int TWv> struct WGeo9 {
    static constexpr int TW = TWv, TH = PT9 / TWv;
    static constexpr int XROW = (TWv + 8) * 2;           // 80 / 48 B per patch row
    static constexpr int PROWS = TH + 2;                 // 4 / 6
    static constexpr int XCH = PROWS * XROW + 16;        // 336 / 304 B = 84 / 76 dwords per ci: 4 * odd -> conflict-free b128 reads
};

template <int TWv, int NP, int XF, int DF>
__global__ __launch_bounds__(NT9, 2) void wgrad9_kernel(const WS p) {
    constexpr bool XPK = XF == 1, DPK = DF == 1;
    static_assert(NP == 2 || (!XPK && !DPK), "split storage is the two-piece fp16 format");
    using GEO = WGeo9<TWv>;
    constexpr int TW = GEO::TW, TH = GEO::TH, XROW = GEO::XROW, PROWS = GEO::PROWS, XCH = GEO::XCH;
    constexpr int SROW = TW / 16, KSTEPS = PT9 / 16;
    constexpr int CT = 64;
    constexpr int DYPLANE = CT * DYROW9, XPLANE = CT * XCH;
    constexpr int DQ = PT9 / 4, DQROW = TW / 4, XQ = TW / 4;
    constexpr int DYSLOTS = CT * DQ / NT9;                            // 4
    constexpr int XQSLOTS = (CT * PROWS * XQ + NT9 - 1) / NT9;        // 8 / 6
    constexpr int XPSLOTS = (CT * PROWS + NT9 - 1) / NT9;             // 1 / 2
    extern __shared__ __attribute__((aligned(16))) char lds[];
    char* dYl = lds;
    char* Xl = lds + NP * DYPLANE;

    const int tid = threadIdx.x, lane = tid & 63, l31 = lane & 31, h = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wn = wave & 1, wm = wave >> 1;
    const int wid = uz::xcd_remap(blockIdx.x, gridDim.x);
    const int nTile = p.nCoT * p.nCiT;
    const int split = wid / nTile, tl = wid - split * nTile;
    const int co0 = (tl / p.nCiT) * CT, ci0 = (tl % p.nCiT) * CT;

    for (int i = tid * 16; i < NP * DYPLANE + NP * XPLANE; i += NT9 * 16) *reinterpret_cast<u32x4*>(lds + i) = u32x4{0u, 0u, 0u, 0u};

    const __amdgpu_buffer_rsrc_t rdy = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(p.dy), 0, (unsigned)(((size_t)(p.N - 1) * p.CoutTot + p.Cout) * p.HW * 4u), 0x00020000);
    const __amdgpu_buffer_rsrc_t rxx = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(p.x), 0, (unsigned)(((size_t)(p.N - 1) * p.CinTot + p.Cin) * p.HW * 4u), 0x00020000);

    f32x4 dreg[DYSLOTS];
    f32x4 xq[XQSLOTS];
    float xp[XPSLOTS][2];
    // the next tile's loads are issued in KSTEPS portions, one per k-step of the MFMA loop (part < 0: all at once, prologue)
    auto gload = [&](int t, int part) __attribute__((always_inline)) {
        const int txi = t % p.tilesX, t2 = t / p.tilesX;
        const int x0 = txi * TW, y0 = (t2 % p.tilesY) * TH, b0 = t2 / p.tilesY;
        const unsigned dbase = 4u * (unsigned)((b0 * p.CoutTot + co0) * p.HW + y0 * p.W + x0);
#pragma unroll
        for (int i = 0; i < DYSLOTS; ++i) {
            if (part >= 0 && i % KSTEPS != part) continue;
            const int e = tid + i * NT9, co = e / DQ, q = e % DQ, row = q / DQROW, c4 = (q % DQROW) * 4;
            const unsigned m = ((y0 + row) < p.H && (co0 + co) < p.Cout) ? 0u : 0xFFFFFFFFu;
            dreg[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rdy, (dbase + 4u * (unsigned)(co * p.HW + row * p.W + c4)) | m, 0, 0));
        }
        const int xbase = (b0 * p.CinTot + ci0) * p.HW + y0 * p.W + x0;
#pragma unroll
        for (int i = 0; i < XQSLOTS; ++i) {
            if (part >= 0 && (i + DYSLOTS) % KSTEPS != part) continue;
            const int e = tid + i * NT9;
            const int q = e % XQ, r = e / XQ, ci = r / PROWS, prow = r - ci * PROWS;
            const int yy = y0 + prow - 1;
            const bool rowok = e < CT * PROWS * XQ && (ci0 + ci) < p.Cin && yy >= 0 && yy < p.H;
            const int sh = (q == 0 && x0 == 0) ? 1 : 0;           // image column -1: load columns 0.. and shift in lstore
            const int off = xbase + ci * p.HW + (prow - 1) * p.W + 4 * q - 1 + sh;
            xq[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rxx, rowok ? 4u * (unsigned)off : 0xFFFFFFFFu, 0, 0));
        }
#pragma unroll
        for (int i = 0; i < XPSLOTS; ++i) {
            if (part >= 0 && (i + DYSLOTS + XQSLOTS) % KSTEPS != part) continue;
            const int e = tid + i * NT9;
            const int ci = e / PROWS, prow = e - ci * PROWS;
            const int yy = y0 + prow - 1;
            const bool rowok = e < CT * PROWS && (ci0 + ci) < p.Cin && yy >= 0 && yy < p.H;
            const int off = xbase + ci * p.HW + (prow - 1) * p.W + TW - 1;
            xp[i][0] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rxx, rowok ? 4u * (unsigned)off : 0xFFFFFFFFu, 0, 0));
            xp[i][1] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rxx, (rowok && x0 + TW < p.W) ? 4u * (unsigned)(off + 1) : 0xFFFFFFFFu, 0, 0));
        }
    };
    const float sdy = NP == 2 ? uz::split_scale(uz::amax_read(p.dy_amax)) : 1.f, sx = NP == 2 ? uz::split_scale(uz::amax_read(p.x_amax)) : 1.f;
    bool xbad = false, dbad = false;
    auto xpieces = [&](float v0, float v1, unsigned (&out)[NP]) __attribute__((always_inline)) {
        if constexpr (XPK) uz::packed_pair(__builtin_bit_cast(unsigned, v0), __builtin_bit_cast(unsigned, v1), out[0], out[1]);
        else {
            if constexpr (NP == 2) xbad |= uz::bound_violated(v0 * sx, v1 * sx);
            pieces<NP>(v0 * sx, v1 * sx, out);
        }
    };
    auto dpieces = [&](float v0, float v1, unsigned (&out)[NP]) __attribute__((always_inline)) {
        if constexpr (DPK) uz::packed_pair(__builtin_bit_cast(unsigned, v0), __builtin_bit_cast(unsigned, v1), out[0], out[1]);
        else {
            if constexpr (NP == 2) dbad |= uz::bound_violated(v0 * sdy, v1 * sdy);
            pieces<NP>(v0 * sdy, v1 * sdy, out);
        }
    };
    auto lstore = [&](int t) __attribute__((always_inline)) {
        const bool left_edge = t % p.tilesX == 0;
#pragma unroll
        for (int i = 0; i < DYSLOTS; ++i) {
            const int e = tid + i * NT9, co = e / DQ, q = e % DQ;
            char* d = dYl + co * DYROW9 + ((q / DQROW) * TW + (q % DQROW) * 4) * 2;
            unsigned pa[NP], pb[NP];
            dpieces(dreg[i][0], dreg[i][1], pa);
            dpieces(dreg[i][2], dreg[i][3], pb);
#pragma unroll
            for (int pl = 0; pl < NP; ++pl) *reinterpret_cast<uint2*>(d + pl * DYPLANE) = make_uint2(pa[pl], pb[pl]);
        }
#pragma unroll
        for (int i = 0; i < XQSLOTS; ++i) {
            const int e = tid + i * NT9;
            if (e < CT * PROWS * XQ) {
                const int q = e % XQ, r = e / XQ, ci = r / PROWS, prow = r - ci * PROWS;
                char* d = Xl + ci * XCH + prow * XROW + q * 8;
                unsigned pa[NP], pb[NP];
                // (left image edge: a zero WORD is a zero value in both formats - split storage holds the two pieces of 0)
                const f32x4 v = (left_edge && q == 0) ? f32x4{0.f, xq[i][0], xq[i][1], xq[i][2]} : xq[i];
                xpieces(v[0], v[1], pa);
                xpieces(v[2], v[3], pb);
#pragma unroll
                for (int pl = 0; pl < NP; ++pl) *reinterpret_cast<uint2*>(d + pl * XPLANE) = make_uint2(pa[pl], pb[pl]);
            }
        }
#pragma unroll
        for (int i = 0; i < XPSLOTS; ++i) {
            const int e = tid + i * NT9;
            if (e < CT * PROWS) {
                const int ci = e / PROWS, prow = e - ci * PROWS;
                unsigned pa[NP];
                xpieces(xp[i][0], xp[i][1], pa);
                char* d = Xl + ci * XCH + prow * XROW + XQ * 8;
#pragma unroll
                for (int pl = 0; pl < NP; ++pl) *reinterpret_cast<unsigned*>(d + pl * XPLANE) = pa[pl];
            }
        }
    };

    const char* Ab = dYl + (wm * 32 + l31) * DYROW9 + h * 16;
    const char* Bb = Xl + (wn * 32 + l31) * XCH + h * 16;

    f32x16 acc[9];
#pragma unroll
    for (int k = 0; k < 9; ++k)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[k][r] = 0.f;
    // No prefetch across the MFMA loop: 144 accumulators + a tile of staging registers do not fit 256 registers (the prefetching build
    // spilled 42 - 67 dwords per lane).  The load latency of a tile is covered by the OTHER workgroup of the CU, which is in its MFMA
    // phase meanwhile; the loads are issued ahead of the barrier so that they also fly while the slower waves finish their MFMAs.
    long long st0 = 0, rt0 = 0, s_wait = 0, s_store = 0, s_mma = 0;
    if (p.stamps) { st0 = __builtin_amdgcn_s_memtime(); rt0 = __builtin_amdgcn_s_memrealtime(); }
    for (int t = split; t < p.T; t += p.S) {
        long long ta = 0, tb = 0, tc = 0;
        if (p.stamps) ta = __builtin_amdgcn_s_memtime();
        gload(t, -1);
        __syncthreads();                   // every wave finished the MFMAs of the previous tile
        if (p.stamps) { __builtin_amdgcn_s_waitcnt(0); tb = __builtin_amdgcn_s_memtime(); }
        lstore(t);
        __syncthreads();
        if (p.stamps) { tc = __builtin_amdgcn_s_memtime(); s_wait += tb - ta; s_store += tc - tb; }
#pragma unroll
        for (int s = 0; s < KSTEPS; ++s) {
            const int srow = s / SROW, scol = (s % SROW) * 16;
            u32x4 a[NP];
#pragma unroll
            for (int q = 0; q < NP; ++q) a[q] = *reinterpret_cast<const u32x4*>(Ab + q * DYPLANE + (srow * TW + scol) * 2);
#pragma unroll
            for (int d = 0; d < 3; ++d) {
                u32x4 v[NP];
                unsigned v4[NP];
#pragma unroll
                for (int q = 0; q < NP; ++q) {
                    const char* src = Bb + q * XPLANE + (srow + d) * XROW + scol * 2;
                    v[q] = *reinterpret_cast<const u32x4*>(src);
                    v4[q] = *reinterpret_cast<const unsigned*>(src + 16);
                }
#pragma unroll
                for (int dx = 0; dx < 3; ++dx) {
                    u32x4 b[NP];
#pragma unroll
                    for (int q = 0; q < NP; ++q) {
                        if (dx == 0) b[q] = v[q];
                        else if (dx == 2) b[q] = u32x4{v[q].y, v[q].z, v[q].w, v4[q]};
                        else b[q] = u32x4{__builtin_amdgcn_alignbit(v[q].y, v[q].x, 16), __builtin_amdgcn_alignbit(v[q].z, v[q].y, 16),
                                          __builtin_amdgcn_alignbit(v[q].w, v[q].z, 16), __builtin_amdgcn_alignbit(v4[q], v[q].w, 16)};
                    }
                    acc[d * 3 + dx] = mma<NP>(acc[d * 3 + dx], a, b);
                }
            }
        }
        if (p.stamps) s_mma += __builtin_amdgcn_s_memtime() - tc;
    }
    long long st2 = 0;
    if (p.stamps) st2 = __builtin_amdgcn_s_memtime();
    if (NP == 2) { uz::raise_flag(p.flags, xbad, uz::FLAG_X_BOUND); uz::raise_flag(p.flags, dbad, uz::FLAG_DY_BOUND); }
    float* out = p.slab + (size_t)split * 9 * p.Cout * p.Cin;
    const int ci = ci0 + wn * 32 + l31;
    const float inv_dy = NP == 2 ? uz::split_inv_scale(uz::amax_read(p.dy_amax)) : 1.f;
    const float inv_x = NP == 2 ? uz::split_inv_scale(uz::amax_read((XPK && p.x_amax2 && ci >= p.seg_channels) ? p.x_amax2 : p.x_amax)) : 1.f;
#pragma unroll
    for (int k = 0; k < 9; ++k)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int co = co0 + wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
            if (co < p.Cout && ci < p.Cin) out[((size_t)k * p.Cout + co) * p.Cin + ci] = acc[k][r] * inv_dy * inv_x;
        }
    if (p.stamps) {                        // rows of 8: {start, load-wait sum, staging sum, loop end, end, real-time start, real-time end, tiles}; MFMA sum in the row + 2048
        __builtin_amdgcn_s_waitcnt(0);
        const long long st3 = __builtin_amdgcn_s_memtime(), rt1 = __builtin_amdgcn_s_memrealtime();
        if (tid == 0 && blockIdx.x < 2048) {
            long long* o = p.stamps + 8 * blockIdx.x;
            o[0] = st0; o[1] = s_wait; o[2] = s_store; o[3] = st2; o[4] = st3; o[5] = rt0; o[6] = rt1; o[7] = (p.T - split + p.S - 1) / p.S;
            p.stamps[8 * (blockIdx.x + 2048)] = s_mma;
        }
    }
}


// ---------------------------------------------------------------------------------------------------------------------------------
// Third form (round 6): the first form's eight waves and tap groups, but TWO LDS images of a 64-pixel tile (2 rows x 32 columns; 2 x 60 KB)
// and the staging INSIDE the MFMA loop.  In-kernel stamps of the first form (224 -> 128 @ 128 x 128): 38 % of a tile's cycles are the phase
// between its two barriers in which all eight waves convert and store the next tile while the matrix pipe idles; the MFMA loop itself runs at
// 92 % of its MFMA cycles.  Here the tile j + 1 is converted and written to the other image in four portions, one per k-step of tile j, the
// registers a portion frees are reloaded at once with the same portion of tile j + 3 (two register sets in flight: a load has two tiles'
// time to arrive, as in the first form), and ONE barrier closes a tile.  Same slabs, same split count, same arithmetic.
constexpr int PTD = 64;
template <int XF, int DF>
__global__ __launch_bounds__(NT) void wgrad_db_kernel(const WS p) {
    constexpr bool XPK = XF == 1, DPK = DF == 1;
    constexpr int NP = 2, TW = 32, TH = PTD / TW, XROW = (TW + 8) * 2, PROWS = TH + 2, XCHD = PROWS * XROW + 16, DYROWD = PTD * 2 + 16;
    constexpr int CT = 64, SROW = TW / 16, KSTEPS = PTD / 16;
    constexpr int DYPLANE = CT * DYROWD, XPLANE = CT * XCHD, IMG = NP * (DYPLANE + XPLANE);
    constexpr int DQ = PTD / 4, DQROW = TW / 4, XQ = TW / 4;
    constexpr int DYSLOTS = CT * DQ / NT;                     // 2
    constexpr int XQSLOTS = CT * PROWS * XQ / NT;             // 4
    static_assert(CT * DQ % NT == 0 && CT * PROWS * XQ % NT == 0 && CT * PROWS <= NT, "staging slots");
    extern __shared__ __attribute__((aligned(16))) char lds[];

    const int tid = threadIdx.x, lane = tid & 63, l31 = lane & 31, h = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int tg = (wave ^ (wave >> 2)) & 1;
    const int wn = (wave >> 1) & 1, wm = wave >> 2;
    const int wid = uz::xcd_remap(blockIdx.x, gridDim.x);
    const int nTile = p.nCoT * p.nCiT;
    const int split = wid / nTile, tl = wid - split * nTile;
    const int co0 = (tl / p.nCiT) * CT, ci0 = (tl % p.nCiT) * CT;

    for (int i = tid * 16; i < 2 * IMG; i += NT * 16) *reinterpret_cast<u32x4*>(lds + i) = u32x4{0u, 0u, 0u, 0u};

    const __amdgpu_buffer_rsrc_t rdy = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(p.dy), 0, (unsigned)(((size_t)(p.N - 1) * p.CoutTot + p.Cout) * p.HW * 4u), 0x00020000);
    const __amdgpu_buffer_rsrc_t rxx = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(p.x), 0, (unsigned)(((size_t)(p.N - 1) * p.CinTot + p.Cin) * p.HW * 4u), 0x00020000);

    struct RS { f32x4 d[DYSLOTS]; f32x4 x[XQSLOTS]; float xp[2]; };
    // slot numbering for the portions: dY slots 0 .. 1, X quads 2 .. 5, the pair slot 6; portion = slot % KSTEPS (part < 0: all)
    auto gload = [&](RS& r, int t, int part) __attribute__((always_inline)) {
        const unsigned tm = t < p.T ? 0u : 0xFFFFFFFFu;          // past the last tile: every load fails the range check (returns 0)
        const int txi = t % p.tilesX, t2 = t / p.tilesX;
        const int x0 = txi * TW, y0 = (t2 % p.tilesY) * TH, b0 = t2 / p.tilesY;
        const unsigned dbase = 4u * (unsigned)((b0 * p.CoutTot + co0) * p.HW + y0 * p.W + x0);
#pragma unroll
        for (int i = 0; i < DYSLOTS; ++i) {
            if (part >= 0 && i % KSTEPS != part) continue;
            const int e = tid + i * NT, co = e / DQ, q = e % DQ, row = q / DQROW, c4 = (q % DQROW) * 4;
            const unsigned m = ((y0 + row) < p.H && (co0 + co) < p.Cout) ? tm : 0xFFFFFFFFu;
            r.d[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rdy, (dbase + 4u * (unsigned)(co * p.HW + row * p.W + c4)) | m, 0, 0));
        }
        const int xbase = (b0 * p.CinTot + ci0) * p.HW + y0 * p.W + x0;
#pragma unroll
        for (int i = 0; i < XQSLOTS; ++i) {
            if (part >= 0 && (i + DYSLOTS) % KSTEPS != part) continue;
            const int e = tid + i * NT;
            const int q = e % XQ, rr = e / XQ, ci = rr / PROWS, prow = rr - ci * PROWS;
            const int yy = y0 + prow - 1;
            const bool rowok = (ci0 + ci) < p.Cin && yy >= 0 && yy < p.H;
            const int sh = (q == 0 && x0 == 0) ? 1 : 0;
            const int off = xbase + ci * p.HW + (prow - 1) * p.W + 4 * q - 1 + sh;
            r.x[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rxx, (rowok ? 4u * (unsigned)off : 0xFFFFFFFFu) | tm, 0, 0));
        }
        if (part < 0 || (DYSLOTS + XQSLOTS) % KSTEPS == part) {
            const int e = tid;
            const int ci = e / PROWS, prow = e - ci * PROWS;
            const int yy = y0 + prow - 1;
            const bool rowok = e < CT * PROWS && (ci0 + ci) < p.Cin && yy >= 0 && yy < p.H;
            const int off = xbase + ci * p.HW + (prow - 1) * p.W + TW - 1;
            r.xp[0] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rxx, (rowok ? 4u * (unsigned)off : 0xFFFFFFFFu) | tm, 0, 0));
            r.xp[1] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rxx, ((rowok && x0 + TW < p.W) ? 4u * (unsigned)(off + 1) : 0xFFFFFFFFu) | tm, 0, 0));
        }
    };
    const float sdy = uz::split_scale(uz::amax_read(p.dy_amax)), sx = uz::split_scale(uz::amax_read(p.x_amax));
    bool xbad = false, dbad = false;
    auto xpieces = [&](float v0, float v1, unsigned (&out)[NP]) __attribute__((always_inline)) {
        if constexpr (XPK) uz::packed_pair(__builtin_bit_cast(unsigned, v0), __builtin_bit_cast(unsigned, v1), out[0], out[1]);
        else { xbad |= uz::bound_violated(v0 * sx, v1 * sx); pieces<NP>(v0 * sx, v1 * sx, out); }
    };
    auto dpieces = [&](float v0, float v1, unsigned (&out)[NP]) __attribute__((always_inline)) {
        if constexpr (DPK) uz::packed_pair(__builtin_bit_cast(unsigned, v0), __builtin_bit_cast(unsigned, v1), out[0], out[1]);
        else { dbad |= uz::bound_violated(v0 * sdy, v1 * sdy); pieces<NP>(v0 * sdy, v1 * sdy, out); }
    };
    auto lstore = [&](const RS& r, int t, char* img, int part) __attribute__((always_inline)) {
        char* dYl = img;
        char* Xl = img + NP * DYPLANE;
        const bool left_edge = t % p.tilesX == 0;
#pragma unroll
        for (int i = 0; i < DYSLOTS; ++i) {
            if (part >= 0 && i % KSTEPS != part) continue;
            const int e = tid + i * NT, co = e / DQ, q = e % DQ;
            char* d = dYl + co * DYROWD + ((q / DQROW) * TW + (q % DQROW) * 4) * 2;
            unsigned pa[NP], pb[NP];
            dpieces(r.d[i][0], r.d[i][1], pa);
            dpieces(r.d[i][2], r.d[i][3], pb);
#pragma unroll
            for (int pl = 0; pl < NP; ++pl) *reinterpret_cast<uint2*>(d + pl * DYPLANE) = make_uint2(pa[pl], pb[pl]);
        }
#pragma unroll
        for (int i = 0; i < XQSLOTS; ++i) {
            if (part >= 0 && (i + DYSLOTS) % KSTEPS != part) continue;
            const int e = tid + i * NT;
            const int q = e % XQ, rr = e / XQ, ci = rr / PROWS, prow = rr - ci * PROWS;
            char* d = Xl + ci * XCHD + prow * XROW + q * 8;
            unsigned pa[NP], pb[NP];
            const f32x4 v = (left_edge && q == 0) ? f32x4{0.f, r.x[i][0], r.x[i][1], r.x[i][2]} : r.x[i];
            xpieces(v[0], v[1], pa);
            xpieces(v[2], v[3], pb);
#pragma unroll
            for (int pl = 0; pl < NP; ++pl) *reinterpret_cast<uint2*>(d + pl * XPLANE) = make_uint2(pa[pl], pb[pl]);
        }
        if (part < 0 || (DYSLOTS + XQSLOTS) % KSTEPS == part) {
            if (tid < CT * PROWS) {
                const int ci = tid / PROWS, prow = tid - ci * PROWS;
                unsigned pa[NP];
                xpieces(r.xp[0], r.xp[1], pa);
                char* d = Xl + ci * XCHD + prow * XROW + XQ * 8;
#pragma unroll
                for (int pl = 0; pl < NP; ++pl) *reinterpret_cast<unsigned*>(d + pl * XPLANE) = pa[pl];
            }
        }
    };

    auto run = [&](auto ntap_c, auto tap0_c) __attribute__((always_inline)) {
        constexpr int NTAP = decltype(ntap_c)::value, TAP0 = decltype(tap0_c)::value;
        constexpr int DY0 = TAP0 / 3;
        f32x16 acc[NTAP];
#pragma unroll
        for (int k = 0; k < NTAP; ++k)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[k][r] = 0.f;
        const int aoff = (wm * 32 + l31) * DYROWD + h * 16;
        const int boff = NP * DYPLANE + (wn * 32 + l31) * XCHD + h * 16;
        // one tile: the MFMAs of tile j out of `rd`, tile j + 1 (registers r) converted into `wr`, r reloaded with tile j + 3
        auto step = [&](const char* rd, char* wr, RS& r, int t_store, int t_load) __attribute__((always_inline)) {
#pragma unroll
            for (int s = 0; s < KSTEPS; ++s) {
                lstore(r, t_store, wr, s);
                gload(r, t_load, s);
                const int srow = s / SROW, scol = (s % SROW) * 16;
                u32x4 a[NP];
#pragma unroll
                for (int q = 0; q < NP; ++q) a[q] = *reinterpret_cast<const u32x4*>(rd + aoff + q * DYPLANE + (srow * TW + scol) * 2);
#pragma unroll
                for (int d = 0; d < 2; ++d) {
                    u32x4 v[NP];
                    unsigned v4[NP];
#pragma unroll
                    for (int q = 0; q < NP; ++q) {
                        const char* src = rd + boff + q * XPLANE + (srow + DY0 + d) * XROW + scol * 2;
                        v[q] = *reinterpret_cast<const u32x4*>(src);
                        v4[q] = *reinterpret_cast<const unsigned*>(src + 16);
                    }
#pragma unroll
                    for (int dx = 0; dx < 3; ++dx) {
                        const int tap = (DY0 + d) * 3 + dx, k = tap - TAP0;
                        if (k >= 0 && k < NTAP) {
                            u32x4 b[NP];
#pragma unroll
                            for (int q = 0; q < NP; ++q) {
                                if (dx == 0) b[q] = v[q];
                                else if (dx == 2) b[q] = u32x4{v[q].y, v[q].z, v[q].w, v4[q]};
                                else b[q] = u32x4{__builtin_amdgcn_alignbit(v[q].y, v[q].x, 16), __builtin_amdgcn_alignbit(v[q].z, v[q].y, 16),
                                                  __builtin_amdgcn_alignbit(v[q].w, v[q].z, 16), __builtin_amdgcn_alignbit(v4[q], v[q].w, 16)};
                            }
                            const int kc = k < 0 ? 0 : (k >= NTAP ? NTAP - 1 : k);
                            acc[kc] = mma<NP>(acc[kc], a, b);
                        }
                    }
                }
            }
        };
        const int J = (p.T - split + p.S - 1) / p.S;          // tiles of this workgroup: split, split + S, ...
        RS r0, r1;
        gload(r0, split, -1);
        gload(r1, split + p.S, -1);
        lstore(r0, split, lds, -1);
        gload(r0, split + 2 * p.S, -1);
        __syncthreads();
        for (int j = 0; j < J; j += 2) {
            step(lds, lds + IMG, r1, split + (j + 1) * p.S, split + (j + 3) * p.S);
            __syncthreads();
            if (j + 1 >= J) break;
            step(lds + IMG, lds, r0, split + (j + 2) * p.S, split + (j + 4) * p.S);
            __syncthreads();
        }
        uz::raise_flag(p.flags, xbad, uz::FLAG_X_BOUND); uz::raise_flag(p.flags, dbad, uz::FLAG_DY_BOUND);
        float* out = p.slab + (size_t)split * 9 * p.Cout * p.Cin;
        const int ci = ci0 + wn * 32 + l31;
        const float inv_dy = uz::split_inv_scale(uz::amax_read(p.dy_amax));
        const float inv_x = uz::split_inv_scale(uz::amax_read((XPK && p.x_amax2 && ci >= p.seg_channels) ? p.x_amax2 : p.x_amax));
#pragma unroll
        for (int k = 0; k < NTAP; ++k)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int co = co0 + wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
                if (co < p.Cout && ci < p.Cin) out[((size_t)(TAP0 + k) * p.Cout + co) * p.Cin + ci] = acc[k][r] * inv_dy * inv_x;
            }
    };
    if (tg == 0) run(std::integral_constant<int, 5>{}, std::integral_constant<int, 0>{});
    else run(std::integral_constant<int, 4>{}, std::integral_constant<int, 5>{});
}


// ---------------------------------------------------------------------------------------------------------------------------------
// Fourth form (round 6): HALF a workgroup of the first form.  256 threads = one wave per SIMD, a 64 x 32 (co x ci) tile of all nine taps
// (waves = co half x tap group 5 + 4), the first form's 128-pixel tile, single LDS image (66.5 KB) and one-tile register prefetch.  Same
// split count, twice the workgroups: two INDEPENDENT workgroups fit a CU (<= 256 VGPRs, 133 KB of LDS), so while one waits at its
// barriers for a tile that was requested ~10 k cycles earlier on a 12 - 13 k-cycle round trip (what the first form's "staging phase"
// consists of, NOTES_r6 section 8) the other one has the matrix pipe - and with one of them per CU the other half of EVERY CU stays free
// for the other lanes' kernels (the first form keeps half the CUs whole).  Price: dY is fetched once per 32 input channels instead of
// once per 64.  This is the form the last review asked for (256 threads, one wave per SIMD, 32 x 64 tile).
// Measured (NOTES_r6 section 8): with PHiSeg's target of 128 its 256 workgroups spread over ALL CUs and an isolated launch is 25 - 31 % faster
// (224 -> 128 @ 128 x 128: 1 218 -> 836 us); with two per CU (target 256) it equals the first form (the clock limit); the STEP is 0 - 1.5 % slower at every target
// (64 ... 256) - what one lane's launch gains by taking the whole chip the other lanes' launches lose.  OFF.
constexpr int NTH = 256;
template <int XF, int DF>
__global__ __launch_bounds__(NTH, 2) void wgrad_half_kernel(const WS p) {
    constexpr bool XPK = XF == 1, DPK = DF == 1;
    constexpr int NP = 2;
    using GEO = WGeo<32>;
    constexpr int TW = GEO::TW, XROW = GEO::XROW, QROWX = GEO::QROWX, PROWS = GEO::PROWS;
    constexpr int SROW = TW / 16;
    constexpr int COT = 64, CIT = 32;
    constexpr int DYPLANE = COT * DYROW, XPLANE = CIT * XCH;
    constexpr int DQ = PT / 4, DQROW = TW / 4, XQ = TW / 4;
    constexpr int DYSLOTS = COT * DQ / NTH;                       // 8
    constexpr int XQSLOTS = CIT * PROWS * XQ / NTH;               // 6
    static_assert(COT * DQ % NTH == 0 && CIT * PROWS * XQ % NTH == 0 && CIT * PROWS <= NTH, "staging slots");
    constexpr int NPARTS = PT / 16;                               // 8 k-steps per tile
    extern __shared__ __attribute__((aligned(16))) char lds[];
    char* dYl = lds;
    char* Xl = lds + NP * DYPLANE;

    const int tid = threadIdx.x, lane = tid & 63, l31 = lane & 31, h = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int tg = (wave ^ (int)blockIdx.x) & 1;                  // two workgroups share a SIMD: neighbours take opposite tap groups
    const int wm = wave >> 1;
    const int wid = uz::xcd_remap(blockIdx.x, gridDim.x);
    const int nTile = p.nCoT * p.nCiT;
    const int split = wid / nTile, tl = wid - split * nTile;
    const int co0 = (tl / p.nCiT) * COT, ci0 = (tl % p.nCiT) * CIT;

    for (int i = tid * 16; i < NP * DYPLANE + NP * XPLANE; i += NTH * 16) *reinterpret_cast<u32x4*>(lds + i) = u32x4{0u, 0u, 0u, 0u};

    const __amdgpu_buffer_rsrc_t rdy = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(p.dy), 0, (unsigned)(((size_t)(p.N - 1) * p.CoutTot + p.Cout) * p.HW * 4u), 0x00020000);
    const __amdgpu_buffer_rsrc_t rxx = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(p.x), 0, (unsigned)(((size_t)(p.N - 1) * p.CinTot + p.Cin) * p.HW * 4u), 0x00020000);

    f32x4 dreg[DYSLOTS];
    f32x4 xq[XQSLOTS];
    float xp[2];
    auto gload = [&](int t, int part) __attribute__((always_inline)) {
        const int txi = t % p.tilesX, t2 = t / p.tilesX;
        const int x0 = txi * TW, y0 = (t2 % p.tilesY) * (PT / TW), b0 = t2 / p.tilesY;
        const unsigned dbase = 4u * (unsigned)((b0 * p.CoutTot + co0) * p.HW + y0 * p.W + x0);
#pragma unroll
        for (int i = 0; i < DYSLOTS; ++i) {
            if (part >= 0 && i % NPARTS != part) continue;
            const int e = tid + i * NTH, co = e / DQ, q = e % DQ, row = q / DQROW, c4 = (q % DQROW) * 4;
            const unsigned m = ((y0 + row) < p.H && (co0 + co) < p.Cout) ? 0u : 0xFFFFFFFFu;
            dreg[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rdy, (dbase + 4u * (unsigned)(co * p.HW + row * p.W + c4)) | m, 0, 0));
        }
        const int xbase = (b0 * p.CinTot + ci0) * p.HW + y0 * p.W + x0;
#pragma unroll
        for (int i = 0; i < XQSLOTS; ++i) {
            if (part >= 0 && (i + DYSLOTS) % NPARTS != part) continue;
            const int e = tid + i * NTH;
            const int q = e % XQ, r = e / XQ, ci = r / PROWS, prow = r - ci * PROWS;
            const int yy = y0 + prow - 1;
            const bool rowok = (ci0 + ci) < p.Cin && yy >= 0 && yy < p.H;
            const int sh = (q == 0 && x0 == 0) ? 1 : 0;
            const int off = xbase + ci * p.HW + (prow - 1) * p.W + 4 * q - 1 + sh;
            xq[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rxx, rowok ? 4u * (unsigned)off : 0xFFFFFFFFu, 0, 0));
        }
        if (part < 0 || (DYSLOTS + XQSLOTS) % NPARTS == part) {
            const int ci = tid / PROWS, prow = tid - ci * PROWS;
            const int yy = y0 + prow - 1;
            const bool rowok = tid < CIT * PROWS && (ci0 + ci) < p.Cin && yy >= 0 && yy < p.H;
            const int off = xbase + ci * p.HW + (prow - 1) * p.W + TW - 1;
            xp[0] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rxx, rowok ? 4u * (unsigned)off : 0xFFFFFFFFu, 0, 0));
            xp[1] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rxx, (rowok && x0 + TW < p.W) ? 4u * (unsigned)(off + 1) : 0xFFFFFFFFu, 0, 0));
        }
    };
    const float sdy = uz::split_scale(uz::amax_read(p.dy_amax)), sx = uz::split_scale(uz::amax_read(p.x_amax));
    bool xbad = false, dbad = false;
    auto xpieces = [&](float v0, float v1, unsigned (&out)[NP]) __attribute__((always_inline)) {
        if constexpr (XPK) uz::packed_pair(__builtin_bit_cast(unsigned, v0), __builtin_bit_cast(unsigned, v1), out[0], out[1]);
        else { xbad |= uz::bound_violated(v0 * sx, v1 * sx); pieces<NP>(v0 * sx, v1 * sx, out); }
    };
    auto dpieces = [&](float v0, float v1, unsigned (&out)[NP]) __attribute__((always_inline)) {
        if constexpr (DPK) uz::packed_pair(__builtin_bit_cast(unsigned, v0), __builtin_bit_cast(unsigned, v1), out[0], out[1]);
        else { dbad |= uz::bound_violated(v0 * sdy, v1 * sdy); pieces<NP>(v0 * sdy, v1 * sdy, out); }
    };
    auto lstore = [&](int t) __attribute__((always_inline)) {
        const bool left_edge = t % p.tilesX == 0;
#pragma unroll
        for (int i = 0; i < DYSLOTS; ++i) {
            const int e = tid + i * NTH, co = e / DQ, q = e % DQ;
            char* d = dYl + co * DYROW + ((q / DQROW) * TW + (q % DQROW) * 4) * 2;
            unsigned pa[NP], pb[NP];
            dpieces(dreg[i][0], dreg[i][1], pa);
            dpieces(dreg[i][2], dreg[i][3], pb);
#pragma unroll
            for (int pl = 0; pl < NP; ++pl) *reinterpret_cast<uint2*>(d + pl * DYPLANE) = make_uint2(pa[pl], pb[pl]);
        }
#pragma unroll
        for (int i = 0; i < XQSLOTS; ++i) {
            const int e = tid + i * NTH;
            const int q = e % XQ, r = e / XQ, ci = r / PROWS, prow = r - ci * PROWS;
            char* d = Xl + ci * XCH + prow * XROW + q * 8;
            unsigned pa[NP], pb[NP];
            const f32x4 v = (left_edge && q == 0) ? f32x4{0.f, xq[i][0], xq[i][1], xq[i][2]} : xq[i];
            xpieces(v[0], v[1], pa);
            xpieces(v[2], v[3], pb);
#pragma unroll
            for (int pl = 0; pl < NP; ++pl) *reinterpret_cast<uint2*>(d + pl * XPLANE) = make_uint2(pa[pl], pb[pl]);
        }
        if (tid < CIT * PROWS) {
            const int ci = tid / PROWS, prow = tid - ci * PROWS;
            unsigned pa[NP];
            xpieces(xp[0], xp[1], pa);
            char* d = Xl + ci * XCH + prow * XROW + QROWX * 8;
#pragma unroll
            for (int pl = 0; pl < NP; ++pl) *reinterpret_cast<unsigned*>(d + pl * XPLANE) = pa[pl];
        }
    };

    const char* Ab = dYl + (wm * 32 + l31) * DYROW + h * 16;
    const char* Bb = Xl + l31 * XCH + h * 16;

    auto run = [&](auto ntap_c, auto tap0_c) __attribute__((always_inline)) {
        constexpr int NTAP = decltype(ntap_c)::value, TAP0 = decltype(tap0_c)::value;
        constexpr int DY0 = TAP0 / 3;
        f32x16 acc[NTAP];
#pragma unroll
        for (int k = 0; k < NTAP; ++k)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[k][r] = 0.f;
        int t = split;
        if (t < p.T) gload(t, -1);
        for (; t < p.T; t += p.S) {
            __syncthreads();
            lstore(t);
            __syncthreads();
#pragma unroll
            for (int si = 0; si < NPARTS; ++si) {
                gload(t + p.S, si);                  // unconditional: behind the last tile the image index is >= N and every offset fails the range check
                const int srow = si / SROW, scol = (si % SROW) * 16;
                u32x4 a[NP];
#pragma unroll
                for (int q = 0; q < NP; ++q) a[q] = *reinterpret_cast<const u32x4*>(Ab + q * DYPLANE + (srow * TW + scol) * 2);
#pragma unroll
                for (int d = 0; d < 2; ++d) {
                    u32x4 v[NP];
                    unsigned v4[NP];
#pragma unroll
                    for (int q = 0; q < NP; ++q) {
                        const char* src = Bb + q * XPLANE + (srow + DY0 + d) * XROW + scol * 2;
                        v[q] = *reinterpret_cast<const u32x4*>(src);
                        v4[q] = *reinterpret_cast<const unsigned*>(src + 16);
                    }
#pragma unroll
                    for (int dx = 0; dx < 3; ++dx) {
                        const int tap = (DY0 + d) * 3 + dx, k = tap - TAP0;
                        if (k >= 0 && k < NTAP) {
                            u32x4 b[NP];
#pragma unroll
                            for (int q = 0; q < NP; ++q) {
                                if (dx == 0) b[q] = v[q];
                                else if (dx == 2) b[q] = u32x4{v[q].y, v[q].z, v[q].w, v4[q]};
                                else b[q] = u32x4{__builtin_amdgcn_alignbit(v[q].y, v[q].x, 16), __builtin_amdgcn_alignbit(v[q].z, v[q].y, 16),
                                                  __builtin_amdgcn_alignbit(v[q].w, v[q].z, 16), __builtin_amdgcn_alignbit(v4[q], v[q].w, 16)};
                            }
                            const int kc = k < 0 ? 0 : (k >= NTAP ? NTAP - 1 : k);
                            acc[kc] = mma<NP>(acc[kc], a, b);
                        }
                    }
                }
            }
        }
        uz::raise_flag(p.flags, xbad, uz::FLAG_X_BOUND); uz::raise_flag(p.flags, dbad, uz::FLAG_DY_BOUND);
        float* out = p.slab + (size_t)split * 9 * p.Cout * p.Cin;
        const int ci = ci0 + l31;
        const float inv_dy = uz::split_inv_scale(uz::amax_read(p.dy_amax));
        const float inv_x = uz::split_inv_scale(uz::amax_read((XPK && p.x_amax2 && ci >= p.seg_channels) ? p.x_amax2 : p.x_amax));
#pragma unroll
        for (int k = 0; k < NTAP; ++k)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int co = co0 + wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
                if (co < p.Cout && ci < p.Cin) out[((size_t)(TAP0 + k) * p.Cout + co) * p.Cin + ci] = acc[k][r] * inv_dy * inv_x;
            }
    };
    if (tg == 0) run(std::integral_constant<int, 5>{}, std::integral_constant<int, 0>{});
    else run(std::integral_constant<int, 4>{}, std::integral_constant<int, 5>{});
}

}  // namespace

namespace uz {

// layers that take the split-fp16 weight gradient: 3x3, rows a multiple of 32 wide or exactly 16 wide (aligned
// float4 / fp16-pair staging) and either at least 64 channels on both sides (64 x 64 tile; a 32-channel side
// would leave half of it empty) or at most 32 on both (32 x 32 tile: the full-resolution 32 -> 32 layers and the
// 1- and 3-channel input layers), with enough pixels to amortise the tile loop
bool wgrad_split_ok(int Cin, int Cout, int N, int H, int W, int ks) {
    const int mode = conv_math_mode();
    if (!mode || ks != 3 || (W % 32 != 0 && W != 16)) return false;
    if (mode == 2) return true;
    if (H < 16 || W < 16) return false;          // the default policy keeps planes below 16 x 16 on the fp32 MFMA (tensor-max-relative error, see tests/test_host_cpu.py)
    const long long px = (long long)N * H * W;
    const int lo = Cin < Cout ? Cin : Cout;
    // 64-channel tiles when both sides have them; 32-channel tiles when one side is narrow (32 -> 32, and the decoder's
    // 96 -> 32 at full resolution: 3 ci tiles - the fp32 kernel ran that layer at 79 TF/s)
    // late round 3 (the volume path's shapes): 64-channel tiles from 2 048 pixels on (768 -> 256 @ 8 x 16 x 16: 99 -> 35 us) and, on
    // tensors of >= 512 k pixels, a narrow side down to 5 channels (the depth window of the 4-channel image / 2-channel latent input:
    // 12 -> 32 and 6 -> 64 @ 128 x 128 x 64: 227 -> 69 and 376 -> 118 us; a 32-wide tile a fifth full still beats the fp32 pipe)
    static const int pxmin = getenv("UZ_WGS_PXMIN") ? atoi(getenv("UZ_WGS_PXMIN")) : 2 * 1024;
    static const int lomin = getenv("UZ_WGS_LOMIN") ? atoi(getenv("UZ_WGS_LOMIN")) : 5;
    return (Cin >= 64 && Cout >= 64 && px >= pxmin) || (lo <= 32 && lo >= 16 && px >= 128 * 1024) || (lo < 16 && lo >= lomin && px >= 512 * 1024);
}

static inline int tile_w(int W) { return W == 16 ? 16 : 32; }
static inline int chan_tile(int Cin, int Cout) { return (Cin <= 32 || Cout <= 32) ? 32 : 64; }

// Workgroups a split-path weight gradient is cut into (uz_set_wgrad_target; UZ_WGS_TARGET overrides; default 256 = one per CU).  Its
// workgroups hold 472 - 508 of a SIMD's 512 registers for as long as the kernel runs, so on the CUs it occupies NOTHING else starts; a grid
// of 128 leaves half the chip to the other lanes' launches - PHiSeg under the lane replay +3.3 % (1 907 -> 1 970 images/s, three
// alternations; 192: +1.8 %, 96: +-0, 64: -7 %), Probabilistic U-Net best at 192 (+0.6 %), the single-lane U-Net at 256 (128: -8 %).
static int g_wgs_target = 0;
extern "C" void uz_set_wgrad_target(int workgroups) { g_wgs_target = workgroups; }
extern "C" int uz_get_wgrad_target(void) {
    static const int env = getenv("UZ_WGS_TARGET") ? atoi(getenv("UZ_WGS_TARGET")) : 0;
    return env > 0 ? env : (g_wgs_target > 0 ? g_wgs_target : 256);
}
// The nine-taps-per-wave form (wgrad9_kernel) serves the 64-channel tiles of the two-piece mode.  UZ_WG9=0: first form everywhere.
static bool wgrad9_on(int Cin, int Cout) {
    static const int on = getenv("UZ_WG9") ? atoi(getenv("UZ_WG9")) : 0;
    return on && conv_np() == 2 && chan_tile(Cin, Cout) == 64;
}
// its workgroups are half the first form's (four waves, two per CU): UZ_WG9_MULT workgroups per unit of the target; slabs may outweigh
// the operands by 1 / UZ_WG9_K
static int wgrad9_splits(int Cin, int Cout, int N, int H, int W) {
    static const int mult = getenv("UZ_WG9_MULT") ? atoi(getenv("UZ_WG9_MULT")) : 2;
    static const double kslab = getenv("UZ_WG9_K") ? atof(getenv("UZ_WG9_K")) : 1.0;
    static const int smin_wg = getenv("UZ_WG9_MINWG") ? atoi(getenv("UZ_WG9_MINWG")) : 256;
    const int nt = ceil_div(Cout, 64) * ceil_div(Cin, 64);
    const int tw = tile_w(W);
    const int T = N * ceil_div(H, PT9 / tw) * (W / tw);
    int s = mult * uz_get_wgrad_target() / nt;
    if (kslab > 0.0) {
        const double operands = 4.0 * (double)N * H * W * ((double)Cin + Cout), slab = 2.0 * 4.0 * 9.0 * (double)Cin * Cout;
        int cap = (int)(operands / (kslab * slab));
        const int floor_s = ceil_div(smin_wg, nt);
        if (cap < floor_s) cap = floor_s;
        if (s > cap) s = cap;
    }
    if (s < 1) s = 1;
    if (s > T) s = T;
    return s;
}
// number of pixel splits: at most one split per pixel tile
int wgrad_split_splits(int Cin, int Cout, int N, int H, int W) {
    if (wgrad9_on(Cin, Cout)) return wgrad9_splits(Cin, Cout, N, H, W);
    const int ct = chan_tile(Cin, Cout);
    const int nt = ceil_div(Cout, ct) * ceil_div(Cin, ct);
    const int tw = tile_w(W);
    const int T = N * ceil_div(H, PT / tw) * (W / tw);
    const int target = uz_get_wgrad_target();
    // one workgroup per CU either way (the 32-channel kernel's 166 VGPRs allow no second one: 512 splits measured 6 % slower).  (192
    // workgroups measured +0.5 % on the STEP - a quarter of the CUs stays with the other lane - but the heaviest launch itself 767 -> 909 us
    // with 1.35x -> 1.77x its algorithmic HBM traffic: not kept.)
    int s = target / nt;
    // ... but every split writes (and the reduction reads back) a full 9 x Cout x Cin slab: on the 32 x 32 and 16 x 16 planes that
    // traffic exceeds the operands' own (192 -> 192 @ 32 x 16 x 16: 12.6 MB of x and dy against 74 MB of slabs at 28 splits).  Cap the
    // splits where the slabs would outweigh the operands by more than 1 / UZ_WGS_K, never below UZ_WGS_MINWG workgroups (round 4: measured
    // on the STEP, not per layer - in isolation more workgroups always win, in the step the other lane fills the CUs a shorter grid leaves
    // free: PHiSeg 1 749 -> 1 775 - 1 780 images/s at K = 1, MINWG = 128; K = 0.5 the same, K = 2 / 4 +0.8 / +0.6 %, MINWG = 32 -0.3 %;
    // a uniform target of 128 instead of 256 workgroups +1.3 %, of 64 -9.6 %)
    static const double kslab = getenv("UZ_WGS_K") ? atof(getenv("UZ_WGS_K")) : 1.0;
    static const int smin_wg = getenv("UZ_WGS_MINWG") ? atoi(getenv("UZ_WGS_MINWG")) : 128;
    if (kslab > 0.0) {
        const double operands = 4.0 * (double)N * H * W * ((double)Cin + Cout), slab = 2.0 * 4.0 * 9.0 * (double)Cin * Cout;
        int cap = (int)(operands / (kslab * slab));
        const int floor_s = ceil_div(smin_wg, nt);
        if (cap < floor_s) cap = floor_s;
        if (s > cap) s = cap;
    }
    if (s < 1) s = 1;
    if (s > T) s = T;
    return s;
}

// UZ_WG_DB=1: the two-image form (wgrad_db_kernel) for the 64-channel tiles of the 32-wide geometry in the two-piece mode
static bool wgrad_db_on() { static const int on = getenv("UZ_WG_DB") ? atoi(getenv("UZ_WG_DB")) : 0; return on != 0; }
template <int XF, int DF>
static int launch_wgrad_db(WS p, int grid, hipStream_t st) {
    constexpr size_t smem = 2 * 2 * (size_t)(64 * (PTD * 2 + 16) + 64 * (4 * 80 + 16));
    static bool attr_done = false;
    if (!attr_done) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(wgrad_db_kernel<XF, DF>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess)
            return fail("wgrad_split: cannot raise dynamic LDS limit");
        attr_done = true;
    }
    p.tilesY = ceil_div(p.H, PTD / 32); p.T = p.N * p.tilesX * p.tilesY;         // 64-pixel tiles; the split count stays the first form's
    hipLaunchKernelGGL((wgrad_db_kernel<XF, DF>), dim3(grid), dim3(NT), smem, st, p);
    return check_launch("wgrad_db_kernel");
}
// UZ_WG_HALF=1: the half-workgroup form (wgrad_half_kernel) for the 64-channel tiles of the 32-wide geometry in the two-piece mode
static bool wgrad_half_on() { static const int on = getenv("UZ_WG_HALF") ? atoi(getenv("UZ_WG_HALF")) : 0; return on != 0; }
template <int XF, int DF>
static int launch_wgrad_half(WS p, hipStream_t st) {
    constexpr size_t smem = 2 * (size_t)(64 * DYROW) + 2 * (size_t)(32 * XCH);
    static bool attr_done = false;
    if (!attr_done) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(wgrad_half_kernel<XF, DF>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess)
            return fail("wgrad_split: cannot raise dynamic LDS limit");
        attr_done = true;
    }
    p.nCiT = ceil_div(p.Cin, 32);                                     // 64 x 32 tiles; the split count stays the first form's
    hipLaunchKernelGGL((wgrad_half_kernel<XF, DF>), dim3(p.nCoT * p.nCiT * p.S), dim3(NTH), smem, st, p);
    return check_launch("wgrad_half_kernel");
}
template <int TWv, int CT, int NP, int XPK = 0, int DPK = 0, int M16 = 0>
static int launch_wgrad_np(const WS& p, int grid, hipStream_t st) {
    if constexpr (!M16 && TWv == 32 && CT == 64 && NP == 2 && XPK == 1 && DPK == 1) {
        if (wgrad_half_on()) return launch_wgrad_half<XPK, DPK>(p, st);
    }
    if constexpr (!M16 && TWv == 32 && CT == 64 && NP == 2 && XPK == 1 && DPK == 1) {       // (fp32 operands: their conversion spills 40 - 50 registers in this form)
        if (wgrad_db_on()) return launch_wgrad_db<XPK, DPK>(p, grid, st);
    }
    if constexpr (!M16 && TWv == 32 && CT == 64 && NP == 2) {
        static const int m16 = getenv("UZ_WG_M16") ? atoi(getenv("UZ_WG_M16")) : 0;
        if (m16) return launch_wgrad_np<TWv, CT, NP, XPK, DPK, 1>(p, grid, st);
    }
    // (the 32-channel kernel folds its pixel quarters through this LDS at the end: three taps x 16 x 64 floats per wave pair)
    constexpr size_t stage = NP * (size_t)(CT * DYROW) + NP * (size_t)(CT * XCH), fold = CT == 32 ? (size_t)4 * 3 * 16 * 64 * 4 : 0;
    constexpr size_t smem = stage > fold ? stage : fold;
    static bool attr_done = false;
    if (!attr_done) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(wgrad_split_kernel<TWv, CT, NP, XPK, DPK, M16>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess)
            return fail("wgrad_split: cannot raise dynamic LDS limit");
        attr_done = true;
    }
    hipLaunchKernelGGL((wgrad_split_kernel<TWv, CT, NP, XPK, DPK, M16>), dim3(grid), dim3(NT), smem, st, p);
    return check_launch("wgrad_split_kernel");
}
template <int TWv, int XF, int DF>
static int launch_wgrad9(const WS& p, int grid, hipStream_t st) {
    constexpr size_t smem = 2 * (size_t)(64 * DYROW9) + 2 * (size_t)(64 * WGeo9<TWv>::XCH);
    hipLaunchKernelGGL((wgrad9_kernel<TWv, 2, XF, DF>), dim3(grid), dim3(NT9), smem, st, p);
    return check_launch("wgrad9_kernel");
}
template <int TWv>
static int launch_wgrad9_f(const WS& p, int grid, hipStream_t st, int xpk, int dpk) {
    if (xpk) return dpk ? launch_wgrad9<TWv, 1, 1>(p, grid, st) : launch_wgrad9<TWv, 1, 0>(p, grid, st);
    return dpk ? launch_wgrad9<TWv, 0, 1>(p, grid, st) : launch_wgrad9<TWv, 0, 0>(p, grid, st);
}
template <int TWv, int CT>
static int launch_wgrad(const WS& p, int grid, hipStream_t st, int xpk, int dpk, int xb16, int db16) {
    if (conv_np() != 2) {
        if (xpk || dpk) return fail("wgrad_split: split storage needs the two-piece fp16 mode");
        if constexpr (TWv == 32) {
            if (xb16) return db16 ? launch_wgrad_np<TWv, CT, 1, 2, 2>(p, grid, st) : launch_wgrad_np<TWv, CT, 1, 2, 0>(p, grid, st);
            if (db16) return launch_wgrad_np<TWv, CT, 1, 0, 2>(p, grid, st);
        } else if (xb16 || db16) return fail("wgrad_split: bf16 storage serves planes whose width is a multiple of 32");
        return launch_wgrad_np<TWv, CT, 1>(p, grid, st);
    }
    if (xb16 || db16) return fail("wgrad_split: bf16 storage needs the single-piece bf16 mode");
    if (xpk) return dpk ? launch_wgrad_np<TWv, CT, 2, 1, 1>(p, grid, st) : launch_wgrad_np<TWv, CT, 2, 1, 0>(p, grid, st);
    return dpk ? launch_wgrad_np<TWv, CT, 2, 0, 1>(p, grid, st) : launch_wgrad_np<TWv, CT, 2, 0, 0>(p, grid, st);
}

int wgrad_split(const float* x, int Cin, int CinTot, const float* dy, int Cout, int CoutTot, float* slab,
                int N, int H, int W, int S, const float* x_amax, const float* dy_amax, hipStream_t st,
                int x_packed, const float* x_amax2, int seg_channels, int dy_packed, int x_b16, int dy_b16) {
    WS p;
    p.x_amax2 = (x_packed && seg_channels > 0) ? x_amax2 : nullptr; p.seg_channels = seg_channels;
    UZ_REQUIRE(!x_packed || seg_channels == 0 || (x_amax2 && seg_channels < Cin), "wgrad_split: the second scale segment needs its bound");
    p.x = x; p.dy = dy; p.slab = slab; p.x_amax = x_amax; p.dy_amax = dy_amax; p.stamps = debug_stamps; p.flags = dev_flags_ptr();
    p.N = N; p.H = H; p.W = W; p.HW = H * W; p.Cin = Cin; p.CinTot = CinTot; p.Cout = Cout; p.CoutTot = CoutTot;
    const int tw = tile_w(W), ct = chan_tile(Cin, Cout);
    const bool nine = wgrad9_on(Cin, Cout);
    p.tilesX = W / tw; p.tilesY = ceil_div(H, (nine ? PT9 : PT) / tw); p.T = N * p.tilesX * p.tilesY; p.S = S;
    p.nCoT = ceil_div(Cout, ct); p.nCiT = ceil_div(Cin, ct);
    UZ_REQUIRE((size_t)N * CinTot * p.HW < (1ull << 30) && (size_t)N * CoutTot * p.HW < (1ull << 30), "wgrad_split: tensor too large for 32-bit offsets (the dispatcher routes such tensors to the fp32 kernel)");
    const int grid = p.nCoT * p.nCiT * S;
    if (nine) {
        UZ_REQUIRE(!x_b16 && !dy_b16, "wgrad_split: bf16 storage needs the single-piece bf16 mode");
        return tw == 16 ? launch_wgrad9_f<16>(p, grid, st, x_packed, dy_packed) : launch_wgrad9_f<32>(p, grid, st, x_packed, dy_packed);
    }
    if (ct == 32) return tw == 16 ? launch_wgrad<16, 32>(p, grid, st, x_packed, dy_packed, x_b16, dy_b16) : launch_wgrad<32, 32>(p, grid, st, x_packed, dy_packed, x_b16, dy_b16);
    return tw == 16 ? launch_wgrad<16, 64>(p, grid, st, x_packed, dy_packed, x_b16, dy_b16) : launch_wgrad<32, 64>(p, grid, st, x_packed, dy_packed, x_b16, dy_b16);
}

}  // namespace uz
