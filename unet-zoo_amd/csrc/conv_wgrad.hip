// Weight gradient of nn.Conv2d(k=3,p=1) / (k=1) on the fp32 matrix cores: the weight half of
// aten::convolution_backward for the reference's conv layers (torchlayers.py:18, unet.py:25-29, ...).
//
//   dW[co][ci][tap] = sum_{b,y,x} dY[b,co,y,x] * X[b,ci,y+dy-1,x+dx-1]
//
// GEMM view: M = co (A operand, lane&31 = co), N = ci (B operand, lane&31 = ci), K = pixels
// (two pixels per v_mfma_f32_32x32x2_f32).  A wave owns a 32co x 32ci block for five (or four) of
// the nine taps (5 x 16 accumulator registers): one dY fragment read feeds five MFMAs whose B
// fragments are the same haloed X patch at constant LDS offsets.  A workgroup is 8 waves:
// (co sub-tile) x (ci sub-tile) x (tap group) x (pixel split); it owns up to 64co x 64ci and walks
// a strided subset of the 64-pixel tiles (deterministic split-K).  The next tile's global loads are
// issued into registers before the MFMA loop of the current tile and written to LDS after it.
// Partial sums go to slabs; an ordered second kernel adds them into dW (bitwise reproducible; no
// float atomics).  LDS rows are padded to an odd stride so that the 32 channel-lanes of a fragment
// read hit 32 different banks.
#include <math.h>
#include <stdlib.h>
#include <type_traits>
#include "uz_common.h"
#include "split_f16.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));

struct WgP {
    const float* x; const float* dy; float* slab;
    int N, H, W, HW;
    int Cin, CinTot, Cout, CoutTot;
    int TW, TH, TB, lgTW, lgTH;
    int tilesX, tilesY, T, S;
    int PW, PSI, PS, PSP, PT, PTP;
    int nCoT, nCiT;
};

constexpr int NT = 512;   // threads per workgroup (8 waves)

// WM x WN waves tile the (co, ci) block of 32*WM x 32*WN; NTG tap groups split the 3x3 taps (5 + 4);
// the remaining WK = 8/(WM*WN*NTG) waves split the pixels of every tile.  Every (tap group is part
// of the same slab, every pixel split writes its own slab.  PF: prefetch the next tile's global
// loads into registers before the MFMA loop of the current tile (needs CIT*PS <= XR*512).
template <int KS, int WM, int WN, bool PF>
__global__ __launch_bounds__(NT, 2) void wgrad_kernel(const WgP p) {
    constexpr int KK = KS * KS, HALO = KS / 2;
    constexpr int NTG = KK == 9 ? 2 : 1, TPG = KK == 9 ? 5 : 1;      // tap groups, taps per group (max)
    constexpr int COT = 32 * WM, CIT = 32 * WN, WK = 8 / (WM * WN * NTG);
    constexpr int DYR = COT / 8;          // dY values staged per thread and tile
    constexpr int XR = PF ? 18 : 1;       // X values staged per thread and tile (upper bound)
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float* dYl = lds;                               // [COT][PTP]
    float* Xl = dYl + COT * p.PTP;                  // [CIT][PSP] (+ slack)
    int* tab = reinterpret_cast<int*>(Xl + CIT * p.PSP + 64);   // [2][PS]: relative offset or -1 (per tile)

    const int tid = threadIdx.x, lane = tid & 63, l31 = lane & 31, h = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);      // provably wave-uniform (scalar loop control below)
    // NTG == 2 (3x3: 5 + 4 taps): waves w and w + 4 share a SIMD and get different tap groups
    const int tg = NTG == 2 ? ((wave ^ (wave >> 2)) & 1) : 0, w2 = wave / NTG;
    const int wmn = w2 % (WM * WN), wk = w2 / (WM * WN);
    const int wm = wmn / WN, wn = wmn % WN;
    const int wid = uz::xcd_remap(blockIdx.x, gridDim.x);
    const int nTile = p.nCoT * p.nCiT;
    const int split = wid / nTile, tl = wid - split * nTile;
    const int co0 = (tl / p.nCiT) * COT, ci0 = (tl % p.nCiT) * CIT;

    {   // zero the whole LDS image once (pad / slack words must be finite)
        const int tot = COT * p.PTP + CIT * p.PSP + 64;
        for (int i = tid; i < tot; i += NT) lds[i] = 0.f;
    }
    const int pl = tid & 63;
    const int ptx = pl & (p.TW - 1), pty = (pl >> p.lgTW) & (p.TH - 1), ptb = pl >> (p.lgTW + p.lgTH);
    const int npix = p.TB << (p.lgTW + p.lgTH);
    const int nsteps = (npix + 1) >> 1;
    const int totalX = CIT * p.PS;
    const int qs = NT / p.PS, rs = NT - qs * p.PS;
    const int ciA = tid / p.PS, rA = tid - ciA * p.PS;

    f32x16 acc[TPG];
#pragma unroll
    for (int t = 0; t < TPG; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;

    auto tile_origin = [&](int t, int& x0, int& y0, int& b0) {
        const int txi = t % p.tilesX, t2 = t / p.tilesX;
        x0 = txi * p.TW; y0 = (t2 % p.tilesY) * p.TH; b0 = (t2 / p.tilesY) * p.TB;
    };
    // per-tile table: patch word r -> offset relative to (b0, channel, y0-HALO, x0-HALO), or -1 outside the image
    auto make_tab = [&](int t, int which) {
        int x0, y0, b0;
        tile_origin(t, x0, y0, b0);
        for (int r = tid; r < p.PS; r += NT) {
            const int tb = r / p.PSI, rr = r - tb * p.PSI;
            const int py = rr / p.PW, px = rr - py * p.PW;
            const int b = b0 + tb, yy = y0 + py - HALO, xx = x0 + px - HALO;
            const bool v = b < p.N && yy >= 0 && yy < p.H && xx >= 0 && xx < p.W;
            tab[which * p.PS + r] = v ? (tb * p.CinTot * p.HW + py * p.W + px) : -1;   // relative to (y0-HALO, x0-HALO): never negative when valid
        }
    };
    float dreg[DYR], xreg[XR];
    auto gload = [&](int t, int which) {
        int x0, y0, b0;
        tile_origin(t, x0, y0, b0);
        const bool pv = pl < npix && (b0 + ptb) < p.N && (y0 + pty) < p.H && (x0 + ptx) < p.W;
        const float* dsrc = p.dy + ((size_t)(b0 + ptb) * p.CoutTot + co0) * p.HW + (y0 + pty) * p.W + (x0 + ptx);
#pragma unroll
        for (int j = 0; j < DYR; ++j) {
            const int co = (tid >> 6) + 8 * j;
            dreg[j] = (pv && (co0 + co) < p.Cout) ? dsrc[(size_t)co * p.HW] : 0.f;
        }
        if (PF) {
            const float* xsrc = p.x + ((size_t)b0 * p.CinTot + ci0) * p.HW + (y0 - HALO) * p.W + (x0 - HALO);
            const int* tb = tab + which * p.PS;
            int ci = ciA, r = rA;
#pragma unroll
            for (int i = 0; i < XR; ++i) {
                float v = 0.f;
                if (tid + i * NT < totalX) {
                    const int off = tb[r];
                    if (off >= 0 && (ci0 + ci) < p.Cin) v = xsrc[(size_t)ci * p.HW + off];
                }
                xreg[i] = v;
                ci += qs; r += rs;
                if (r >= p.PS) { r -= p.PS; ++ci; }
            }
        }
    };
    auto lstore = [&](int t, int which) {
#pragma unroll
        for (int j = 0; j < DYR; ++j) dYl[((tid >> 6) + 8 * j) * p.PTP + pl] = dreg[j];
        if (PF) {
            int ci = ciA, r = rA;
#pragma unroll
            for (int i = 0; i < XR; ++i) {
                if (tid + i * NT < totalX) Xl[ci * p.PSP + r] = xreg[i];
                ci += qs; r += rs;
                if (r >= p.PS) { r -= p.PS; ++ci; }
            }
        } else {   // direct staging (large patches of the tiny-resolution levels)
            int x0, y0, b0;
            tile_origin(t, x0, y0, b0);
            const float* xsrc = p.x + ((size_t)b0 * p.CinTot + ci0) * p.HW + (y0 - HALO) * p.W + (x0 - HALO);
            const int* tb = tab + which * p.PS;
            int ci = ciA, r = rA;
            for (int e = tid; e < totalX; e += NT) {
                const int off = tb[r];
                Xl[ci * p.PSP + r] = (off >= 0 && (ci0 + ci) < p.Cin) ? xsrc[(size_t)ci * p.HW + off] : 0.f;
                ci += qs; r += rs;
                if (r >= p.PS) { r -= p.PS; ++ci; }
            }
        }
    };

    int t = split;
    if (t < p.T) make_tab(t, 0);
    __syncthreads();
    if (t < p.T) gload(t, 0);
    for (int it = 0; t < p.T; t += p.S, ++it) {
        const int cur = it & 1;
        const int tn = t + p.S;
        __syncthreads();                       // every wave finished the MFMAs of the previous tile
        lstore(t, cur);
        if (tn < p.T) make_tab(tn, cur ^ 1);
        __syncthreads();
        if (tn < p.T) gload(tn, cur ^ 1);      // in flight during the MFMA loop below
        {
            // pixel pair (2s, 2s+1): the second pixel sits one word to the right (or one image further when TW == 1)
            const float* Ab = dYl + (wm * 32 + l31) * p.PTP + h;
            const float* Bb = Xl + (wn * 32 + l31) * p.PSP + (p.TW > 1 ? h : h * p.PSI);
            auto mma_loop = [&](auto ntap_c, auto tap0_c) {
                constexpr int NTAP = decltype(ntap_c)::value, TAP0 = decltype(tap0_c)::value;
#pragma unroll 4
                for (int s = wk; s < nsteps; s += WK) {
                    const int p2 = 2 * s;
                    const int po = (p2 >> (p.lgTW + p.lgTH)) * p.PSI + ((p2 >> p.lgTW) & (p.TH - 1)) * p.PW + (p2 & (p.TW - 1));
                    const float a = Ab[p2];
                    float b[NTAP];
#pragma unroll
                    for (int k = 0; k < NTAP; ++k) b[k] = Bb[po + ((TAP0 + k) / KS) * p.PW + (TAP0 + k) % KS];
#pragma unroll
                    for (int k = 0; k < NTAP; ++k) acc[k] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b[k], acc[k], 0, 0, 0);
                }
            };
            if (KK == 9) {
                if (tg == 0) mma_loop(std::integral_constant<int, 5>{}, std::integral_constant<int, 0>{});
                else mma_loop(std::integral_constant<int, 4>{}, std::integral_constant<int, 5>{});
            } else {
                mma_loop(std::integral_constant<int, 1>{}, std::integral_constant<int, 0>{});
            }
        }
    }

    // ---- partial slab [split * WK + wk][tap][co][ci]
    float* out = p.slab + (size_t)(split * WK + wk) * KK * p.Cout * p.Cin;
    const int ci = ci0 + wn * 32 + l31;
    const int tap0 = tg * TPG, ntap = (KK == 9) ? (tg == 0 ? 5 : 4) : 1;
#pragma unroll
    for (int k = 0; k < TPG; ++k)
        if (k < ntap) {
            const int tap = tap0 + k;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int co = co0 + wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
                if (co < p.Cout && ci < p.Cin) out[((size_t)tap * p.Cout + co) * p.Cin + ci] = acc[k][r];
            }
        }
}

// ------------------------------------------------------------------------------------------------
// Specialisation for the layers that carry almost all of the work: 3x3, tile = THF rows x TW columns
// of one image (TW = 32 for W >= 32, else W = 16 or 8; patch (THF+2) x (TW+2)).  Every LDS offset of the MFMA loop is a compile-time
// immediate, the 32 pixel-pair steps of a tile are fully unrolled (LDS reads are scheduled far ahead
// of the MFMAs that consume them), and the staging map is one fixed patch word per thread
// (408 of 512 threads active; a thread walks the input channels with a constant stride).
template <int WM, int WN, int THF, int TW>
__global__ __launch_bounds__(NT, 2) void wgrad_fast_kernel(const WgP p) {
    constexpr int PW = TW + 2, PS = (THF + 2) * PW, PSP = PS | 1, PT = TW * THF, PTP = PT + 1;
    static_assert(TW == 32 || TW == 16 || TW == 8, "tile width");
    constexpr int XG = NT / PS;             // channel groups of the X staging map (3 for 4x34 patches, 2 for 6x34)
    constexpr int COT = 32 * WM, CIT = 32 * WN, WK = 8 / (WM * WN * 2);
    constexpr int DYR = COT * PT / NT, XRN = (CIT + XG - 1) / XG, NSTEP = (PT / 2) / WK, DCS = NT / PT;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float* dYl = lds;                       // [COT][65]
    float* Xl = dYl + COT * PTP;            // [CIT][PSP]

    const int tid = threadIdx.x, lane = tid & 63, l31 = lane & 31, h = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    // waves w and w + 4 share a SIMD: give them different tap groups (5 + 4 taps) so every SIMD carries 9 taps' worth
    const int tg = (wave ^ (wave >> 2)) & 1, w2 = wave >> 1;
    const int wmn = w2 % (WM * WN), wk = w2 / (WM * WN);
    const int wm = wmn / WN, wn = wmn % WN;
    const int wid = uz::xcd_remap(blockIdx.x, gridDim.x);
    const int nTile = p.nCoT * p.nCiT;
    const int split = wid / nTile, tl = wid - split * nTile;
    const int co0 = (tl / p.nCiT) * COT, ci0 = (tl % p.nCiT) * CIT;

    for (int i = tid; i < COT * PTP + CIT * PSP + 64; i += NT) lds[i] = 0.f;

    // staging maps
    const int pl = tid & (PT - 1), ptx = pl & (TW - 1), pty = pl / TW, dco = tid / PT;
    const bool xact = tid < XG * PS;
    const int xg = tid / PS, xr = tid - xg * PS;
    const int xpy = xr / PW, xpx = xr - xpy * PW;

    // Branch-free global staging through raw buffer loads: an all-ones offset fails the hardware range check
    // and returns 0 (padding pixels, channel-tile overhang).
    const __amdgpu_buffer_rsrc_t rdy = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(p.dy), 0, (unsigned)(((size_t)(p.N - 1) * p.CoutTot + p.Cout) * p.HW * sizeof(float)), 0x00020000);
    const __amdgpu_buffer_rsrc_t rxx = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(p.x), 0, (unsigned)(((size_t)(p.N - 1) * p.CinTot + p.Cin) * p.HW * sizeof(float)), 0x00020000);
    const unsigned dlane = 4u * (unsigned)(dco * p.HW + pty * p.W + ptx);               // this thread's dY word inside a tile
    const unsigned xlane = 4u * (unsigned)(xg * p.HW + (xpy - 1) * p.W + (xpx - 1));    // may wrap below zero: added to the tile base
    const int cmaxo = min(COT, p.Cout - co0), cmaxi = min(CIT, p.Cin - ci0);
    float dreg[DYR], xreg[XRN];
    auto gload = [&](int t) {
        const int txi = t % p.tilesX, t2 = t / p.tilesX;
        const int x0 = txi * TW, y0 = (t2 % p.tilesY) * THF, b0 = t2 / p.tilesY;
        {
            const unsigned pm = ((y0 + pty) < p.H && (x0 + ptx) < p.W) ? 0u : 0xFFFFFFFFu;
            const unsigned base = 4u * (unsigned)((b0 * p.CoutTot + co0) * p.HW + y0 * p.W + x0) + dlane;
#pragma unroll
            for (int j = 0; j < DYR; ++j) {
                const unsigned cm = (dco + DCS * j) < cmaxo ? 0u : 0xFFFFFFFFu;
                dreg[j] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rdy, (base + 4u * (unsigned)(DCS * j * p.HW)) | pm | cm, 0, 0));
            }
        }
        {
            const int yy = y0 + xpy - 1, xx = x0 + xpx - 1;
            const unsigned pm = (xact && yy >= 0 && yy < p.H && xx >= 0 && xx < p.W) ? 0u : 0xFFFFFFFFu;
            const unsigned base = 4u * (unsigned)((b0 * p.CinTot + ci0) * p.HW + y0 * p.W + x0) + xlane;
#pragma unroll
            for (int i = 0; i < XRN; ++i) {
                const unsigned cm = (xg + XG * i) < cmaxi ? 0u : 0xFFFFFFFFu;
                xreg[i] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rxx, (base + 4u * (unsigned)(XG * i * p.HW)) | pm | cm, 0, 0));
            }
        }
    };
    auto lstore = [&]() {
#pragma unroll
        for (int j = 0; j < DYR; ++j) dYl[(dco + DCS * j) * PTP + pl] = dreg[j];
        if (xact) {
#pragma unroll
            for (int i = 0; i < XRN; ++i)
                if (xg + XG * i < CIT) Xl[(xg + XG * i) * PSP + xr] = xreg[i];
        }
    };

    const float* Ab = dYl + (wm * 32 + l31) * PTP + h + 2 * wk;
    const float* Bb = Xl + (wn * 32 + l31) * PSP + h + 2 * wk;

    auto run = [&](auto ntap_c, auto tap0_c) {
        constexpr int NTAP = decltype(ntap_c)::value, TAP0 = decltype(tap0_c)::value;
        f32x16 acc[NTAP];
#pragma unroll
        for (int k = 0; k < NTAP; ++k)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[k][r] = 0.f;
        int t = split;
        if (t < p.T) gload(t);
        for (; t < p.T; t += p.S) {
            __syncthreads();                   // every wave finished the MFMAs of the previous tile
            lstore();
            __syncthreads();
            if (t + p.S < p.T) gload(t + p.S); // in flight during the MFMA loop below
#pragma unroll
            for (int si = 0; si < NSTEP; ++si) {
                constexpr int dummy = 0; (void)dummy;
                const int s0 = si * WK;                                  // + wk folded into Ab / Bb
                const int po = ((2 * s0) / TW) * PW + ((2 * s0) & (TW - 1));
                const float a = Ab[2 * s0];
                float b[NTAP];
#pragma unroll
                for (int k = 0; k < NTAP; ++k) b[k] = Bb[po + ((TAP0 + k) / 3) * PW + (TAP0 + k) % 3];
#pragma unroll
                for (int k = 0; k < NTAP; ++k) acc[k] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b[k], acc[k], 0, 0, 0);
            }
        }
        // The WK pixel-pair groups of the workgroup hold partial sums of the same outputs: fold them
        // pairwise through LDS (fixed order ((0+2)+(1+3)), three taps per round so the staging area
        // suffices) and let group 0 write ONE slab per workgroup instead of WK.
        if constexpr (WK > 1) {
            constexpr int NW = 2 * WM * WN, RT = 3;             // waves per wk group, taps per round
            float* red = lds;
            const int slot = wmn * 2 + tg;
#pragma unroll
            for (int stride = WK / 2; stride >= 1; stride >>= 1) {
#pragma unroll
                for (int k0 = 0; k0 < 5; k0 += RT) {
                    __syncthreads();
                    if (wk >= stride && wk < 2 * stride) {
                        float* dstp = red + (size_t)(((wk - stride) * NW + slot) * RT) * 16 * 64 + lane;
#pragma unroll
                        for (int kk = 0; kk < RT; ++kk)
                            if (k0 + kk < NTAP) {
#pragma unroll
                                for (int r = 0; r < 16; ++r) dstp[(kk * 16 + r) * 64] = acc[k0 + kk][r];
                            }
                    }
                    __syncthreads();
                    if (wk < stride) {
                        const float* srcp = red + (size_t)((wk * NW + slot) * RT) * 16 * 64 + lane;
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
        const int ci = ci0 + wn * 32 + l31;
#pragma unroll
        for (int k = 0; k < NTAP; ++k)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int co = co0 + wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
                if (co < p.Cout && ci < p.Cin) out[((size_t)(TAP0 + k) * p.Cout + co) * p.Cin + ci] = acc[k][r];
            }
    };
    if (tg == 0) run(std::integral_constant<int, 5>{}, std::integral_constant<int, 0>{});
    else run(std::integral_constant<int, 4>{}, std::integral_constant<int, 5>{});
}

// Ordered two-stage reduction of the partial slabs (bitwise reproducible).
// stage 1: slab[g*RG] += slab[g*RG + 1 .. g*RG + RG-1]  for every group g of RG consecutive slabs
__global__ __launch_bounds__(256) void wgrad_reduce_groups(float* __restrict__ slab, int S, int RG, int n) {
    const int g = blockIdx.y;
    const int lo = g * RG, hi = min(S, lo + RG);
    for (int i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) {
        float s = 0.f;
        for (int k = lo; k < hi; ++k) s += slab[(size_t)k * n + i];
        slab[(size_t)lo * n + i] = s;
    }
}
// stage 2: dW[co][ci][tap] = sum_g slab[g*RG][tap][co][ci]  (fixed order).  One thread per (co, ci) and tap (blockIdx.y):
// the slab reads of a wave are coalesced (lane = consecutive ci), a thread's S loads are independent of each other (issued
// eight at a time, added in slab order), and 9x the workgroups of a per-(co, ci) sweep fill the chip - this kernel runs once
// per layer and step, 106 times for PHiSeg.
// volC > 0: the layer is a Conv3d run as a depth window (Cin = 3 volC contraction channels, k = kd volC + ci): the sum goes
// straight into the PyTorch [Cout][volC][3][3][3] gradient layout (no separate permutation launch).
template <int KK>
__global__ __launch_bounds__(256) void wgrad_reduce_kernel(const float* __restrict__ slab, float* __restrict__ dw,
                                                            int S, int RG, int Cout, int Cin, int volC) {
    const int m = Cout * Cin, n = KK * m;
    const int j = blockIdx.x * 256 + threadIdx.x, t = blockIdx.y;
    if (j >= m) return;
    const float* src = slab + (size_t)t * m + j;
    const size_t step = (size_t)RG * n;
    const int cnt = (S + RG - 1) / RG;
    float s = 0.f;
    int k = 0;
    for (; k + 8 <= cnt; k += 8) {
        float v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = src[(size_t)(k + u) * step];
#pragma unroll
        for (int u = 0; u < 8; ++u) s += v[u];
    }
    for (; k < cnt; ++k) s += src[(size_t)k * step];
    if (volC > 0) {
        const int co = j / Cin, kc = j - co * Cin, kd = kc / volC, ci = kc - kd * volC;
        dw[(((size_t)co * volC + ci) * 3 + kd) * KK + t] = s;
    } else {
        dw[(size_t)j * KK + t] = s;
    }
}

// db[c] = sum_{b,y,x} dy[b,c,y,x]: CSB blocks per channel write fp64 partials, then one wave per
// channel adds them in a fixed order.
constexpr int CSB = 64;
__global__ __launch_bounds__(256) void channel_sum_partial(const float* __restrict__ dy, int Ctot, int N, int HW,
                                                            double* __restrict__ part) {
    __shared__ double sm[4];
    const int c = blockIdx.x, blk = blockIdx.y;
    double s[1] = {0.0};
    const long long total = (long long)N * HW;
    for (long long i = (long long)blk * 256 + threadIdx.x; i < total; i += (long long)CSB * 256) {
        const int b = (int)(i / HW), q = (int)(i - (long long)b * HW);
        s[0] += dy[((size_t)b * Ctot + c) * HW + q];
    }
    uz::block_sum_d<1>(s, sm);
    if (threadIdx.x == 0) part[(size_t)c * CSB + blk] = s[0];
}
__global__ __launch_bounds__(64) void channel_sum_final(const double* __restrict__ part, float* __restrict__ db) {
    const int c = blockIdx.x;
    double s = part[(size_t)c * CSB + threadIdx.x];
    s = uz::wave_sum_d(s);
    if (threadIdx.x == 0) db[c] = (float)s;
}


// ---------------------------------------------------------------- 3x3 weight gradient of the 1..4-channel input layers
// dW[co][ci][tap] with Cin <= 4 (the first convolution of every encoder: image / image + one-hot mask) is a streaming
// reduction over dY (N * Cout * HW * 4 bytes read once), not a GEMM: an MFMA tile would be 3 % full.  A workgroup walks
// tiles of 2 rows x up to 128 columns of one image: dY tile [32 co][256 px] and the haloed X patch go to LDS (coalesced
// float4 reads), thread (co, g) owns 32 consecutive pixels of the tile, slides a 3 x 3 window of X over them in registers
// (all 32 co-lanes read the same X address: broadcast) and accumulates its 9 Cin sums; the 8 pixel groups are folded in a
// fixed order through LDS and the workgroup writes ONE slab [tap][co][ci] - the usual ordered reduce follows.
constexpr int THIN_CO = 32, THIN_G = 8, THIN_ROWS = 2, THIN_WMAX = 128, THIN_STRIDE = THIN_ROWS * THIN_WMAX + 1;
template <int CIN>
__global__ __launch_bounds__(256) void wgrad_thin_kernel(const float* __restrict__ x, int CinTot, const float* __restrict__ dy, int CoutTot, int Cout,
                                                          float* __restrict__ slab, int N, int H, int W, int tiles_per_image, int tiles_per_wg) {
    __shared__ __attribute__((aligned(16))) float dyL[THIN_CO * THIN_STRIDE];
    __shared__ float xL[CIN * (THIN_ROWS + 2) * (THIN_WMAX + 2)];
    const int tid = threadIdx.x, co = tid & 31, g = tid >> 5;
    const int co0 = blockIdx.y * THIN_CO;
    const int HW = H * W, PWp = W + 2;
    const int npx = THIN_ROWS * W;                           // pixels of a tile (<= 256), W % 32 == 0 checked by the host
    const int per = npx / THIN_G;                            // pixels per thread: a run inside one row (W >= per)
    float acc[CIN][9];
#pragma unroll
    for (int c = 0; c < CIN; ++c)
#pragma unroll
        for (int t = 0; t < 9; ++t) acc[c][t] = 0.f;
    const int total = N * tiles_per_image;
    const int t_lo = blockIdx.x * tiles_per_wg, t_hi = min(total, t_lo + tiles_per_wg);
    for (int tl = t_lo; tl < t_hi; ++tl) {
        const int b = tl / tiles_per_image, y0 = (tl - b * tiles_per_image) * THIN_ROWS;
        __syncthreads();                                     // previous tile fully consumed
        // dY tile: 32 co x (2 rows x W) - float4 along x
        const int q4 = npx / 4;
        for (int e = tid; e < THIN_CO * q4; e += 256) {
            const int c = e / q4, q = e - c * q4, px = 4 * q, r = px / W, col = px - r * W;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (co0 + c < Cout && y0 + r < H) v = *reinterpret_cast<const float4*>(dy + ((size_t)b * CoutTot + co0 + c) * HW + (size_t)(y0 + r) * W + col);
            float* d = dyL + c * THIN_STRIDE + px;
            d[0] = v.x; d[1] = v.y; d[2] = v.z; d[3] = v.w;
        }
        // X patch: Cin x 4 rows x (W + 2), zero outside the image
        for (int e = tid; e < CIN * (THIN_ROWS + 2) * PWp; e += 256) {
            const int c = e / ((THIN_ROWS + 2) * PWp), r2 = e - c * (THIN_ROWS + 2) * PWp, r = r2 / PWp, col = r2 - r * PWp;
            const int yy = y0 + r - 1, xx = col - 1;
            float v = 0.f;
            if (yy >= 0 && yy < H && xx >= 0 && xx < W) v = x[((size_t)b * CinTot + c) * HW + (size_t)yy * W + xx];
            xL[(c * (THIN_ROWS + 2) + r) * (THIN_WMAX + 2) + col] = v;
        }
        __syncthreads();
        const int p0 = g * per, r = p0 / W, c0 = p0 - r * W;
        const float* dyrow = dyL + co * THIN_STRIDE + p0;
#pragma unroll
        for (int c = 0; c < CIN; ++c) {
            const float* xr = xL + (c * (THIN_ROWS + 2) + r) * (THIN_WMAX + 2) + c0;     // patch column of pixel column c0 - 1
            float w0[3], w1[3], w2[3];
#pragma unroll
            for (int k = 0; k < 3; ++k) { w0[k] = xr[k * (THIN_WMAX + 2)]; w1[k] = xr[k * (THIN_WMAX + 2) + 1]; }
            for (int i = 0; i < per; ++i) {
#pragma unroll
                for (int k = 0; k < 3; ++k) w2[k] = xr[k * (THIN_WMAX + 2) + i + 2];
                const float v = dyrow[i];
#pragma unroll
                for (int k = 0; k < 3; ++k) {
                    acc[c][3 * k + 0] += v * w0[k];
                    acc[c][3 * k + 1] += v * w1[k];
                    acc[c][3 * k + 2] += v * w2[k];
                    w0[k] = w1[k]; w1[k] = w2[k];
                }
            }
        }
    }
    // fold the 8 pixel groups in a fixed order, then one slab per workgroup: slab[blockIdx.x][tap][co][ci] (all co tiles)
    __syncthreads();
    float* red = dyL;                                        // 8 x 32 x (CIN * 9) floats <= 9216 < the dY tile
    for (int c = 0; c < CIN; ++c)
        for (int t = 0; t < 9; ++t) red[(g * THIN_CO + co) * (CIN * 9) + c * 9 + t] = acc[c][t];
    __syncthreads();
    for (int e = tid; e < THIN_CO * CIN * 9; e += 256) {
        const int c_o = e / (CIN * 9), r2 = e - c_o * (CIN * 9), c = r2 / 9, t = r2 - c * 9;
        float s2 = 0.f;
#pragma unroll
        for (int k = 0; k < THIN_G; ++k) s2 += red[(k * THIN_CO + c_o) * (CIN * 9) + r2];
        if (co0 + c_o < Cout) slab[((size_t)blockIdx.x * 9 + t) * Cout * CIN + (size_t)(co0 + c_o) * CIN + c] = s2;
    }
}
inline bool wgrad_thin_ok(int Cin, int Cout, int N, int H, int W, int ks) {
    return ks == 3 && Cin <= 4 && W % 32 == 0 && W <= THIN_WMAX && H % THIN_ROWS == 0 && (long long)N * H * W >= 64 * 1024;
}
constexpr int THIN_SLABS = 512;
inline int wgrad_thin_slabs(int N, int H) { const int t = N * (H / THIN_ROWS); return t < THIN_SLABS ? t : THIN_SLABS; }

struct WGeom { int TW, TH, TB, PW, PSI, PS, tilesX, tilesY, tilesB, T, S, nCoT, nCiT, WM, WN, WK, pf, fast, SW; };   // SW: slabs per pixel split

WGeom pick_wgeom(int Cin, int Cout, int N, int H, int W, int halo) {
    WGeom g;
    g.TW = W >= 32 ? 32 : uz::pow2_ceil(W);
    const int npx = (halo == 1 && W >= 32 && H >= 64) ? 128 : 64;   // fast kernel: 4 x 32 tiles on the large planes
    g.TH = uz::pow2_ceil(H);
    if (g.TH > npx / g.TW) g.TH = npx / g.TW;
    if (g.TH < 1) g.TH = 1;
    g.TB = npx / (g.TW * g.TH);
    if (g.TB > uz::pow2_ceil(N)) g.TB = uz::pow2_ceil(N);
    g.PW = g.TW + 2 * halo;
    g.PSI = (g.TH + 2 * halo) * g.PW;
    while (g.TB > 1 && g.TB * g.PSI > 288) g.TB >>= 1;
    g.PS = g.TB * g.PSI;
    g.tilesX = uz::ceil_div(W, g.TW);
    g.tilesY = uz::ceil_div(H, g.TH);
    g.tilesB = uz::ceil_div(N, g.TB);
    g.T = g.tilesX * g.tilesY * g.tilesB;
    g.WM = Cout > 32 ? 2 : 1;
    g.WN = Cin > 32 ? 2 : 1;
    // (round 4, measured and dropped: 32 x 32 channel tiles on the 4 x 4 / 2 x 2 levels, where 64 x 64 tiles leave 18 ... 72 workgroups:
    //  192 -> 192 @ 2 x 2 28 -> 26 us, @ 4 x 4 31 -> 37 us - these launches are bound by their fixed phases, not by workgroup count)
    g.WK = 8 / (g.WM * g.WN * (halo ? 2 : 1));
    g.nCoT = uz::ceil_div(Cout, 32 * g.WM);
    g.nCiT = uz::ceil_div(Cin, 32 * g.WN);
    g.pf = (32 * g.WN * g.PS <= 18 * 512) ? 1 : 0;
    // the fast kernels address a whole tensor through one buffer resource (32-bit byte offsets): tensors of 2^30 elements
    // or more take the generic kernel (64-bit pointers)
    const bool huge = (size_t)N * Cin * H * W >= (1ull << 30) || (size_t)N * Cout * H * W >= (1ull << 30);
    g.fast = halo == 1 && g.TB == 1 && !huge &&
             ((g.TW == 32 && (g.TH == 2 || g.TH == 4)) || (g.TW == 16 && g.TH == 4) || (g.TW == 8 && g.TH == 8));
    g.SW = g.fast ? 1 : g.WK;                 // the fast kernels fold their WK partial sums in LDS
    // pixel splits S: small cost model instead of a fixed target.  More splits = more workgroups in flight
    // (the only parallelism a low-resolution layer has) but S*WK partial slabs of Cout*Cin*k*k floats to
    // write and re-read; fewer splits = longer serial tile loops per workgroup.
    {
        const int nt = g.nCoT * g.nCiT;
        const double kk = halo ? 9.0 : 1.0, n_out = kk * Cout * Cin;
        const double px = (double)g.TW * g.TH * g.TB;
        const double t_tile = (32.0 * g.WM) * (32.0 * g.WN) * kk * px * 2.0 / (613e9 * 0.6) + 1.0e-6;   // s per tile per workgroup
        const int slots = 256 * ((g.WM * g.WN == 4 && halo) ? 1 : 2);
        static const double slabcost = getenv("UZ_WG_SLABCOST") ? atof(getenv("UZ_WG_SLABCOST")) : 1.0;      // weight of the slab traffic in the model (step-level tuning knob)
        double best = 1e30;
        int s = 1;
        for (int c = 1; c <= g.T && c <= 4096; ++c) {       // any split count: nt * c should land just under a whole number of rounds
            const double rounds = ceil((double)nt * c / slots);
            const double t = rounds * ceil((double)g.T / c) * t_tile + slabcost * (double)c * g.SW * n_out * 4.0 * 2.0 / 2.5e12 + 4e-6 * (c * g.SW > 64 ? 2 : 1);
            if (t < best) { best = t; s = c; }
        }
        g.S = s;
    }
    return g;
}

}  // namespace

extern "C" size_t uz_conv_bwd_weight_workspace(int Cin, int Cout, int N, int H, int W, int ks) {
    const WGeom g = pick_wgeom(Cin, Cout, N, H, W, ks / 2);
    const size_t slabs = (size_t)g.S * g.SW * ks * ks * Cout * Cin * sizeof(float);
    const size_t dbp = (size_t)Cout * CSB * sizeof(double);
    size_t need = slabs > dbp ? slabs : dbp;
    if (wgrad_thin_ok(Cin, Cout, N, H, W, ks)) {
        const size_t th = (size_t)wgrad_thin_slabs(N, H) * 9 * Cout * Cin * sizeof(float);
        if (th > need) need = th;
    }
    if (uz::wgrad_split_ok(Cin, Cout, N, H, W, ks)) {
        const size_t sp = (size_t)uz::wgrad_split_splits(Cin, Cout, N, H, W) * ks * ks * Cout * Cin * sizeof(float) + 2 * uz::AMAX_FLOATS * sizeof(float);    // + fallback bound slots
        if (sp > need) need = sp;
    }
    if (ks == 1 && uz::conv1x1_small_ok(Cin, Cout)) {
        const size_t sm = uz::conv1x1_small_bwd_weight_ws(Cin, Cout, N, H, W);
        if (sm > need) need = sm;
    }
    return need;
}

// Which kernel family a convolution call takes under the current math mode (diagnostics / roofline bookkeeping):
// kind 0 = forward, 1 = data gradient, 2 = weight gradient; returns 0 fp32 MFMA, 1 split-fp16 MFMA, 2 streaming VALU (1x1 heads).
extern "C" int uz_conv_route(int kind, int Cin, int Cout, int N, int H, int W, int ks) {
    if (ks == 1 && uz::conv1x1_small_ok(Cin, Cout)) return 2;
    const bool thin = ks == 3 && Cin <= 4 && (long long)N * H * W >= 64 * 1024;       // 1..4-channel input layers: streaming kernels
    if (kind == 0 && thin && Cout <= 256 && N <= 65535) return 2;
    if (kind == 2 && wgrad_thin_ok(Cin, Cout, N, H, W, ks)) return 2;
    if (kind == 0) return uz::conv_split_ok(Cin, Cout, N, H, W, ks, 0) ? 1 : 0;
    if (kind == 1) return uz::conv_split_ok(Cout, Cin, N, H, W, ks, 1) ? 1 : 0;
    return uz::wgrad_split_ok(Cin, Cout, N, H, W, ks) ? 1 : 0;
}

// Number of partial-sum slabs [ks*ks][Cout][Cin] a weight-gradient call of this shape writes before its reduction (0: the call does
// not go through slabs - the streaming 1x1 heads).  Mirrors the dispatch of uz_conv_bwd_weight_ex below.
extern "C" int uz_conv_bwd_weight_slabs(int Cin, int Cout, int N, int H, int W, int ks) {
    if ((ks != 1 && ks != 3) || Cin <= 0 || Cout <= 0 || N <= 0 || H <= 0 || W <= 0) return 0;
    if (ks == 1 && uz::conv1x1_small_ok(Cin, Cout)) return 0;
    if (wgrad_thin_ok(Cin, Cout, N, H, W, ks)) {
        const int total = N * (H / THIN_ROWS);
        const int per_wg = uz::ceil_div(total, wgrad_thin_slabs(N, H));
        return uz::ceil_div(total, per_wg);
    }
    const bool huge = (size_t)N * Cin * H * W >= (1ull << 30) || (size_t)N * Cout * H * W >= (1ull << 30);
    if (!huge && uz::wgrad_split_ok(Cin, Cout, N, H, W, ks)) return uz::wgrad_split_splits(Cin, Cout, N, H, W);
    const WGeom g = pick_wgeom(Cin, Cout, N, H, W, ks / 2);
    return g.S * g.SW;
}
namespace {
// All slab reductions of a tape in ONE launch.  table: n_layers rows of 8 int64 {slabs (address), dw (address), S, Cout, Cin, ks*ks,
// first block, volC}; a workgroup finds its layer by bisection over the first-block column; 256 (co, ci) pairs of one tap per
// workgroup.  Same order of additions as the per-layer launches: groups of 32 slabs first when S > 64, then the group sums.
// volC > 0: the row is a depth window (Cin = 3 volC, k = kd volC + ci) and the sum leaves in the Conv3d layout [Cout][volC][3][3][3].
__global__ __launch_bounds__(256) void wgrad_reduce_table_k(const long long* __restrict__ table, int n_layers) {
    int lo = 0, hi = n_layers - 1;
    while (lo < hi) { const int mid = (lo + hi + 1) >> 1; if ((int)table[8 * mid + 6] <= (int)blockIdx.x) lo = mid; else hi = mid - 1; }
    const long long* t = table + 8 * lo;
    const float* slab = reinterpret_cast<const float*>(t[0]);
    float* dw = reinterpret_cast<float*>(t[1]);
    const int S = (int)t[2], Cout = (int)t[3], Cin = (int)t[4], KK = (int)t[5], blk = (int)blockIdx.x - (int)t[6];
    const int m = Cout * Cin, n = KK * m, per_tap = (m + 255) / 256;
    const int tap = blk / per_tap, j = (blk - tap * per_tap) * 256 + threadIdx.x;
    if (j >= m) return;
    const float* src = slab + (size_t)tap * m + j;
    const int RG = S > 64 ? 32 : 1;
    float s = 0.f;
    if (RG == 1) {
        int k = 0;
        for (; k + 8 <= S; k += 8) {
            float v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = src[(size_t)(k + u) * n];
#pragma unroll
            for (int u = 0; u < 8; ++u) s += v[u];
        }
        for (; k < S; ++k) s += src[(size_t)k * n];
    } else {
        for (int g0 = 0; g0 < S; g0 += RG) {
            const int ge = min(S, g0 + RG);
            float sg = 0.f;
            int k = g0;
            for (; k + 8 <= ge; k += 8) {
                float v[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) v[u] = src[(size_t)(k + u) * n];
#pragma unroll
                for (int u = 0; u < 8; ++u) sg += v[u];
            }
            for (; k < ge; ++k) sg += src[(size_t)k * n];
            s += sg;
        }
    }
    const int volC = (int)t[7];
    if (volC > 0) {
        const int co = j / Cin, kc = j - co * Cin, kd = kc / volC, ci = kc - kd * volC;
        dw[(((size_t)co * volC + ci) * 3 + kd) * KK + tap] = s;
    } else {
        dw[(size_t)j * KK + tap] = s;
    }
}
}  // namespace
extern "C" int uz_wgrad_reduce_blocks(int Cin, int Cout, int ks) { return ks * ks * uz::ceil_div(Cout * Cin, 256); }
extern "C" int uz_wgrad_reduce_table(const int64_t* table, int n_layers, int total_blocks, void* stream) {
    UZ_REQUIRE(table && n_layers > 0 && total_blocks > 0, "wgrad_reduce_table: bad arguments");
    hipLaunchKernelGGL(wgrad_reduce_table_k, dim3(total_blocks), dim3(256), 0, uz::S(stream), reinterpret_cast<const long long*>(table), n_layers);
    return uz::check_launch("wgrad_reduce_table_k");
}

extern "C" int uz_conv_bwd_weight(const float* x, int Cin, int CinTot, const float* dy, int Cout, int CoutTot,
                                  float* dw, float* db, int N, int H, int W, int ks,
                                  const float* x_amax, const float* dy_amax,
                                  void* workspace, size_t workspace_bytes, void* stream) {
    return uz_conv_bwd_weight_ex(x, Cin, CinTot, dy, Cout, CoutTot, dw, db, N, H, W, ks, x_amax, dy_amax, workspace, workspace_bytes, 0, nullptr, 0, 0, nullptr, stream);
}
// ... with either operand in split storage (include/uz_api.h, round 4): only on the split-fp16 path, bias gradient not available
// (db reads dy as fp32).
static int conv_bwd_weight_impl(const float* x, int Cin, int CinTot, const float* dy, int Cout, int CoutTot,
                                float* dw, float* db, int N, int H, int W, int ks,
                                const float* x_amax, const float* dy_amax, void* workspace, size_t workspace_bytes,
                                int x_packed, const float* x_amax2, int seg_channels, int dy_packed, float* slabs_out, void* stream,
                                int x_b16, int dy_b16);
extern "C" int uz_conv_bwd_weight_ex(const float* x, int Cin, int CinTot, const float* dy, int Cout, int CoutTot,
                                     float* dw, float* db, int N, int H, int W, int ks,
                                     const float* x_amax, const float* dy_amax, void* workspace, size_t workspace_bytes,
                                     int x_packed, const float* x_amax2, int seg_channels, int dy_packed, float* slabs_out, void* stream) {
    return conv_bwd_weight_impl(x, Cin, CinTot, dy, Cout, CoutTot, dw, db, N, H, W, ks, x_amax, dy_amax, workspace, workspace_bytes,
                                x_packed, x_amax2, seg_channels, dy_packed, slabs_out, stream, 0, 0);
}
// bf16 STORAGE (include/uz_api.h, "bf16 storage"): x and / or dy hold 2-byte bf16 elements; single-piece bf16 mode, 3x3 shapes on the
// matrix-pipe path with rows a multiple of 32 wide; the gradient itself stays fp32 (fp32 accumulation, fp32 slabs, ordered reduce).
extern "C" int uz_conv_bwd_weight_b16(const void* x, int Cin, int CinTot, const void* dy, int Cout, int CoutTot,
                                      float* dw, int N, int H, int W, int ks, void* workspace, size_t workspace_bytes,
                                      int x_b16, int dy_b16, float* slabs_out, void* stream) {
    UZ_REQUIRE(uz::conv_np() == 1 && ks == 3 && W % 32 == 0 && uz_conv_route(2, Cin, Cout, N, H, W, ks) == 1,
               "conv_bwd_weight_b16: bf16 storage needs uz_set_conv_math(3), a 3x3 shape on the matrix-pipe path and rows a multiple of 32 wide");
    return conv_bwd_weight_impl(static_cast<const float*>(x), Cin, CinTot, static_cast<const float*>(dy), Cout, CoutTot, dw, nullptr, N, H, W, ks,
                                nullptr, nullptr, workspace, workspace_bytes, 0, nullptr, 0, 0, slabs_out, stream, x_b16 != 0, dy_b16 != 0);
}
static int conv_bwd_weight_impl(const float* x, int Cin, int CinTot, const float* dy, int Cout, int CoutTot,
                                float* dw, float* db, int N, int H, int W, int ks,
                                const float* x_amax, const float* dy_amax, void* workspace, size_t workspace_bytes,
                                int x_packed, const float* x_amax2, int seg_channels, int dy_packed, float* slabs_out, void* stream,
                                int x_b16, int dy_b16) {
    UZ_REQUIRE(ks == 1 || ks == 3, "conv_bwd_weight: kernel size %d unsupported", ks);
    // slabs_out: the call stops behind its main kernel and leaves its uz_conv_bwd_weight_slabs() partial-sum slabs [S][ks*ks][Cout][Cin]
    // there; uz_wgrad_reduce_table adds the slabs of MANY layers in one launch (the plans: one at the end of the backward tape
    // instead of one or two reduction launches behind every weight gradient).  dw is not written, db must be NULL.
    // (a depth window - 3 C view channels over a C-channel buffer - may leave slabs too: its table row carries C, and the table kernel
    //  writes the Conv3d parameter layout exactly as this call's own reduction does)
    UZ_REQUIRE(!slabs_out || (!db && uz_conv_bwd_weight_slabs(Cin, Cout, N, H, W, ks) > 0),
               "conv_bwd_weight_ex: slabs_out on a call that writes no slabs (uz_conv_bwd_weight_slabs() == 0) or asks for a bias gradient");
    const bool any_packed = x_packed || dy_packed;
    UZ_REQUIRE(!any_packed || (uz_conv_route(2, Cin, Cout, N, H, W, ks) == 1 && uz::conv_np() == 2 && !(dy_packed && db)),
               "conv_bwd_weight_ex: an operand in split storage, but this shape / math mode does not take the split-fp16 path (or a bias gradient was requested)");
    UZ_REQUIRE((!x_packed || x_amax) && (!dy_packed || dy_amax), "conv_bwd_weight_ex: an operand in split storage needs the bound it was scaled from");
    UZ_REQUIRE(N > 0 && H > 0 && W > 0 && Cin > 0 && Cout > 0, "conv_bwd_weight: empty tensor");
    UZ_REQUIRE(H < 1024 && W < 1024, "conv_bwd_weight: spatial size too large");
    if (ks == 1 && uz::conv1x1_small_ok(Cin, Cout)) {                 // 2..8-output heads: streaming VALU kernel
        UZ_REQUIRE(workspace && workspace_bytes >= uz::conv1x1_small_bwd_weight_ws(Cin, Cout, N, H, W), "conv_bwd_weight: workspace too small");
        const int rc = uz::conv1x1_small_bwd_weight(x, Cin, CinTot, dy, Cout, CoutTot, dw, db, N, H, W, workspace, uz::S(stream));
        if (rc != -2) return rc;
    }
    const WGeom g = pick_wgeom(Cin, Cout, N, H, W, ks / 2);
    const size_t need = (size_t)g.S * g.SW * ks * ks * Cout * Cin * sizeof(float);
    UZ_REQUIRE(workspace && workspace_bytes >= need, "conv_bwd_weight: workspace %zu < %zu bytes", workspace_bytes, need);
    hipStream_t st = uz::S(stream);
    WgP p;
    p.x = x; p.dy = dy; p.slab = slabs_out ? slabs_out : static_cast<float*>(workspace);
    p.N = N; p.H = H; p.W = W; p.HW = H * W;
    p.Cin = Cin; p.CinTot = CinTot; p.Cout = Cout; p.CoutTot = CoutTot;
    p.TW = g.TW; p.TH = g.TH; p.TB = g.TB; p.lgTW = uz::ilog2(g.TW); p.lgTH = uz::ilog2(g.TH);
    p.tilesX = g.tilesX; p.tilesY = g.tilesY; p.T = g.T; p.S = g.S;
    p.PW = g.PW; p.PSI = g.PSI; p.PS = g.PS; p.PSP = g.PS | 1;
    p.PT = g.TW * g.TH * g.TB > 64 ? 128 : 64; p.PTP = p.PT + 1;
    p.nCoT = g.nCoT; p.nCiT = g.nCiT;
    const int cot = 32 * g.WM, cit = 32 * g.WN;
    size_t smem = ((size_t)cot * p.PTP + (size_t)cit * p.PSP + 64 + 2 * (size_t)p.PS) * sizeof(float);
    if (g.fast && g.WK > 1) {                 // LDS for the in-workgroup fold: WK/2 writer groups x waves x 3 taps x 16 x 64 floats
        const size_t fold = (size_t)(g.WK / 2) * (2 * g.WM * g.WN) * 3 * 16 * 64 * sizeof(float);
        if (fold > smem) smem = fold;
    }
    const int grid = g.nCoT * g.nCiT * g.S;
    int Stot = g.S * g.SW;
    const bool thin = wgrad_thin_ok(Cin, Cout, N, H, W, ks);
    if (thin) {                                   // 1..4 input channels: streaming reduction (same slab layout, ordered reduce below)
        const int tiles_img = H / THIN_ROWS, total = N * tiles_img;
        Stot = wgrad_thin_slabs(N, H);
        const int per_wg = uz::ceil_div(total, Stot);
        Stot = uz::ceil_div(total, per_wg);
        UZ_REQUIRE(workspace_bytes >= (size_t)Stot * 9 * Cout * Cin * sizeof(float), "conv_bwd_weight: workspace too small for the thin-input path");
        const dim3 tg(Stot, uz::ceil_div(Cout, THIN_CO));
#define UZ_THIN(C_) hipLaunchKernelGGL(wgrad_thin_kernel<C_>, tg, dim3(256), 0, st, x, CinTot, dy, CoutTot, Cout, p.slab, N, H, W, tiles_img, per_wg)
        if (Cin == 1) UZ_THIN(1); else if (Cin == 2) UZ_THIN(2); else if (Cin == 3) UZ_THIN(3); else UZ_THIN(4);
#undef UZ_THIN
    }
    // The split-fp16 and the fast fp32 kernels address a whole tensor through one buffer resource (32-bit byte offsets): tensors
    // of 2^30 elements or more take the generic fp32 kernel (64-bit pointers) instead of failing.
    const bool huge = (size_t)N * CinTot * H * W >= (1ull << 30) || (size_t)N * CoutTot * H * W >= (1ull << 30);
    UZ_REQUIRE(!(huge && g.fast), "conv_bwd_weight: a channel-slice view of a buffer of 2^30 elements or more is not supported by the tiled kernels");
    const bool split_math = !thin && !huge && uz::wgrad_split_ok(Cin, Cout, N, H, W, ks);
    //  the slab-count query sees the VIEW's channels, this dispatch the buffer's: a channel slice of a >= 2^30-element
    // buffer leaves the split path here but not there - a caller that sized slabs_out from the query must not be handed more slabs
    UZ_REQUIRE(!slabs_out || (split_math ? uz::wgrad_split_splits(Cin, Cout, N, H, W) : Stot) == uz_conv_bwd_weight_slabs(Cin, Cout, N, H, W, ks),
               "conv_bwd_weight_ex: slabs_out was sized for %d slabs (uz_conv_bwd_weight_slabs), this call writes another number (a channel-slice view of a huge buffer?)",
               uz_conv_bwd_weight_slabs(Cin, Cout, N, H, W, ks));
    UZ_REQUIRE(!any_packed || split_math, "conv_bwd_weight_ex: split storage on a call that left the split-fp16 path");
    UZ_REQUIRE(!(x_b16 || dy_b16) || (split_math && !db), "conv_bwd_weight_b16: bf16 storage on a call that left the matrix-pipe path");
    if (split_math) {                              // large layers: split-fp16 matrix pipe (conv_wgrad_split.hip), same slab layout
        Stot = uz::wgrad_split_splits(Cin, Cout, N, H, W);
        const size_t slab_bytes = (size_t)Stot * 9 * Cout * Cin * sizeof(float);
        constexpr size_t slot_bytes = uz::AMAX_FLOATS * sizeof(float);
        UZ_REQUIRE(workspace_bytes >= slab_bytes + 2 * slot_bytes, "conv_bwd_weight: workspace too small for the split path");
        if (uz::conv_np() == 2 && (!x_amax || !dy_amax)) {             // no bounds from the caller: measure them (stand-alone C-ABI path; the bf16 mode has no scales)
            float* slots = reinterpret_cast<float*>(static_cast<char*>(workspace) + slab_bytes);
            if (hipMemsetAsync(slots, 0, 2 * slot_bytes, st) != hipSuccess) return uz::fail("conv_bwd_weight: memset failed");
            if (!x_amax) { if (int rc = uz::absmax_view(x, Cin, CinTot, N, H * W, slots, st)) return rc; x_amax = slots; }
            if (!dy_amax) { if (int rc = uz::absmax_view(dy, Cout, CoutTot, N, H * W, slots + uz::AMAX_FLOATS, st)) return rc; dy_amax = slots + uz::AMAX_FLOATS; }
        }
        if (int rc = uz::wgrad_split(x, Cin, CinTot, dy, Cout, CoutTot, p.slab, N, H, W, Stot, x_amax, dy_amax, st, x_packed, x_amax2, seg_channels, dy_packed, x_b16, dy_b16)) return rc;
    }
#define UZ_WG_LAUNCH(KS_, WM_, WN_, PF_)                                                                         \
    do {                                                                                                         \
        static bool attr = false;                                                                                \
        auto kern = wgrad_kernel<KS_, WM_, WN_, PF_>;                                                            \
        if (!attr) {                                                                                             \
            if (hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess) \
                return uz::fail("wgrad: cannot raise dynamic LDS limit");                                        \
            attr = true;                                                                                         \
        }                                                                                                        \
        hipLaunchKernelGGL(kern, dim3(grid), dim3(NT), smem, st, p);                                            \
    } while (0)
#define UZ_WG_PF(KS_, WM_, WN_) UZ_WG_LAUNCH(KS_, WM_, WN_, false)
#define UZ_WG_TILE(KS_)                                                              \
    do {                                                                             \
        if (g.WM == 2 && g.WN == 2) UZ_WG_PF(KS_, 2, 2);                             \
        else if (g.WM == 2) UZ_WG_PF(KS_, 2, 1);                                     \
        else if (g.WN == 2) UZ_WG_PF(KS_, 1, 2);                                     \
        else UZ_WG_PF(KS_, 1, 1);                                                    \
    } while (0)
    const bool fast = g.fast != 0;
#define UZ_WG_FAST(WM_, WN_, THF_, TW_)                                                                            \
    do {                                                                                                         \
        static bool attr = false;                                                                                \
        auto kern = wgrad_fast_kernel<WM_, WN_, THF_, TW_>;                                                      \
        if (!attr) {                                                                                             \
            if (hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess) \
                return uz::fail("wgrad: cannot raise dynamic LDS limit");                                        \
            attr = true;                                                                                         \
        }                                                                                                        \
        hipLaunchKernelGGL(kern, dim3(grid), dim3(NT), smem, st, p);                                             \
    } while (0)
#define UZ_WG_FAST_T(THF_, TW_)                                                      \
    do {                                                                             \
        if (g.WM == 2 && g.WN == 2) UZ_WG_FAST(2, 2, THF_, TW_);                     \
        else if (g.WM == 2) UZ_WG_FAST(2, 1, THF_, TW_);                             \
        else if (g.WN == 2) UZ_WG_FAST(1, 2, THF_, TW_);                             \
        else UZ_WG_FAST(1, 1, THF_, TW_);                                            \
    } while (0)
    if (split_math || thin) {
    } else if (fast) {
        if (g.TW == 32 && g.TH == 4) UZ_WG_FAST_T(4, 32);
        else if (g.TW == 32) UZ_WG_FAST_T(2, 32);
        else if (g.TW == 16) UZ_WG_FAST_T(4, 16);
        else UZ_WG_FAST_T(8, 8);
    } else if (ks == 3) UZ_WG_TILE(3); else UZ_WG_TILE(1);
    if (int rc = uz::check_launch("wgrad_kernel")) return rc;
    if (slabs_out) return 0;
    const int n = ks * ks * Cout * Cin;
    int rgrid = uz::ceil_div(n, 256);
    if (rgrid > 4096) rgrid = 4096;
    int RG = 1;
    if (Stot > 64) {
        RG = 32;
        hipLaunchKernelGGL(wgrad_reduce_groups, dim3(rgrid, uz::ceil_div(Stot, RG)), dim3(256), 0, st, p.slab, Stot, RG, n);
        if (int rc = uz::check_launch("wgrad_reduce_groups")) return rc;
    }
    const int fgrid = uz::ceil_div(Cout * Cin, 256);
    // a depth window (csrc/vol.hip: Conv3d = the 2-D kernel over D slices with 3 C contraction channels on a C-channel buffer) is the
    // only call with more view channels than the buffer has: its gradient leaves in the Conv3d parameter layout
    const int volC = (ks == 3 && Cin == 3 * CinTot) ? CinTot : 0;
    if (ks == 3) hipLaunchKernelGGL(wgrad_reduce_kernel<9>, dim3(fgrid, 9), dim3(256), 0, st, p.slab, dw, Stot, RG, Cout, Cin, volC);
    else hipLaunchKernelGGL(wgrad_reduce_kernel<1>, dim3(fgrid, 1), dim3(256), 0, st, p.slab, dw, Stot, RG, Cout, Cin, 0);
    if (int rc = uz::check_launch("wgrad_reduce_kernel")) return rc;
    if (db) {
        // the slab workspace is free again after the reduction above; it holds the fp64 partials
        UZ_REQUIRE(workspace_bytes >= (size_t)Cout * CSB * sizeof(double), "conv_bwd_weight: workspace too small for the bias gradient");
        double* part = static_cast<double*>(workspace);
        hipLaunchKernelGGL(channel_sum_partial, dim3(Cout, CSB), dim3(256), 0, st, dy, CoutTot, N, H * W, part);
        if (int rc = uz::check_launch("channel_sum_partial")) return rc;
        hipLaunchKernelGGL(channel_sum_final, dim3(Cout), dim3(64), 0, st, part, db);
        if (int rc = uz::check_launch("channel_sum_final")) return rc;
    }
    return 0;
}
