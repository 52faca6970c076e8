// Weight gradient of nn.Conv2d(k=3,p=1) / (k=1) on the fp32 matrix cores: the weight half of
// aten::convolution_backward for the reference's conv layers (torchlayers.py:18, unet.py:25-29, ...).
//
//   dW[co][ci][tap] = sum_{b,y,x} dY[b,co,y,x] * X[b,ci,y+dy-1,x+dx-1]
//
// GEMM view: M = co (A operand, lane&31 = co), N = ci (B operand, lane&31 = ci), K = pixels
// (two pixels per v_mfma_f32_32x32x2_f32).  A wave owns a 32co x 32ci block for ALL nine taps
// (9 x 16 accumulator registers): one dY fragment read feeds nine MFMAs whose B fragments are the
// same haloed X patch at nine constant LDS offsets.  A workgroup (2x2 waves) owns 64co x 64ci and
// walks a strided subset of the 64-pixel tiles (deterministic split-K): it writes its partial sum
// to a slab, and an ordered second kernel adds the slabs into dW (bitwise reproducible; no float
// atomics).  LDS rows are padded to an odd stride so that the 32 channel-lanes of a fragment read
// hit 32 different banks.
#include "uz_common.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
constexpr int CT = 64;   // co and ci tile of a workgroup

struct WgP {
    const float* x; const float* dy; float* slab;
    int N, H, W, HW;
    int Cin, CinTot, Cout, CoutTot;
    int TW, TH, TB, lgTW, lgTH;
    int tilesX, tilesY, T, S;
    int PW, PSI, PS, PSP, PT, PTP;
    int nCoT, nCiT;
};

template <int KS>
__global__ __launch_bounds__(256, 2) void wgrad_kernel(const WgP p) {
    constexpr int KK = KS * KS, HALO = KS / 2;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float* dYl = lds;                               // [CT][PTP]
    float* Xl = dYl + CT * p.PTP;                   // [CT][PSP] (+ slack)
    int* tabOff = reinterpret_cast<int*>(Xl + CT * p.PSP + 64);
    int* tabCrd = tabOff + p.PS;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, h = lane >> 5;
    const int wm = wave >> 1, wn = wave & 1;
    const int wid = uz::xcd_remap(blockIdx.x, gridDim.x);
    const int nTile = p.nCoT * p.nCiT;
    const int split = wid / nTile, tl = wid - split * nTile;
    const int co0 = (tl / p.nCiT) * CT, ci0 = (tl % p.nCiT) * CT;

    // zero the whole LDS image once (pad / slack / not-yet-staged words must be finite)
    {
        const int tot = CT * p.PTP + CT * p.PSP + 64;
        for (int i = tid; i < tot; i += 256) lds[i] = 0.f;
    }
    // relative offset + packed coordinates of every patch word (identical for all tiles)
    for (int r = tid; r < p.PS; r += 256) {
        const int tb = r / p.PSI, rr = r - tb * p.PSI;
        const int py = rr / p.PW, px = rr - py * p.PW;
        tabOff[r] = tb * p.CinTot * p.HW + (py - HALO) * p.W + (px - HALO);
        tabCrd[r] = (tb << 20) | (py << 10) | px;
    }
    // this thread's dY pixel inside a tile
    const int pl = tid & 63;
    const int ptx = pl & (p.TW - 1), pty = (pl >> p.lgTW) & (p.TH - 1), ptb = pl >> (p.lgTW + p.lgTH);
    const int npix = p.TB << (p.lgTW + p.lgTH);

    f32x16 acc[KK];
#pragma unroll
    for (int t = 0; t < KK; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;

    const int q256 = 256 / p.PS, r256 = 256 - q256 * p.PS;
    const int nsteps = (npix + 1) >> 1;
    __syncthreads();

    for (int t = split; t < p.T; t += p.S) {
        const int txi = t % p.tilesX, t2 = t / p.tilesX;
        const int tyi = t2 % p.tilesY, tbi = t2 / p.tilesY;
        const int x0 = txi * p.TW, y0 = tyi * p.TH, b0 = tbi * p.TB;
        // ---- stage dY tile: [co][pixel]
        {
            const bool pv = pl < npix && (b0 + ptb) < p.N && (y0 + pty) < p.H && (x0 + ptx) < p.W;
            const float* src = p.dy + ((size_t)(b0 + ptb) * p.CoutTot + co0) * p.HW + (y0 + pty) * p.W + (x0 + ptx);
#pragma unroll 4
            for (int j = 0; j < CT / 4; ++j) {
                const int co = (tid >> 6) + 4 * j;
                const float v = (pv && (co0 + co) < p.Cout) ? src[(size_t)co * p.HW] : 0.f;
                dYl[co * p.PTP + pl] = v;
            }
        }
        // ---- stage X patch: [ci][patch word]
        {
            const float* src = p.x + ((size_t)b0 * p.CinTot + ci0) * p.HW + y0 * p.W + x0;
            int ci = tid / p.PS, r = tid - ci * p.PS;
            const int total = CT * p.PS;
#pragma unroll 4
            for (int e = tid; e < total; e += 256) {
                const int crd = tabCrd[r];
                const int tb = crd >> 20, py = (crd >> 10) & 1023, px = crd & 1023;
                const int b = b0 + tb, yy = y0 + py - HALO, xx = x0 + px - HALO;
                const bool v = b < p.N && yy >= 0 && yy < p.H && xx >= 0 && xx < p.W && (ci0 + ci) < p.Cin;
                Xl[ci * p.PSP + r] = v ? src[(size_t)ci * p.HW + tabOff[r]] : 0.f;
                ci += q256; r += r256;
                if (r >= p.PS) { r -= p.PS; ++ci; }
            }
        }
        __syncthreads();
        // ---- MFMA over the tile's pixels, two per instruction
        {
            const float* Ab = dYl + (wm * 32 + l31) * p.PTP + h;
            const float* Bb = Xl + (wn * 32 + l31) * p.PSP;
#pragma unroll 2
            for (int s = 0; s < nsteps; ++s) {
                const int pp = 2 * s + h;
                const int tx = pp & (p.TW - 1), ty = (pp >> p.lgTW) & (p.TH - 1), tb = pp >> (p.lgTW + p.lgTH);
                const float a = Ab[2 * s];
                const float* bp = Bb + tb * p.PSI + ty * p.PW + tx;
#pragma unroll
                for (int tap = 0; tap < KK; ++tap) {
                    const float b = bp[(tap / KS) * p.PW + (tap % KS)];
                    acc[tap] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[tap], 0, 0, 0);
                }
            }
        }
        __syncthreads();
    }

    // ---- partial slab [split][tap][co][ci]
    float* out = p.slab + (size_t)split * KK * p.Cout * p.Cin;
    const int ci = ci0 + wn * 32 + l31;
#pragma unroll
    for (int tap = 0; tap < KK; ++tap)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int co = co0 + wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
            if (co < p.Cout && ci < p.Cin) out[((size_t)tap * p.Cout + co) * p.Cin + ci] = acc[tap][r];
        }
}

// dW[co][ci][tap] = sum_s slab[s][tap][co][ci]  (fixed order)
__global__ __launch_bounds__(256) void wgrad_reduce_kernel(const float* __restrict__ slab, float* __restrict__ dw,
                                                            int S, int KK, int Cout, int Cin) {
    const int n = KK * Cout * Cin;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) {
        float s = 0.f;
        for (int k = 0; k < S; ++k) s += slab[(size_t)k * n + i];
        const int tap = i / (Cout * Cin), rem = i - tap * (Cout * Cin);
        dw[(size_t)rem * KK + tap] = s;
    }
}

// db[c] = sum_{b,y,x} dy[b,c,y,x]: one block per channel, ordered.
__global__ __launch_bounds__(256) void channel_sum_kernel(const float* __restrict__ dy, int Ctot, int N, int HW,
                                                           float* __restrict__ db) {
    __shared__ double sm[4];
    const int c = blockIdx.x;
    double s[1] = {0.0};
    const int total = N * HW;
    for (int i = threadIdx.x; i < total; i += 256) {
        const int b = i / HW, q = i - b * HW;
        s[0] += dy[((size_t)b * Ctot + c) * HW + q];
    }
    uz::block_sum_d<1>(s, sm);
    if (threadIdx.x == 0) db[c] = (float)s[0];
}

struct WGeom { int TW, TH, TB, PW, PSI, PS, tilesX, tilesY, tilesB, T, S, nCoT, nCiT; };

WGeom pick_wgeom(int Cin, int Cout, int N, int H, int W, int halo) {
    WGeom g;
    g.TW = W >= 32 ? 32 : uz::pow2_ceil(W);
    g.TH = uz::pow2_ceil(H);
    if (g.TH > 64 / g.TW) g.TH = 64 / g.TW;
    if (g.TH < 1) g.TH = 1;
    g.TB = 64 / (g.TW * g.TH);
    if (g.TB > uz::pow2_ceil(N)) g.TB = uz::pow2_ceil(N);
    g.PW = g.TW + 2 * halo;
    g.PSI = (g.TH + 2 * halo) * g.PW;
    while (g.TB > 1 && g.TB * g.PSI > 288) g.TB >>= 1;
    g.PS = g.TB * g.PSI;
    g.tilesX = uz::ceil_div(W, g.TW);
    g.tilesY = uz::ceil_div(H, g.TH);
    g.tilesB = uz::ceil_div(N, g.TB);
    g.T = g.tilesX * g.tilesY * g.tilesB;
    g.nCoT = uz::ceil_div(Cout, CT);
    g.nCiT = uz::ceil_div(Cin, CT);
    int s = 1024 / (g.nCoT * g.nCiT);
    if (s < 1) s = 1;
    if (s > g.T) s = g.T;
    if (s > 512) s = 512;
    g.S = s;
    return g;
}

}  // namespace

extern "C" size_t uz_conv_bwd_weight_workspace(int Cin, int Cout, int N, int H, int W, int ks) {
    const WGeom g = pick_wgeom(Cin, Cout, N, H, W, ks / 2);
    return (size_t)g.S * ks * ks * Cout * Cin * sizeof(float);
}

extern "C" int uz_conv_bwd_weight(const float* x, int Cin, int CinTot, const float* dy, int Cout, int CoutTot,
                                  float* dw, float* db, int N, int H, int W, int ks,
                                  void* workspace, size_t workspace_bytes, void* stream) {
    UZ_REQUIRE(ks == 1 || ks == 3, "conv_bwd_weight: kernel size %d unsupported", ks);
    UZ_REQUIRE(N > 0 && H > 0 && W > 0 && Cin > 0 && Cout > 0, "conv_bwd_weight: empty tensor");
    UZ_REQUIRE(H < 1024 && W < 1024, "conv_bwd_weight: spatial size too large");
    const WGeom g = pick_wgeom(Cin, Cout, N, H, W, ks / 2);
    const size_t need = (size_t)g.S * ks * ks * Cout * Cin * sizeof(float);
    UZ_REQUIRE(workspace && workspace_bytes >= need, "conv_bwd_weight: workspace %zu < %zu bytes", workspace_bytes, need);
    hipStream_t st = uz::S(stream);
    WgP p;
    p.x = x; p.dy = dy; p.slab = static_cast<float*>(workspace);
    p.N = N; p.H = H; p.W = W; p.HW = H * W;
    p.Cin = Cin; p.CinTot = CinTot; p.Cout = Cout; p.CoutTot = CoutTot;
    p.TW = g.TW; p.TH = g.TH; p.TB = g.TB; p.lgTW = uz::ilog2(g.TW); p.lgTH = uz::ilog2(g.TH);
    p.tilesX = g.tilesX; p.tilesY = g.tilesY; p.T = g.T; p.S = g.S;
    p.PW = g.PW; p.PSI = g.PSI; p.PS = g.PS; p.PSP = g.PS | 1;
    p.PT = 64; p.PTP = 65;
    p.nCoT = g.nCoT; p.nCiT = g.nCiT;
    const size_t smem = ((size_t)CT * p.PTP + (size_t)CT * p.PSP + 64 + 2 * (size_t)p.PS) * sizeof(float);
    const int grid = g.nCoT * g.nCiT * g.S;
    static bool attr3 = false, attr1 = false;
    if (ks == 3) {
        if (!attr3) {
            if (hipFuncSetAttribute(reinterpret_cast<const void*>(wgrad_kernel<3>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess)
                return uz::fail("wgrad: cannot raise dynamic LDS limit");
            attr3 = true;
        }
        hipLaunchKernelGGL(wgrad_kernel<3>, dim3(grid), dim3(256), smem, st, p);
    } else {
        if (!attr1) {
            if (hipFuncSetAttribute(reinterpret_cast<const void*>(wgrad_kernel<1>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess)
                return uz::fail("wgrad: cannot raise dynamic LDS limit");
            attr1 = true;
        }
        hipLaunchKernelGGL(wgrad_kernel<1>, dim3(grid), dim3(256), smem, st, p);
    }
    if (int rc = uz::check_launch("wgrad_kernel")) return rc;
    const int n = ks * ks * Cout * Cin;
    int rgrid = uz::ceil_div(n, 256);
    if (rgrid > 4096) rgrid = 4096;
    hipLaunchKernelGGL(wgrad_reduce_kernel, dim3(rgrid), dim3(256), 0, st, p.slab, dw, g.S, ks * ks, Cout, Cin);
    if (int rc = uz::check_launch("wgrad_reduce_kernel")) return rc;
    if (db) {
        hipLaunchKernelGGL(channel_sum_kernel, dim3(Cout), dim3(256), 0, st, dy, CoutTot, N, H * W, db);
        if (int rc = uz::check_launch("channel_sum_kernel")) return rc;
    }
    return 0;
}
