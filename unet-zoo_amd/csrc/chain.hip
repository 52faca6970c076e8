// Deep-level chain (round 6): the 16 x 16 ... 2 x 2 levels of PHiSeg as PHASES of one persistent launch.
//
// Reference: phiseg.py:14-39 (DownConvolutionalBlock, levels 3 - 6), :42-73 (UpConvolutionalBlock), :76-106 (SampleZBlock),
// :209-221 / :269-277 (the likelihood's small planes), each Conv2D unit = torchlayers.py:18-21.  Per training step these levels
// are ~340 launches of 5 - 40 us on the step's critical chains, and beside the device-filling convolutions of the other lanes
// every one of them waits 80 - 180 us for a register slot (profiles/NOTES_r5.md section 4).  Here a sub-DAG of such ops is
// ONE launch of resident workgroups (1024 threads = four waves per SIMD at <= 128 VGPRs, one workgroup per CU: the first form of this
// file ran 256-thread workgroups with one wave per SIMD and was latency-bound everywhere - a tile is a chain of dependent trips to
// L2, and only other waves hide them) that walk a phase table built by the host (unet-zoo_amd/_plan.py, Plan._chain_pass):
//
//   for phase:  for tile = workgroup; tile < tiles of the phase; tile += workgroups:  run the tile of the sub-op it falls into
//               grid barrier among the launch's own workgroups
//
// Sub-ops of one phase are independent (the host levels the sub-DAG), so the tiny planes of the posterior, the prior and the
// likelihood fill a phase together.  Design points:
//   * 3 x 3 convolutions: fp16 matrix pipe with two-piece split operands (split_f16.h: a 22-bit emulation, three products per
//     operand pair, fp32 accumulation - the arithmetic of conv_split.hip).  A WAVE owns a 64-channel x 32-pixel output tile over a
//     slice of the contraction; operands go global memory -> registers -> MFMA, no LDS and no workgroup barrier inside a tile:
//     the weights come from an image packed once per tape in fragment order (one 16-byte load per lane and fragment), the
//     activations as eight channel-strided dwords per lane (coalesced over the 32 pixels of a fragment column), split on the way.
//     Planes this small live in L2; what the launch is sized by is latency, not bytes.
//   * split-K: the host cuts a convolution's contraction into S slices so that a phase has enough tiles; slices leave partial
//     sums (slabs) that the unit's BatchNorm adds in slab order (bias first) - deterministic, no atomics on data.
//   * BatchNorm: one workgroup per channel, the channel's batch (<= 8192 values) in registers, fp64 statistics in a fixed order.
//   * cross-workgroup visibility, two forms (template SC1; UZ_CHAIN_SC1 picks): (0) every workgroup ends a phase with s_waitcnt
//     vmcnt(0) + an agent-scope release fence before it arrives on the barrier counter and starts the next with an agent-scope
//     acquire fence (MI355X_MICROARCH.md, barrier-counter) - every phase then starts with cold caches; (1) every tensor element that
//     one workgroup writes and another reads inside the launch is stored write-through and loaded at agent scope (relaxed agent-scope
//     atomics = global_store / global_load ... sc1), every storing wave drains (vmcnt(0)) before its workgroup's one lane arrives, and
//     no cache is flushed: parameters, weight images and the op table stay cached (MI355X_MICROARCH.md, hand-off table, first row).
//     Bound slots and the barrier words are only touched with agent-scope atomics in both forms.  Every spin is bounded: a barrier
//     that does not complete raises the status word, every workgroup leaves, uz_chain_status reports the phase.
#include <stdlib.h>
#include "uz_common.h"
#include "split_f16.h"

namespace {

using uz::f32x16; using uz::f16x8; using uz::u32x4;

constexpr int NT = 1024, NW = NT / 64;   // threads / waves per workgroup
constexpr int PXB = 32;            // pixels per wave tile (one MFMA column block)
constexpr int COB = 64;            // output channels per wave tile (two MFMA row blocks)
constexpr int BN_MAX_EPT = 8;      // values per thread of a BatchNorm tile: N*H*W <= 8192
constexpr int MIN_KSTEPS = 6;      // 16-deep k-steps a split-K slice keeps at least
constexpr int MAX_OPS = 640, MAX_PHASES = 256;      // directory kept in LDS

// state words (unsigned), each on a 64-byte line of its own
constexpr int ST_ARRIVE = 0, ST_STATUS = 16, ST_STAMPS = 32, MAX_STAMPS = 512, ST_DBG = ST_STAMPS + 2 * MAX_STAMPS, MAX_DBG = 64, ST_WORDS = ST_DBG + 2 * MAX_DBG;      // stamps: 100 MHz clock of workgroup 0 at launch and behind every phase (diagnostics)
constexpr unsigned SPIN_LIMIT = 4u << 20;     // polls of ~0.5 us: ~2 s

// diagnostics (UZ_CHAIN_DEBUG_PHASE): in-tile stamps of ONE workgroup in ONE phase; null otherwise
__shared__ unsigned long long* g_dbg;
#define STAMP(id) do { if (g_dbg && threadIdx.x == 0) g_dbg[id] = wall_clock64(); } while (0)

// every shared word is a GLOBAL agent-scope access (address_space(1): global_load / global_store ... sc1, never flat)
typedef __attribute__((address_space(1))) unsigned gu32;
typedef __attribute__((address_space(1))) const unsigned gcu32;
__device__ __forceinline__ unsigned ld_agent(const unsigned* p) { return __hip_atomic_load((gcu32*)p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
// hand-off accesses: data another workgroup of this launch wrote / will read
template <bool SC1> __device__ __forceinline__ float ldh(const float* p) {
    if constexpr (SC1) return __builtin_bit_cast(float, __hip_atomic_load((gcu32*)p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
    else return *(__attribute__((address_space(1))) const float*)p;
}
template <bool SC1> __device__ __forceinline__ void sth(float* p, float v) {
    if constexpr (SC1) __hip_atomic_store((gu32*)p, __builtin_bit_cast(unsigned, v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    else *(__attribute__((address_space(1))) float*)p = v;
}
__device__ __forceinline__ float amax_read_agent(const float* slot) {
    float m = 0.f;
#pragma unroll
    for (int i = 0; i < uz::AMAX_SUB; ++i)
        m = fmaxf(m, __builtin_bit_cast(float, ld_agent(reinterpret_cast<const unsigned*>(slot) + i * uz::AMAX_STRIDE)));
    return m;
}
__device__ __forceinline__ void amax_publish_key(float m, float* slot, unsigned key) {
    if (m > 0.f) uz::amax_publish_one(m, slot, key);
}

// Grid barrier among the launch's workgroups.  Returns false when the launch is to be abandoned (status word raised).
template <bool SC1>
__device__ __forceinline__ bool grid_barrier(unsigned* st, unsigned n_wg, unsigned epoch, unsigned* verdict) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");           // every wave: its stores have left
    __syncthreads();
    if (threadIdx.x == 0) {
        if constexpr (!SC1) {
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // (the compiler may drop the wait behind the write-back: cdna guide, compiler hazard)
        }
        __hip_atomic_fetch_add((gu32*)(st + ST_ARRIVE), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const unsigned target = n_wg * epoch;
        unsigned ok = 1, spins = 0;
        while (ld_agent(st + ST_ARRIVE) < target) {
            __builtin_amdgcn_s_sleep(1);
            if (++spins > SPIN_LIMIT || ((spins & 1023u) == 0u && ld_agent(st + ST_STATUS) != 0u)) {
                __hip_atomic_fetch_max((gu32*)(st + ST_STATUS), epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                ok = 0;
                break;
            }
        }
        if constexpr (!SC1) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        *verdict = ok;
    }
    __syncthreads();
    return *verdict != 0u;
}

// ---------------------------------------------------------------------------------------------- 3 x 3 convolution, matrix pipe
// wave tile wt -> (slice s, channel block cb, pixel block pb), pixel block fastest
template <bool SC1>
__device__ __forceinline__ void conv3_wave_tile(const uz_chain_op& o, int wt, int lane) {
    const int Kc = o.i[0], KcTot = o.i[1], Mc = o.i[2], McTot = o.i[3], N = o.i[4], H = o.i[5], W = o.i[6], S = o.i[7], accumulate = o.i[8];
    const int HW = H * W, P = N * HW, PB = (P + PXB - 1) / PXB, CB = (Mc + COB - 1) / COB;
    const int pb = wt % PB, r = wt / PB, cb = r % CB, s = r / CB;
    if (s >= S) return;
    const float* x = static_cast<const float*>(o.p[0]);
    const char* __restrict__ img = static_cast<const char*>(o.p[1]);
    const int half = lane >> 5, l31 = lane & 31;
    const int p = pb * PXB + l31;
    const bool pvalid = p < P;
    const int pc = pvalid ? p : P - 1;
    const int n = pc / HW, hw = pc - n * HW, h = hw / W, w = hw - h * W;
    const float ax = amax_read_agent(static_cast<const float*>(o.p[5])), aw = amax_read_agent(static_cast<const float*>(o.p[6]));
    const float sx = uz::split_scale(ax);
    const int KB = Kc / 16, M32 = Mc / 32, T = 9 * KB;
    const int q0 = (int)((long long)s * T / S), q1 = (int)((long long)(s + 1) * T / S);
    const float* xb = x + ((size_t)n * KcTot + 8 * half) * HW;
    const bool second = cb * 2 + 1 < M32;                       // the tile's upper 32 channels exist (Mc % 64 == 32: last block is half)
    f32x16 acc0, acc1;
#pragma unroll
    for (int i = 0; i < 16; ++i) { acc0[i] = 0.f; acc1[i] = 0.f; }
    int tap = q0 / KB, kb = q0 - tap * KB;
    for (int q = q0; q < q1; ++q) {
        const int dy = tap / 3 - 1, dx = tap - (tap / 3) * 3 - 1;
        const int hh = h + dy, ww = w + dx;
        const bool inb = pvalid && hh >= 0 && hh < H && ww >= 0 && ww < W;
        const float* src = xb + (size_t)(kb * 16) * HW + (inb ? hh * W + ww : hw);
        float v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = ldh<SC1>(src + (size_t)j * HW);
        // weight fragments: [(tap * KB + kb)][m32][piece][lane][8 halfs]
        const char* wsrc = img + ((size_t)(tap * KB + kb) * M32 + cb * 2) * 2048 + lane * 16;
        u32x4 a0[2], a1[2];
        a0[0] = *reinterpret_cast<const u32x4*>(wsrc);
        a0[1] = *reinterpret_cast<const u32x4*>(wsrc + 1024);
        if (second) {
            a1[0] = *reinterpret_cast<const u32x4*>(wsrc + 2048);
            a1[1] = *reinterpret_cast<const u32x4*>(wsrc + 3072);
        }
        u32x4 b[2];
        {
            unsigned p1[4], p2[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float v0 = inb ? v[2 * j] * sx : 0.f, v1 = inb ? v[2 * j + 1] * sx : 0.f;
                uz::split2(v0, v1, p1[j], p2[j]);
            }
            b[0] = u32x4{p1[0], p1[1], p1[2], p1[3]};
            b[1] = u32x4{p2[0], p2[1], p2[2], p2[3]};
        }
        // smallest products first (conv_split.hip: a2 b1, a1 b2, a1 b1)
        acc0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a0[1]), __builtin_bit_cast(f16x8, b[0]), acc0, 0, 0, 0);
        acc0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a0[0]), __builtin_bit_cast(f16x8, b[1]), acc0, 0, 0, 0);
        acc0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a0[0]), __builtin_bit_cast(f16x8, b[0]), acc0, 0, 0, 0);
        if (second) {
            acc1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a1[1]), __builtin_bit_cast(f16x8, b[0]), acc1, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a1[0]), __builtin_bit_cast(f16x8, b[1]), acc1, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a1[0]), __builtin_bit_cast(f16x8, b[0]), acc1, 0, 0, 0);
        }
        if (++kb == KB) { kb = 0; ++tap; }
    }
    if (!pvalid) return;
    const float inv = uz::split_inv_scale(ax) * uz::split_inv_scale(aw);
    const float* __restrict__ bias = static_cast<const float*>(o.p[2]);
    if (S == 1) {
        float* y = static_cast<float*>(o.p[3]) + (size_t)n * McTot * HW + hw;
#pragma unroll
        for (int m = 0; m < 2; ++m) {
            if (m == 1 && !second) break;
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const int co = cb * COB + m * 32 + (i >> 2) * 8 + 4 * half + (i & 3);
                float val = (m == 0 ? acc0[i] : acc1[i]) * inv + (bias ? bias[co] : 0.f);
                float* dst = y + (size_t)co * HW;
                if (accumulate) val += ldh<SC1>(dst);
                sth<SC1>(dst, val);
            }
        }
    } else {
        float* sl = static_cast<float*>(o.p[4]) + ((size_t)s * N + n) * Mc * HW + hw;
#pragma unroll
        for (int m = 0; m < 2; ++m) {
            if (m == 1 && !second) break;
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const int co = cb * COB + m * 32 + (i >> 2) * 8 + 4 * half + (i & 3);
                sth<SC1>(sl + (size_t)co * HW, (m == 0 ? acc0[i] : acc1[i]) * inv);
            }
        }
    }
}
__host__ __device__ inline int conv3_wave_tiles(const uz_chain_op& o) {
    const int P = o.i[4] * o.i[5] * o.i[6];
    return ((P + PXB - 1) / PXB) * ((o.i[2] + COB - 1) / COB) * o.i[7];
}

// ---------------------------------------------------------------------------------------------- 3 x 3 convolution, <= 4 input channels
// tile = NT pixels x 16 output channels; a thread keeps its pixel's 9 Kc inputs in registers
constexpr int SMALL_CO = 16;
template <bool SC1>
__device__ __forceinline__ void conv3_small_tile(const uz_chain_op& o, int tile) {
    const int Kc = o.i[0], KcTot = o.i[1], Mc = o.i[2], McTot = o.i[3], N = o.i[4], H = o.i[5], W = o.i[6];
    const int HW = H * W, P = N * HW, PT = (P + NT - 1) / NT;
    const int pt = tile % PT, ct = tile / PT;
    const int p = pt * NT + threadIdx.x;
    if (p >= P) return;
    const int n = p / HW, hw = p - n * HW, h = hw / W, w = hw - h * W;
    const float* x = static_cast<const float*>(o.p[0]) + (size_t)n * KcTot * HW;
    const float* __restrict__ wt = static_cast<const float*>(o.p[1]);
    const float* __restrict__ bias = static_cast<const float*>(o.p[2]);
    float xv[4][9];
#pragma unroll
    for (int c = 0; c < 4; ++c)
#pragma unroll
        for (int t = 0; t < 9; ++t) {
            const int hh = h + t / 3 - 1, ww = w + t % 3 - 1;
            const bool inb = c < Kc && hh >= 0 && hh < H && ww >= 0 && ww < W;
            const float ld = ldh<SC1>(x + (inb ? (size_t)c * HW + hh * W + ww : (size_t)hw));      // (every load unconditional: a branch per load serialises the trips)
            xv[c][t] = inb ? ld : 0.f;
        }
    float* y = static_cast<float*>(o.p[3]) + (size_t)n * McTot * HW + hw;
    const int co0 = ct * SMALL_CO, co1 = min(Mc, co0 + SMALL_CO);
    for (int co = co0; co < co1; ++co) {
        float acc = bias ? bias[co] : 0.f;
        const float* wr = wt + (size_t)co * Kc * 9;             // uniform address: scalar loads (parameters: never written inside the launch)
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            if (c < Kc) {
#pragma unroll
                for (int t = 0; t < 9; ++t) acc = fmaf(wr[c * 9 + t], xv[c][t], acc);
            }
        }
        sth<SC1>(y + (size_t)co * HW, acc);
    }
}
// data gradient of the same layers: dx (Kc <= 4 channels) = sum over Mc channels and taps; tile = 64 pixels, the waves split the
// Mc channels and add their partial sums through LDS in wave order
template <bool SC1>
__device__ __forceinline__ void conv3_small_bwd_tile(const uz_chain_op& o, int tile, float* sm /* >= NW * 4 * 64 */) {
    const int Kc = o.i[0], KcTot = o.i[1], Mc = o.i[2], McTot = o.i[3], N = o.i[4], H = o.i[5], W = o.i[6], accumulate = o.i[7];
    const int HW = H * W, P = N * HW;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int p = tile * 64 + lane;
    const bool pvalid = p < P;
    const int pc = pvalid ? p : P - 1;
    const int n = pc / HW, hw = pc - n * HW, h = hw / W, w = hw - h * W;
    const float* dy = static_cast<const float*>(o.p[0]) + (size_t)n * McTot * HW;
    const float* __restrict__ wt = static_cast<const float*>(o.p[1]);
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
    const int cpw = (Mc + NW - 1) / NW, c0 = wave * cpw, c1 = min(Mc, c0 + cpw);
    for (int co = c0; co < c1; ++co) {
        const float* wr = wt + (size_t)co * Kc * 9;
#pragma unroll
        for (int t = 0; t < 9; ++t) {
            // dx[p] += dy[p - tap offset] * w[co][ci][t]  <=>  gather dy at (h - dy_t, w - dx_t)
            const int hh = h - (t / 3 - 1), ww = w - (t % 3 - 1);
            const bool inb = hh >= 0 && hh < H && ww >= 0 && ww < W;
            const float gl = ldh<SC1>(dy + (size_t)co * HW + (inb ? hh * W + ww : hw));
            const float g = inb ? gl : 0.f;
#pragma unroll
            for (int c = 0; c < 4; ++c)
                if (c < Kc) acc[c] = fmaf(wr[c * 9 + t], g, acc[c]);
        }
    }
#pragma unroll
    for (int c = 0; c < 4; ++c) sm[(wave * 4 + c) * 64 + lane] = acc[c];
    __syncthreads();
    if (wave == 0 && pvalid) {
        float* dx = static_cast<float*>(o.p[2]) + (size_t)n * KcTot * HW + hw;
        for (int c = 0; c < Kc; ++c) {
            float v = 0.f;
            for (int g = 0; g < NW; ++g) v += sm[(g * 4 + c) * 64 + lane];
            if (accumulate) v += ldh<SC1>(dx + (size_t)c * HW);
            sth<SC1>(dx + (size_t)c * HW, v);
        }
    }
    __syncthreads();
}

// block-wide sums of NV doubles over NT threads; result valid in thread 0 (wave partial sums added in wave order)
template <int NV>
__device__ __forceinline__ void block_sum(double (&v)[NV], double* smem /* >= NW * NV */) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int i = 0; i < NV; ++i) v[i] = uz::wave_sum_d(v[i]);
    __syncthreads();
    if (lane == 0) {
#pragma unroll
        for (int i = 0; i < NV; ++i) smem[wave * NV + i] = v[i];
    }
    __syncthreads();
    if (threadIdx.x == 0) {
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            double t = smem[i];
            for (int g = 1; g < NW; ++g) t += smem[g * NV + i];
            v[i] = t;
        }
    }
}

// ---------------------------------------------------------------------------------------------- BatchNorm + ReLU, forward
// one workgroup per channel; element i = tid + NT j of the channel's batch lives in register j; fp64 statistics, fixed order
template <bool SC1>
__device__ __forceinline__ void bn_fwd_tile(const uz_chain_op& o, int c, double* smd /* >= 2 NW */, float* smf /* >= 2 */) {
    const int C = o.i[0], CtotY = o.i[1], CtotA = o.i[2], N = o.i[3], HW = o.i[4], relu = o.i[5], S = o.i[6];
    const float eps = o.f[0], momentum = o.f[1];
    const int total = N * HW, tid = threadIdx.x;
    STAMP(9);
    float* y = static_cast<float*>(o.p[0]);
    float v[BN_MAX_EPT];
    unsigned ey[BN_MAX_EPT];                   // element offsets (tensors of the chain are far below 2^32 elements)
#pragma unroll
    for (int j = 0; j < BN_MAX_EPT; ++j) {
        const int i = tid + NT * j, ic = i < total ? i : 0;
        const int b = ic / HW, q = ic - b * HW;
        ey[j] = (unsigned)(((size_t)b * CtotY + c) * HW + q);
    }
    if (S > 1) {
        const float* slab = static_cast<const float*>(o.p[7]);
        const float* cbias = static_cast<const float*>(o.p[9]);
        const float bv = cbias ? cbias[c] : 0.f;
        const size_t nsl = (size_t)N * C * HW;
        unsigned es[BN_MAX_EPT];
#pragma unroll
        for (int j = 0; j < BN_MAX_EPT; ++j) {
            const int i = tid + NT * j, ic = i < total ? i : 0;
            const int b = ic / HW, q = ic - b * HW;
            es[j] = (unsigned)(((size_t)b * C + c) * HW + q);
            v[j] = bv;
        }
        // slabs added in slab order; a chunk's loads are all in flight together (a load of handed-off data is a trip past L2: the loop
        // is paid in trips, not bytes): 24 slabs at a time where the thread holds one element (planes up to 1024 values), else 2
        if (total <= NT) {
            for (int k0 = 0; k0 < S; k0 += 24) {
                float t[24];
#pragma unroll
                for (int u = 0; u < 24; ++u) t[u] = ldh<SC1>(slab + (size_t)min(k0 + u, S - 1) * nsl + es[0]);
#pragma unroll
                for (int u = 0; u < 24; ++u) v[0] += (k0 + u < S) ? t[u] : 0.f;
            }
        } else {
            for (int k0 = 0; k0 < S; k0 += 2) {
                float t[2][BN_MAX_EPT];
#pragma unroll
                for (int u = 0; u < 2; ++u)
#pragma unroll
                    for (int j = 0; j < BN_MAX_EPT; ++j) t[u][j] = ldh<SC1>(slab + (size_t)min(k0 + u, S - 1) * nsl + es[j]);
#pragma unroll
                for (int u = 0; u < 2; ++u)
#pragma unroll
                    for (int j = 0; j < BN_MAX_EPT; ++j) v[j] += (k0 + u < S) ? t[u][j] : 0.f;
            }
        }
        STAMP(10);
#pragma unroll
        for (int j = 0; j < BN_MAX_EPT; ++j)
            if (tid + NT * j < total) sth<SC1>(y + ey[j], v[j]);
    } else {
#pragma unroll
        for (int j = 0; j < BN_MAX_EPT; ++j) { const float ld = ldh<SC1>(y + ey[j]); v[j] = tid + NT * j < total ? ld : 0.f; }
    }
    STAMP(1);
    double v2[2] = {0.0, 0.0};
#pragma unroll
    for (int j = 0; j < BN_MAX_EPT; ++j)
        if (tid + NT * j < total) { const double d = v[j]; v2[0] += d; v2[1] += d * d; }
    STAMP(2);
    block_sum<2>(v2, smd);
    STAMP(3);
    if (tid == 0) {
        float* save = static_cast<float*>(o.p[5]);
        float* rmean = static_cast<float*>(o.p[3]);
        float* rvar = static_cast<float*>(o.p[4]);
        const double n = (double)total;
        const double mean = v2[0] / n;
        double var = v2[1] / n - mean * mean;
        if (var < 0.0) var = 0.0;
        save[c] = (float)mean;
        save[C + c] = (float)(1.0 / sqrt(var + (double)eps));
        if (rmean) {
            const double unb = n > 1.0 ? var * n / (n - 1.0) : var;
            rmean[c] = (float)((1.0 - momentum) * rmean[c] + momentum * mean);
            rvar[c] = (float)((1.0 - momentum) * rvar[c] + momentum * unb);
        }
        smf[0] = save[c]; smf[1] = save[C + c];
    }
    STAMP(4);
    __syncthreads();
    STAMP(5);
    const float mean = smf[0], rstd = smf[1];
    const float* gamma = static_cast<const float*>(o.p[1]);
    const float* beta = static_cast<const float*>(o.p[2]);
    const float g = gamma ? gamma[c] : 1.f, bb = beta ? beta[c] : 0.f;
    const float alpha = g * rstd, beta_ = bb - mean * alpha;
    const float floor_ = relu ? 0.f : -INFINITY;
    float* a = static_cast<float*>(o.p[6]);
    float vmax = 0.f;
#pragma unroll
    for (int j = 0; j < BN_MAX_EPT; ++j) {
        const int i = tid + NT * j;
        if (i < total) {
            const int b = i / HW, q = i - b * HW;
            const float rr = fmaxf(fmaf(v[j], alpha, beta_), floor_);
            sth<SC1>(a + ((size_t)b * CtotA + c) * HW + q, rr);
            vmax = fmaxf(vmax, fabsf(rr));
        }
    }
    STAMP(6);
    float* amax = static_cast<float*>(o.p[8]);
    if (amax) {
#pragma unroll
        for (int ofs = 32; ofs > 0; ofs >>= 1) vmax = fmaxf(vmax, __shfl_xor(vmax, ofs, 64));
        if ((tid & 63) == 0) amax_publish_key(vmax, amax, (unsigned)c * NW + (tid >> 6));
    }
    STAMP(7);
    __syncthreads();                                            // smd / smf are reused by the workgroup's next tile
    STAMP(8);
}

// ---------------------------------------------------------------------------------------------- BatchNorm + ReLU, backward
// dz = dA * (a > 0); dgamma = sum dz x_hat, dbeta = sum dz; dy = alpha (dz - mean(dz) - x_hat mean(dz x_hat)); dbias = sum dy
// (the arithmetic of bn_fused_small_bwd, bn.hip)
template <bool SC1>
__device__ __forceinline__ void bn_bwd_tile(const uz_chain_op& o, int c, double* smd /* >= 2 NW */, float* smf /* >= 2 */) {
    const int C = o.i[0], CtotDa = o.i[1], CtotY = o.i[2], N = o.i[3], HW = o.i[4], relu = o.i[5], S = o.i[6];
    const int total = N * HW, tid = threadIdx.x;
    const float* da = static_cast<const float*>(o.p[0]);
    const float* __restrict__ y = static_cast<const float*>(o.p[1]);      // (written by the forward tape: an earlier launch)
    const float* gamma = static_cast<const float*>(o.p[2]);
    const float* save = static_cast<const float*>(o.p[3]);
    const float* beta = static_cast<const float*>(o.p[10]);
    const float mean = save[c], rstd = save[C + c];
    const float g = gamma ? gamma[c] : 1.f, bb = beta ? beta[c] : 0.f;
    const float alpha = g * rstd, beta_ = bb - mean * alpha;
    float dz[BN_MAX_EPT], xh[BN_MAX_EPT];
    const float* slab = static_cast<const float*>(o.p[9]);
    const size_t nsl = (size_t)N * C * HW;
    unsigned es[BN_MAX_EPT];
    unsigned masked = 0u;
#pragma unroll
    for (int j = 0; j < BN_MAX_EPT; ++j) {
        const int i = tid + NT * j, ic = i < total ? i : 0;
        const int b = ic / HW, q = ic - b * HW;
        const bool ok = i < total;
        const float yl = y[((size_t)b * CtotY + c) * HW + q];
        const float yv = ok ? yl : 0.f;
        xh[j] = (yv - mean) * rstd;
        es[j] = (unsigned)(((size_t)b * C + c) * HW + q);
        const float dl = ldh<SC1>(S <= 1 ? da + ((size_t)b * CtotDa + c) * HW + q : slab + es[j]);      // (S > 1: a dummy address, value unused)
        dz[j] = (ok && S <= 1) ? dl : 0.f;
        if (relu && !(fmaf(yv, alpha, beta_) > 0.f)) masked |= 1u << j;               // ReLU mask: dz = 0 there (x_hat still enters dy)
    }
    if (S > 1) {
        if (total <= NT) {
            for (int k0 = 0; k0 < S; k0 += 24) {
                float t[24];
#pragma unroll
                for (int u = 0; u < 24; ++u) t[u] = ldh<SC1>(slab + (size_t)min(k0 + u, S - 1) * nsl + es[0]);
#pragma unroll
                for (int u = 0; u < 24; ++u) dz[0] += (k0 + u < S) ? t[u] : 0.f;
            }
        } else {
            for (int k0 = 0; k0 < S; k0 += 2) {
                float t[2][BN_MAX_EPT];
#pragma unroll
                for (int u = 0; u < 2; ++u)
#pragma unroll
                    for (int j = 0; j < BN_MAX_EPT; ++j) t[u][j] = ldh<SC1>(slab + (size_t)min(k0 + u, S - 1) * nsl + es[j]);
#pragma unroll
                for (int u = 0; u < 2; ++u)
#pragma unroll
                    for (int j = 0; j < BN_MAX_EPT; ++j) dz[j] += (k0 + u < S) ? t[u][j] : 0.f;
            }
        }
    }
    double v2[2] = {0.0, 0.0};
#pragma unroll
    for (int j = 0; j < BN_MAX_EPT; ++j) {
        if ((masked >> j) & 1u) dz[j] = 0.f;
        if (tid + NT * j < total) { v2[0] += (double)dz[j]; v2[1] += (double)(dz[j] * xh[j]); }
    }
    block_sum<2>(v2, smd);
    if (tid == 0) {
        float* dgamma = static_cast<float*>(o.p[5]);
        float* dbeta = static_cast<float*>(o.p[6]);
        if (dgamma) dgamma[c] = (float)v2[1];
        if (dbeta) dbeta[c] = (float)v2[0];
        smf[0] = (float)(v2[0] / total);
        smf[1] = (float)(v2[1] / total);
    }
    __syncthreads();
    const float m1 = smf[0], m2 = smf[1];
    float* dyo = static_cast<float*>(o.p[4]);
    float vmax = 0.f;
    double sd[1] = {0.0};
#pragma unroll
    for (int j = 0; j < BN_MAX_EPT; ++j) {
        if (tid + NT * j < total) {
            const float r = alpha * (dz[j] - m1 - xh[j] * m2);
            sth<SC1>(dyo + es[j], r);                              // dy: a contiguous [N][C][HW] tensor of its own
            vmax = fmaxf(vmax, fabsf(r));
            sd[0] += (double)r;
        }
    }
    float* dbias = static_cast<float*>(o.p[7]);
    block_sum<1>(sd, smd);
    if (tid == 0 && dbias) dbias[c] = (float)sd[0];
    float* amax = static_cast<float*>(o.p[8]);
    if (amax) {
#pragma unroll
        for (int ofs = 32; ofs > 0; ofs >>= 1) vmax = fmaxf(vmax, __shfl_xor(vmax, ofs, 64));
        if ((tid & 63) == 0) amax_publish_key(vmax, amax, (unsigned)c * NW + (tid >> 6));
    }
    __syncthreads();
}

// ---------------------------------------------------------------------------------------------- pooling / interpolation
constexpr int RS_PER_TILE = 4 * NT;          // outputs per tile: four per thread
__device__ __forceinline__ void forward_bound(const uz_chain_op& o, int tile) {
    const float* xa = static_cast<const float*>(o.p[2]);
    float* ya = static_cast<float*>(o.p[3]);
    if (xa && ya && tile == 0 && threadIdx.x == 0) amax_publish_key(amax_read_agent(xa), ya, 0u);
}
template <bool SC1>
__device__ __forceinline__ void avgpool_fwd_tile(const uz_chain_op& o, int tile) {
    const int C = o.i[0], CtotX = o.i[1], CtotY = o.i[2], N = o.i[3], Hi = o.i[4], Wi = o.i[5];
    const int H = (Hi + 1) / 2, W = (Wi + 1) / 2;
    forward_bound(o, tile);
    const float* x = static_cast<const float*>(o.p[0]);
    float* y = static_cast<float*>(o.p[1]);
    const long long total = (long long)N * C * H * W;
    // (all sixteen loads of a thread are issued before the first use: no branch around a load)
    float v[4][4];
    long long ee[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const long long e = (long long)tile * RS_PER_TILE + k * NT + threadIdx.x;
        ee[k] = e;
        const long long ec = e < total ? e : total - 1;
        const int ox = (int)(ec % W); long long t = ec / W;
        const int oy = (int)(t % H); t /= H;
        const int c = (int)(t % C), b = (int)(t / C);
        const float* s = x + ((size_t)b * CtotX + c) * Hi * Wi;
        const int y0 = 2 * oy, x0 = 2 * ox, y1 = min(y0 + 1, Hi - 1), x1 = min(x0 + 1, Wi - 1);
        v[k][0] = ldh<SC1>(s + y0 * Wi + x0); v[k][1] = ldh<SC1>(s + y0 * Wi + x1);
        v[k][2] = ldh<SC1>(s + y1 * Wi + x0); v[k][3] = ldh<SC1>(s + y1 * Wi + x1);
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        if (ee[k] >= total) continue;
        const int ox = (int)(ee[k] % W); long long t = ee[k] / W;
        const int oy = (int)(t % H); t /= H;
        const int c = (int)(t % C), b = (int)(t / C);
        const bool hx = 2 * ox + 1 < Wi, hy = 2 * oy + 1 < Hi;          // ceil mode: the last window of an odd plane is partial
        float acc = v[k][0];
        if (hx) acc += v[k][1];
        if (hy) { acc += v[k][2]; if (hx) acc += v[k][3]; }
        sth<SC1>(y + ((size_t)b * CtotY + c) * H * W + oy * W + ox, acc / (float)((hx ? 2 : 1) * (hy ? 2 : 1)));
    }
}
template <bool SC1>
__device__ __forceinline__ void avgpool_bwd_tile(const uz_chain_op& o, int tile) {
    const int C = o.i[0], CtotDy = o.i[1], CtotDx = o.i[2], N = o.i[3], Ho = o.i[4], Wo = o.i[5], accumulate = o.i[6];
    const int H = (Ho + 1) / 2, W = (Wo + 1) / 2;
    const float* dy = static_cast<const float*>(o.p[0]);
    float* dx = static_cast<float*>(o.p[1]);
    const long long total = (long long)N * C * Ho * Wo;
    float g[4], old[4];
    float* dst[4];
    int cnt[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const long long e = (long long)tile * RS_PER_TILE + k * NT + threadIdx.x;
        const long long ec = e < total ? e : total - 1;
        const int xq = (int)(ec % Wo); long long t = ec / Wo;
        const int yq = (int)(t % Ho); t /= Ho;
        const int c = (int)(t % C), b = (int)(t / C);
        const int oy = yq >> 1, ox = xq >> 1;
        cnt[k] = (min(2 * oy + 2, Ho) - 2 * oy) * (min(2 * ox + 2, Wo) - 2 * ox);
        g[k] = ldh<SC1>(dy + ((size_t)b * CtotDy + c) * H * W + oy * W + ox);
        dst[k] = e < total ? dx + ((size_t)b * CtotDx + c) * Ho * Wo + yq * Wo + xq : nullptr;
        old[k] = accumulate ? ldh<SC1>(dx + ((size_t)b * CtotDx + c) * Ho * Wo + yq * Wo + xq) : 0.f;
    }
#pragma unroll
    for (int k = 0; k < 4; ++k)
        if (dst[k]) sth<SC1>(dst[k], old[k] + g[k] / (float)cnt[k]);
}
__device__ __forceinline__ void src_index(int o, float scale, int ac, int in, int& i0, int& ip, float& l0, float& l1) {
    float r;
    if (ac) r = scale * (float)o;
    else { r = scale * ((float)o + 0.5f) - 0.5f; if (r < 0.f) r = 0.f; }
    i0 = (int)r;
    if (i0 > in - 1) i0 = in - 1;
    ip = (i0 < in - 1) ? 1 : 0;
    l1 = r - (float)i0;
    l0 = 1.f - l1;
}
__device__ __forceinline__ void bil_scales(int H, int W, int ac, float& sh, float& sw) {
    const int Ho = 2 * H, Wo = 2 * W;
    if (ac) { sh = Ho > 1 ? (float)(H - 1) / (float)(Ho - 1) : 0.f; sw = Wo > 1 ? (float)(W - 1) / (float)(Wo - 1) : 0.f; }
    else { sh = 0.5f; sw = 0.5f; }
}
template <bool SC1>
__device__ __forceinline__ void bilinear_fwd_tile(const uz_chain_op& o, int tile) {
    const int C = o.i[0], CtotX = o.i[1], CtotY = o.i[2], N = o.i[3], H = o.i[4], W = o.i[5], ac = o.i[6];
    const int Ho = 2 * H, Wo = 2 * W;
    forward_bound(o, tile);
    float sh, sw;
    bil_scales(H, W, ac, sh, sw);
    const float* x = static_cast<const float*>(o.p[0]);
    float* y = static_cast<float*>(o.p[1]);
    const long long total = (long long)N * C * Ho * Wo;
    float v[4][4], wgt[4][4];
    float* dst[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const long long e = (long long)tile * RS_PER_TILE + k * NT + threadIdx.x;
        const long long ec = e < total ? e : total - 1;
        const int ox = (int)(ec % Wo); long long t = ec / Wo;
        const int oy = (int)(t % Ho); t /= Ho;
        const int c = (int)(t % C), b = (int)(t / C);
        const float* s = x + ((size_t)b * CtotX + c) * H * W;
        int h1, hp, w1, wp;
        src_index(oy, sh, ac, H, h1, hp, wgt[k][0], wgt[k][1]);
        src_index(ox, sw, ac, W, w1, wp, wgt[k][2], wgt[k][3]);
        const float* r0 = s + h1 * W + w1;
        const float* r1 = r0 + hp * W;
        v[k][0] = ldh<SC1>(r0); v[k][1] = ldh<SC1>(r0 + wp); v[k][2] = ldh<SC1>(r1); v[k][3] = ldh<SC1>(r1 + wp);
        dst[k] = e < total ? y + ((size_t)b * CtotY + c) * Ho * Wo + oy * Wo + ox : nullptr;
    }
#pragma unroll
    for (int k = 0; k < 4; ++k)
        if (dst[k]) sth<SC1>(dst[k], wgt[k][0] * (wgt[k][2] * v[k][0] + wgt[k][3] * v[k][1]) + wgt[k][1] * (wgt[k][2] * v[k][2] + wgt[k][3] * v[k][3]));
}
__device__ __forceinline__ float tap_weight(int o, int osize, float scale, int ac, int isize, int i) {
    if (o < 0 || o >= osize) return 0.f;
    int i0, ip; float l0, l1;
    src_index(o, scale, ac, isize, i0, ip, l0, l1);
    return (i0 == i ? l0 : 0.f) + (i0 + ip == i ? l1 : 0.f);
}
// gather form (bilinear_bwd_k of resample.hip): low-resolution pixel i receives from high-resolution 2i-2 .. 2i+4
template <bool SC1>
__device__ __forceinline__ void bilinear_bwd_tile(const uz_chain_op& o, int tile) {
    const int C = o.i[0], CtotDy = o.i[1], CtotDx = o.i[2], N = o.i[3], H = o.i[4], W = o.i[5], ac = o.i[6], accumulate = o.i[7];
    const int Ho = 2 * H, Wo = 2 * W;
    float sh, sw;
    bil_scales(H, W, ac, sh, sw);
    const float* dy = static_cast<const float*>(o.p[0]);
    float* dx = static_cast<float*>(o.p[1]);
    const long long total = (long long)N * C * H * W;
#pragma unroll 1
    for (int k = 0; k < 4; ++k) {
        const long long e = (long long)tile * RS_PER_TILE + k * NT + threadIdx.x;
        if (e >= total) continue;
        const int ix = (int)(e % W); long long t = e / W;
        const int iy = (int)(t % H); t /= H;
        const int c = (int)(t % C), b = (int)(t / C);
        const float* s = dy + ((size_t)b * CtotDy + c) * Ho * Wo;
        const int oy0 = 2 * iy - 2, ox0 = 2 * ix - 2;
        float wx[7];
#pragma unroll
        for (int kx = 0; kx < 7; ++kx) wx[kx] = tap_weight(ox0 + kx, Wo, sw, ac, W, ix);
        float acc = 0.f;
#pragma unroll
        for (int ky = 0; ky < 7; ++ky) {
            const float wy = tap_weight(oy0 + ky, Ho, sh, ac, H, iy);
            if (wy != 0.f) {
                const float* row = s + (oy0 + ky) * Wo + ox0;
                float ra = 0.f;
#pragma unroll
                for (int kx = 0; kx < 7; ++kx)
                    if (wx[kx] != 0.f) ra += wx[kx] * ldh<SC1>(row + kx);
                acc += wy * ra;
            }
        }
        float* d = dx + ((size_t)b * CtotDx + c) * H * W + iy * W + ix;
        sth<SC1>(d, accumulate ? ldh<SC1>(d) + acc : acc);
    }
}

// ---------------------------------------------------------------------------------------------- latent heads (L == 2)
__device__ __forceinline__ float head_softplus(float x) { return x > 20.f ? x : log1pf(expf(x)); }
__device__ __forceinline__ float head_sigma(float pre, int act) { return act ? expf(pre) : head_softplus(pre); }
// tile = 64 pixels; wave g accumulates channels g, g + NW, ... ; the partial sums are added in wave order through LDS
template <bool SC1>
__device__ __forceinline__ void heads_fwd_tile(const uz_chain_op& o, int tile, float* sm /* >= NW * 4 * 64 */) {
    const int Cin = o.i[0], CinTot = o.i[1], N = o.i[2], HW = o.i[3], act = o.i[4];
    const int P = N * HW, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int p = tile * 64 + lane;
    const bool pvalid = p < P;
    const int pc = pvalid ? p : P - 1;
    const int n = pc / HW, q = pc - n * HW;
    const float* h = static_cast<const float*>(o.p[0]) + (size_t)n * CinTot * HW + q;
    const float* __restrict__ wm = static_cast<const float*>(o.p[1]);
    const float* __restrict__ wsg = static_cast<const float*>(o.p[3]);
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll 4
    for (int c = wave; c < Cin; c += NW) {
        const float v = ldh<SC1>(h + (size_t)c * HW);
        acc[0] = fmaf(wm[c], v, acc[0]); acc[1] = fmaf(wm[Cin + c], v, acc[1]);
        acc[2] = fmaf(wsg[c], v, acc[2]); acc[3] = fmaf(wsg[Cin + c], v, acc[3]);
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) sm[(wave * 4 + k) * 64 + lane] = acc[k];
    __syncthreads();
    if (wave == 0 && pvalid) {
        const float* bm = static_cast<const float*>(o.p[2]);
        const float* bs = static_cast<const float*>(o.p[4]);
        const float* eps = static_cast<const float*>(o.p[5]);
        float* mu = static_cast<float*>(o.p[6]);
        float* pre = static_cast<float*>(o.p[7]);
        float* sigma = static_cast<float*>(o.p[8]);
        float* z = static_cast<float*>(o.p[9]);
        float t[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            float a = 0.f;
            for (int g = 0; g < NW; ++g) a += sm[(g * 4 + k) * 64 + lane];
            t[k] = a;
        }
#pragma unroll
        for (int l = 0; l < 2; ++l) {
            const size_t e = ((size_t)n * 2 + l) * HW + q;
            const float m = t[l] + (bm ? bm[l] : 0.f), pr = t[2 + l] + (bs ? bs[l] : 0.f);
            const float sg = head_sigma(pr, act);
            sth<SC1>(mu + e, m); sth<SC1>(pre + e, pr); sth<SC1>(sigma + e, sg);
            if (z) sth<SC1>(z + e, m + sg * eps[e]);
        }
    }
    __syncthreads();
}
// Backward of a SampleZBlock's tail (phiseg.py:95-105): uz_latent_sample_bwd + uz_latent_heads_bwd_data in one sub-op.
//   g_mu = kl_dmu + dz, g_sigma = kl_dsigma + dz eps, g_pre = g_sigma * softplus'(pre)  [= 1 - exp(-sigma)]          (written for the heads' weight gradient)
//   dh[c] (+)= w_sigma[0][c] g_pre[0] + w_sigma[1][c] g_pre[1] + w_mu[0][c] g_mu[0] + w_mu[1][c] g_mu[1]              (the sigma head's rows first)
// tile = 64 pixels: wave g takes channels g, g + NW, ...; wave 0 stores g_mu / g_pre
template <bool SC1>
__device__ __forceinline__ void latent_heads_bwd_tile(const uz_chain_op& o, int tile) {
    const int Cin = o.i[0], CinTot = o.i[1], N = o.i[2], HW = o.i[3], act = o.i[4], accumulate = o.i[5];
    const int P = N * HW;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int p = tile * 64 + lane;
    if (p >= P) return;
    const int n = p / HW, q = p - n * HW;
    const float* kdm = static_cast<const float*>(o.p[0]);
    const float* kds = static_cast<const float*>(o.p[1]);
    const float* dz = static_cast<const float*>(o.p[2]);
    const float* eps = static_cast<const float*>(o.p[3]);
    const float* sigma = static_cast<const float*>(o.p[4]);
    float* gmu = static_cast<float*>(o.p[5]);
    float* gpre = static_cast<float*>(o.p[6]);
    const float* __restrict__ wa = static_cast<const float*>(o.p[7]);      // sigma head
    const float* __restrict__ wb = static_cast<const float*>(o.p[8]);      // mu head
    float gm[2], gp[2];
#pragma unroll
    for (int l = 0; l < 2; ++l) {
        const size_t e = ((size_t)n * 2 + l) * HW + q;
        const float gz = dz ? ldh<SC1>(dz + e) : 0.f;
        gm[l] = (kdm ? kdm[e] : 0.f) + gz;
        const float gs = (kds ? kds[e] : 0.f) + gz * eps[e];
        const float sg = sigma[e];
        gp[l] = gs * (act ? sg : (sg > 20.f ? 1.f : (1.f - expf(-sg))));
        if (wave == 0) { sth<SC1>(gmu + e, gm[l]); sth<SC1>(gpre + e, gp[l]); }
    }
    float* dh = static_cast<float*>(o.p[9]);
    if (!dh) return;
    dh += (size_t)n * CinTot * HW + q;
    for (int c = wave; c < Cin; c += NW) {
        float v = accumulate ? ldh<SC1>(dh + (size_t)c * HW) : 0.f;
        v = fmaf(wa[c], gp[0], v); v = fmaf(wa[Cin + c], gp[1], v);
        v = fmaf(wb[c], gm[0], v); v = fmaf(wb[Cin + c], gm[1], v);
        sth<SC1>(dh + (size_t)c * HW, v);
    }
}
// dst (+)= sum of S slabs [S][N][C][HW] in slab order (a split-K data gradient whose output has other writers)
template <bool SC1>
__device__ __forceinline__ void slab_sum_tile(const uz_chain_op& o, int tile) {
    const int S = o.i[0], N = o.i[1], C = o.i[2], Ctot = o.i[3], HW = o.i[4], accumulate = o.i[5];
    const float* slab = static_cast<const float*>(o.p[0]);
    float* dst = static_cast<float*>(o.p[1]);
    const size_t nsl = (size_t)N * C * HW;
    const long long e = (long long)tile * NT + threadIdx.x;
    if (e >= (long long)nsl) return;
    const int q = (int)(e % HW); long long t = e / HW;
    const int c = (int)(t % C), b = (int)(t / C);
    float* d = dst + ((size_t)b * Ctot + c) * HW + q;
    const float dl = ldh<SC1>(d);
    float v = accumulate ? dl : 0.f;
    for (int k0 = 0; k0 < S; k0 += 24) {
        float tt[24];
#pragma unroll
        for (int u = 0; u < 24; ++u) tt[u] = ldh<SC1>(slab + (size_t)min(k0 + u, S - 1) * nsl + e);
#pragma unroll
        for (int u = 0; u < 24; ++u) v += (k0 + u < S) ? tt[u] : 0.f;
    }
    sth<SC1>(d, v);
}

// ---------------------------------------------------------------------------------------------- tiles per sub-op (host and device agree)
// CONV3 counts WAVE tiles (dealt round-robin to the workgroups, then to their waves); every other sub-op counts workgroup tiles
__host__ __device__ inline int op_tiles(const uz_chain_op& o) {
    const int32_t* i = o.i;
    switch (o.code) {
        case UZ_CH_CONV3: return conv3_wave_tiles(o);
        case UZ_CH_CONV3_SMALL: return ((i[4] * i[5] * i[6] + NT - 1) / NT) * ((i[2] + SMALL_CO - 1) / SMALL_CO);
        case UZ_CH_CONV3_SMALL_BWD_DATA: return (i[4] * i[5] * i[6] + 63) / 64;
        case UZ_CH_BN_FWD: case UZ_CH_BN_BWD: return i[0];
        case UZ_CH_AVGPOOL_FWD: return (int)(((long long)i[3] * i[0] * ((i[4] + 1) / 2) * ((i[5] + 1) / 2) + RS_PER_TILE - 1) / RS_PER_TILE);
        case UZ_CH_AVGPOOL_BWD: return (int)(((long long)i[3] * i[0] * i[4] * i[5] + RS_PER_TILE - 1) / RS_PER_TILE);
        case UZ_CH_BILINEAR_FWD: return (int)(((long long)i[3] * i[0] * 4 * i[4] * i[5] + RS_PER_TILE - 1) / RS_PER_TILE);
        case UZ_CH_BILINEAR_BWD: return (int)(((long long)i[3] * i[0] * i[4] * i[5] + RS_PER_TILE - 1) / RS_PER_TILE);
        case UZ_CH_HEADS_FWD: case UZ_CH_LATENT_HEADS_BWD: return (i[2] * i[3] + 63) / 64;
        case UZ_CH_SLAB_SUM: return (int)(((long long)i[1] * i[2] * i[4] + NT - 1) / NT);
        default: return -1;
    }
}

template <bool SC1>
__global__ __launch_bounds__(NT) void chain_kernel(const uz_chain_op* __restrict__ ops, const int32_t* __restrict__ phases, int n_phases, int n_ops,
                                                   unsigned* state, int dbg_phase, int dbg_wg) {
    __shared__ double smd[2 * NW];
    __shared__ float smf[NW * 4 * 64 + 8];
    __shared__ int d_code[MAX_OPS], d_t0[MAX_OPS], d_nt[MAX_OPS], d_ph[2 * MAX_PHASES];
    __shared__ unsigned verdict;
    const int wg = blockIdx.x, G = gridDim.x;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int k = threadIdx.x; k < n_ops; k += NT) { d_code[k] = ops[k].code; d_t0[k] = ops[k].tile0; d_nt[k] = ops[k].ntiles; }
    for (int k = threadIdx.x; k < 2 * n_phases; k += NT) d_ph[k] = phases[k];
    __syncthreads();
    unsigned long long* stamps = reinterpret_cast<unsigned long long*>(state + ST_STAMPS);
    if (wg == 0 && threadIdx.x == 0) stamps[0] = wall_clock64();
    if (threadIdx.x == 0) g_dbg = nullptr;
    for (int ph = 0; ph < n_phases; ++ph) {
        const int op0 = d_ph[2 * ph], nops = d_ph[2 * ph + 1];
        if (dbg_phase >= 0) {
            __syncthreads();
            if (threadIdx.x == 0) g_dbg = (ph == dbg_phase && wg == dbg_wg) ? reinterpret_cast<unsigned long long*>(state + ST_DBG) : nullptr;
            __syncthreads();
            STAMP(0);
        }
        for (int k = op0; k < op0 + nops; ++k) {
            const int code = d_code[k], nt = d_nt[k];
            int first = wg - d_t0[k];                          // the op's tiles start at workgroup tile0 (mod G)
            if (first < 0) first += G;
            const uz_chain_op& o = ops[k];
            switch (code) {
                case UZ_CH_CONV3:
                    for (int wt = first + G * wave; wt < nt; wt += G * NW) conv3_wave_tile<SC1>(o, wt, lane);
                    break;
                case UZ_CH_CONV3_SMALL: for (int t = first; t < nt; t += G) conv3_small_tile<SC1>(o, t); break;
                case UZ_CH_CONV3_SMALL_BWD_DATA: for (int t = first; t < nt; t += G) conv3_small_bwd_tile<SC1>(o, t, smf); break;
                case UZ_CH_BN_FWD: for (int t = first; t < nt; t += G) bn_fwd_tile<SC1>(o, t, smd, smf); break;
                case UZ_CH_BN_BWD: for (int t = first; t < nt; t += G) bn_bwd_tile<SC1>(o, t, smd, smf); break;
                case UZ_CH_AVGPOOL_FWD: for (int t = first; t < nt; t += G) avgpool_fwd_tile<SC1>(o, t); break;
                case UZ_CH_AVGPOOL_BWD: for (int t = first; t < nt; t += G) avgpool_bwd_tile<SC1>(o, t); break;
                case UZ_CH_BILINEAR_FWD: for (int t = first; t < nt; t += G) bilinear_fwd_tile<SC1>(o, t); break;
                case UZ_CH_BILINEAR_BWD: for (int t = first; t < nt; t += G) bilinear_bwd_tile<SC1>(o, t); break;
                case UZ_CH_HEADS_FWD: for (int t = first; t < nt; t += G) heads_fwd_tile<SC1>(o, t, smf); break;
                case UZ_CH_LATENT_HEADS_BWD: for (int t = first; t < nt; t += G) latent_heads_bwd_tile<SC1>(o, t); break;
                case UZ_CH_SLAB_SUM: for (int t = first; t < nt; t += G) slab_sum_tile<SC1>(o, t); break;
                default: break;
            }
        }
        STAMP(30);
        if (ph + 1 < n_phases && !grid_barrier<SC1>(state, (unsigned)G, (unsigned)(ph + 1), &verdict)) return;
        STAMP(31);
        if (wg == 0 && threadIdx.x == 0 && ph + 1 < MAX_STAMPS) stamps[ph + 1] = wall_clock64();
    }
}

// ---------------------------------------------------------------------------------------------- weight images
// one thread per (tap * KB + kb, m32, lane): eight weights -> the lane's 16 bytes of either piece
struct PackRow { const float* w; char* img; int Mc, Kc, Cin, dgrad, blk0; };
__global__ __launch_bounds__(256) void chain_pack_kernel(const int64_t* __restrict__ table, int n_layers, const float* __restrict__ w_amax, int* flags) {
    // find the layer of this block (table rows are 7 int64; blk0 ascending)
    int lo = 0, hi = n_layers - 1;
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if ((int)table[(size_t)mid * 7 + 6] <= (int)blockIdx.x) lo = mid; else hi = mid - 1;
    }
    const int64_t* row = table + (size_t)lo * 7;
    const float* __restrict__ w = reinterpret_cast<const float*>(row[0]);
    char* img = reinterpret_cast<char*>(row[1]);
    const int Mc = (int)row[2], Kc = (int)row[3], Cin = (int)row[4], dgrad = (int)row[5], blk0 = (int)row[6];
    const int KB = Kc / 16, M32 = Mc / 32;
    const long long units = (long long)9 * KB * M32 * 64;
    const long long u = (long long)(blockIdx.x - blk0) * 256 + threadIdx.x;
    if (u >= units) return;
    const int lane = (int)(u & 63);
    long long t = u >> 6;
    const int m32 = (int)(t % M32); t /= M32;
    const int kb = (int)(t % KB), tap = (int)(t / KB);
    const int m = m32 * 32 + (lane & 31), k0 = kb * 16 + 8 * (lane >> 5);
    const float s = uz::split_scale(uz::amax_read(w_amax));
    float v[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        // forward: w[co = m][ci = k][tap]; data gradient: w[co = k][ci = m][8 - tap]   (parameter layout [Cout][Cin][3][3])
        const size_t e = dgrad ? ((size_t)(k0 + j) * Cin + m) * 9 + (8 - tap) : ((size_t)m * Cin + (k0 + j)) * 9 + tap;
        v[j] = w[e] * s;
    }
    unsigned p1[4], p2[4];
    bool bad = false;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        bad |= uz::bound_violated(v[2 * j], v[2 * j + 1]);
        uz::split2(v[2 * j], v[2 * j + 1], p1[j], p2[j]);
    }
    if (bad && flags) atomicOr(flags, uz::FLAG_W_BOUND);
    char* dst = img + ((size_t)(tap * KB + kb) * M32 + m32) * 2048 + lane * 16;
    *reinterpret_cast<u32x4*>(dst) = u32x4{p1[0], p1[1], p1[2], p1[3]};
    *reinterpret_cast<u32x4*>(dst + 1024) = u32x4{p2[0], p2[1], p2[2], p2[3]};
}

}  // namespace

extern "C" int uz_chain_op_tiles(const uz_chain_op* op) {
    if (!op) return -1;
    const int32_t* i = op->i;
    switch (op->code) {
        case UZ_CH_CONV3:
            if (i[0] % 16 || i[2] % 32 || i[0] < 16 || i[7] < 1 || i[7] > 9 * (i[0] / 16)) return -1;
            break;
        case UZ_CH_CONV3_SMALL: case UZ_CH_CONV3_SMALL_BWD_DATA:
            if (i[0] < 1 || i[0] > 4) return -1;
            break;
        case UZ_CH_BN_FWD: case UZ_CH_BN_BWD:
            if ((long long)i[3] * i[4] > (long long)BN_MAX_EPT * NT) return -1;
            break;
        default: break;
    }
    return op_tiles(*op);
}
extern "C" int uz_chain_conv_ksplit(int Kc, int Mc, int N, int H, int W, int n_workgroups) {
    if (Kc % 16 || Kc < 16) return 1;
    const int P = N * H * W, base = ((P + PXB - 1) / PXB) * ((Mc + COB - 1) / COB);        // wave tiles without split
    const int T = 9 * (Kc / 16);
    const int want = 2 * NW * n_workgroups / 4;                                            // wave tiles wanted: half the launch's waves busy at least
    int S = (want + base - 1) / base;
    const int smax = T / MIN_KSTEPS > 0 ? T / MIN_KSTEPS : 1;
    if (S > smax) S = smax;
    if (S < 1) S = 1;
    return S;
}
extern "C" size_t uz_chain_packed_bytes(int Kc, int Mc) { return (size_t)9 * (Kc / 16) * (Mc / 32) * 2048; }
extern "C" int uz_chain_pack_blocks(int Kc, int Mc) { return (int)(((long long)9 * (Kc / 16) * (Mc / 32) * 64 + 255) / 256); }
extern "C" int uz_chain_pack_weights(const int64_t* table, int n_layers, int total_blocks, const float* w_amax, void* stream) {
    UZ_REQUIRE(table && w_amax && n_layers > 0 && total_blocks > 0, "chain_pack_weights: empty table");
    hipLaunchKernelGGL(chain_pack_kernel, dim3(total_blocks), dim3(256), 0, uz::S(stream), table, n_layers, w_amax, uz::dev_flags_ptr());
    return uz::check_launch("chain_pack_kernel");
}
extern "C" size_t uz_chain_state_bytes(void) { return ST_WORDS * sizeof(unsigned); }
extern "C" int uz_chain_run(const uz_chain_op* ops, const int32_t* phases, int n_phases, int n_ops, int n_workgroups, void* state, void* stream) {
    UZ_REQUIRE(ops && phases && state && n_phases > 0 && n_ops > 0, "chain_run: null table");
    UZ_REQUIRE(n_phases <= MAX_PHASES && n_ops <= MAX_OPS, "chain_run: %d phases / %d sub-ops beyond the directory (%d / %d)", n_phases, n_ops, MAX_PHASES, MAX_OPS);
    UZ_REQUIRE(n_workgroups >= 1 && n_workgroups <= 1024, "chain_run: %d workgroups outside [1, 1024]", n_workgroups);
    // a grid barrier needs every workgroup resident: one 1024-thread workgroup per CU at most
    static const int n_cu = [] { int n = 0; if (uz_device_info(&n, nullptr, 0) != 0) n = 0; return n; }();
    UZ_REQUIRE(n_cu <= 0 || n_workgroups <= n_cu, "chain_run: %d workgroups on %d CUs (a workgroup holds a whole CU's wave slots)", n_workgroups, n_cu);
    static const int sc1 = [] { const char* e = getenv("UZ_CHAIN_SC1"); return e ? atoi(e) : 1; }();
    static const int dbg_phase = [] { const char* e = getenv("UZ_CHAIN_DEBUG_PHASE"); return e ? atoi(e) : -1; }();
    static const int dbg_wg = [] { const char* e = getenv("UZ_CHAIN_DEBUG_WG"); return e ? atoi(e) : 0; }();
    if (hipMemsetAsync(state, 0, ST_STAMPS * sizeof(unsigned), uz::S(stream)) != hipSuccess) return uz::fail("chain_run: hipMemsetAsync failed");
    if (sc1) hipLaunchKernelGGL(chain_kernel<true>, dim3(n_workgroups), dim3(NT), 0, uz::S(stream), ops, phases, n_phases, n_ops, static_cast<unsigned*>(state), dbg_phase, dbg_wg);
    else hipLaunchKernelGGL(chain_kernel<false>, dim3(n_workgroups), dim3(NT), 0, uz::S(stream), ops, phases, n_phases, n_ops, static_cast<unsigned*>(state), dbg_phase, dbg_wg);
    return uz::check_launch("chain_kernel");
}
extern "C" int uz_chain_status(const void* state, int* out, void* stream) {
    UZ_REQUIRE(state && out, "chain_status: null argument");
    unsigned w = 0;
    if (hipStreamSynchronize(uz::S(stream)) != hipSuccess) return uz::fail("chain_status: synchronize failed");
    if (hipMemcpy(&w, static_cast<const unsigned*>(state) + ST_STATUS, sizeof(w), hipMemcpyDeviceToHost) != hipSuccess) return uz::fail("chain_status: copy failed");
    *out = (int)w;
    return 0;
}
