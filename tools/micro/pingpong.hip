// Micro-benchmark for DESIGN.md section 9 item 0: does overlapping a staging phase with a matrix phase pay on MI355X, or does the clock
// give it back?  One 512-thread workgroup per CU (8 waves, 2 per SIMD), per iteration and wave:
//   M = 108 x v_mfma_f32_32x32x16_f16 on four accumulator tiles, fragments re-read from LDS (4 ds_read_b128 per 3 MFMAs), like one
//       16-channel chunk of conv_split's 64-channel tile;
//   S = 6 x 16-byte global loads (a 64 MB buffer, different lines every iteration), wait, 12 byte permutes, 6 x ds_write_b128.
// Kernels: M only, S only, "serial" (all waves S, barrier, all waves M, barrier - the structure of the product kernels) and "ping-pong"
// (the two waves of a SIMD in opposite phases, a barrier at every hand-over).  Prints the time per iteration and the shader clock
// (s_memtime cycles per s_memrealtime tick at 100 MHz).  Build + run: hipcc --offload-arch=gfx950 -O3 pingpong.hip -o /tmp/pingpong && /tmp/pingpong
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <math.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

constexpr int NT = 512, LDSB = 128 * 1024;     // 128 KB of LDS per workgroup: one workgroup per CU
constexpr int IMG = 73728;                     // bytes of fragments the matrix phase reads (a: 36 x 1 KB, b: 36 x 1 KB)

struct St { f32x16 acc[4]; u32x4 hold[6]; unsigned mix; };

#ifndef INTERLEAVE
#define INTERLEAVE 0
#endif
#ifndef LDSBAR
#define LDSBAR 1           // 1: the ping-pong hand-over waits for LDS traffic only (0: __syncthreads(), which drains the global loads too)
#endif
#ifndef PREFM
#define PREFM 0            // 1: the matrix phase reads the fragments of tap r + 1 before it issues the MFMAs of tap r (software pipelined)
#endif
struct Frag { u32x4 a[2][2], b[2][2]; };
__device__ __forceinline__ void frag_load(Frag& f, const char* lds, int lane, int r) {
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            f.a[i][q] = *reinterpret_cast<const u32x4*>(lds + ((r * 4 + i * 2 + q) * 64 + lane) * 16);
            f.b[i][q] = *reinterpret_cast<const u32x4*>(lds + 36864 + ((r * 4 + i * 2 + q) * 64 + lane) * 16);
        }
}
__device__ __forceinline__ void frag_mma(St& s, const Frag& f) {
#pragma unroll
    for (int pr = 0; pr < 3; ++pr)
#pragma unroll
        for (int m = 0; m < 2; ++m)
#pragma unroll
            for (int n = 0; n < 2; ++n)
                s.acc[m * 2 + n] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, f.a[m][pr == 0]), __builtin_bit_cast(f16x8, f.b[n][pr == 1]),
                                                                          s.acc[m * 2 + n], 0, 0, 0);
}
__device__ __forceinline__ void phase_m_pref(St& s, const char* lds, int lane) {
    asm volatile("" ::: "memory");
    Frag f[2];
    frag_load(f[0], lds, lane, 0);
#pragma unroll
    for (int r = 0; r < 9; ++r) {
        if (r + 1 < 9) frag_load(f[(r + 1) & 1], lds, lane, r + 1);
        __builtin_amdgcn_sched_barrier(0);                          // keep the next tap's reads ahead of this tap's MFMAs
        frag_mma(s, f[r & 1]);
    }
}
__device__ __forceinline__ void phase_m(St& s, const char* lds, int lane) {
    if (PREFM) { phase_m_pref(s, lds, lane); return; }
    asm volatile("" ::: "memory");                                // the fragments are re-read every iteration (no hoisting in the M-only kernel)
#pragma unroll
    for (int r = 0; r < 9; ++r) {                                  // 9 taps x 4 tiles x 3 products = 108 MFMAs
        u32x4 a[2][2], b[2][2];                                    // [tile half][piece]
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                a[i][q] = *reinterpret_cast<const u32x4*>(lds + ((r * 4 + i * 2 + q) * 64 + lane) * 16);
                b[i][q] = *reinterpret_cast<const u32x4*>(lds + 36864 + ((r * 4 + i * 2 + q) * 64 + lane) * 16);
            }
        if (INTERLEAVE) {                                          // consecutive MFMAs on DIFFERENT accumulator tiles (a single wave keeps the pipe full)
#pragma unroll
            for (int pr = 0; pr < 3; ++pr)
#pragma unroll
                for (int m = 0; m < 2; ++m)
#pragma unroll
                    for (int n = 0; n < 2; ++n)
                        s.acc[m * 2 + n] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a[m][pr == 0]), __builtin_bit_cast(f16x8, b[n][pr == 1]),
                                                                                  s.acc[m * 2 + n], 0, 0, 0);
        } else {
#pragma unroll
            for (int m = 0; m < 2; ++m)
#pragma unroll
                for (int n = 0; n < 2; ++n) {
                    f32x16 t = s.acc[m * 2 + n];
                    t = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a[m][1]), __builtin_bit_cast(f16x8, b[n][0]), t, 0, 0, 0);
                    t = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a[m][0]), __builtin_bit_cast(f16x8, b[n][1]), t, 0, 0, 0);
                    t = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a[m][0]), __builtin_bit_cast(f16x8, b[n][0]), t, 0, 0, 0);
                    s.acc[m * 2 + n] = t;
                }
        }
    }
}

__device__ __forceinline__ void phase_s(St& s, const u32x4* g, size_t nvec, int it, char* lds, int tid, int slot) {
    const size_t base = ((size_t)blockIdx.x * 9973 + (size_t)it * 1237) * NT * 6;
#pragma unroll
    for (int j = 0; j < 6; ++j) s.hold[j] = g[(base + (size_t)j * NT + tid) % nvec];
#pragma unroll
    for (int j = 0; j < 6; ++j) {
        u32x4 v = s.hold[j];
        v.x = __builtin_amdgcn_perm(v.x, v.y, 0x05040100u); v.z = __builtin_amdgcn_perm(v.z, v.w, 0x07060302u);
        s.mix ^= v.x + v.z;
        v.x = (v.x & 0x03ff03ffu) | 0x3c003c00u; v.y = (v.y & 0x03ff03ffu) | 0x3c003c00u;       // keep the fp16 pieces finite (1.0 .. 2.0)
        v.z = (v.z & 0x03ff03ffu) | 0x3c003c00u; v.w = (v.w & 0x03ff03ffu) | 0x3c003c00u;
        *reinterpret_cast<u32x4*>(lds + ((size_t)j * NT + slot) * 16) = v;
    }
}

// consume the loads issued half a period ago (permutes + LDS writes), then issue the next ones: their latency hides behind this group's own matrix phase
__device__ __forceinline__ void stage_prefetched(St& s, const u32x4* g, size_t nvec, int it, char* lds, int tid, int slot) {
#pragma unroll
    for (int j = 0; j < 6; ++j) {
        u32x4 v = s.hold[j];
        v.x = __builtin_amdgcn_perm(v.x, v.y, 0x05040100u); v.z = __builtin_amdgcn_perm(v.z, v.w, 0x07060302u);
        s.mix ^= v.x + v.z;
        *reinterpret_cast<u32x4*>(lds + ((size_t)j * NT + slot) * 16) = v;
    }
    const size_t base = ((size_t)blockIdx.x * 9973 + (size_t)(it + 1) * 1237) * NT * 6;
#pragma unroll
    for (int j = 0; j < 6; ++j) s.hold[j] = g[(base + (size_t)j * NT + tid) % nvec];
}

template <int MODE>      // 0 = M only (constant fragments), 1 = S only, 2 = serial, 3 = ping-pong, 4 = M only on random fragments, 5 = serial with the loads issued one iteration ahead,
                         // 6 = M only on REALISTIC fragments (split pieces of post-ReLU activations x small weights), 7 = serial on them (S writes elsewhere),
                         // 8 = ping-pong on them with the loads issued half a period ahead (the proposal of DESIGN section 9 item 0)
__global__ __launch_bounds__(NT, 1) void kern(const u32x4* g, size_t nvec, float* out, long long* clk, int iters, const u32x4* img) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, group = wave >> 2;      // waves 0-3 / 4-7: one of each per SIMD
    St s;
    for (int i = 0; i < 4; ++i) s.acc[i] = f32x16{};
    s.mix = 0;
    for (int i = tid; i < LDSB / 16; i += NT) *reinterpret_cast<u32x4*>(lds + (size_t)i * 16) = u32x4{0x3c003c00u, 0x3c003c00u, 0x3c003c00u, 0x3c003c00u};
    __syncthreads();
    if (MODE == 4 || MODE == 5) {                       // random mantissas in every fragment the matrix phase reads
        for (int k = 0; k < 2; ++k) { phase_s(s, g, nvec, 7 + k, lds + k * 49152 * 0 + k * 24576, tid, tid); }
        __syncthreads();
    }
    if (MODE == 6 || MODE == 7 || MODE == 8) {
        for (int i = tid; i < IMG / 16; i += NT) *reinterpret_cast<u32x4*>(lds + (size_t)i * 16) = img[i];
        __syncthreads();
    }
    long long t0 = 0, r0 = 0;
    if (blockIdx.x == 0 && tid == 0) { t0 = __builtin_amdgcn_s_memtime(); r0 = __builtin_amdgcn_s_memrealtime(); }
    for (int it = 0; it < iters; ++it) {
        if (MODE == 0 || MODE == 4 || MODE == 6) { phase_m(s, lds, lane); }
        else if (MODE == 8) {
            if (it == 0) for (int j = 0; j < 6; ++j) s.hold[j] = g[((size_t)blockIdx.x * 6 * NT + (size_t)j * NT + tid) % nvec];
            // LDS-only barrier (s_waitcnt lgkmcnt(0) + s_barrier): __syncthreads() also waits for vmcnt(0), i.e. for the loads that were
            // issued to stay in flight across the hand-over - with it the half period equals the load round trip (LDSBAR=0 shows that)
            if (group == 0) phase_m(s, lds, lane); else stage_prefetched(s, g, nvec, it, lds + IMG, tid, tid & 255);
            if (LDSBAR) { __builtin_amdgcn_s_waitcnt(0xC07F); __builtin_amdgcn_s_barrier(); } else __syncthreads();
            if (group == 0) stage_prefetched(s, g, nvec, it, lds + IMG, tid, tid & 255); else phase_m(s, lds, lane);
            if (LDSBAR) { __builtin_amdgcn_s_waitcnt(0xC07F); __builtin_amdgcn_s_barrier(); } else __syncthreads();
        }
        else if (MODE == 7) { phase_s(s, g, nvec, it, lds + IMG, tid, tid); __syncthreads(); phase_m(s, lds, lane); __syncthreads(); }
        else if (MODE == 5) {
            // loads of iteration it + 1 in flight during the matrix phase of iteration it; only the permutes and LDS writes stay exposed
            u32x4 nxt[6];
            const size_t base = ((size_t)blockIdx.x * 9973 + (size_t)(it + 1) * 1237) * NT * 6;
#pragma unroll
            for (int j = 0; j < 6; ++j) nxt[j] = g[(base + (size_t)j * NT + tid) % nvec];
            phase_m(s, lds, lane);
            __syncthreads();
#pragma unroll
            for (int j = 0; j < 6; ++j) {
                u32x4 v = nxt[j];
                v.x = __builtin_amdgcn_perm(v.x, v.y, 0x05040100u); v.z = __builtin_amdgcn_perm(v.z, v.w, 0x07060302u);
                s.mix ^= v.x + v.z;
                v.x = (v.x & 0x03ff03ffu) | 0x3c003c00u; v.y = (v.y & 0x03ff03ffu) | 0x3c003c00u;
                v.z = (v.z & 0x03ff03ffu) | 0x3c003c00u; v.w = (v.w & 0x03ff03ffu) | 0x3c003c00u;
                *reinterpret_cast<u32x4*>(lds + ((size_t)j * NT + tid) * 16) = v;
            }
            __syncthreads();
        }
        else if (MODE == 1) { phase_s(s, g, nvec, it, lds, tid, tid); __syncthreads(); }
        else if (MODE == 2) { phase_s(s, g, nvec, it, lds, tid, tid); __syncthreads(); phase_m(s, lds, lane); __syncthreads(); }
        else {
            if (group == 0) phase_m(s, lds, lane); else phase_s(s, g, nvec, it, lds + 49152, tid, tid & 255);
            __syncthreads();
            if (group == 0) phase_s(s, g, nvec, it, lds + 49152, tid, tid & 255); else phase_m(s, lds, lane);
            __syncthreads();
        }
    }
    if (blockIdx.x == 0 && tid == 0) { clk[0] = __builtin_amdgcn_s_memtime() - t0; clk[1] = __builtin_amdgcn_s_memrealtime() - r0; }
    float v = (float)s.mix;
    for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) v += s.acc[i][r];
    out[(size_t)blockIdx.x * NT + tid] = v;
}

template <int MODE>
void run(const char* name, const u32x4* g, size_t nvec, float* out, long long* clk, int iters, int grid, const u32x4* img = nullptr) {
    hipFuncSetAttribute(reinterpret_cast<const void*>(kern<MODE>), hipFuncAttributeMaxDynamicSharedMemorySize, LDSB);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    float best = 1e9f;
    long long h[2] = {0, 0};
    for (int rep = 0; rep < 4; ++rep) {
        hipEventRecord(e0, 0);
        hipLaunchKernelGGL(kern<MODE>, dim3(grid), dim3(NT), LDSB, 0, g, nvec, out, clk, iters, img);
        hipEventRecord(e1, 0);
        hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        if (rep && ms < best) { best = ms; hipMemcpy(h, clk, sizeof(h), hipMemcpyDeviceToHost); }
    }
    const double rounds = (double)grid / 256.0;
    const double us_it = best * 1e3 / iters / rounds;
    const double mfma = MODE == 1 ? 0.0 : 108.0 * 8 * 256 * 32768.0 / (us_it * 1e-6) / 1e15;        // PFLOP/s of fp16 MFMAs over the chip
    printf("%-14s %8.3f ms  %7.3f us per iteration and workgroup  clock %.2f GHz  fp16 MFMA %.2f PF/s\n", name, best, us_it,
           h[1] ? (double)h[0] / (double)h[1] * 0.1 : 0.0, mfma);
}

int main() {
    const size_t bytes = 64ull << 20, nvec = bytes / 16;
    u32x4* g; float* out; long long* clk;
    hipMalloc(&g, bytes); hipMalloc(&out, 4096 * NT * sizeof(float)); hipMalloc(&clk, 16);
    unsigned* hbuf = (unsigned*)malloc(bytes);
    srand(1);
    for (size_t i = 0; i < bytes / 4; ++i) hbuf[i] = (unsigned)rand() * 2654435761u;
    hipMemcpy(g, hbuf, bytes, hipMemcpyHostToDevice);
    // realistic fragments: a = the two fp16 pieces (hi | lo, scaled by 2^11) of post-ReLU activations (|N(0,1)|, half of them zero),
    // b = the pieces of weights ~ N(0, 0.05) scaled by 2^14; fragment f = tap * 4 + half * 2 + piece
    _Float16* himg = (_Float16*)malloc(IMG);
    for (int side = 0; side < 2; ++side)
        for (int f = 0; f < 36; ++f)
            for (int e = 0; e < 512; ++e) {
                float u1 = (rand() + 1.0f) / (RAND_MAX + 2.0f), u2 = (rand() + 1.0f) / (RAND_MAX + 2.0f);
                float n = sqrtf(-2.f * logf(u1)) * cosf(6.2831853f * u2);
                float v = side == 0 ? ((rand() & 1) ? fabsf(n) : 0.f) * 2048.f : n * 0.05f * 16384.f;
                _Float16 hi = (_Float16)v, lo = (_Float16)(v - (float)hi);
                himg[(side * 36 + f) * 512 + e] = (f & 1) ? lo : hi;
            }
    u32x4* img; hipMalloc(&img, IMG); hipMemcpy(img, himg, IMG, hipMemcpyHostToDevice);
    const int iters = 400, grid = 1024;            // 4 rounds of 256 workgroups
    run<0>("M only", g, nvec, out, clk, iters, grid);
    run<1>("S only", g, nvec, out, clk, iters, grid);
    run<2>("serial", g, nvec, out, clk, iters, grid);
    run<3>("ping-pong", g, nvec, out, clk, iters, grid);
    run<0>("M only", g, nvec, out, clk, iters, grid);
    run<2>("serial", g, nvec, out, clk, iters, grid);
    run<3>("ping-pong", g, nvec, out, clk, iters, grid);
    run<4>("M random", g, nvec, out, clk, iters, grid);
    run<5>("serial+pf", g, nvec, out, clk, iters, grid);
    run<4>("M random", g, nvec, out, clk, iters, grid);
    run<5>("serial+pf", g, nvec, out, clk, iters, grid);
    for (int k = 0; k < 2; ++k) {
        run<6>("M real", g, nvec, out, clk, iters, grid, img);
        run<7>("serial real", g, nvec, out, clk, iters, grid, img);
        run<8>("pingpong real", g, nvec, out, clk, iters, grid, img);
    }
    return 0;
}
