// Shared helpers for the libuz_hip.so kernels (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stddef.h>
#include "uz_api.h"

namespace uz {

void set_error(const char* fmt, ...);
int  fail(const char* fmt, ...);          // set_error + return -1
int  check_launch(const char* what);      // hipGetLastError -> status
extern long long* debug_stamps;           // diagnostics: per-workgroup cycle stamps of the split kernels (uz_debug_stamps), normally null
int  conv_math_mode();                    // 0 fp32 MFMA only | 1 split-fp16 where it pays | 2 split-fp16 wherever eligible | 3 bf16 (one piece, one product) where 1 would split
int  conv_np();                           // operand planes of the split kernels under the current mode: 2 (fp16 split) or 1 (bf16)

static inline hipStream_t S(void* s) { return reinterpret_cast<hipStream_t>(s); }
static inline int ceil_div(int a, int b) { return (a + b - 1) / b; }
static inline int pow2_ceil(int v) { int p = 1; while (p < v) p <<= 1; return p; }
static inline int ilog2(int v) { int l = 0; while ((1 << l) < v) ++l; return l; }

// The dispatcher deals consecutive workgroup ids round-robin over the 8 XCDs (each with a
// private L2).  Remap so that every XCD walks a contiguous chunk of the work list: blocks that
// share operand panels then hit the same L2.  Bijective for any nwg (cdna guide 5.5 T1).
__device__ __forceinline__ int xcd_remap(int bid, int nwg) {
    const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7, slot = bid >> 3;
    return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + slot;
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ double wave_sum_d(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// Block-wide sum of NV doubles (blockDim.x == 256).  Result valid in thread 0.
template <int NV>
__device__ __forceinline__ void block_sum_d(double (&v)[NV], double* smem /* >= 4*NV */) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int i = 0; i < NV; ++i) v[i] = wave_sum_d(v[i]);
    __syncthreads();
    if (lane == 0) {
#pragma unroll
        for (int i = 0; i < NV; ++i) smem[wave * NV + i] = v[i];
    }
    __syncthreads();
    if (threadIdx.x == 0) {
#pragma unroll
        for (int i = 0; i < NV; ++i) v[i] = smem[i] + smem[NV + i] + smem[2 * NV + i] + smem[3 * NV + i];
    }
}

// conv_split.hip: 3x3 forward / data gradient on the fp16 matrix pipe with two-piece split operands (fp32-accurate)
bool conv_split_ok(int Kc, int Mc, int N, int H, int W, int ks, int dgrad);
size_t conv_split_workspace(int Kc, int Mc, int N, int H, int W);
int splitk_reduce(const float* slab, int ksplit, const float* bias, float* y, int Mc, int McTot, int N, int HW, int relu, int accumulate,
                  float* y_amax, hipStream_t st);      // conv_mfma.hip
int conv_split(const float* x, int Kc, int KcTot, const float* w, int wCi, const float* bias,
               float* y, int Mc, int McTot, int N, int H, int W, int dgrad, int relu, int accumulate,
               const float* x_amax, const float* w_amax, float* y_amax, void* workspace, const void* packed_w, float* bn_partials, hipStream_t st);
// round-4 options of the split path
struct SplitOpts {
    int x_packed = 0;                       // the input operand (x, or dy for a data gradient) is split storage (split_f16.h)
    const float* x_amax2 = nullptr;         // ... whose channels from seg_channels on were scaled from this bound instead of x_amax
    int seg_channels = 0;
    int mk = 0;                             // epilogue: 0 plain, 1 ReLU mask of the producing unit (mask = its activation), 2 BatchNorm-backward reduction
    const float* mask = nullptr; int maskCtot = 0;
    const float* mk_save = nullptr;         // mk == 2: [4][C] {mean, rstd, alpha, beta'} of the producing unit (uz_bn_relu_fwd_ex)
    int mk_relu = 1;
    // bf16 STORAGE (single-piece bf16 mode only, unsplit chunk loops): the input operand / the output tensor hold 2-byte bf16 elements
    int x_b16 = 0, y_b16 = 0;
    // round 6: x is the pre-normalisation output of the Conv -> BatchNorm -> ReLU unit in front (fp32) and aff its [4][Kc] {mean, rstd, alpha, beta'} table:
    // the forward convolution's staging applies the unit's BatchNorm + ReLU itself (x_amax = the bound of the APPLIED activation)
    const float* aff = nullptr; int aff_relu = 0;
};
int conv_split_ex(const float* x, int Kc, int KcTot, const float* w, int wCi, const float* bias,
                  float* y, int Mc, int McTot, int N, int H, int W, int dgrad, int relu, int accumulate,
                  const float* x_amax, const float* w_amax, float* y_amax, void* workspace, const void* packed_w, float* bn_partials, hipStream_t st,
                  const SplitOpts& o);
int conv_split_dgrad_relu(const float* dy, int Kc, int KcTot, const float* w, int wCi, float* dx, int Mc, int McTot, int N, int H, int W, int accumulate,
                          const float* dy_amax, const float* w_amax, float* dx_amax, void* workspace, const void* packed_w,
                          const float* a, int aCtot, float* partials, hipStream_t st);   // conv_split.hip
int conv_split_bn_partials(int Kc, int Mc, int N, int H, int W);

// conv_wgrad_split.hip: 3x3 weight gradient on the fp16 matrix pipe with two-piece split operands
bool wgrad_split_ok(int Cin, int Cout, int N, int H, int W, int ks);
int wgrad_split_splits(int Cin, int Cout, int N, int H, int W);
int wgrad_split(const float* x, int Cin, int CinTot, const float* dy, int Cout, int CoutTot, float* slab,
                int N, int H, int W, int S, const float* x_amax, const float* dy_amax, hipStream_t st,
                int x_packed = 0, const float* x_amax2 = nullptr, int seg_channels = 0, int dy_packed = 0,       // *_packed: operand in split storage
                int x_b16 = 0, int dy_b16 = 0);                                                                    // *_b16: operand stored as bf16 (single-piece mode)

// conv1x1_small.hip: streaming VALU kernels for 1x1 convolutions with <= 8 outputs (-2 = shape not covered)
bool conv1x1_small_ok(int Cin, int Cout);
int conv1x1_small_fwd(const float* x, int Cin, int CinTot, const float* w, const float* bias, float* y, int Cout, int CoutTot,
                      int N, int H, int W, hipStream_t st, int x_b16 = 0);
int conv1x1_small_bwd_data(const float* dy, int Cout, int CoutTot, const float* w, float* dx, int Cin, int CinTot,
                           int N, int H, int W, int accumulate, hipStream_t st, int dx_b16 = 0);
size_t conv1x1_small_bwd_weight_ws(int Cin, int Cout, int N, int H, int W);
int conv1x1_small_bwd_weight(const float* x, int Cin, int CinTot, const float* dy, int Cout, int CoutTot, float* dw, float* db,
                             int N, int H, int W, void* ws, hipStream_t st, int x_b16 = 0);

}  // namespace uz

#define UZ_REQUIRE(cond, ...) do { if (!(cond)) return uz::fail(__VA_ARGS__); } while (0)
