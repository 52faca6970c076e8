// Two-piece fp16 split of fp32 values, shared by conv_split.hip and conv_wgrad_split.hip.
//
//   v * s = h1 + h2 + r,   h1 = fp16(v * s),  h2 = fp16(v * s - h1),   |r| <= 2^-22 |v * s|
// with s a power of two chosen from an upper bound `amax` of the tensor's magnitudes so that |v * s| < 2^14 (no fp16
// overflow; every element above 2^-17 * amax keeps both pieces in the normal fp16 range, smaller ones are off by at
// most 2^-39 * amax in absolute terms).  A product a*b is accumulated in fp32 from THREE piece products
//   a2*b1, a1*b2, a1*b1          (11-bit x 11-bit significands: exact in fp32)
// the dropped a2*b2 is <= 2^-22 |ab|.  Fewer, larger partial sums than a k-ordered fp32 fmaf chain: measured against fp64
// the result is as accurate as the fp32-MFMA kernels (tests/test_full_configs_gpu.py, tools/sim_split_accuracy.py).
// Three fp16 MFMAs per 16-deep k-step cost 96 cycles against 512 for the eight fp32 MFMAs they replace.
#pragma once
#include <hip/hip_runtime.h>

namespace uz {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

// power-of-two scale s with amax * s < 2^14 (amax >= 0 finite), and its inverse
__device__ __forceinline__ float split_scale(float amax) {
    int e = (int)((__builtin_bit_cast(unsigned, amax) >> 23) & 0xFFu);      // amax < 2^(e - 126)
    int se = 267 - e;                                                       // biased exponent of 2^(140 - e)
    se = se < 1 ? 1 : (se > 253 ? 253 : se);
    return __builtin_bit_cast(float, (unsigned)se << 23);
}
__device__ __forceinline__ float split_inv_scale(float amax) {
    int e = (int)((__builtin_bit_cast(unsigned, amax) >> 23) & 0xFFu);
    int se = 267 - e;
    se = se < 1 ? 1 : (se > 253 ? 253 : se);
    return __builtin_bit_cast(float, (unsigned)(254 - se) << 23);           // 2^-(se - 127)
}

// splits two (already scaled) values at once; p1 / p2 hold the pieces of (v0, v1) as packed fp16 pairs (v0 in the low half)
__device__ __forceinline__ void split2(float v0, float v1, unsigned& p1, unsigned& p2) {
    const f32x2 a = {v0, v1};
    const f16x2 h1 = __builtin_convertvector(a, f16x2);                     // round to nearest even
    const f32x2 b = __builtin_convertvector(h1, f32x2);
    const f32x2 r = {v0 - b.x, v1 - b.y};                                   // exact
    const f16x2 h2 = __builtin_convertvector(r, f16x2);
    p1 = __builtin_bit_cast(unsigned, h1);
    p2 = __builtin_bit_cast(unsigned, h2);
}

// running |v| maximum -> one atomic per wave; slot holds the float bits of a non-negative value (ordered like unsigned)
__device__ __forceinline__ void amax_publish(float m, float* slot) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
    if ((threadIdx.x & 63) == 0 && m > 0.f) atomicMax(reinterpret_cast<unsigned*>(slot), __builtin_bit_cast(unsigned, m));
}

// conv_split.hip: absmax of a channel-slice view (fallback when the caller supplies no bound); slot must be zeroed
int absmax_view(const float* x, int C, int Ctot, int N, int HW, float* slot, hipStream_t st);
int absmax_flat(const float* x, size_t n, float* slot, hipStream_t st);

}  // namespace uz
