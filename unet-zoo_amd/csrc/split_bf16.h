// Exact three-way bf16 split of fp32 values, shared by conv_split.hip and conv_wgrad_split.hip.
//   v = h1 + h2 + h3 with h_i bf16 (8 + 8 + 8 significand bits; round-to-nearest pieces, exact remainders).
// A product a*b is then accumulated in fp32 from the six piece products of weight >= 2^-16:
//   a1b1, a1b2, a2b1, a1b3, a3b1, a2b2   (bf16 x bf16 is exact in fp32; the dropped terms are <= 2^-24 |ab| each).
#pragma once
#include <hip/hip_runtime.h>

namespace uz {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

// splits two values at once; p1 / p2 / p3 hold the pieces of (v0, v1) as packed bf16 pairs (v0 in the low half)
__device__ __forceinline__ void split3(float v0, float v1, unsigned& p1, unsigned& p2, unsigned& p3) {
    const f32x2 a = {v0, v1};
    p1 = __builtin_bit_cast(unsigned, __builtin_convertvector(a, bf16x2));
    const f32x2 r1 = {v0 - __builtin_bit_cast(float, p1 << 16), v1 - __builtin_bit_cast(float, p1 & 0xFFFF0000u)};
    p2 = __builtin_bit_cast(unsigned, __builtin_convertvector(r1, bf16x2));
    const f32x2 r2 = {r1.x - __builtin_bit_cast(float, p2 << 16), r1.y - __builtin_bit_cast(float, p2 & 0xFFFF0000u)};
    p3 = __builtin_bit_cast(unsigned, __builtin_convertvector(r2, bf16x2));
}

}  // namespace uz
