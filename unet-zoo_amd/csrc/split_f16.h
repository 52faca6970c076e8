// Two-piece fp16 split of fp32 values, shared by conv_split.hip and conv_wgrad_split.hip.
//
//   v * s = h1 + h2 + r,   h1 = fp16(v * s),  h2 = fp16(v * s - h1),   |r| <= 2^-22 |v * s|
// with s a power of two chosen from an upper bound `amax` of the tensor's magnitudes so that |v * s| < 2^14 (no fp16
// overflow; every element above 2^-17 * amax keeps both pieces in the normal fp16 range, smaller ones are off by at
// most 2^-39 * amax in absolute terms).  A product a*b is accumulated in fp32 from THREE piece products
//   a2*b1, a1*b2, a1*b1          (11-bit x 11-bit significands: exact in fp32)
// the dropped a2*b2 is <= 2^-22 |ab|.  A 22-bit emulation with fewer, larger partial sums than a k-ordered fp32 fmaf chain:
// measured against fp64, per layer 0.5 - 0.8x the error of the fp32-MFMA kernels (gate <= 2x: tests/test_full_configs_gpu.py,
// tools/sim_split_accuracy.py); end to end the gradients' median error is 1.6x the fp32 reference's own (tests/test_phiseg_gpu.py).
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

// Largest finite fp16.  A scaled value beyond it can only come from a magnitude bound that is more than 4x too small (stale or
// wrong bound handed in through the C ABI): it is clamped, so that the matrix pipe never sees an infinity (inf - inf = NaN
// would poison a whole output tile), and reported through the device flag word (uz_device_flags) where a kernel checks it.
constexpr float F16_MAX = 65504.f;
// bits of the device flag word
enum { FLAG_X_BOUND = 1, FLAG_W_BOUND = 2, FLAG_DY_BOUND = 4 };
extern int* dev_flags_ptr();                    // conv_split.hip: lazily allocated device word (nullptr when allocation failed)
__device__ __forceinline__ bool bound_violated(float v0, float v1) { return fabsf(v0) > F16_MAX || fabsf(v1) > F16_MAX; }

// splits two (already scaled) values at once; p1 / p2 hold the pieces of (v0, v1) as packed fp16 pairs (v0 in the low half)
__device__ __forceinline__ void split2(float v0, float v1, unsigned& p1, unsigned& p2) {
    v0 = __builtin_amdgcn_fmed3f(v0, -F16_MAX, F16_MAX);
    v1 = __builtin_amdgcn_fmed3f(v1, -F16_MAX, F16_MAX);
    const f32x2 a = {v0, v1};
    const f16x2 h1 = __builtin_convertvector(a, f16x2);                     // round to nearest even
    const f32x2 b = __builtin_convertvector(h1, f32x2);
    const f32x2 r = {v0 - b.x, v1 - b.y};                                   // exact
    const f16x2 h2 = __builtin_convertvector(r, f16x2);
    p1 = __builtin_bit_cast(unsigned, h1);
    p2 = __builtin_bit_cast(unsigned, h2);
}

// ---- split storage ("packed" operands, round 4).  A producer that knows its tensor's bound before it writes the first element
// (BatchNorm forward: the exact output range follows from the statistics; BatchNorm backward: the analytic bound; pooling and
// interpolation: their input's bound) stores every element as ONE 32-bit word holding the two fp16 pieces of v * s
// (low half = h1, high half = h2, s = split_scale(bound)) instead of the fp32 value.  Same bytes; the consuming matrix kernels
// stage the word as it is (two byte permutes per pair of elements instead of scale / clamp / convert / subtract / convert), and
// h1 + h2 reproduces v * s to 2^-22 - exactly the operand the consumer would have formed from the fp32 value itself.
// (round 5: the scaled value is clamped to the finite fp16 range like split2's - a bound more than 4x too small must not put an
//  infinity into a stored operand - and the three-argument form tells the caller, who raises the device flag word once per thread)
__device__ __forceinline__ unsigned pack_split(float v, float s) {
    const float t = __builtin_amdgcn_fmed3f(v * s, -F16_MAX, F16_MAX);
    const _Float16 h1 = (_Float16)t;                                        // round to nearest even
    const _Float16 h2 = (_Float16)(t - (float)h1);                          // the subtraction is exact
    return (unsigned)__builtin_bit_cast(unsigned short, h1) | ((unsigned)__builtin_bit_cast(unsigned short, h2) << 16);
}
__device__ __forceinline__ unsigned pack_split(float v, float s, bool& bad) {
    bad |= fabsf(v * s) > F16_MAX;
    return pack_split(v, s);
}
__device__ __forceinline__ void raise_flag(int* flags, bool bad, int bit) { if (bad && flags) atomicOr(flags, bit); }
__device__ __forceinline__ float unpack_split(unsigned w, float inv_s) {
    const _Float16 h1 = __builtin_bit_cast(_Float16, (unsigned short)(w & 0xFFFFu)), h2 = __builtin_bit_cast(_Float16, (unsigned short)(w >> 16));
    return ((float)h1 + (float)h2) * inv_s;
}
// two packed words (elements k, k + 1) -> the pair's h1 plane word and h2 plane word (element k in the low half)
__device__ __forceinline__ void packed_pair(unsigned w0, unsigned w1, unsigned& p1, unsigned& p2) {
    p1 = __builtin_amdgcn_perm(w1, w0, 0x05040100u);
    p2 = __builtin_amdgcn_perm(w1, w0, 0x07060302u);
}

// ---- bf16 storage (round 4, the volume path: BASELINE config 5).  A tensor kept as bf16 has the same NCHW shape with 2-byte
// elements; values are rounded to nearest even when they are written (v_cvt_pk_bf16_f32) and widened exactly when they are read.
typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
// bits of a float passed BY VALUE.  (__builtin_bit_cast applied directly to an element of an ext_vector - v[1], v.y - reads element 0
// with this compiler: it copies from the vector's base address.  Going through a by-value parameter reads the right element.)
__device__ __forceinline__ unsigned fbits(float v) { return __builtin_bit_cast(unsigned, v); }
__device__ __forceinline__ unsigned pack_bf16x2(float a, float b) { return __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2{a, b}, bf16x2_t)); }
__device__ __forceinline__ float bf16_lo(unsigned w) { return __builtin_bit_cast(float, w << 16); }
__device__ __forceinline__ float bf16_hi(unsigned w) { return __builtin_bit_cast(float, w & 0xFFFF0000u); }
__device__ __forceinline__ float bf16_round(float v) { return bf16_lo(pack_bf16x2(v, 0.f) & 0xFFFFu); }
__device__ __forceinline__ f32x4 bf16x4_widen(uint2 w) { return f32x4{bf16_lo(w.x), bf16_hi(w.x), bf16_lo(w.y), bf16_hi(w.y)}; }
__device__ __forceinline__ uint2 bf16x4_pack(f32x4 v) { return make_uint2(pack_bf16x2(v.x, v.y), pack_bf16x2(v.z, v.w)); }
// element e of a tensor stored as fp32 (b16 == 0) or bf16 (b16 != 0) behind the same pointer type
__device__ __forceinline__ float ld_elem(const float* base, size_t e, int b16) {
    return b16 ? __builtin_bit_cast(float, (unsigned)reinterpret_cast<const unsigned short*>(base)[e] << 16) : base[e];
}
__device__ __forceinline__ void st_elem(float* base, size_t e, float v, int b16) {
    if (b16) reinterpret_cast<unsigned short*>(base)[e] = (unsigned short)(pack_bf16x2(v, 0.f) & 0xFFFFu);
    else base[e] = v;
}
// four consecutive elements starting at element e (e % 4 == 0, row 16-byte / 8-byte aligned)
__device__ __forceinline__ f32x4 ld_elem4(const float* base, size_t e, int b16) {
    if (b16) return bf16x4_widen(*reinterpret_cast<const uint2*>(reinterpret_cast<const unsigned short*>(base) + e));
    return *reinterpret_cast<const f32x4*>(base + e);
}
__device__ __forceinline__ void st_elem4(float* base, size_t e, f32x4 v, int b16) {
    if (b16) *reinterpret_cast<uint2*>(reinterpret_cast<unsigned short*>(base) + e) = bf16x4_pack(v);
    else *reinterpret_cast<f32x4*>(base + e) = v;
}

// eight consecutive elements starting at element e (e % 8 == 0; 16-byte aligned rows in bf16, 32-byte in fp32): ONE 16-byte access per
// bf16 tensor, two per fp32 tensor - the streaming kernels keep full-width memory instructions in either format
__device__ __forceinline__ void ld_elem8(const float* base, size_t e, int b16, f32x4& lo, f32x4& hi) {
    if (b16) {
        const uint4 w = *reinterpret_cast<const uint4*>(reinterpret_cast<const unsigned short*>(base) + e);
        lo = bf16x4_widen(make_uint2(w.x, w.y)); hi = bf16x4_widen(make_uint2(w.z, w.w));
    } else {
        lo = *reinterpret_cast<const f32x4*>(base + e); hi = *reinterpret_cast<const f32x4*>(base + e + 4);
    }
}
__device__ __forceinline__ void st_elem8(float* base, size_t e, f32x4 lo, f32x4 hi, int b16) {
    if (b16) {
        const uint2 a = bf16x4_pack(lo), b = bf16x4_pack(hi);
        *reinterpret_cast<uint4*>(reinterpret_cast<unsigned short*>(base) + e) = make_uint4(a.x, a.y, b.x, b.y);
    } else {
        *reinterpret_cast<f32x4*>(base + e) = lo; *reinterpret_cast<f32x4*>(base + e + 4) = hi;
    }
}

// ---- magnitude-bound slots.  A slot is AMAX_SUB sub-slots AMAX_STRIDE floats (64 bytes) apart; its value is the maximum
// over the sub-slots.  Same-address float atomics serialise at the memory side (MI355X_MICROARCH.md, Global float atomics:
// every workgroup into one row is 14x slower), so producers (a) spread their updates over the sub-slots and (b) read the
// sub-slot first and skip the atomic when it already covers their value - after the first wave of updates almost all do.
// Values are non-negative floats, whose bit patterns order like unsigned integers.
constexpr int AMAX_SUB = 16, AMAX_STRIDE = 16, AMAX_FLOATS = AMAX_SUB * AMAX_STRIDE;

__device__ __forceinline__ float amax_read(const float* slot) {
    float m = 0.f;
#pragma unroll
    for (int i = 0; i < AMAX_SUB; ++i) m = fmaxf(m, slot[i * AMAX_STRIDE]);      // uniform address: scalar loads
    return m;
}
// one lane publishes m (key selects the sub-slot)
__device__ __forceinline__ void amax_publish_one(float m, float* slot, unsigned key) {
    unsigned* p = reinterpret_cast<unsigned*>(slot) + (key % AMAX_SUB) * AMAX_STRIDE;
    const unsigned bits = __builtin_bit_cast(unsigned, m);
    if (bits > __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax(p, bits);
}
// every wave of the workgroup calls this with its lanes' running maxima: wave reduce, lane 0 publishes
__device__ __forceinline__ void amax_publish(float m, float* slot) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
    if ((threadIdx.x & 63) == 0 && m > 0.f)
        amax_publish_one(m, slot, blockIdx.x * 7u + blockIdx.y * 3u + blockIdx.z + (threadIdx.x >> 6));
}

// conv_split.hip: absmax of a channel-slice view (fallback when the caller supplies no bound); slot (AMAX_FLOATS floats) must be zeroed
int absmax_view(const float* x, int C, int Ctot, int N, int HW, float* slot, hipStream_t st);
int absmax_flat(const float* x, size_t n, float* slot, hipStream_t st);

}  // namespace uz
