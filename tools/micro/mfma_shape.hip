// Micro-benchmark: sustained rate of v_mfma_f32_32x32x16_bf16 vs v_mfma_f32_16x16x32_bf16 on random register operands
// (same output tile per wave: 64 x 64, two waves per SIMD).  Build: hipcc --offload-arch=gfx950 -O3 mfma_shape.hip -o mfma_shape
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

__global__ __launch_bounds__(512) void k32(const bf16x8* in, float* out, int iters) {
    bf16x8 a[2], b[2];
    for (int i = 0; i < 2; ++i) { a[i] = in[threadIdx.x + 512 * i]; b[i] = in[threadIdx.x + 512 * (2 + i)]; }
    f32x16 acc[2][2] = {};
    for (int it = 0; it < iters; ++it)
#pragma unroll
        for (int r = 0; r < 6; ++r)
#pragma unroll
            for (int m = 0; m < 2; ++m)
#pragma unroll
                for (int n = 0; n < 2; ++n) acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[m], b[n], acc[m][n], 0, 0, 0);
    float s = 0.f;
    for (int m = 0; m < 2; ++m) for (int n = 0; n < 2; ++n) for (int r = 0; r < 16; ++r) s += acc[m][n][r];
    out[blockIdx.x * 512 + threadIdx.x] = s;
}
__global__ __launch_bounds__(512) void k16(const bf16x8* in, float* out, int iters) {
    bf16x8 a[4], b[4];
    for (int i = 0; i < 4; ++i) { a[i] = in[threadIdx.x + 512 * i]; b[i] = in[threadIdx.x + 512 * (4 + i)]; }
    f32x4 acc[4][4] = {};
    for (int it = 0; it < iters; ++it)
#pragma unroll
        for (int r = 0; r < 3; ++r)        // 16 blocks x 3 = 48 MFMAs of K=32 = same FLOPs as 24 MFMAs of 32x32x16
#pragma unroll
            for (int m = 0; m < 4; ++m)
#pragma unroll
                for (int n = 0; n < 4; ++n) acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[m], b[n], acc[m][n], 0, 0, 0);
    float s = 0.f;
    for (int m = 0; m < 4; ++m) for (int n = 0; n < 4; ++n) for (int r = 0; r < 4; ++r) s += acc[m][n][r];
    out[blockIdx.x * 512 + threadIdx.x] = s;
}
int main() {
    const int blocks = 256 * 4, iters = 4000;
    bf16x8* in; float* out;
    hipMalloc(&in, 512 * 8 * sizeof(bf16x8)); hipMalloc(&out, blocks * 512 * sizeof(float));
    unsigned short* h = (unsigned short*)malloc(512 * 8 * 16);
    for (int i = 0; i < 512 * 8 * 8; ++i) h[i] = 0x3c00 + (rand() & 0x3ff) + ((rand() & 1) << 15);   // random bf16 around +-0.01..0.03
    hipMemcpy(in, h, 512 * 8 * 16, hipMemcpyHostToDevice);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int v = 0; v < 2; ++v) {
        for (int rep = 0; rep < 2; ++rep) {
            hipEventRecord(e0);
            if (v == 0) hipLaunchKernelGGL(k32, dim3(blocks), dim3(512), 0, 0, in, out, iters);
            else hipLaunchKernelGGL(k16, dim3(blocks), dim3(512), 0, 0, in, out, iters);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            const double flops = (double)blocks * 8 * iters * 24 * 32768.0;     // per wave and iteration: 24 x (32*32*16*2)
            if (rep) printf("%s: %.2f ms  %.1f TFLOP/s bf16\n", v == 0 ? "32x32x16" : "16x16x32", ms, flops / ms / 1e9);
        }
    }
    return 0;
}
