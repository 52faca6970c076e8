// Do raw buffer loads of 8 / 16 bytes work at 2-byte aligned offsets on gfx950 (bf16 storage: a quad of 4 pixels starting at an odd column)?
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
__global__ void k(const unsigned short* src, int n, unsigned long long* out, int shift) {
    const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned short*>(src), 0, (unsigned)(n * 2), 0x00020000);
    const int tid = threadIdx.x;
    const unsigned off = 2u * (unsigned)(4 * tid + shift);
    u32x2 v = __builtin_bit_cast(u32x2, __builtin_amdgcn_raw_buffer_load_b64(r, off, 0, 0));
    out[tid] = (unsigned long long)v.x | ((unsigned long long)v.y << 32);
    u32x4 w = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(r, off, 0, 0));
    out[256 + 2 * tid] = (unsigned long long)w.x | ((unsigned long long)w.y << 32);
    out[256 + 2 * tid + 1] = (unsigned long long)w.z | ((unsigned long long)w.w << 32);
    // LDS: 8-byte store at a 4-byte aligned address, 8-byte read at 2-byte aligned address
    __shared__ __attribute__((aligned(16))) unsigned short l[4096];
    for (int i = tid; i < 4096; i += 256) l[i] = (unsigned short)i;
    __syncthreads();
    unsigned long long t;
    __builtin_memcpy(&t, reinterpret_cast<const char*>(l) + 2 * (4 * tid + shift), 8);
    out[768 + tid] = t;
}
int main() {
    const int n = 4096;
    unsigned short* h = (unsigned short*)malloc(n * 2);
    for (int i = 0; i < n; ++i) h[i] = (unsigned short)i;
    unsigned short* d; unsigned long long* o;
    hipMalloc(&d, n * 2); hipMalloc(&o, 1024 * 8); hipMemcpy(d, h, n * 2, hipMemcpyHostToDevice);
    for (int shift = 0; shift < 4; ++shift) {
        hipMemset(o, 0, 1024 * 8);
        k<<<1, 256>>>(d, n, o, shift);
        unsigned long long ho[1024];
        hipError_t e = hipMemcpy(ho, o, sizeof(ho), hipMemcpyDeviceToHost);
        int bad64 = 0, bad128 = 0, badl = 0;
        for (int t = 0; t < 256; ++t) {
            unsigned long long want = 0, want2 = 0;
            for (int j = 0; j < 4; ++j) { want |= (unsigned long long)(unsigned short)(4 * t + shift + j) << (16 * j); want2 |= (unsigned long long)(unsigned short)(4 * t + shift + 4 + j) << (16 * j); }
            if (ho[t] != want) ++bad64;
            if (ho[256 + 2 * t] != want || ho[256 + 2 * t + 1] != want2) ++bad128;
            if (ho[768 + t] != want) ++badl;
        }
        printf("shift %d: err %d b64 mismatches %d, b128 mismatches %d, lds mismatches %d (first b64 %016llx)\n", shift, (int)e, bad64, bad128, badl, ho[1]);
    }
    return 0;
}
