// Validation metrics of the reference harness on the device (train_model.py:186-230):
//   - pair counts |A == l and B == l|, |A == l|, |B == l| for every (sample, ground truth) pair of integer
//     label maps: the integer core of generalised_energy_distance (utils.py:148-200, IoU via medpy jc) and of the
//     per-label Dice (train_model.py:212-224, medpy dc).  Exact integer arithmetic -> bit-exact parity.
//   - the two pixel-wise cross-entropy maps of variance_ncc_dist (utils.py:202-247) and the normalised
//     cross-correlation of a map pair (utils.py:130-145).
#include "uz_common.h"

namespace {

__global__ __launch_bounds__(256) void pair_counts_k(const uint8_t* __restrict__ a, const uint8_t* __restrict__ b, int HW, int label,
                                                      int* __restrict__ out) {
    __shared__ int sm[3][4];
    const int i = blockIdx.x, j = blockIdx.y, nb = gridDim.y;
    const uint8_t* pa = a + (size_t)i * HW;
    const uint8_t* pb = b + (size_t)j * HW;
    int inter = 0, ca = 0, cb = 0;
    for (int q = threadIdx.x; q < HW; q += 256) {
        const int x = pa[q] == label, y = pb[q] == label;
        inter += x & y; ca += x; cb += y;
    }
    for (int o = 32; o > 0; o >>= 1) { inter += __shfl_xor(inter, o, 64); ca += __shfl_xor(ca, o, 64); cb += __shfl_xor(cb, o, 64); }
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    if (lane == 0) { sm[0][w] = inter; sm[1][w] = ca; sm[2][w] = cb; }
    __syncthreads();
    if (threadIdx.x < 3) out[((size_t)i * nb + j) * 3 + threadIdx.x] = sm[threadIdx.x][0] + sm[threadIdx.x][1] + sm[threadIdx.x][2] + sm[threadIdx.x][3];
}

// E_ss[p] = mean_i( -sum_k mean_seg[k,p] * log(s_i[k,p] + eps) ),  E_sy[j,p] = mean_i( -sum_k gt_j[k,p] * log(s_i[k,p] + eps) )
__global__ __launch_bounds__(256) void ncc_maps_k(const float* __restrict__ soft, const float* __restrict__ gt, int N, int M, int K, int HW,
                                                   float* __restrict__ Ess, float* __restrict__ Esy) {
    const int q = blockIdx.x * 256 + threadIdx.x;
    if (q >= HW) return;
    double ess = 0.0;
    for (int k = 0; k < K; ++k) {
        float ms = 0.f;
        for (int i = 0; i < N; ++i) ms += soft[((size_t)i * K + k) * HW + q];
        ms /= (float)N;
        double acc = 0.0;
        for (int i = 0; i < N; ++i) acc += (double)(ms * logf(soft[((size_t)i * K + k) * HW + q] + 1e-8f));
        ess -= acc;
    }
    Ess[q] = (float)(ess / N);
    for (int j = 0; j < M; ++j) {
        double acc = 0.0;
        for (int k = 0; k < K; ++k) {
            const float g = gt[((size_t)j * K + k) * HW + q];
            if (g != 0.f)
                for (int i = 0; i < N; ++i) acc -= (double)(g * logf(soft[((size_t)i * K + k) * HW + q] + 1e-8f));
        }
        Esy[(size_t)j * HW + q] = (float)(acc / N);
    }
}

// ncc(a, v) with zero_norm=True (utils.py:130-145) = Pearson correlation of the two maps
__global__ __launch_bounds__(256) void ncc_k(const float* __restrict__ a, const float* __restrict__ v, int HW, float* __restrict__ out) {
    __shared__ double sm[4 * 5];
    const float* vj = v + (size_t)blockIdx.x * HW;
    double s[5] = {0, 0, 0, 0, 0};
    for (int q = threadIdx.x; q < HW; q += 256) {
        const double x = a[q], y = vj[q];
        s[0] += x; s[1] += x * x; s[2] += y; s[3] += y * y; s[4] += x * y;
    }
    uz::block_sum_d<5>(s, sm);
    if (threadIdx.x == 0) {
        const double n = HW, ma = s[0] / n, mv = s[2] / n;
        const double sa = sqrt(fmax(s[1] / n - ma * ma, 0.0)), sv = sqrt(fmax(s[3] / n - mv * mv, 0.0));
        out[blockIdx.x] = (float)((s[4] / n - ma * mv) / (sa * sv));
    }
}

}  // namespace

extern "C" int uz_label_pair_counts(const uint8_t* a, int Na, const uint8_t* b, int Nb, int HW, int label, int32_t* out, void* stream) {
    UZ_REQUIRE(Na > 0 && Nb > 0 && HW > 0 && Nb <= 65535, "label_pair_counts: bad sizes");
    hipLaunchKernelGGL(pair_counts_k, dim3(Na, Nb), dim3(256), 0, uz::S(stream), a, b, HW, label, out);
    return uz::check_launch("pair_counts_k");
}
extern "C" int uz_ncc_maps(const float* soft, const float* gt_onehot, int N, int M, int K, int HW, float* E_ss, float* E_sy, void* stream) {
    UZ_REQUIRE(N > 0 && M > 0 && K > 0 && HW > 0, "ncc_maps: bad sizes");
    hipLaunchKernelGGL(ncc_maps_k, dim3(uz::ceil_div(HW, 256)), dim3(256), 0, uz::S(stream), soft, gt_onehot, N, M, K, HW, E_ss, E_sy);
    return uz::check_launch("ncc_maps_k");
}
extern "C" int uz_ncc(const float* a, const float* v, int M, int HW, float* out, void* stream) {
    UZ_REQUIRE(M > 0 && HW > 0, "ncc: bad sizes");
    hipLaunchKernelGGL(ncc_k, dim3(M), dim3(256), 0, uz::S(stream), a, v, HW, out);
    return uz::check_launch("ncc_k");
}
