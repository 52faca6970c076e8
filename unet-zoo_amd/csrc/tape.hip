// Error reporting, device info and the op-tape runner of libuz_hip.so.
// A forward or backward pass of a model is a static list of C-ABI calls ("tape") that the host
// builds once per (model, N, H, W); uz_run_tape replays it with one FFI call, uz_graph_* capture
// it into a hipGraph so that the ~1000 launches of a PHiSeg step cost one graph launch.
#include <stdarg.h>
#include <stdio.h>
#include <string.h>
#include "uz_common.h"

namespace uz {

static thread_local char g_err[768] = "";

void set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}
int fail(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return -1;
}
int check_launch(const char* what) {
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail("%s: launch failed: %s", what, hipGetErrorString(e));
    return 0;
}

}  // namespace uz

extern "C" int uz_version(void) { return UZ_VERSION; }
extern "C" const char* uz_last_error(void) { return uz::g_err; }

extern "C" int uz_device_info(int* n_cu, char* name, int name_cap) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return uz::fail("device_info: no HIP device");
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, dev) != hipSuccess) return uz::fail("device_info: hipGetDeviceProperties failed");
    if (n_cu) *n_cu = prop.multiProcessorCount;
    if (name && name_cap > 0) { strncpy(name, prop.gcnArchName, name_cap - 1); name[name_cap - 1] = 0; }
    return 0;
}

static int run_one(const uz_op& o, void* st) {
    const int32_t* i = o.i;
    const float* f = o.f;
    void* const* p = o.p;
#define FP(k) static_cast<float*>(p[k])
#define CFP(k) static_cast<const float*>(p[k])
    switch (o.code) {
        case UZ_OP_CONV_FWD:
            return uz_conv_fwd(CFP(0), i[0], i[1], CFP(1), CFP(2), FP(3), i[2], i[3], i[4], i[5], i[6], i[7], i[8], p[4], (size_t)o.n, st);
        case UZ_OP_CONV_BWD_DATA:
            return uz_conv_bwd_data(CFP(0), i[0], i[1], CFP(1), FP(2), i[2], i[3], i[4], i[5], i[6], i[7], i[8], p[3], (size_t)o.n, st);
        case UZ_OP_CONV_BWD_WEIGHT:
            return uz_conv_bwd_weight(CFP(0), i[0], i[1], CFP(1), i[2], i[3], FP(2), FP(3), i[4], i[5], i[6], i[7], p[4], (size_t)o.n, st);
        case UZ_OP_BN_RELU_FWD:
            return uz_bn_relu_fwd(CFP(0), i[0], i[1], CFP(1), CFP(2), FP(3), FP(4), FP(5), FP(6), i[2], i[3], i[4], i[5], f[0], f[1], i[6], i[7], p[7], st);
        case UZ_OP_BN_RELU_BWD:
            return uz_bn_relu_bwd(CFP(0), i[0], CFP(1), i[1], i[2], CFP(2), CFP(3), CFP(4), FP(5), i[3], FP(6), FP(7), FP(8), i[4], i[5], i[6], i[7], p[9], st);
        case UZ_OP_RELU_BWD:
            return uz_relu_bwd(CFP(0), i[0], CFP(1), i[1], i[2], FP(2), i[3], FP(3), i[4], i[5], i[6], p[4], st);
        case UZ_OP_AVGPOOL_FWD:
            return uz_avgpool2_fwd(CFP(0), i[0], i[1], FP(1), i[2], i[3], i[4], i[5], st);
        case UZ_OP_AVGPOOL_BWD:
            return uz_avgpool2_bwd(CFP(0), i[0], i[1], FP(1), i[2], i[3], i[4], i[5], i[6], st);
        case UZ_OP_BILINEAR_FWD:
            return uz_bilinear2x_fwd(CFP(0), i[0], i[1], FP(1), i[2], i[3], i[4], i[5], i[6], st);
        case UZ_OP_BILINEAR_BWD:
            return uz_bilinear2x_bwd(CFP(0), i[0], i[1], FP(1), i[2], i[3], i[4], i[5], i[6], i[7], st);
        case UZ_OP_NEAREST_FWD:
            return uz_nearest_fwd(CFP(0), i[0], i[1], FP(1), i[2], i[3], i[4], i[5], i[6], st);
        case UZ_OP_NEAREST_BWD:
            return uz_nearest_bwd(CFP(0), i[0], i[1], FP(1), i[2], i[3], i[4], i[5], i[6], i[7], st);
        case UZ_OP_SPATIAL_MEAN_FWD:
            return uz_spatial_mean_fwd(CFP(0), i[0], i[1], FP(1), i[2], i[3], i[4], st);
        case UZ_OP_SPATIAL_MEAN_BWD:
            return uz_spatial_mean_bwd(CFP(0), i[0], FP(1), i[1], i[2], i[3], i[4], i[5], st);
        case UZ_OP_POSTERIOR_INPUT:
            return uz_posterior_input(CFP(0), i[0], CFP(1), i[1], FP(2), i[2], i[3], i[4], st);
        case UZ_OP_LATENT_FWD:
            return uz_latent_sample_fwd(CFP(0), CFP(1), CFP(2), FP(3), FP(4), (size_t)o.n, i[0], st);
        case UZ_OP_LATENT_BWD:
            return uz_latent_sample_bwd(CFP(0), CFP(1), CFP(2), CFP(3), CFP(4), FP(5), FP(6), (size_t)o.n, i[0], st);
        case UZ_OP_KL_FWD:
            return uz_kl_fwd(CFP(0), CFP(1), CFP(2), CFP(3), i[0], i[1], f[0], FP(4), st);
        case UZ_OP_KL_BWD:
            return uz_kl_bwd(CFP(0), CFP(1), CFP(2), CFP(3), i[0], i[1], f[0], CFP(4), FP(5), FP(6), FP(7), FP(8), st);
        case UZ_OP_CE_FWD:
            return uz_residual_ce_fwd(static_cast<const float* const*>(p[0]), i[0], i[1], CFP(1), i[2], i[3], i[4], FP(2), p[3], st);
        case UZ_OP_CE_BWD:
            return uz_residual_ce_bwd(static_cast<const float* const*>(p[0]), static_cast<float* const*>(p[1]), i[0], i[1], CFP(2), i[2], i[3], i[4], CFP(3), st);
        case UZ_OP_SUM_TERMS:
            return uz_sum_terms(CFP(0), i[0], FP(1), st);
        case UZ_OP_ACC_SOFTMAX_ARGMAX:
            return uz_accumulate_softmax_argmax(static_cast<const float* const*>(p[0]), i[0], i[1], i[2], i[3], i[4], FP(1), FP(2), static_cast<uint8_t*>(p[3]), st);
        case UZ_OP_ADAM:
            return uz_adam_step(FP(0), CFP(1), FP(2), FP(3), (size_t)o.n, (int64_t)i[0], f[0], f[1], f[2], f[3], *reinterpret_cast<const float*>(&i[1]), 1.0f, st);
        case UZ_OP_AXPY:
            return uz_axpy(FP(0), CFP(1), f[0], (size_t)o.n, st);
        case UZ_OP_SCALE:
            return uz_scale(FP(0), f[0], (size_t)o.n, st);
        case UZ_OP_L2_NORMS:
            return uz_l2_norms(CFP(0), static_cast<const int64_t*>(p[1]), i[0], FP(2), st);
        case UZ_OP_L2_NORMS_BWD:
            return uz_l2_norms_bwd(CFP(0), static_cast<const int64_t*>(p[1]), i[0], CFP(2), CFP(3), FP(4), st);
        case UZ_OP_MEMSET:
            if (hipMemsetAsync(p[0], 0, (size_t)o.n, uz::S(st)) != hipSuccess) return uz::fail("memset failed");
            return 0;
        case UZ_OP_COPY:
            if (hipMemcpyAsync(p[0], p[1], (size_t)o.n, hipMemcpyDeviceToDevice, uz::S(st)) != hipSuccess) return uz::fail("copy failed");
            return 0;
        case UZ_OP_BCAST_CHANNELS:
            return uz_bcast_channels_fwd(CFP(0), i[0], FP(1), i[1], i[2], i[3], i[4], st);
        case UZ_OP_BCAST_CHANNELS_BWD:
            return uz_bcast_channels_bwd(CFP(0), i[0], i[1], FP(1), i[2], i[3], i[4], st);
        default:
            return uz::fail("run_tape: unknown op code %d", o.code);
    }
#undef FP
#undef CFP
}

extern "C" int uz_run_tape(const uz_op* ops, int n_ops, void* stream) {
    for (int k = 0; k < n_ops; ++k) {
        const int rc = run_one(ops[k], stream);
        if (rc != 0) {
            char prev[600];
            strncpy(prev, uz::g_err, sizeof(prev) - 1);
            prev[sizeof(prev) - 1] = 0;
            return uz::fail("tape op %d (code %d): %s", k, ops[k].code, prev);
        }
    }
    return 0;
}

extern "C" int uz_graph_create(const uz_op* ops, int n_ops, void* stream, void** graph_exec_out) {
    hipStream_t st = uz::S(stream);
    UZ_REQUIRE(st != nullptr, "graph_create: capture needs a non-default stream");
    if (hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal) != hipSuccess) return uz::fail("graph_create: hipStreamBeginCapture failed");
    const int rc = uz_run_tape(ops, n_ops, stream);
    hipGraph_t graph = nullptr;
    const hipError_t e = hipStreamEndCapture(st, &graph);
    if (rc != 0) { if (graph) (void)hipGraphDestroy(graph); return rc; }
    if (e != hipSuccess || !graph) return uz::fail("graph_create: hipStreamEndCapture failed: %s", hipGetErrorString(e));
    hipGraphExec_t exec = nullptr;
    const hipError_t e2 = hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0);
    (void)hipGraphDestroy(graph);
    if (e2 != hipSuccess) return uz::fail("graph_create: hipGraphInstantiate failed: %s", hipGetErrorString(e2));
    *graph_exec_out = exec;
    return 0;
}
extern "C" int uz_graph_launch(void* graph_exec, void* stream) {
    if (hipGraphLaunch(static_cast<hipGraphExec_t>(graph_exec), uz::S(stream)) != hipSuccess) return uz::fail("graph_launch failed");
    return 0;
}
extern "C" void uz_graph_destroy(void* graph_exec) {
    if (graph_exec) (void)hipGraphExecDestroy(static_cast<hipGraphExec_t>(graph_exec));
}
