// Error reporting, device info and the op-tape runner of libuz_hip.so.
// A forward or backward pass of a model is a static list of C-ABI calls ("tape") that the host
// builds once per (model, N, H, W); uz_run_tape replays it with one FFI call, uz_graph_* capture
// it into a hipGraph so that the ~1000 launches of a PHiSeg step cost one graph launch.
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
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

// 0 = fp32 MFMA only, 1 = split-fp16 where it pays (default), 2 = split-fp16 on every eligible 3x3 shape, 3 = bf16: the layers of
// mode 1 with ONE bf16 piece per operand and one product (bf16 arithmetic, fp32 accumulation - BASELINE config 5).
// Process default from UZ_CONV_MATH (f32 | split | bf16), overridable at run time through uz_set_conv_math (tests, diagnostics).
static int g_conv_math = -1;
int conv_math_mode() {
    if (g_conv_math >= 0) return g_conv_math;
    static const int env = [] { const char* e = getenv("UZ_CONV_MATH"); return !e ? 1 : !strcmp(e, "f32") ? 0 : !strcmp(e, "split") ? 2 : !strcmp(e, "bf16") ? 3 : 1; }();
    return env;
}

}  // namespace uz

extern "C" int uz_set_conv_math(int mode) {
    if (mode < -1 || mode > 3) return uz::fail("set_conv_math: mode %d not in -1..3", mode);
    uz::g_conv_math = mode;
    return 0;
}
extern "C" int uz_get_conv_math(void) { return uz::conv_math_mode(); }

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

// What this binary is (uz_build_info): a product build multiplies every operand pair with three piece products and carries no
// experiment code; `make VARIANT=... XFLAGS="-DUZ_EXP_..."` builds say so here, and bench.py refuses to report a line from them.
#ifndef UZ_VARIANT
#define UZ_VARIANT ""
#endif
#ifndef UZ_SRC_HASH
#define UZ_SRC_HASH "unknown"
#endif
extern "C" int uz_build_info(char* out, int cap) {
    int products = 3, experiment = 0;
#ifdef UZ_EXP_PRODUCTS
    products = UZ_EXP_PRODUCTS; experiment |= (UZ_EXP_PRODUCTS != 3);
#endif
    int patch_dma = 0, pref_all = 0, diag = 0;
#ifdef UZ_EXP_PATCH_DMA
    patch_dma = 1; experiment = 1;
#endif
#ifdef UZ_EXP_PREF_ALL
    pref_all = 1; experiment = 1;
#endif
#ifdef UZ_DIAG
    diag = 1; experiment = 1;
#endif
    if (UZ_VARIANT[0]) experiment = 1;
    if (out && cap > 0)
        snprintf(out, (size_t)cap, "{\"variant\": \"%s\", \"products_per_mac\": %d, \"exp_patch_dma\": %d, \"exp_pref_all\": %d, \"diag_skip_compiled\": %d, "
                                   "\"source_hash\": \"%s\", \"experiment\": %d}", UZ_VARIANT, products, patch_dma, pref_all, diag, UZ_SRC_HASH, experiment);
    return experiment;
}

#ifdef UZ_DIAG
// Diagnostics only, compiled in by `make VARIANT=diag XFLAGS=-DUZ_DIAG` (tools/what_if.sh), never in the product library:
// UZ_DIAG_SKIP="conv:8,bn:8,resample:4,convmin:64" drops every convolution / BatchNorm / resampling
// op on planes up to (convmin: from) that height from a tape - the results are garbage, the step time shows what those ops cost
// on the critical path (an upper bound for any optimisation of them).
// The first UZ_DIAG_WARM tape replays (default 8 = four steps) run everything: the skipped ops' outputs then hold the values of a real step
// (bench.py feeds the same batch every step), so what runs downstream sees realistic operands.  Without it their buffers stay all-zero and the
// clock-limited kernels downstream run 14 - 20 % faster on zeros (NOTES_r6 section 8) - the first what-if tables of round 6 carried that bias.
static int g_diag_tapes = 0;
static bool diag_skip(const uz_op& o) {
    static const char* env = getenv("UZ_DIAG_SKIP");
    if (!env) return false;
    static const int warm = getenv("UZ_DIAG_WARM") ? atoi(getenv("UZ_DIAG_WARM")) : 8;
    if (g_diag_tapes <= warm) return false;
    static int conv = -1, bn = -1, rs = -1, convmin = 1 << 30, init = 0, fwd = -1, dgrad = -1, wgrad = -1, bnf = -1, bnb = -1, only = 0;
    if (!init) {
        init = 1;
        const char* q;
        if ((q = strstr(env, "conv:"))) conv = atoi(q + 5);
        if ((q = strstr(env, "bn:"))) bn = atoi(q + 3);
        if ((q = strstr(env, "resample:"))) rs = atoi(q + 9);
        if ((q = strstr(env, "convmin:"))) convmin = atoi(q + 8);
        if ((q = strstr(env, "fwd:"))) fwd = atoi(q + 4);            // per direction: fwd:8 = forward convolutions on planes up to 8 rows
        if ((q = strstr(env, "dgrad:"))) dgrad = atoi(q + 6);
        if ((q = strstr(env, "wgrad:"))) wgrad = atoi(q + 6);
        if ((q = strstr(env, "bnf:"))) bnf = atoi(q + 4);
        if ((q = strstr(env, "bnb:"))) bnb = atoi(q + 4);
        if ((q = strstr(env, "only:"))) only = atoi(q + 5);          // only:16 = the limits above select planes of EXACTLY that height
    }
    auto hit = [&](int h, int lim) { return only ? (h == only && lim >= only) : h <= lim; };
    switch (o.code) {
        case UZ_OP_CONV_FWD: return hit(o.i[5], conv) || hit(o.i[5], fwd) || o.i[5] >= convmin;
        case UZ_OP_CONV_BWD_DATA: return hit(o.i[5], conv) || hit(o.i[5], dgrad) || o.i[5] >= convmin;
        case UZ_OP_CONV_BWD_WEIGHT: return hit(o.i[5], conv) || hit(o.i[5], wgrad) || o.i[5] >= convmin;
        case UZ_OP_BN_RELU_FWD: return hit(o.i[4], bn) || hit(o.i[4], bnf);
        case UZ_OP_BN_RELU_BWD: return hit(o.i[5], bn) || hit(o.i[5], bnb);
        case UZ_OP_AVGPOOL_FWD: case UZ_OP_AVGPOOL_BWD: case UZ_OP_BILINEAR_FWD: case UZ_OP_BILINEAR_BWD: return hit(o.i[4], rs);
        default: return false;
    }
}
#endif

static int run_one(const uz_op& o, void* st) {
#ifdef UZ_DIAG
    if (diag_skip(o)) return 0;
#endif
    const int32_t* i = o.i;
    const float* f = o.f;
    void* const* p = o.p;
#define FP(k) static_cast<float*>(p[k])
#define CFP(k) static_cast<const float*>(p[k])
    switch (o.code) {
        case UZ_OP_CONV_FWD:
            // i[13]: bf16 STORAGE bits (Plan._b16_pass) - bit 0 = x, bit 1 = y hold 2-byte bf16 elements
            if (i[13] && i[7] == 1) return uz_conv1x1_fwd_b16(p[0], i[0], i[1], CFP(1), CFP(2), FP(3), i[2], i[3], i[4], i[5], i[6], i[13] & 1, st);      // a 1x1 head: only x may be bf16
            if (i[13]) return uz_conv_fwd_b16(p[0], i[0], i[1], CFP(1), CFP(2), p[3], i[2], i[3], i[4], i[5], i[6], i[7], p[4], (size_t)o.n, p[8], FP(9), i[13] & 1, (i[13] >> 1) & 1, st);
            if (i[9]) return uz_conv_fwd_slabs(CFP(0), i[0], i[1], CFP(1), i[2], i[4], i[5], i[6], i[7], p[4], (size_t)o.n, st);      // the unit's BatchNorm adds the slabs
            // i[12] = 1 | relu << 1: x is the producing unit's PRE-normalisation output, p[11] its statistics table - the staging applies BatchNorm + ReLU
            if (i[12]) return uz_conv_fwd_bn_ex(CFP(0), i[0], i[1], CFP(11), (i[12] >> 1) & 1, CFP(1), CFP(2), FP(3), i[2], i[3], i[4], i[5], i[6], i[7], CFP(5), CFP(6), FP(7), p[4], (size_t)o.n, p[8], FP(9), st);
            // i[10] = input in split storage, p[10] / i[11] = bound and first channel of its second scale segment
            return uz_conv_fwd_ex(CFP(0), i[0], i[1], CFP(1), CFP(2), FP(3), i[2], i[3], i[4], i[5], i[6], i[7], i[8], CFP(5), CFP(6), FP(7), p[4], (size_t)o.n, p[8], FP(9),
                                  i[10], CFP(10), i[11], st);
        case UZ_OP_CONV_BWD_DATA:
            // i[10]: fold of the unit that produced the output's forward twin - 1 ReLU mask (p[7] = its activation), 2 BatchNorm-backward
            // reduction (p[7] = its pre-normalisation output, p[10] = its statistics table, i[12] = its relu flag); i[11] = dy in split storage
            if (i[13] && i[7] == 1) return uz_conv1x1_bwd_data_b16(CFP(0), i[0], i[1], CFP(1), p[2], i[2], i[3], i[4], i[5], i[6], i[8], (i[13] >> 1) & 1, st);
            if (i[13]) return uz_conv_bwd_data_b16(p[0], i[0], i[1], CFP(1), p[2], i[2], i[3], i[4], i[5], i[6], i[7], i[8], p[3], (size_t)o.n, p[6], i[13] & 1, (i[13] >> 1) & 1, st);      // bit 0 = dy, bit 1 = dx in bf16 storage
            if (i[10] == 3) return uz_conv_bwd_data_slabs(CFP(0), i[0], i[1], CFP(1), i[2], i[4], i[5], i[6], i[7], FP(7), st);      // slabs only: the consumer's BatchNorm backward adds them
            if (i[10] == 1) return uz_conv_bwd_data_relu(CFP(0), i[0], i[1], CFP(1), FP(2), i[2], i[3], i[4], i[5], i[6], i[7], i[8], CFP(4), CFP(5), p[3], (size_t)o.n, p[6], CFP(7), i[9], FP(8), FP(9), st);
            return uz_conv_bwd_data_ex(CFP(0), i[0], i[1], CFP(1), FP(2), i[2], i[3], i[4], i[5], i[6], i[7], i[8], CFP(4), CFP(5), p[3], (size_t)o.n, p[6], i[11],
                                       i[10] == 2 ? CFP(7) : nullptr, i[9], CFP(10), i[12], FP(8), st);
        case UZ_OP_CONV_BWD_WEIGHT:
            // i[8] = x in split storage, p[7] / i[9] = second scale segment of x, i[10] = dy in split storage
            // i[11]: slabs only, into p[8] (added by UZ_OP_WGRAD_REDUCE_TABLE at the end of the tape); i[12]: the slab count p[8] and the
            // table row were SIZED for when the plan was built.  uz_set_wgrad_target is process state: a caller that replays this tape
            // under another target would write more slabs than were allocated and reduce the wrong number - refuse instead
            if (i[11] && i[12] > 0 && uz_conv_bwd_weight_slabs(i[0], i[2], i[4], i[5], i[6], i[7]) != i[12])
                return uz::fail("conv_bwd_weight: the plan sized %d slabs for this layer, the library would now write %d (uz_set_wgrad_target / "
                                "uz_set_conv_math changed between plan build and replay: rebuild the plan or restore the setting)",
                                i[12], uz_conv_bwd_weight_slabs(i[0], i[2], i[4], i[5], i[6], i[7]));
            if (i[13] && i[7] == 1) return uz_conv1x1_bwd_weight_b16(p[0], i[0], i[1], CFP(1), i[2], i[3], FP(2), FP(3), i[4], i[5], i[6], p[4], (size_t)o.n, i[13] & 1, st);
            if (i[13]) return uz_conv_bwd_weight_b16(p[0], i[0], i[1], p[1], i[2], i[3], FP(2), i[4], i[5], i[6], i[7], p[4], (size_t)o.n, i[13] & 1, (i[13] >> 1) & 1, i[11] ? FP(8) : nullptr, st);      // bit 0 = x, bit 1 = dy in bf16 storage
            return uz_conv_bwd_weight_ex(CFP(0), i[0], i[1], CFP(1), i[2], i[3], FP(2), FP(3), i[4], i[5], i[6], i[7], CFP(5), CFP(6), p[4], (size_t)o.n,
                                         i[8], CFP(7), i[9], i[10], i[11] ? FP(8) : nullptr, st);
        case UZ_OP_BN_RELU_FWD:
            if (i[13]) return uz_bn_relu_fwd_b16(p[0], i[0], i[1], CFP(1), CFP(2), FP(3), FP(4), FP(5), p[6], i[2], i[3], i[4], i[5], f[0], f[1], i[6], i[7], p[7], CFP(9), i[8], i[13] & 1, (i[13] >> 1) & 1, st);      // bit 0 = y, bit 1 = a in bf16 storage
            if (i[9] > 1) return uz_bn_relu_fwd_slabs(CFP(10), i[9], CFP(11), FP(0), i[0], i[1], CFP(1), CFP(2), FP(3), FP(4), FP(5), FP(6), i[2], i[3], i[4], i[5], f[0], f[1], i[6], i[7], FP(8), st);
            // i[11] = phase (Plan._bn_offchain_pass): 1 statistics only, 2 apply only
            if (i[11]) return uz_bn_relu_fwd_phase(CFP(0), i[0], i[1], CFP(1), CFP(2), FP(3), FP(4), FP(5), FP(6), i[2], i[3], i[4], i[5], f[0], f[1], i[7], FP(8), CFP(9), i[8], i[10], i[11], st);
            // (save holds 4 C floats in plans; i[10] = write the activation as split storage)
            return uz_bn_relu_fwd_ex(CFP(0), i[0], i[1], CFP(1), CFP(2), FP(3), FP(4), FP(5), FP(6), i[2], i[3], i[4], i[5], f[0], f[1], i[6], i[7], FP(8), p[7], CFP(9), i[8], i[10], st);
        case UZ_OP_BN_RELU_BWD:
            // p[11] / i[8] = reduction partials left by the data gradient that wrote dA last, i[9] = write dy as split storage
            // i[10]: p[8] holds the rows for the conv-bias gradient's partial sums instead of the gradient itself (added by UZ_OP_CHAN_SUM_TABLE)
            // i[11] > 0: p[11] holds that many split-K slabs of dA left by the data gradient (small planes) instead of reduction partials
            if (i[13]) return uz_bn_relu_bwd_b16(p[0], i[0], p[1], i[1], i[2], CFP(2), CFP(3), CFP(4), p[5], i[3], FP(6), FP(7), FP(8), i[4], i[5], i[6], i[7], p[9], i[13] & 1, (i[13] >> 1) & 1, (i[13] >> 2) & 1, st);      // bits: dA, y, dy
            return uz_bn_relu_bwd_ex(CFP(0), i[0], CFP(1), i[1], i[2], CFP(2), CFP(3), CFP(4), FP(5), i[3], FP(6), FP(7), i[10] ? nullptr : FP(8), i[4], i[5], i[6], i[7], FP(10), p[9],
                                     i[11] ? nullptr : CFP(11), i[8], i[9], i[10] ? static_cast<double*>(p[8]) : nullptr, i[11] ? CFP(11) : nullptr, i[11], st);
        case UZ_OP_RELU_BWD:
            return uz_relu_bwd(CFP(0), i[0], CFP(1), i[1], i[2], FP(2), i[3], FP(3), i[4], i[5], i[6], FP(5), p[4], st);
        case UZ_OP_AVGPOOL_FWD:
            return uz_avgpool2_fwd_ex(CFP(0), i[0], i[1], FP(1), i[2], i[3], i[4], i[5], CFP(2), FP(3), i[6], st);
        case UZ_OP_AVGPOOL_BWD:
            if (p[2]) return uz_avgpool2_bwd_relu(CFP(0), i[0], i[1], FP(1), i[2], i[3], i[4], i[5], i[6], CFP(2), i[7], static_cast<double*>(p[3]), FP(4), st);
            return uz_avgpool2_bwd(CFP(0), i[0], i[1], FP(1), i[2], i[3], i[4], i[5], i[6], st);
        case UZ_OP_BILINEAR_FWD:
            if (i[13]) return uz_bilinear2x_fwd_b16(CFP(0), i[0], i[1], p[1], i[2], i[3], i[4], i[5], i[6], (i[13] >> 1) & 1, st);      // bf16 storage of the high-resolution output
            return uz_bilinear2x_fwd_ex(CFP(0), i[0], i[1], FP(1), i[2], i[3], i[4], i[5], i[6], CFP(2), FP(3), i[7], st);
        case UZ_OP_BILINEAR_BWD:
            if (i[13]) return uz_bilinear2x_bwd_b16(p[0], i[0], i[1], FP(1), i[2], i[3], i[4], i[5], i[6], i[7], i[13] & 1, st);      // bf16 storage of the incoming (high-resolution) gradient
            if (p[2]) return uz_bilinear2x_bwd_relu(CFP(0), i[0], i[1], FP(1), i[2], i[3], i[4], i[5], i[6], i[7], CFP(2), i[8], static_cast<double*>(p[3]), FP(4), st);
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
            return uz_kl_fwd_ws(CFP(0), CFP(1), CFP(2), CFP(3), i[0], i[1], f[0], FP(4), p[5], st);
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
        case UZ_OP_CHAN_SUM_TABLE:
            return uz_chan_sum_table(static_cast<const int64_t*>(p[0]), i[0], i[1], st);
        case UZ_OP_WGRAD_REDUCE_TABLE:
            return uz_wgrad_reduce_table(static_cast<const int64_t*>(p[0]), i[0], i[1], st);
        case UZ_OP_CHAN_SUM_PARTIALS:
            return i[2] ? uz_chan_sum_partials_d(static_cast<const double*>(p[0]), i[0], i[1], FP(1), st) : uz_chan_sum_partials(CFP(0), i[0], i[1], FP(1), st);
        case UZ_OP_PACK_WEIGHTS:
            return uz_conv_pack_weights(static_cast<const int64_t*>(p[0]), i[0], i[1], CFP(1), st);
        case UZ_OP_MEMSET:
            return uz_zero_f32(FP(0), (size_t)o.n / 4, st);               /* n = bytes, always whole floats */
        case UZ_OP_COPY:
            return uz_copy_f32(FP(0), CFP(1), (size_t)o.n / 4, st);
        case UZ_OP_BCAST_CHANNELS:
            return uz_bcast_channels_fwd(CFP(0), i[0], FP(1), i[1], i[2], i[3], i[4], st);
        case UZ_OP_BCAST_CHANNELS_BWD:
            return uz_bcast_channels_bwd(CFP(0), i[0], i[1], FP(1), i[2], i[3], i[4], st);
        case UZ_OP_ABSMAX_COPY:
            return uz_absmax_copy(CFP(0), FP(1), st);
        case UZ_OP_W3D_PERMUTE:
            return uz_w3d_permute(CFP(0), FP(1), i[0], i[1], i[2], st);
        case UZ_OP_AVGPOOL3D_FWD:
            if (i[13]) return uz_avgpool3d_fwd_b16(p[0], i[0], i[1], p[1], i[2], i[3], i[4], i[5], i[13] & 1, (i[13] >> 1) & 1, st);
            return uz_avgpool3d_fwd(CFP(0), i[0], i[1], FP(1), i[2], i[3], i[4], i[5], st);
        case UZ_OP_AVGPOOL3D_BWD:
            if (i[13]) return uz_avgpool3d_bwd_b16(p[0], i[0], i[1], p[1], i[2], i[3], i[4], i[5], i[6], i[13] & 1, (i[13] >> 1) & 1, st);
            return uz_avgpool3d_bwd(CFP(0), i[0], i[1], FP(1), i[2], i[3], i[4], i[5], i[6], st);
        case UZ_OP_DEPTH_LERP_FWD:
            if (i[13]) return uz_depth_lerp2x_fwd_b16(p[0], i[0], i[1], p[1], i[2], i[3], i[4], i[5], i[13] & 1, (i[13] >> 1) & 1, st);
            return uz_depth_lerp2x_fwd(CFP(0), i[0], i[1], FP(1), i[2], i[3], i[4], i[5], st);
        case UZ_OP_DEPTH_LERP_BWD:
            if (i[13]) return uz_depth_lerp2x_bwd_b16(p[0], i[0], i[1], p[1], i[2], i[3], i[4], i[5], i[6], i[13] & 1, (i[13] >> 1) & 1, st);
            return uz_depth_lerp2x_bwd(CFP(0), i[0], i[1], FP(1), i[2], i[3], i[4], i[5], i[6], st);
        case UZ_OP_NEAREST3D_FWD:
            return uz_nearest3d_fwd(CFP(0), i[0], i[1], FP(1), i[2], i[3], i[4], i[5], i[6], i[7], st);
        case UZ_OP_NEAREST3D_BWD:
            return uz_nearest3d_bwd(CFP(0), i[0], i[1], FP(1), i[2], i[3], i[4], i[5], i[6], i[7], i[8], st);
        case UZ_OP_ADD_VIEWS:
            return uz_add_views(CFP(0), i[0], CFP(1), i[1], FP(2), i[2], i[3], i[4], i[5], i[6], f[0], i[7], CFP(3), CFP(4), FP(5), st);
        case UZ_OP_ABSMAX:
            return uz_absmax(CFP(0), (size_t)o.n, FP(1), st);
        case UZ_OP_EVENT_RECORD:
            return uz_event_record(p[0], st);
        case UZ_OP_LATENT_HEADS_FWD:
            return uz_latent_heads_fwd(CFP(0), i[0], i[1], CFP(1), CFP(2), CFP(3), CFP(4), CFP(5), FP(6), FP(7), FP(8), FP(9), i[2], i[3], i[4], i[5], i[6], st);
        case UZ_OP_LATENT_HEADS_BWD_DATA:
            return uz_latent_heads_bwd_data(CFP(0), CFP(1), i[0], CFP(2), CFP(3), FP(4), i[1], i[2], i[3], i[4], i[5], i[6], st);
        case UZ_OP_LATENT_HEADS_BWD_WEIGHT:
            return uz_latent_heads_bwd_weight(CFP(0), i[0], i[1], CFP(1), CFP(2), i[2], FP(3), FP(4), FP(5), FP(6), i[3], i[4], i[5], p[7], (size_t)o.n, st);
        case UZ_OP_CHAIN:
            return uz_chain_run(static_cast<const uz_chain_op*>(p[0]), static_cast<const int32_t*>(p[1]), i[0], i[2], i[1], p[2], st);
        case UZ_OP_CHAIN_PACK:
            return uz_chain_pack_weights(static_cast<const int64_t*>(p[0]), i[0], i[1], CFP(1), st);
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
// Lane capture: the tape is captured on ONE stream, but before each scheduling group the capture
// dependency set is replaced (hipStreamUpdateCaptureDependencies) by the tail node of the group's
// lane plus the tail nodes of the groups it waits for, so the captured graph carries the tape's true
// DAG instead of one chain.  Independent chains (posterior / prior encoders, likelihood branches)
// then overlap on the device at replay.  sched[k].lane also selects the scratch copy the host
// resolved into the op, which is why groups sharing a lane must stay ordered.
extern "C" int uz_graph_create_lanes(const uz_op* ops, const uz_sched* sched, int n_ops, int n_lanes,
                                     void* stream, void** graph_exec_out) {
    hipStream_t st = uz::S(stream);
    UZ_REQUIRE(st != nullptr, "graph_create_lanes: capture needs a non-default stream");
    UZ_REQUIRE(n_lanes >= 1 && n_lanes <= UZ_MAX_LANES, "graph_create_lanes: n_lanes %d outside [1,%d]", n_lanes, UZ_MAX_LANES);
    for (int k = 0; k < n_ops; ++k) {
        UZ_REQUIRE(sched[k].lane >= 0 && sched[k].lane < n_lanes, "graph_create_lanes: op %d lane %d", k, sched[k].lane);
        UZ_REQUIRE(sched[k].n_wait >= 0 && sched[k].n_wait <= UZ_MAX_LANES, "graph_create_lanes: op %d n_wait %d", k, sched[k].n_wait);
        for (int w = 0; w < sched[k].n_wait; ++w)
            UZ_REQUIRE(sched[k].wait[w] >= 0 && sched[k].wait[w] < k && sched[sched[k].wait[w]].signal,
                       "graph_create_lanes: op %d waits on op %d which is not an earlier signalling op", k, sched[k].wait[w]);
    }
    constexpr int TAILCAP = 4;                      // capture tail of one op: normally exactly one node
    struct Tail { hipGraphNode_t n[TAILCAP]; int cnt; };
    Tail* tails = static_cast<Tail*>(calloc(static_cast<size_t>(n_ops) + 1, sizeof(Tail)));
    if (!tails) return uz::fail("graph_create_lanes: out of host memory");
    Tail lane_tail[UZ_MAX_LANES] = {};
    bool lane_used[UZ_MAX_LANES] = {};
    int rc = 0;
    auto ok = [&](hipError_t e, const char* what) {
        if (e != hipSuccess && rc == 0) rc = uz::fail("graph_create_lanes: %s: %s", what, hipGetErrorString(e));
        return e == hipSuccess;
    };
    if (!ok(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal), "hipStreamBeginCapture")) { free(tails); return rc; }
    auto current_tail = [&](Tail& t) {
        hipStreamCaptureStatus status = hipStreamCaptureStatusNone;
        unsigned long long id = 0;
        hipGraph_t g = nullptr;
        const hipGraphNode_t* deps = nullptr;
        size_t nd = 0;
        if (!ok(hipStreamGetCaptureInfo_v2(st, &status, &id, &g, &deps, &nd), "hipStreamGetCaptureInfo_v2")) return;
        if (nd > TAILCAP) { rc = uz::fail("graph_create_lanes: capture tail of %zu nodes", nd); return; }
        t.cnt = static_cast<int>(nd);
        for (size_t j = 0; j < nd; ++j) t.n[j] = deps[j];
    };
    for (int k = 0; k < n_ops && rc == 0; ++k) {
        const int lane = sched[k].lane;
        const bool group_start = (k == 0) || sched[k].n_wait > 0 || sched[k - 1].lane != lane || sched[k - 1].signal;
        if (group_start) {
            hipGraphNode_t deps[TAILCAP * (UZ_MAX_LANES + 1)];
            int nd = 0;
            if (lane_used[lane]) for (int j = 0; j < lane_tail[lane].cnt; ++j) deps[nd++] = lane_tail[lane].n[j];
            for (int w = 0; w < sched[k].n_wait; ++w) {
                const Tail& t = tails[sched[k].wait[w]];
                for (int j = 0; j < t.cnt; ++j) deps[nd++] = t.n[j];
            }
            if (!ok(hipStreamUpdateCaptureDependencies(st, deps, static_cast<size_t>(nd), hipStreamSetCaptureDependencies), "hipStreamUpdateCaptureDependencies")) break;
        }
        if (run_one(ops[k], stream) != 0) {
            char prev[600];
            strncpy(prev, uz::g_err, sizeof(prev) - 1);
            prev[sizeof(prev) - 1] = 0;
            rc = uz::fail("tape op %d (code %d): %s", k, ops[k].code, prev);
            break;
        }
        current_tail(lane_tail[lane]);
        lane_used[lane] = true;
        if (sched[k].signal) tails[k] = lane_tail[lane];
    }
    if (rc == 0) {                                   // join: the capture ends with every lane's tail as a dependency
        hipGraphNode_t deps[TAILCAP * UZ_MAX_LANES];
        int nd = 0;
        for (int l = 0; l < n_lanes; ++l)
            if (lane_used[l]) for (int j = 0; j < lane_tail[l].cnt; ++j) deps[nd++] = lane_tail[l].n[j];
        ok(hipStreamUpdateCaptureDependencies(st, deps, static_cast<size_t>(nd), hipStreamSetCaptureDependencies), "join dependencies");
    }
    hipGraph_t graph = nullptr;
    const hipError_t e = hipStreamEndCapture(st, &graph);
    if (rc == 0 && (e != hipSuccess || !graph)) rc = uz::fail("graph_create_lanes: hipStreamEndCapture failed: %s", hipGetErrorString(e));
    free(tails);
    hipGraphExec_t exec = nullptr;
    if (rc == 0) ok(hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0), "hipGraphInstantiate");
    if (graph) (void)hipGraphDestroy(graph);
    if (rc != 0) { if (exec) (void)hipGraphExecDestroy(exec); return rc; }
    *graph_exec_out = exec;
    return 0;
}
// Lane replay WITHOUT a hipGraph (round 5): the same DAG, issued by the host on one HIP stream per lane with one event per
// cross-lane edge - lane 0 is `stream` itself, the other lanes are library-owned non-blocking streams that fork from and join
// `stream`.  Every wait names exactly the op it waits for, so an op starts as soon as ITS predecessors are done (the ROCm 7.2
// graph executor enqueues a captured DAG in an order of its own and an in-order hardware queue then holds ready nodes behind
// unrelated waits: the prior's encoder started 1.3 ms late in a replayed PHiSeg step).
namespace {
struct LanePool {
    hipStream_t lanes[UZ_MAX_LANES] = {};
    hipEvent_t* ev = nullptr; int n_ev = 0;
    hipEvent_t fork = nullptr, join[UZ_MAX_LANES] = {};
};
LanePool g_lp;
// diagnostics (uz_lane_trace): a timing event behind every op of the last uz_run_tape_lanes call
struct LaneTrace { bool on = false; hipEvent_t base = nullptr; hipEvent_t* ev = nullptr; int cap = 0, n = 0; };
LaneTrace g_lt;
int lane_pool_reserve(int n_lanes, int n_events) {
    for (int l = 1; l < n_lanes; ++l) {
        if (!g_lp.lanes[l]) {
            // (stream priorities and a CU-masked stream for one lane were measured in round 5 and removed: priorities change nothing, a masked
            //  stream runs the heaviest convolution 27 % slower even with an all-ones mask - profiles/NOTES_r5.md section 3, commit 0cbd7b7)
            if (hipStreamCreateWithFlags(&g_lp.lanes[l], hipStreamNonBlocking) != hipSuccess) return uz::fail("run_tape_lanes: cannot create lane stream");
        }
        if (!g_lp.join[l] && hipEventCreateWithFlags(&g_lp.join[l], hipEventDisableTiming) != hipSuccess) return uz::fail("run_tape_lanes: cannot create event");
    }
    if (!g_lp.fork && hipEventCreateWithFlags(&g_lp.fork, hipEventDisableTiming) != hipSuccess) return uz::fail("run_tape_lanes: cannot create event");
    if (n_events > g_lp.n_ev) {
        hipEvent_t* ne = static_cast<hipEvent_t*>(realloc(g_lp.ev, sizeof(hipEvent_t) * static_cast<size_t>(n_events)));
        if (!ne) return uz::fail("run_tape_lanes: out of host memory");
        g_lp.ev = ne;
        for (int k = g_lp.n_ev; k < n_events; ++k)
            if (hipEventCreateWithFlags(&g_lp.ev[k], hipEventDisableTiming) != hipSuccess) { g_lp.n_ev = k; return uz::fail("run_tape_lanes: cannot create event"); }
        g_lp.n_ev = n_events;
    }
    return 0;
}
}  // namespace
extern "C" int uz_run_tape_lanes(const uz_op* ops, const uz_sched* sched, int n_ops, int n_lanes, void* stream) {
#ifdef UZ_DIAG
    if (n_ops > 16) ++g_diag_tapes;
#endif
    UZ_REQUIRE(n_lanes >= 1 && n_lanes <= UZ_MAX_LANES, "run_tape_lanes: n_lanes %d outside [1,%d]", n_lanes, UZ_MAX_LANES);
    int n_sig = 0;
    for (int k = 0; k < n_ops; ++k) {
        UZ_REQUIRE(sched[k].lane >= 0 && sched[k].lane < n_lanes, "run_tape_lanes: op %d lane %d", k, sched[k].lane);
        UZ_REQUIRE(sched[k].n_wait >= 0 && sched[k].n_wait <= UZ_MAX_LANES, "run_tape_lanes: op %d n_wait %d", k, sched[k].n_wait);
        for (int w = 0; w < sched[k].n_wait; ++w)
            UZ_REQUIRE(sched[k].wait[w] >= 0 && sched[k].wait[w] < k && sched[sched[k].wait[w]].signal,
                       "run_tape_lanes: op %d waits on op %d which is not an earlier signalling op", k, sched[k].wait[w]);
        n_sig += sched[k].signal != 0;
    }
    if (int rc = lane_pool_reserve(n_lanes, n_sig)) return rc;
    hipStream_t main = uz::S(stream);
    int* ev_of = static_cast<int*>(malloc(sizeof(int) * static_cast<size_t>(n_ops > 0 ? n_ops : 1)));
    if (!ev_of) return uz::fail("run_tape_lanes: out of host memory");
    bool forked[UZ_MAX_LANES] = {};
    int rc = 0, next_ev = 0;
    auto ok = [&](hipError_t e, const char* what) {
        if (e != hipSuccess && rc == 0) rc = uz::fail("run_tape_lanes: %s: %s", what, hipGetErrorString(e));
        return e == hipSuccess;
    };
    if (n_lanes > 1) ok(hipEventRecord(g_lp.fork, main), "fork record");
    if (g_lt.on) { (void)hipEventRecord(g_lt.base, main); g_lt.n = 0; }
    for (int k = 0; k < n_ops && rc == 0; ++k) {
        const int lane = sched[k].lane;
        hipStream_t s = lane == 0 ? main : g_lp.lanes[lane];
        if (lane != 0 && !forked[lane]) { forked[lane] = true; if (!ok(hipStreamWaitEvent(s, g_lp.fork, 0), "fork wait")) break; }
        for (int w = 0; w < sched[k].n_wait; ++w)
            if (!ok(hipStreamWaitEvent(s, g_lp.ev[ev_of[sched[k].wait[w]]], 0), "wait")) break;
        if (rc) break;
        if (run_one(ops[k], s) != 0) {
            char prev[600];
            strncpy(prev, uz::g_err, sizeof(prev) - 1);
            prev[sizeof(prev) - 1] = 0;
            rc = uz::fail("tape op %d (code %d): %s", k, ops[k].code, prev);
            break;
        }
        if (sched[k].signal) { ev_of[k] = next_ev; ok(hipEventRecord(g_lp.ev[next_ev++], s), "signal record"); }
        if (g_lt.on && k < g_lt.cap) { (void)hipEventRecord(g_lt.ev[k], s); g_lt.n = k + 1; }
    }
    for (int l = 1; l < n_lanes; ++l)               // join, also on errors: the lanes' work must not outlive the call's stream order
        if (forked[l] && hipEventRecord(g_lp.join[l], g_lp.lanes[l]) == hipSuccess) (void)hipStreamWaitEvent(main, g_lp.join[l], 0);
    free(ev_of);
    return rc;
}
// diagnostics: enable = 1 arms a timing event behind each of the first `capacity` ops of every following uz_run_tape_lanes call;
// enable = 0 with out != NULL synchronises and writes the end time (ms since the call's fork) of every op of the LAST call, returns their count
extern "C" int uz_lane_trace(int enable, int capacity, float* out, int n_out) {
    if (enable) {
        if (!g_lt.base && hipEventCreate(&g_lt.base) != hipSuccess) return uz::fail("lane_trace: cannot create event");
        if (capacity > g_lt.cap) {
            hipEvent_t* ne = static_cast<hipEvent_t*>(realloc(g_lt.ev, sizeof(hipEvent_t) * static_cast<size_t>(capacity)));
            if (!ne) return uz::fail("lane_trace: out of host memory");
            g_lt.ev = ne;
            for (int k = g_lt.cap; k < capacity; ++k) if (hipEventCreate(&g_lt.ev[k]) != hipSuccess) { g_lt.cap = k; return uz::fail("lane_trace: cannot create event"); }
            g_lt.cap = capacity;
        }
        g_lt.on = true;
        return 0;
    }
    g_lt.on = false;
    if (!out) return 0;
    if (hipDeviceSynchronize() != hipSuccess) return uz::fail("lane_trace: synchronize failed");
    const int n = g_lt.n < n_out ? g_lt.n : n_out;
    for (int k = 0; k < n; ++k) if (hipEventElapsedTime(&out[k], g_lt.base, g_lt.ev[k]) != hipSuccess) out[k] = -1.f;
    return n;
}
extern "C" int uz_graph_launch(void* graph_exec, void* stream) {
    if (hipGraphLaunch(static_cast<hipGraphExec_t>(graph_exec), uz::S(stream)) != hipSuccess) return uz::fail("graph_launch failed");
    return 0;
}
extern "C" void uz_graph_destroy(void* graph_exec) {
    if (graph_exec) (void)hipGraphExecDestroy(static_cast<hipGraphExec_t>(graph_exec));
}
