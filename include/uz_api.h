/*
 * uz_api.h - C ABI of libuz_hip.so: the MI355X (gfx950) implementation of the
 * U-Net / PHiSeg / Probabilistic-U-Net forward+backward hot path of
 * gigantenbein/UNet-Zoo.
 *
 * The reference has NO native / FFI boundary of its own: the path
 * sits behind Python nn.Module classes that dispatch ATen ops.  Each entry point
 * below therefore replaces one ATen call site of the reference; the call site
 * it replaces is cited (file:line relative to the reference repository).
 *
 * Conventions
 *  - plain C, no torch types: device pointers + sizes.  Tensors are fp32 NCHW.
 *    A tensor argument is a *channel-slice view*: `ptr` points at channel 0 of
 *    the view, `C` is the number of channels of the view and `Ctot` the channel
 *    count of the underlying buffer (batch stride = Ctot*H*W floats).  Writing a
 *    producer's output at a channel offset of a wider buffer is how torch.cat
 *    (phiseg.py:71,315; unet.py:72) is eliminated.
 *  - every call is asynchronous on the given hipStream_t (passed as void*); no
 *    call synchronises the device or allocates device memory.
 *  - return 0 on success, <0 on error; uz_last_error() gives the message
 *    (thread local, valid until the next call on that thread).
 *  - `accumulate != 0` means dst += result (gradient fan-in), else dst = result.
 */
#ifndef UZ_API_H
#define UZ_API_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define UZ_VERSION 100

int         uz_version(void);
const char* uz_last_error(void);
/* number of MI355X compute units / device name of the current device (diagnostics) */
int         uz_device_info(int* n_cu, char* name, int name_cap);

/* ---------------------------------------------------------------- convolution
 * nn.Conv2d(k=3, stride 1, pad 1) / nn.Conv2d(k=1): torchlayers.py:18, unet.py:25-29,
 * phiseg.py:95-96,281-284, probabilistic_unet.py:95,156-163.  w is the PyTorch
 * parameter itself, layout [Cout][Cin][ks][ks]; bias may be NULL.
 * fp32 in, fp32 out, fp32 accumulation everywhere.  Two implicit-GEMM kernel families:
 *  - fp32 MFMA (v_mfma_f32_32x32x2_f32) for every shape, including the 1..3-channel image /
 *    latent inputs (channel tiles are zero padded in LDS); 1x1 heads with <= 8 outputs run on
 *    streaming VALU kernels instead;
 *  - for the large 3x3 layers, the fp16 matrix pipe with split operands (each fp32 value, scaled by a power of two,
 *    = two fp16 pieces of 11 significand bits; three piece products, fp32 accumulate: split_f16.h, conv_split.hip,
 *    conv_wgrad_split.hip) - 22-bit operands (per layer within 2x the error of the fp32-MFMA kernels against fp64 (measured 0.5 - 0.8x on the heaviest layer); end to end at batch 32 the gradients' median error against fp64 is 1.6x the fp32 reference's own, logits within 1e-4, argmax bit-equal: tests/test_full_configs_gpu.py, tests/test_phiseg_gpu.py).  Environment switch, read once per process:
 *    UZ_CONV_MATH=f32 (fp32 MFMA only) | split (split path on every eligible 3x3 shape) | unset.
 *    The forward / data-gradient split path keeps its packed weight image in `workspace`.
 * Deep, low-resolution levels (2x2 .. 16x16) cannot fill 256 CUs with output tiles alone: with a
 * workspace of uz_conv_workspace() bytes the input-channel loop is split over workgroups and summed
 * in a fixed order (bitwise reproducible); workspace may be NULL (no split, slower, same result up
 * to summation order).                                                                          */
/* run-time override of UZ_CONV_MATH for tests / diagnostics: -1 = environment, 0 = f32, 1 = default policy (fp32-accurate fp16
 * split where it pays), 2 = split on every eligible shape, 3 = bf16: the layers of mode 1 with one bf16 piece per operand and one
 * MFMA product, fp32 accumulation (bf16 arithmetic, fp32 storage; BASELINE configs[4] is quoted in bf16).  Workspace sizes depend
 * on the mode: query them after switching.  Not thread safe.                                                            */
int uz_set_conv_math(int mode);
int uz_get_conv_math(void);
/* kernel family a call takes under the current mode: kind 0 fwd / 1 bwd_data / 2 bwd_weight -> 0 fp32 MFMA, 1 split-fp16 MFMA,
 * 2 streaming VALU (1x1 heads with <= 8 outputs).  Used by bench.py to price each op against the roof it runs on.          */
int uz_conv_route(int kind, int Cin, int Cout, int N, int H, int W, int ks);
size_t uz_conv_workspace(int Cin, int Cout, int N, int H, int W, int ks);   /* covers fwd and bwd_data */
/* Magnitude bounds (`*_amax`, all nullable): device scalars holding an UPPER BOUND of max|tensor| (any bound within ~2^10 of
 * the true maximum keeps full accuracy).  The split-fp16 kernels scale their operands by a power of two derived from the bound
 * before splitting them into fp16 pieces; NULL makes the call measure the tensor itself (one extra pass - the stand-alone
 * path; the model plans pass bounds that the producing kernels maintain: y_amax / a_amax / dy_amax outputs below are
 * atomic-max accumulated into a slot the caller zeroes once per pass).  The fp32-MFMA kernels ignore the input bounds.   */
int uz_conv_fwd(const float* x, int Cin, int CinTot,
                const float* w, const float* bias,
                float* y, int Cout, int CoutTot,
                int N, int H, int W, int ks, int relu,
                const float* x_amax, const float* w_amax, float* y_amax,
                void* workspace, size_t workspace_bytes, void* stream);
/* Weight images of the split path packed ONCE per step for a whole tape (instead of inside every call): the caller keeps a
 * buffer of uz_conv_packed_bytes() per (layer, direction), describes them in a device table of 8 int64 per layer
 * {w address, image address, Mc, Kc, wCi, uz_conv_pack_cot(), dgrad, first row} with rows counted by uz_conv_pack_rows()
 * (Mc / Kc = output / contraction channels of the direction: forward Cout / Cin, data gradient Cin / Cout; wCi = Cin), runs
 * uz_conv_pack_weights once (w_amax = the bound every image is scaled with) and hands each image to the *_packed calls,
 * which are otherwise uz_conv_fwd / uz_conv_bwd_data (packed_w may be NULL: same behaviour as those).                      */
size_t uz_conv_packed_bytes(int Cin, int Cout, int W, int dgrad);
int uz_conv_pack_rows(int Cin, int Cout, int W, int dgrad);
int uz_conv_pack_cot(int Cin, int Cout, int W, int dgrad);
int uz_conv_pack_weights(const int64_t* table, int n_layers, int total_rows, const float* w_amax, void* stream);
int uz_conv_fwd_packed(const float* x, int Cin, int CinTot,
                       const float* w, const float* bias,
                       float* y, int Cout, int CoutTot,
                       int N, int H, int W, int ks, int relu,
                       const float* x_amax, const float* w_amax, float* y_amax,
                       void* workspace, size_t workspace_bytes, const void* packed_w, void* stream);
/* Forward convolution that also reduces the BatchNorm statistics of its output while the tile is in registers
 * (torchlayers.py:18-21: every Conv2d of a Conv2D unit feeds a BatchNorm2d): bn_partials (nullable) receives
 * uz_conv_bn_partials(...) x Cout x 4 floats for uz_bn_relu_fwd_pre.  uz_conv_bn_partials returns 0 for shapes whose kernel
 * cannot do this (off the split-fp16 path, split-K); relu must be 0 with bn_partials.                                     */
int uz_conv_bn_partials(int Cin, int Cout, int N, int H, int W, int ks);
int uz_conv_fwd_bnstats(const float* x, int Cin, int CinTot, const float* w, const float* bias,
                        float* y, int Cout, int CoutTot, int N, int H, int W, int ks, int relu,
                        const float* x_amax, const float* w_amax, float* y_amax,
                        void* workspace, size_t workspace_bytes, const void* packed_w, float* bn_partials, void* stream);
int uz_conv_bwd_data_packed(const float* dy, int Cout, int CoutTot,
                            const float* w,
                            float* dx, int Cin, int CinTot,
                            int N, int H, int W, int ks, int accumulate,
                            const float* dy_amax, const float* w_amax,
                            void* workspace, size_t workspace_bytes, const void* packed_w, void* stream);
/* autograd of the above w.r.t. its input (aten::convolution_backward, input part):
 * dx[b,ci] (+)= sum_co sum_tap dy[b,co,.] * w[co,ci,flip(tap)]                 */
int uz_conv_bwd_data(const float* dy, int Cout, int CoutTot,
                     const float* w,
                     float* dx, int Cin, int CinTot,
                     int N, int H, int W, int ks, int accumulate,
                     const float* dy_amax, const float* w_amax,
                     void* workspace, size_t workspace_bytes, void* stream);
/* Data gradient with the ReLU backward of the unit that produced A folded in (vanilla U-Net blocks, unet.py:25-30: Conv -> ReLU ->
 * Conv): dx = (a > 0) ? conv_T(dy, w) (+ dx) : 0 - i.e. dx leaves as the gradient w.r.t. that unit's convolution output, no
 * separate uz_relu_bwd pass.  partials: uz_conv_bwd_relu_partials() x Cin x 4 floats whose .x components uz_chan_sum_partials adds
 * up to the unit's bias gradient; dx_amax: bound slot of dx.  uz_conv_bwd_relu_partials() == 0: shape not supported (off the
 * split path or split-K) - use uz_conv_bwd_data + uz_relu_bwd. */
int uz_conv_bwd_relu_partials(int Cin, int Cout, int N, int H, int W, int ks);
int uz_conv_bwd_data_relu(const float* dy, int Cout, int CoutTot, const float* w, float* dx, int Cin, int CinTot,
                          int N, int H, int W, int ks, int accumulate, const float* dy_amax, const float* w_amax,
                          void* workspace, size_t workspace_bytes, const void* packed_w,
                          const float* a, int aCtot, float* partials, float* dx_amax, void* stream);
int uz_chan_sum_partials(const float* partials, int n_rows, int C, float* out, void* stream);
/* The same fold for the OTHER last writers of a unit's dA: the backward of the pooling / interpolation that consumed A (the third
 * unit of every U-Net block).  partials: uz_resample_bwd_relu_rows(kind, C, N, H, W) x C doubles (kind 0 = avgpool2 with H x W the
 * high-resolution plane, 1 = bilinear2x with H x W the low-resolution plane), summed by uz_chan_sum_partials_d. */
int uz_resample_bwd_relu_rows(int kind, int C, int N, int H, int W);
int uz_avgpool2_bwd_relu(const float* dy, int C, int CtotDy, float* dx, int CtotDx, int N, int H, int W, int accumulate,
                         const float* a, int CtotA, double* partials, float* dx_amax, void* stream);
int uz_bilinear2x_bwd_relu(const float* dy, int C, int CtotDy, float* dx, int CtotDx, int N, int H, int W, int align_corners, int accumulate,
                           const float* a, int CtotA, double* partials, float* dx_amax, void* stream);
int uz_chan_sum_partials_d(const double* partials, int n_rows, int C, float* out, void* stream);
/* every bias gradient of a tape in one launch: table = n_entries rows of 5 int64 {partials pointer, out pointer, n_rows, C, partials are double} */
int uz_chan_sum_table(const int64_t* table, int n_entries, int max_channels, void* stream);
/* autograd w.r.t. the weight: dw[co,ci,tap] = sum_{b,y,x} dy * x_shifted.
 * Deterministic split-K: partial slabs in `workspace` (uz_conv_bwd_weight_workspace
 * bytes), then an ordered reduction.  db (nullable) = sum_{b,y,x} dy.
 * Depth window (Cin == 3 * CinTot, ks 3: a Conv3d run over D slices, see the volume section): dw is written in the Conv3d
 * parameter layout [Cout][CinTot][3][3][3] (contraction channel k = kd * CinTot + ci), no separate permutation needed.   */
size_t uz_conv_bwd_weight_workspace(int Cin, int Cout, int N, int H, int W, int ks);
int uz_conv_bwd_weight(const float* x, int Cin, int CinTot,
                       const float* dy, int Cout, int CoutTot,
                       float* dw, float* db,
                       int N, int H, int W, int ks,
                       const float* x_amax, const float* dy_amax,
                       void* workspace, size_t workspace_bytes, void* stream);

/* ---------------------------------------------------------------- BatchNorm2d(eps, momentum) + ReLU
 * torchlayers.py:20-21 (nn.BatchNorm2d(eps=1e-3, momentum=0.01) -> nn.ReLU).
 * training != 0: batch statistics (biased var) normalise, running stats get the
 * momentum update with the unbiased var; save_mean_rstd[2*C] keeps (mean, rstd) for
 * backward.  training == 0: running stats normalise.  workspace >= uz_bn_workspace(). */
size_t uz_bn_workspace(int C, int N, int H, int W);
int uz_bn_relu_fwd(const float* y, int C, int CtotY,
                   const float* gamma, const float* beta,
                   float* running_mean, float* running_var,
                   float* save_mean_rstd,
                   float* a, int CtotA,
                   int N, int H, int W, float eps, float momentum, int training, int relu,
                   float* a_amax, void* workspace, void* stream);
/* The same with the batch statistics taken from the partials the producing convolution wrote in its epilogue
 * (uz_conv_fwd_bnstats): conv_partials = [n_partials][C][4] floats {sum, sum of squares, max, max of the negated}; NULL / 0 =
 * uz_bn_relu_fwd.  Training mode, N*H*W > 4096 only.                                                                     */
int uz_bn_relu_fwd_pre(const float* y, int C, int CtotY, const float* gamma, const float* beta,
                       float* running_mean, float* running_var, float* save_mean_rstd,
                       float* a, int CtotA, int N, int H, int W, float eps, float momentum,
                       int training, int relu, float* a_amax, void* workspace,
                       const float* conv_partials, int n_partials, void* stream);
/* Small planes (N*H*W <= 4096: the 8x8 ... 2x2 levels), where a Conv2D unit (torchlayers.py:7-29) is a chain of short launches:
 * the convolution's split-K reduce folded into the BatchNorm.  uz_conv_splitk_parts() = number of partial-sum slabs the fp32 forward
 * kernel produces for a shape when given its workspace (1: unsplit or another kernel family - not foldable); uz_conv_fwd_slabs runs
 * that kernel only and leaves [parts][N][Cout][H*W] floats at the start of `workspace` (>= uz_conv_workspace()); uz_bn_relu_fwd_slabs
 * then forms y = conv_bias + slabs (slab order: bit-identical to uz_conv_fwd), WRITES y (the backward pass reads it) and does what
 * uz_bn_relu_fwd does.                                                                                                    */
int uz_conv_splitk_parts(int Cin, int Cout, int N, int H, int W, int ks);
int uz_conv_fwd_slabs(const float* x, int Cin, int CinTot, const float* w, int Cout, int N, int H, int W, int ks,
                      void* workspace, size_t workspace_bytes, void* stream);
int uz_conv_bwd_splitk_parts(int Cin, int Cout, int N, int H, int W, int ks);      /* the data gradient's split count (1: not split, or another kernel family) */
int uz_conv_bwd_data_slabs(const float* dy, int Cout, int CoutTot, const float* w, int Cin, int N, int H, int W, int ks,
                           float* slabs_out, void* stream);                        /* [parts][N][Cin][H*W] partial sums; summed by uz_bn_relu_bwd_ex(da_slabs) */
int uz_bn_relu_fwd_slabs(const float* slabs, int n_slabs, const float* conv_bias, float* y, int C, int CtotY,
                         const float* gamma, const float* beta, float* running_mean, float* running_var, float* save_mean_rstd,
                         float* a, int CtotA, int N, int H, int W, float eps, float momentum,
                         int training, int relu, float* a_amax, void* stream);
/* native_batch_norm_backward + threshold_backward: da -> dy, dgamma, dbeta, and the
 * conv-bias gradient dbias = sum dy (nullable).  dy may alias da.               */
int uz_bn_relu_bwd(const float* da, int CtotDa,
                   const float* y, int C, int CtotY,
                   const float* gamma, const float* beta, const float* save_mean_rstd,
                   float* dy, int CtotDy,
                   float* dgamma, float* dbeta, float* dbias,
                   int N, int H, int W, int relu,
                   float* dy_amax, void* workspace, void* stream);
/* ReLU-only units of the vanilla U-Net (unet.py:26,28,30): da * [a > 0] (threshold_backward)
 * fused with the conv-bias gradient.                                              */
int uz_relu_bwd(const float* da, int CtotDa, const float* a, int C, int CtotA,
                float* dy, int CtotDy, float* dbias,
                int N, int H, int W, float* dy_amax, void* workspace, void* stream);

/* ---------------------------------------------------------------- split storage + folded BatchNorm backward (round 4)
 * Conv2D unit = Conv2d -> BatchNorm2d -> ReLU (torchlayers.py:18-21), 106 of them per PHiSeg step.  Two things the producer of
 * a tensor can do for the matrix kernels that consume it:
 *
 * Split storage.  A tensor whose bound is known BEFORE its first element is written (BatchNorm forward: exact output range from
 * the statistics; BatchNorm backward: analytic bound; pooling / interpolation: the input's bound) may be stored as one 32-bit
 * word per element holding the two fp16 pieces of v * s (low half h1 = fp16(v s), high half h2 = fp16(v s - h1), s = the power of
 * two the split kernels derive from the bound slot) - the operand pieces the consuming convolution would otherwise form from the
 * fp32 value in its staging (scale, clamp, two conversions, subtract, convert per element).  Same bytes, same values
 * (h1 + h2 = v s to 2^-22); the `out_packed` producers below write it, the `*_packed` flags of the convolution calls read it.
 * A concat buffer with two producers carries two bounds: channels [0, seg_channels) were scaled from x_amax, the rest from
 * x_amax2 (seg_channels a multiple of 16; 0 = one bound).  uz_pack_split / uz_unpack_split convert whole tensors (tests, tools).
 *
 * Folded backward reduction.  The data gradient that writes dA of a unit LAST can apply the unit's ReLU mask (alpha y + beta' > 0)
 * and leave the unit's BatchNorm-backward sums {sum dz, sum dz x_hat, max |dz|, max |x_hat|} per (tile, channel) in its epilogue
 * (bn_y = the unit's pre-normalisation output, bn_save = the 4 C floats uz_bn_relu_fwd_ex saved); uz_bn_relu_bwd_ex then runs a
 * one-workgroup-per-channel finalise instead of the reduction pass over dA and y.  uz_conv_bwd_relu_partials() gives the number of
 * partial rows (0: shape not supported).                                                                                       */
int uz_pack_split(const float* x, float* packed, size_t n, const float* amax, void* stream);
int uz_unpack_split(const float* packed, float* x, size_t n, const float* amax, void* stream);
int uz_bn_relu_fwd_ex(const float* y, int C, int CtotY, const float* gamma, const float* beta,
                      float* running_mean, float* running_var, float* save_mean_rstd_ab,
                      float* a, int CtotA, int N, int H, int W, float eps, float momentum,
                      int training, int relu, float* a_amax, void* workspace,
                      const float* conv_partials, int n_partials, int out_packed, void* stream);
int uz_bn_relu_bwd_ex(const float* da, int CtotDa, const float* y, int C, int CtotY,
                      const float* gamma, const float* beta, const float* save_mean_rstd,
                      float* dy, int CtotDy, float* dgamma, float* dbeta, float* dbias,
                      int N, int H, int W, int relu, float* dy_amax, void* workspace,
                      const float* conv_partials, int n_partials, int out_packed, double* dbias_partials,
                      const float* da_slabs, int n_da_slabs, void* stream);   /* da_slabs: dA = the sum of these [n][N][C][H*W] slabs (small planes, uz_conv_bwd_data_slabs); da is then not read */
/* uz_bn_relu_fwd_ex as two calls (large planes, training, statistics from the convolution's partials): phase 1 = statistics only (table of 4 C floats,
 * running buffers, the activation's bound; y / a untouched and nullable), phase 2 = the apply pass alone from what phase 1 left.  A following
 * uz_conv_fwd_bn_ex depends on phase 1 only: the apply pass then runs beside that convolution instead of in front of it. */
int uz_bn_relu_fwd_phase(const float* y, int C, int CtotY, const float* gamma, const float* beta,
                         float* running_mean, float* running_var, float* save_mean_rstd_ab,
                         float* a, int CtotA, int N, int H, int W, float eps, float momentum,
                         int relu, float* a_amax, const float* conv_partials, int n_partials, int out_packed, int phase, void* stream);
int uz_bn_fwd_fused_limit(int H, int W);         /* N*H*W up to which the training-mode uz_bn_relu_fwd(_ex) WITHOUT conv_partials is one launch (statistics + apply from registers) */
int uz_bn_bwd_fused_limit(int H, int W);         /* N*H*W up to which uz_bn_relu_bwd(_ex) is one launch with the channel's batch on chip: no out_packed / dbias_partials there */
int uz_bn_bwd_dbias_rows(int N, int H, int W);   /* rows of dbias_partials ([rows][C] doubles, summed by uz_chan_sum_table); 0: small-plane path */
int uz_avgpool2_fwd_ex(const float* x, int C, int CtotX, float* y, int CtotY, int N, int H, int W,
                       const float* x_amax, float* y_amax, int out_packed, void* stream);
int uz_bilinear2x_fwd_ex(const float* x, int C, int CtotX, float* y, int CtotY, int N, int H, int W, int align_corners,
                         const float* x_amax, float* y_amax, int out_packed, void* stream);
int uz_conv_fwd_ex(const float* x, int Cin, int CinTot, const float* w, const float* bias,
                   float* y, int Cout, int CoutTot, int N, int H, int W, int ks, int relu,
                   const float* x_amax, const float* w_amax, float* y_amax,
                   void* workspace, size_t workspace_bytes, const void* packed_w, float* bn_partials,
                   int x_packed, const float* x_amax2, int seg_channels, void* stream);
/* uz_conv_fwd_ex for a layer that follows a Conv -> BatchNorm -> ReLU unit (reference torchlayers.py:18-21) and reads that unit's
 * PRE-normalisation output y_prev with its statistics table bn_save ([4][Cin]: mean, rstd, alpha, beta' - uz_bn_relu_fwd_ex): the staging
 * applies a = max(alpha y_prev + beta', 0) (bn_relu) itself and splits with the scale of a_amax, the bound of the APPLIED activation.
 * Same values as reading the unit's stored activation; the unit's apply pass leaves the chain of dependent launches.  Split path only. */
int uz_conv_fwd_bn_ex(const float* y_prev, int Cin, int CinTot, const float* bn_save, int bn_relu,
                      const float* w, const float* bias, float* y, int Cout, int CoutTot, int N, int H, int W, int ks,
                      const float* a_amax, const float* w_amax, float* y_amax,
                      void* workspace, size_t workspace_bytes, const void* packed_w, float* bn_partials, void* stream);
int uz_conv_bwd_data_ex(const float* dy, int Cout, int CoutTot, const float* w, float* dx, int Cin, int CinTot,
                        int N, int H, int W, int ks, int accumulate, const float* dy_amax, const float* w_amax,
                        void* workspace, size_t workspace_bytes, const void* packed_w, int dy_packed,
                        const float* bn_y, int bn_yCtot, const float* bn_save, int bn_relu, float* bn_partials, void* stream);
int uz_conv_bwd_weight_ex(const float* x, int Cin, int CinTot, const float* dy, int Cout, int CoutTot,
                          float* dw, float* db, int N, int H, int W, int ks,
                          const float* x_amax, const float* dy_amax, void* workspace, size_t workspace_bytes,
                          int x_packed, const float* x_amax2, int seg_channels, int dy_packed, float* slabs_out, void* stream);
/* Weight-gradient slab reductions of many layers in ONE launch.  slabs_out (above, nullable): the call leaves its
 * uz_conv_bwd_weight_slabs() x ks*ks x Cout x Cin partial sums there instead of reducing them into dw; uz_wgrad_reduce_table adds
 * them later - table rows of 8 int64 {slabs, dw, S, Cout, Cin, ks*ks, first block, volC}, blocks counted by uz_wgrad_reduce_blocks();
 * volC = 0, or - the call was a depth window (Cin = 3 volC view channels over a volC-channel buffer) - the window's C: the sum
 * then leaves in the Conv3d parameter layout [Cout][volC][3][3][3] exactly as the call's own reduction writes it.
 * Same order of additions as the in-call reduction (bitwise the same dw).                                                        */
int uz_conv_bwd_weight_slabs(int Cin, int Cout, int N, int H, int W, int ks);
int uz_wgrad_reduce_blocks(int Cin, int Cout, int ks);
int uz_wgrad_reduce_table(const int64_t* table, int n_layers, int total_blocks, void* stream);

/* ---------------------------------------------------------------- resampling
 * nn.AvgPool2d(2, 2, padding=0, ceil_mode=True): phiseg.py:23, unet.py:22,
 * probabilistic_unet.py:56.  Ho = ceil(H/2); partial windows divide by the
 * in-bounds element count.                                                        */
int uz_avgpool2_fwd(const float* x, int C, int CtotX, float* y, int CtotY,
                    int N, int H, int W, const float* x_amax, float* y_amax, void* stream);   /* |y| <= bound of |x| is forwarded */
int uz_avgpool2_bwd(const float* dy, int C, int CtotDy, float* dx, int CtotDx,
                    int N, int H, int W, int accumulate, void* stream);
/* F.interpolate(mode='bilinear', scale_factor=2, align_corners=ac): phiseg.py:66,213-216,
 * 305-309 (ac=1), unet.py:67 (ac=0).  (H, W) are the INPUT sizes.                 */
int uz_bilinear2x_fwd(const float* x, int C, int CtotX, float* y, int CtotY,
                      int N, int H, int W, int align_corners, const float* x_amax, float* y_amax, void* stream);
int uz_bilinear2x_bwd(const float* dy, int C, int CtotDy, float* dx, int CtotDx,
                      int N, int H, int W, int align_corners, int accumulate, void* stream);
/* F.interpolate(size=[Ho,Wo], mode='nearest'), integer factor: phiseg.py:321        */
int uz_nearest_fwd(const float* x, int C, int CtotX, float* y, int CtotY,
                   int N, int H, int W, int factor, void* stream);
int uz_nearest_bwd(const float* dy, int C, int CtotDy, float* dx, int CtotDx,
                   int N, int H, int W, int factor, int accumulate, void* stream);
/* torch.mean over H then W (probabilistic_unet.py:114-115): (N,C,H,W) -> (N,C)       */
int uz_spatial_mean_fwd(const float* x, int C, int CtotX, float* y, int N, int H, int W, void* stream);
int uz_spatial_mean_bwd(const float* dy, int C, float* dx, int CtotDx, int N, int H, int W,
                        int accumulate, void* stream);

/* ---------------------------------------------------------------- latent heads / losses
 * mask (N,1,H,W) float {0,1,..} -> cat([patch, onehot(mask)-0.5], 1): utils.py:289-311,
 * phiseg.py:178-183, probabilistic_unet.py:103-109.  out has in_ch + nlabels channels. */
int uz_posterior_input(const float* patch, int in_ch, const float* mask, int nlabels,
                       float* out, int N, int H, int W, void* stream);
/* Reparameterised latent sample.  act = 0: SampleZBlock tail (phiseg.py:100-105), sigma =
 * softplus(pre_sigma) (beta 1, threshold 20); act = 1: AxisAlignedConvGaussian + rsample
 * (probabilistic_unet.py:124-129,350), sigma = exp(pre_sigma).  z = mu + sigma * eps (z nullable:
 * the prior's own draw is discarded in training, phiseg.py:200-202).  n = number of elements. */
int uz_latent_sample_fwd(const float* mu, const float* pre_sigma, const float* eps,
                         float* sigma, float* z, size_t n, int act, void* stream);
/* dmu_pre = dmu + dz ; dpre_sigma = (dsigma + dz*eps) * dsigma/dpre, with dsigma/dpre =
 * 1 - exp(-sigma) (softplus) or sigma (exp); dmu / dsigma / dz are nullable (treated as 0).  */
int uz_latent_sample_bwd(const float* dmu, const float* dsigma, const float* dz,
                         const float* eps, const float* sigma,
                         float* dmu_pre, float* dpre_sigma, size_t n, int act, void* stream);
/* The two heads of a SampleZBlock and its sampling tail as ONE launch (phiseg.py:95-105): mu = mu_conv(h), pre_sigma =
 * sigma_conv(h) (1x1 convolutions, L latent channels each, weights [L][Cin], biases nullable), sigma = softplus(pre_sigma)
 * (act = 0; act = 1: exp), z = mu + sigma * eps (z nullable).  h is a view of Cin channels in a buffer of CinTot; mu,
 * pre_sigma, sigma, z, eps are contiguous [N][L][H][W].  Bit-identical to uz_conv_fwd x 2 + uz_latent_sample_fwd; h is read
 * once.  1 <= L <= 4, Cin <= 512 (uz_latent_heads_ok). */
int uz_latent_heads_ok(int Cin, int L);
int uz_latent_heads_fwd(const float* h, int Cin, int CinTot, const float* w_mu, const float* b_mu,
                        const float* w_sigma, const float* b_sigma, const float* eps, float* mu, float* pre_sigma,
                        float* sigma, float* z, int L, int N, int H, int W, int act, void* stream);
/* Backward of the two heads: dh (+)= w_a^T dy_a + w_b^T dy_b in one pass (head a's rows are added first: pass the head whose
 * separate data gradient would have run first), and both heads' weight / bias gradients from one read of h (fp64 ordered
 * partial sums in `workspace`, uz_latent_heads_bwd_weight_workspace bytes; db_* nullable). */
int uz_latent_heads_bwd_data(const float* dy_a, const float* dy_b, int L, const float* w_a, const float* w_b,
                             float* dh, int Cin, int CinTot, int N, int H, int W, int accumulate, void* stream);
size_t uz_latent_heads_bwd_weight_workspace(int Cin, int L, int N, int H, int W);
int uz_latent_heads_bwd_weight(const float* h, int Cin, int CinTot, const float* dy_a, const float* dy_b, int L,
                               float* dw_a, float* db_a, float* dw_b, float* db_b, int N, int H, int W,
                               void* workspace, size_t workspace_bytes, void* stream);
/* KL_two_gauss_with_diag_cov with the reference's sigma1*sigma0 quirk (phiseg.py:436-453),
 * times `weight` (4**level, phiseg.py:463).  Tensors are (N, per_sample) contiguous.
 * fwd writes loss_out[0] (single block, ordered); bwd writes the four gradients * scale. */
int uz_kl_fwd(const float* mu0, const float* s0, const float* mu1, const float* s1,
              int N, int per_sample, float weight, float* loss_out, void* stream);
/* uz_kl_fwd with a scratch of >= 512 bytes (nullable): tensors beyond 128 k elements (a volume's full-resolution latent level is ONE
 * sample of 2 M elements) are summed by up to 64 workgroups + an ordered final pass instead of one workgroup. */
int uz_kl_fwd_ws(const float* mu0, const float* s0, const float* mu1, const float* s1, int N, int per_sample, float weight,
                 float* loss_out, void* workspace, void* stream);
int uz_kl_bwd(const float* mu0, const float* s0, const float* mu1, const float* s1,
              int N, int per_sample, float weight, const float* loss_scale,
              float* dmu0, float* ds0, float* dmu1, float* ds1, void* stream);
/* residual_multinoulli_loss (phiseg.py:481-513), 2..8 classes, L levels of (N,K,H,W) logits
 * given as an array of L device pointers held in DEVICE memory (s_ptrs).  level l loss =
 * mean_b sum_pix CE(sum_{j>=l} s_j, mask).  loss_out[l], l=0..L-1.                      */
size_t uz_ce_workspace(int N, int H, int W, int L);
int uz_residual_ce_fwd(const float* const* s_ptrs, int L, int K, const float* mask,
                       int N, int H, int W, float* loss_out, void* workspace, void* stream);
int uz_residual_ce_bwd(const float* const* s_ptrs, float* const* ds_ptrs, int L, int K, const float* mask,
                       int N, int H, int W, const float* loss_scale, void* stream);
/* nn.CrossEntropyLoss() mean over all pixels (unet.py:159-165) = residual CE with L=1 scaled 1/(H*W) */
/* loss_terms[0..n-1] -> total[0] = sum (ordered)                                         */
int uz_sum_terms(const float* terms, int n, float* total, void* stream);
/* accumulate_output(use_softmax) + argmax (phiseg.py:428-434, train_model.py:186,195):
 * acc = sum_l s_l ; soft = softmax_c(acc) ; label = argmax_c.  soft / label nullable.     */
int uz_accumulate_softmax_argmax(const float* const* s_ptrs, int L, int K, int N, int H, int W,
                                 float* acc, float* soft, uint8_t* label, void* stream);

/* ---------------------------------------------------------------- validation metrics (train_model.py:186-230)
 * out[i][j][0..2] = |a_i==label & b_j==label|, |a_i==label|, |b_j==label| over HW pixels (int32, exact): the integer core of
 * utils.generalised_energy_distance (utils.py:148-200; IoU = medpy.metric.jc, MedPy 0.4.0) and of the per-label Dice
 * (train_model.py:212-224; medpy.metric.dc).  a: (Na, HW) uint8 label maps, b: (Nb, HW).                               */
int uz_label_pair_counts(const uint8_t* a, int Na, const uint8_t* b, int Nb, int HW, int label, int32_t* out, void* stream);
/* utils.variance_ncc_dist (utils.py:202-247): pixel-wise cross-entropy maps E_ss (HW) and E_sy (M, HW) of N softmax samples
 * (N,K,HW) against M one-hot ground truths (M,K,HW); then ncc(E_ss, E_sy[j]) for every j (utils.py:130-145).              */
int uz_ncc_maps(const float* soft, const float* gt_onehot, int N, int M, int K, int HW, float* E_ss, float* E_sy, void* stream);
int uz_ncc(const float* a, const float* v, int M, int HW, float* out, void* stream);

/* ---------------------------------------------------------------- optimiser / vector ops
 * torch.optim.Adam(lr, betas, eps, weight_decay as L2 added to the gradient): train_model.py:49,122.
 * One launch over a contiguous range of the flat parameter buffer.                        */
int uz_adam_step(float* params, const float* grads, float* exp_avg, float* exp_avg_sq, size_t n,
                 int64_t step, float lr, float beta1, float beta2, float eps, float weight_decay,
                 float grad_scale, void* stream);
int uz_axpy(float* y, const float* x, float alpha, size_t n, void* stream);     /* y += alpha*x */
/* Additive coupling of revtorch's ReversibleBlock (torchlayers.py:55-82: y1 = x1 + F(x2), y2 = x2 + G(y1), and its inversion
 * x2 = y2 - G(y1), x1 = y1 - F(x2) in the recomputing backward pass): y = (accumulate ? y : 0) + a + alpha * b on
 * channel-slice views of C channels (b nullable: plain strided copy / gradient fan-in).                                    */
int uz_add_views(const float* a, int CtotA, const float* b, int CtotB, float* y, int CtotY, int C, int N, int H, int W,
                 float alpha, int accumulate, const float* a_amax, const float* b_amax, float* y_amax, void* stream);
/* slot[0] = max(slot[0], max_i |x_i|) (slot holds a non-negative float; zero it first for a plain maximum) */
int uz_absmax(const float* x, size_t n, float* slot, void* stream);
/* dst bound slot = max(dst, value of src bound slot): forwards a magnitude bound through ops that cannot raise it (pooling, interpolation) */
int uz_absmax_copy(const float* src_slot, float* dst_slot, void* stream);
/* Latent noise eps ~ N(0, 1) (phiseg.py:104 torch.randn_like, probabilistic_unet.py:276 rsample) on the device: Philox4x32-10 +
 * Box-Muller keyed by rng_state = {seed, offset} (two uint64 in DEVICE memory; offset counts 4-float blocks).  uz_step_counters
 * runs behind it: counters[idx[k]] += 1 for the BatchNorm batch counters (num_batches_tracked) of the pass and
 * rng_state.offset += advance - so a replayed launch never repeats a draw.  (uint64_t parameters are passed as pointers / int64.) */
int uz_randn_fill(float* dst, size_t n, const void* rng_state, void* stream);
int uz_step_counters(int64_t* counters, const int64_t* idx, int n_idx, void* rng_state, int64_t advance, void* stream);
/* Device flag word of the split-fp16 convolution path.  The kernels scale every operand by a power of two taken from an upper
 * bound of its tensor's magnitudes; a bound that is too small by more than 4x (a stale parameter bound, a wrong bound passed to a
 * stand-alone call) would overflow fp16.  Such values are CLAMPED to the largest finite fp16 (the result is wrong but never inf /
 * NaN) and a bit is raised here: 1 = activation, 2 = weight, 4 = output gradient.  Synchronises the stream; clear != 0 resets. */
int uz_device_flags(int* out, int clear, void* stream);
/* diagnostics: when buf is non-null the split-fp16 convolution kernels write 8 int64 per workgroup (first 4096 workgroups):
 * shader-clock stamps at start / first tile staged / sum of the staging phases / main loop end / kernel end, the
 * 100 MHz real-time counter at start and end, and the workgroup's tile count (tools/stamp_conv.py).  NULL switches it off. */
void uz_debug_stamps(void* buf);
int uz_scale(float* y, float alpha, size_t n, void* stream);                    /* y *= alpha   */
int uz_zero_f32(float* p, size_t n, void* stream);                              /* p[0..n) = 0  (kernel launch, graph-capturable) */
int uz_copy_f32(float* dst, const float* src, size_t n, void* stream);          /* dst = src    (kernel launch, graph-capturable) */
/* sqrt(sum x^2) terms of utils.l2_regularisation (utils.py:93-101): one norm per tensor of a
 * table of (offset, count) pairs over the flat parameter buffer; out[i] = ||p_i||_2.       */
int uz_l2_norms(const float* flat, const int64_t* offs_counts, int n_tensors, float* out, void* stream);
int uz_l2_norms_bwd(const float* flat, const int64_t* offs_counts, int n_tensors, const float* norms,
                    const float* scale, float* grad_flat, void* stream);          /* g += scale * p / ||p|| */

/* ---------------------------------------------------------------- op tape
 * A forward or backward pass is a static list of the calls above ("tape"), built once by
 * the host for a (model, N, H, W) and replayed with ONE FFI call per pass - or captured
 * into a hipGraph.  uz_op mirrors the argument lists above positionally: p[] pointers,
 * i[] ints, f[] floats, in declaration order of the corresponding function.               */
enum {
  UZ_OP_CONV_FWD = 1, UZ_OP_CONV_BWD_DATA, UZ_OP_CONV_BWD_WEIGHT,
  UZ_OP_BN_RELU_FWD, UZ_OP_BN_RELU_BWD, UZ_OP_RELU_BWD,
  UZ_OP_AVGPOOL_FWD, UZ_OP_AVGPOOL_BWD, UZ_OP_BILINEAR_FWD, UZ_OP_BILINEAR_BWD,
  UZ_OP_NEAREST_FWD, UZ_OP_NEAREST_BWD, UZ_OP_SPATIAL_MEAN_FWD, UZ_OP_SPATIAL_MEAN_BWD,
  UZ_OP_POSTERIOR_INPUT, UZ_OP_LATENT_FWD, UZ_OP_LATENT_BWD, UZ_OP_KL_FWD, UZ_OP_KL_BWD,
  UZ_OP_CE_FWD, UZ_OP_CE_BWD, UZ_OP_SUM_TERMS, UZ_OP_ACC_SOFTMAX_ARGMAX,
  UZ_OP_ADAM, UZ_OP_AXPY, UZ_OP_SCALE, UZ_OP_L2_NORMS, UZ_OP_L2_NORMS_BWD,
  UZ_OP_MEMSET, UZ_OP_COPY, UZ_OP_BCAST_CHANNELS, UZ_OP_BCAST_CHANNELS_BWD,
  UZ_OP_ABSMAX,          /* p[0] = src, p[1] = slot, n = count */
  UZ_OP_ABSMAX_COPY,     /* p[0] = src slot, p[1] = dst slot */
  UZ_OP_W3D_PERMUTE,     /* p = src, dst; i = Cout, Cin, mode */
  UZ_OP_AVGPOOL3D_FWD, UZ_OP_AVGPOOL3D_BWD, UZ_OP_DEPTH_LERP_FWD, UZ_OP_DEPTH_LERP_BWD, UZ_OP_NEAREST3D_FWD, UZ_OP_NEAREST3D_BWD,
  UZ_OP_ADD_VIEWS,       /* p = a, b, y, a_amax, b_amax, y_amax; i = CtotA, CtotB, CtotY, C, N, H, W, accumulate; f[0] = alpha */
  UZ_OP_EVENT_RECORD,    /* p[0] = event (uz_event_create): marks "every earlier op this one depends on is done" */
  UZ_OP_PACK_WEIGHTS,    /* p[0] = table, p[1] = w_amax; i = n_layers, total_rows (uz_conv_pack_weights); CONV_FWD p[8] / CONV_BWD_DATA p[6] = image */
  UZ_OP_CHAN_SUM_TABLE,    /* p[0] = table (uz_chan_sum_table), p[1] = the gradient regions it writes; i = n_entries, max_channels */
  UZ_OP_WGRAD_REDUCE_TABLE, /* p[0] = table (uz_wgrad_reduce_table), p[1] = the gradient regions it writes; i = n_layers, total_blocks */
  UZ_OP_CHAN_SUM_PARTIALS, /* p = partials, out; i = n_rows, C (uz_chan_sum_partials); CONV_BWD_DATA p[7] = a (folded ReLU backward), p[8] = partials, p[9] = dx bound; i[9] = CtotA */
  UZ_OP_LATENT_HEADS_FWD,        /* p = h, w_mu, b_mu, w_sigma, b_sigma, eps, mu, pre, sigma, z; i = Cin, CinTot, L, N, H, W, act (uz_latent_heads_fwd) */
  UZ_OP_LATENT_HEADS_BWD_DATA,   /* p = dy_a, dy_b, w_a, w_b, dh; i = L, Cin, CinTot, N, H, W, accumulate */
  UZ_OP_LATENT_HEADS_BWD_WEIGHT, /* p = h, dy_a, dy_b, dw_a, db_a, dw_b, db_b, workspace; i = Cin, CinTot, L, N, H, W; n = workspace bytes */
  UZ_OP_CHAIN,           /* p = sub-op table, phase table, state; i = n_phases, n_workgroups, n_sub_ops (uz_chain_run); p[3..] = the buffers it touches (scheduling only) */
  UZ_OP_CHAIN_PACK,      /* p = table, w_amax, images; i = n_layers, total_blocks (uz_chain_pack_weights) */
  UZ_OP__COUNT
};
typedef struct uz_op {
  int32_t  code;
  int32_t  i[15];
  float    f[4];
  int64_t  n;          /* element / byte count for the vector ops */
  void*    p[12];
} uz_op;
int uz_run_tape(const uz_op* ops, int n_ops, void* stream);
/* What this binary is: writes a JSON object {variant, products_per_mac, exp_patch_dma, exp_pref_all, diag_skip_compiled, source_hash,
 * experiment} into out (cap bytes) and returns `experiment` (0 for a product build: three piece products per operand pair, no experiment or
 * diagnostic code compiled in).  bench.py prints it in config.build and refuses to report a line from an experiment build. */
int uz_build_info(char* out, int cap);
/* Capture the tape into a hipGraph on `stream` and return an opaque executable handle. */
int  uz_graph_create(const uz_op* ops, int n_ops, void* stream, void** graph_exec_out);
/* Lane capture: like uz_graph_create, but the captured graph carries the tape's dependency DAG
 * instead of one chain.  Ops are partitioned into lanes (each with its own scratch copy, resolved
 * by the host into the op's pointers); op k follows the previous op of its lane and the earlier
 * ops listed in wait[0..n_wait) (which must have signal != 0).  Independent chains of the tape
 * overlap on the device at replay; results are bit-identical to uz_run_tape.                   */
#define UZ_MAX_LANES 8
typedef struct uz_sched {
  int32_t lane;
  int32_t signal;
  int32_t n_wait;
  int32_t wait[UZ_MAX_LANES];
} uz_sched;
int  uz_graph_create_lanes(const uz_op* ops, const uz_sched* sched, int n_ops, int n_lanes,
                           void* stream, void** graph_exec_out);
/* Workgroups a split-path weight gradient (uz_conv_bwd_weight*, 3x3, >= 64 channels) is cut into: a process setting the host makes BEFORE it
 * sizes slab buffers with uz_conv_bwd_weight_slabs and keeps while it runs the calls sized with it (the Python face sets it per model in
 * front of every tape: PHISeg 128, ProbabilisticUnet 192, others 256 = one workgroup per CU).  No counterpart in the reference. */
void uz_set_wgrad_target(int workgroups);   /* a tape whose slab buffers were sized under another target refuses to run (UZ_OP_CONV_BWD_WEIGHT i[12]) */
int  uz_get_wgrad_target(void);
/* The same DAG replayed WITHOUT a hipGraph: lane 0 runs on `stream`, every other lane on a library-owned stream that forks from
 * and joins `stream`; every cross-lane edge is one event that names exactly the op it waits for.  Replaces the reference's
 * implicit stream order (it has none: PyTorch issues phiseg.py:326-537 op by op on one stream). */
int  uz_run_tape_lanes(const uz_op* ops, const uz_sched* sched, int n_ops, int n_lanes, void* stream);
/* diagnostics (tools/lane_trace.py): enable != 0 arms one timing event behind each of the first `capacity` ops of every following
 * uz_run_tape_lanes call; enable == 0 disarms and, with out != NULL, writes the end time (ms since the call began) of every op of the
 * last call and returns their count */
int  uz_lane_trace(int enable, int capacity, float* out, int n_out);
int  uz_graph_launch(void* graph_exec, void* stream);
void uz_graph_destroy(void* graph_exec);

/* ---------------------------------------------------------------- deep-level chain (round 6)
 * The 16 x 16 ... 2 x 2 levels of PHiSeg (phiseg.py:14-39 encoder levels 3 - 6, :42-73 UpConvolutionalBlock, :76-106 SampleZBlock,
 * :209-221 / :269-277 the likelihood's small planes) are ~340 launches of 5 - 40 us per step on the step's critical chains; beside the
 * device-filling convolutions of the other lanes each of them waits 80 - 180 us for a register slot.  uz_chain_run executes a whole
 * sub-DAG of such ops as PHASES of ONE persistent launch: n_workgroups resident workgroups (1024 threads = four waves per SIMD at
 * <= 128 VGPRs, one per CU at most) walk the phase table; the sub-ops of one phase are independent and are cut into tiles (ntiles,
 * counted by uz_chain_op_tiles: WAVE tiles for UZ_CH_CONV3, workgroup tiles otherwise) dealt round-robin to the workgroups starting
 * at workgroup tile0; phases are separated by a grid barrier among
 * the launch's own workgroups (bounded spin: a barrier that does not complete within ~2 s raises the status word and every
 * workgroup leaves - uz_chain_status).  Sub-ops mirror the per-op entry points above (same operands, same results to fp32 rounding;
 * the 3 x 3 convolutions run on the fp16 matrix pipe with two-piece split operands like uz_conv_fwd_ex's split path, staged straight
 * from global memory: no LDS).  Device tables: `ops` = n_ops uz_chain_op in phase order, `phases` = n_phases pairs {first op, op
 * count}; `state` = uz_chain_state_bytes() bytes (zeroed by the call).                                                              */
enum {
  UZ_CH_CONV3 = 1,      /* 3x3 pad 1, contraction channels % 16 == 0, output channels % 32 == 0 (forward and data gradient differ only in the packed image).
                           p = x, image (uz_chain_pack_weights), bias|NULL, y, slabs|NULL, x_amax, w_amax; i = Kc, KcTot, Mc, McTot, N, H, W, S, accumulate.
                           S == 1: y (+)= conv + bias; S > 1: slabs[S][N][Mc][HW] = partial sums (the consumer adds bias + slabs in order) */
  UZ_CH_CONV3_SMALL,    /* 3x3 pad 1 on the vector pipe, contraction channels <= 4 (the 2-channel latents), forward only.
                           p = x, w [Mc][Kc][3][3], bias|NULL, y; i = Kc, KcTot, Mc, McTot, N, H, W */
  UZ_CH_BN_FWD,         /* training-mode BatchNorm + ReLU of one Conv2D unit (torchlayers.py:18-21), N*H*W <= 8192.
                           p = y, gamma, beta, running_mean, running_var, save, a, slabs|NULL, a_amax|NULL, conv bias|NULL; i = C, CtotY, CtotA, N, HW, relu, S; f = eps, momentum */
  UZ_CH_AVGPOOL_FWD,    /* p = x, y, x_amax|NULL, y_amax|NULL; i = C, CtotX, CtotY, N, H, W (input plane) */
  UZ_CH_BILINEAR_FWD,   /* p = x, y, x_amax|NULL, y_amax|NULL; i = C, CtotX, CtotY, N, H, W (input plane), align_corners */
  UZ_CH_HEADS_FWD,      /* uz_latent_heads_fwd with L == 2: p = h, w_mu, b_mu, w_sigma, b_sigma, eps, mu, pre, sigma, z|NULL; i = Cin, CinTot, N, HW, act */
  UZ_CH_BN_BWD,         /* backward of UZ_CH_BN_FWD. p = dA, y, gamma, save, dy [N][C][HW], dgamma, dbeta, dbias|NULL, dy_amax|NULL, slabs|NULL (of dA), beta; i = C, CtotDa, CtotY, N, HW, relu, S */
  UZ_CH_AVGPOOL_BWD,    /* p = dy, dx; i = C, CtotDy, CtotDx, N, H, W (high-resolution plane), accumulate */
  UZ_CH_BILINEAR_BWD,   /* p = dy, dx; i = C, CtotDy, CtotDx, N, H, W (low-resolution plane), align_corners, accumulate */
  UZ_CH_LATENT_HEADS_BWD, /* uz_latent_sample_bwd + uz_latent_heads_bwd_data (L == 2): p = kl_dmu|NULL, kl_dsigma|NULL, dz|NULL, eps, sigma, g_mu (out), g_pre (out), w_sigma, w_mu, dh|NULL; i = Cin, CinTot, N, HW, act, accumulate */
  UZ_CH_CONV3_SMALL_BWD_DATA, /* data gradient of UZ_CH_CONV3_SMALL: p = dy, w [Mc][Kc][3][3], dx; i = Kc (<= 4, channels of dx), KcTot, Mc, McTot, N, H, W, accumulate */
  UZ_CH_SLAB_SUM,       /* dst (+)= sum_s slabs[s] in slab order: p = slabs [S][N][C][HW], dst; i = S, N, C, CtotDst, HW, accumulate */
  UZ_CH__COUNT
};
typedef struct uz_chain_op {
  int32_t code;        /* UZ_CH_* */
  int32_t tile0;       /* first workgroup tile of this sub-op inside its phase */
  int32_t ntiles;      /* uz_chain_op_tiles */
  int32_t rsv;
  int32_t i[16];
  float   f[4];
  void*   p[12];
} uz_chain_op;
int    uz_chain_op_tiles(const uz_chain_op* op);                 /* host: workgroup tiles of one sub-op, < 0 on a shape it does not cover */
int    uz_chain_conv_ksplit(int Kc, int Mc, int N, int H, int W, int n_workgroups);   /* S a UZ_CH_CONV3 of this shape should run with */
size_t uz_chain_packed_bytes(int Kc, int Mc);                     /* bytes of one packed image */
int    uz_chain_pack_blocks(int Kc, int Mc);                      /* 256-thread blocks uz_chain_pack_weights spends on one image */
/* table (device, int64): per layer {w, image, Mc, Kc, Cin of the parameter tensor, dgrad, first block}: image = the two fp16 pieces of
 * w * split_scale(*w_amax) in MFMA fragment order [tap][Kc / 16][Mc / 32][piece][lane][8]; dgrad: rows = input channels, taps flipped */
int    uz_chain_pack_weights(const int64_t* table, int n_layers, int total_blocks, const float* w_amax, void* stream);
size_t uz_chain_state_bytes(void);
int    uz_chain_run(const uz_chain_op* ops, const int32_t* phases, int n_phases, int n_ops, int n_workgroups, void* state, void* stream);
int    uz_chain_status(const void* state, int* out, void* stream);   /* synchronises; out = 0 ok, else 1 + the phase whose barrier timed out */

/* ---------------------------------------------------------------- data-parallel gradient exchange (RCCL over xGMI)
 * New component (the reference has no communication layer): one process per GPU, full replica, and
 * between loss.backward() and optimizer.step() (train_model.py:121-122) the flat fp32 gradient buffer is averaged over the
 * ranks.  librccl.so is bound at run time (uz_comm_load: path or NULL for the default search); the 128-byte unique id is
 * created on rank 0 (uz_comm_unique_id) and handed to the other ranks by the host (torch.distributed store / file / pipe);
 * uz_comm_init is collective.  Every call below is asynchronous on the given stream.                                       */
int  uz_comm_load(const char* librccl_path);
int  uz_comm_version(void);                                   /* RCCL version code, < 0 when RCCL cannot be loaded */
int  uz_comm_unique_id(void* out_128B);
int  uz_comm_init(int rank, int nranks, const void* unique_id_128B, void** comm_out);
void uz_comm_destroy(void* comm);
int  uz_comm_size(void* comm);
int  uz_allreduce_mean_f32(void* comm, float* flat, size_t count, void* stream);         /* in place, ncclAvg */
/* several slices {offset, count} (in floats, pairs in HOST memory) of one buffer in ONE RCCL group launch */
int  uz_allreduce_mean_f32_multi(void* comm, float* flat, const int64_t* offs_counts, int n_slices, void* stream);
int  uz_broadcast_f32(void* comm, float* flat, size_t count, int root, void* stream);
/* streams / events for overlapping the collective with the rest of the backward tape: a UZ_OP_EVENT_RECORD op inside the
 * backward tape (also when the tape is replayed as a hipGraph: external event-record node) marks a gradient bucket final;
 * the communication stream waits for it (uz_stream_wait_event) and runs the bucket's all-reduce beside the remaining
 * backward kernels; the compute stream finally waits for an event recorded behind the last all-reduce.                     */
int  uz_stream_create(void** stream_out, int high_priority);
void uz_stream_destroy(void* stream);
int  uz_stream_synchronize(void* stream);
int  uz_event_create(void** event_out, int timing);
void uz_event_destroy(void* event);
int  uz_event_record(void* event, void* stream);
int  uz_stream_wait_event(void* stream, void* event);
int  uz_event_elapsed_ms(void* start, void* stop, float* ms_out);

/* ---------------------------------------------------------------- input pipeline (data/batch_provider.py:43-67,140-271)
 * One training batch from a dataset resident in HBM: row gather (idx), random annotator (ann), rotation, random crop + resize
 * and flips per sample, images bilinear, labels as one-hot maps with an argmax after each resampling stage (OpenCV rules
 * restated, utils.py:16-36).  X (M,H,W) f32, Y (M,H,W,A) u8; params (B,8) f32 = {do_rot, cos, sin, do_scale, p_x, p_y, r, flips}
 * drawn by the host in the reference's order; outputs x (B,1,H,W) f32 and s (B,H,W) f32.                                    */
int uz_augment_batch(const float* X, const uint8_t* Y, int H, int W, int A, const int* idx, const int* ann,
                     const float* params, int B, int nlabels, float* x_out, float* s_out, void* stream);

/* ---------------------------------------------------------------- volumes (models/phiseg3D.py), one sample
 * A volume is stored [D + 2][C][H][W] (zero slice before and after the D real ones) = a batch of D 2-D images.  Conv3d(3x3x3,
 * pad 1) (phiseg3D.py:24) then is uz_conv_fwd / uz_conv_bwd_* with the depth window as 3 C input channels (pointer = slice d-1,
 * Cin = 3 C, CinTot = C) and permuted weights: mode 0 forward [co][kd][ci][9], mode 1 data gradient [(j, co)][ci][9] (kd = 2 - j),
 * mode 2 weight gradient back to the parameter layout [co][ci][kd][9].  BatchNorm3d / ReLU / 1x1x1 heads run their 2-D entry
 * points over the D slices.  AvgPool3d(2, 2, ceil_mode) (phiseg3D.py:101); depth stage of F.interpolate(mode='trilinear',
 * scale_factor=2, align_corners=True) (phiseg3D.py:146,306,376 - the in-plane stage is uz_bilinear2x_*); nearest volume resize.
 * Pointers address slice 0 of the REAL slices; (D, H, W) are input sizes.                                                     */
int uz_w3d_permute(const float* src, float* dst, int Cout, int Cin, int mode, void* stream);
int uz_avgpool3d_fwd(const float* x, int C, int CtotX, float* y, int CtotY, int D, int H, int W, void* stream);
int uz_avgpool3d_bwd(const float* dy, int C, int CtotDy, float* dx, int CtotDx, int D, int H, int W, int accumulate, void* stream);
int uz_depth_lerp2x_fwd(const float* x, int C, int CtotX, float* y, int CtotY, int D, int H, int W, void* stream);
int uz_depth_lerp2x_bwd(const float* dy, int C, int CtotDy, float* dx, int CtotDx, int D, int H, int W, int accumulate, void* stream);
int uz_nearest3d_fwd(const float* x, int C, int CtotX, float* y, int CtotY, int D, int H, int W, int f, int fz, void* stream);
int uz_nearest3d_bwd(const float* dy, int C, int CtotDy, float* dx, int CtotDx, int D, int H, int W, int f, int fz, int accumulate, void* stream);

/* ---------------------------------------------------------------- bf16 storage (BASELINE config 5: PHiSeg3D "bf16"; phiseg3D.py:13-35)
 * A tensor in bf16 storage has the same N x C x H x W shape with 2-byte elements (round to nearest even when written, widened exactly
 * when read); every arithmetic stays what the fp32-storage entry points do in the single-piece bf16 mode (uz_set_conv_math(3)):
 * bf16 operands, fp32 accumulation, fp32 / fp64 BatchNorm statistics (of the STORED values), fp32 parameters and gradients.  Each
 * tensor operand of these entry points carries its own flag (0 = fp32, 1 = bf16), so a plan may keep any subset of its buffers in
 * bf16.  Scope: the large planes of a volume - 3x3 convolutions on the matrix-pipe path with an unsplit chunk loop (weight
 * gradient: rows a multiple of 32 wide), the large-plane BatchNorm path (N*H*W > 32 768, H*W % 4 == 0), AvgPool3d and the depth
 * stage of the trilinear interpolation.  Replaces, for such tensors: uz_conv_fwd_packed / uz_conv_bwd_data_packed /
 * uz_conv_bwd_weight_ex, uz_bn_relu_fwd_pre / uz_bn_relu_bwd, uz_avgpool3d_* / uz_depth_lerp2x_*.                               */
int uz_conv_fwd_b16(const void* x, int Cin, int CinTot, const float* w, const float* bias,
                    void* y, int Cout, int CoutTot, int N, int H, int W, int ks,
                    void* workspace, size_t workspace_bytes, const void* packed_w, float* bn_partials,
                    int x_b16, int y_b16, void* stream);
int uz_conv_bwd_data_b16(const void* dy, int Cout, int CoutTot, const float* w, void* dx, int Cin, int CinTot,
                         int N, int H, int W, int ks, int accumulate,
                         void* workspace, size_t workspace_bytes, const void* packed_w, int dy_b16, int dx_b16, void* stream);
int uz_conv_split_parts(int kind, int Cin, int Cout, int N, int H, int W);   /* chunk-loop split of the matrix-pipe kernel (kind 0 forward, 1 data gradient); the *_b16 convolutions need 1 */
int uz_conv_bwd_weight_b16(const void* x, int Cin, int CinTot, const void* dy, int Cout, int CoutTot,
                           float* dw, int N, int H, int W, int ks, void* workspace, size_t workspace_bytes,
                           int x_b16, int dy_b16, float* slabs_out, void* stream);
int uz_bn_relu_fwd_b16(const void* y, int C, int CtotY, const float* gamma, const float* beta,
                       float* running_mean, float* running_var, float* save_mean_rstd,
                       void* a, int CtotA, int N, int H, int W, float eps, float momentum, int training, int relu,
                       void* workspace, const float* conv_partials, int n_partials, int y_b16, int a_b16, void* stream);
int uz_bn_relu_bwd_b16(const void* da, int CtotDa, const void* y, int C, int CtotY,
                       const float* gamma, const float* beta, const float* save_mean_rstd,
                       void* dy, int CtotDy, float* dgamma, float* dbeta, float* dbias,
                       int N, int H, int W, int relu, void* workspace, int da_b16, int y_b16, int dy_b16, void* stream);
int uz_avgpool3d_fwd_b16(const void* x, int C, int CtotX, void* y, int CtotY, int D, int H, int W, int x_b16, int y_b16, void* stream);
int uz_avgpool3d_bwd_b16(const void* dy, int C, int CtotDy, void* dx, int CtotDx, int D, int H, int W, int accumulate, int dy_b16, int dx_b16, void* stream);
int uz_depth_lerp2x_fwd_b16(const void* x, int C, int CtotX, void* y, int CtotY, int D, int H, int W, int x_b16, int y_b16, void* stream);
int uz_depth_lerp2x_bwd_b16(const void* dy, int C, int CtotDy, void* dx, int CtotDx, int D, int H, int W, int accumulate, int dy_b16, int dx_b16, void* stream);
/* in-plane stage of the trilinear interpolation: the HIGH-resolution side (y; dy) in bf16 storage, the low-resolution side fp32 */
int uz_bilinear2x_fwd_b16(const float* x, int C, int CtotX, void* y, int CtotY, int N, int H, int W, int align_corners, int y_b16, void* stream);
int uz_bilinear2x_bwd_b16(const void* dy, int C, int CtotDy, float* dx, int CtotDx, int N, int H, int W, int align_corners, int accumulate, int dy_b16, void* stream);
/* 1x1(x1) heads with 1 .. 8 outputs (mu_conv / sigma_conv phiseg3D.py:83-84, s_layer): the many-channel operand (x; dx of the data
 * gradient) in bf16 storage, the few-channel side (y, dy) fp32.  Replace uz_conv_fwd / uz_conv_bwd_data / uz_conv_bwd_weight (ks = 1). */
int uz_conv1x1_fwd_b16(const void* x, int Cin, int CinTot, const float* w, const float* bias, float* y, int Cout, int CoutTot,
                       int N, int H, int W, int x_b16, void* stream);
int uz_conv1x1_bwd_data_b16(const float* dy, int Cout, int CoutTot, const float* w, void* dx, int Cin, int CinTot,
                            int N, int H, int W, int accumulate, int dx_b16, void* stream);
int uz_conv1x1_bwd_weight_b16(const void* x, int Cin, int CinTot, const float* dy, int Cout, int CoutTot, float* dw, float* db,
                              int N, int H, int W, void* workspace, size_t workspace_bytes, int x_b16, void* stream);
/* element-wise conversions between the two storage formats (n elements, contiguous) */
int uz_cvt_f32_to_b16(const float* src, void* dst, size_t n, void* stream);
int uz_cvt_b16_to_f32(const void* src, float* dst, size_t n, void* stream);

/* Fcomb input (probabilistic_unet.py:172-197) */
/* z (N,L) tiled over HxW into channels of a (N,Ctot,H,W) buffer, and its backward (sum over pixels) */
int uz_bcast_channels_fwd(const float* z, int L, float* y, int CtotY, int N, int H, int W, void* stream);
int uz_bcast_channels_bwd(const float* dy, int CtotDy, int L, float* dz, int N, int H, int W, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* UZ_API_H */
