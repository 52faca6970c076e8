/* CPU twins of the C-ABI entry points of include/uz_api.h - TEST INFRASTRUCTURE (see oracle/__init__.py).
 *
 * "every GPU entry point has a uz_cpu_* twin (scalar restatement) with identical signature minus the stream".
 * Each function below takes exactly the arguments of its uz_* namesake without the trailing `void* stream`; magnitude-bound
 * slots and workspaces keep their positions and are ignored (they are plumbing of the HIP kernels, not semantics).  Plain C,
 * scalar loops, double accumulation for reductions: written for obviousness, not speed.  The twins restate the SAME reference
 * call sites as the header does (cited there); they are pinned twice - against torch.nn.functional on the CPU
 * (tests/test_cpu_twins.py, which in turn is what the pinned oracle/refgraph.py is made of) and they serve as the checker of
 * the HIP kernels through identical argument lists (tests/test_cpu_twins.py -m gpu).
 *
 * Built by `make -C oracle` (gcc) into oracle/_build/libuz_cpu.so; never linked into or loaded by the product. */
#include <math.h>
#include <stddef.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define IDX4(b, c, y, x, Ctot, H, W) ((((size_t)(b) * (Ctot) + (c)) * (H) + (y)) * (W) + (x))

/* ------------------------------------------------------------------ convolution family (uz_conv_fwd / _bwd_data / _bwd_weight)
 * nn.Conv2d(k, stride 1, pad k/2), w = [Cout][Cin][k][k] */
int uz_cpu_conv_fwd(const float* x, int Cin, int CinTot, const float* w, const float* bias, float* y, int Cout, int CoutTot,
                    int N, int H, int W, int ks, int relu, const float* x_amax, const float* w_amax, float* y_amax,
                    void* workspace, size_t workspace_bytes) {
    (void)x_amax; (void)w_amax; (void)y_amax; (void)workspace; (void)workspace_bytes;
    const int pad = ks / 2;
    for (int b = 0; b < N; ++b)
        for (int co = 0; co < Cout; ++co)
            for (int yy = 0; yy < H; ++yy)
                for (int xx = 0; xx < W; ++xx) {
                    double acc = bias ? bias[co] : 0.0;
                    for (int ci = 0; ci < Cin; ++ci)
                        for (int ky = 0; ky < ks; ++ky)
                            for (int kx = 0; kx < ks; ++kx) {
                                const int sy = yy + ky - pad, sx = xx + kx - pad;
                                if (sy < 0 || sy >= H || sx < 0 || sx >= W) continue;
                                acc += (double)x[IDX4(b, ci, sy, sx, CinTot, H, W)] * w[(((size_t)co * Cin + ci) * ks + ky) * ks + kx];
                            }
                    float v = (float)acc;
                    if (relu && v < 0.f) v = 0.f;
                    y[IDX4(b, co, yy, xx, CoutTot, H, W)] = v;
                }
    return 0;
}

int uz_cpu_conv_bwd_data(const float* dy, int Cout, int CoutTot, const float* w, float* dx, int Cin, int CinTot,
                         int N, int H, int W, int ks, int accumulate, const float* dy_amax, const float* w_amax,
                         void* workspace, size_t workspace_bytes) {
    (void)dy_amax; (void)w_amax; (void)workspace; (void)workspace_bytes;
    const int pad = ks / 2;
    for (int b = 0; b < N; ++b)
        for (int ci = 0; ci < Cin; ++ci)
            for (int yy = 0; yy < H; ++yy)
                for (int xx = 0; xx < W; ++xx) {
                    double acc = 0.0;
                    for (int co = 0; co < Cout; ++co)
                        for (int ky = 0; ky < ks; ++ky)
                            for (int kx = 0; kx < ks; ++kx) {
                                const int oy = yy - ky + pad, ox = xx - kx + pad;      /* output pixel that read (yy, xx) through tap (ky, kx) */
                                if (oy < 0 || oy >= H || ox < 0 || ox >= W) continue;
                                acc += (double)dy[IDX4(b, co, oy, ox, CoutTot, H, W)] * w[(((size_t)co * Cin + ci) * ks + ky) * ks + kx];
                            }
                    float* d = dx + IDX4(b, ci, yy, xx, CinTot, H, W);
                    *d = accumulate ? *d + (float)acc : (float)acc;
                }
    return 0;
}

int uz_cpu_conv_bwd_weight(const float* x, int Cin, int CinTot, const float* dy, int Cout, int CoutTot, float* dw, float* db,
                           int N, int H, int W, int ks, const float* x_amax, const float* dy_amax, void* workspace, size_t workspace_bytes) {
    (void)x_amax; (void)dy_amax; (void)workspace; (void)workspace_bytes;
    const int pad = ks / 2;
    for (int co = 0; co < Cout; ++co) {
        for (int ci = 0; ci < Cin; ++ci)
            for (int ky = 0; ky < ks; ++ky)
                for (int kx = 0; kx < ks; ++kx) {
                    double acc = 0.0;
                    for (int b = 0; b < N; ++b)
                        for (int yy = 0; yy < H; ++yy) {
                            const int sy = yy + ky - pad;
                            if (sy < 0 || sy >= H) continue;
                            for (int xx = 0; xx < W; ++xx) {
                                const int sx = xx + kx - pad;
                                if (sx < 0 || sx >= W) continue;
                                acc += (double)dy[IDX4(b, co, yy, xx, CoutTot, H, W)] * x[IDX4(b, ci, sy, sx, CinTot, H, W)];
                            }
                        }
                    dw[(((size_t)co * Cin + ci) * ks + ky) * ks + kx] = (float)acc;
                }
        if (db) {
            double s = 0.0;
            for (int b = 0; b < N; ++b)
                for (int q = 0; q < H * W; ++q) s += dy[((size_t)b * CoutTot + co) * H * W + q];
            db[co] = (float)s;
        }
    }
    return 0;
}

/* ------------------------------------------------------------------ BatchNorm2d + ReLU (uz_bn_relu_fwd / _bwd), ReLU backward */
int uz_cpu_bn_relu_fwd(const float* y, int C, int CtotY, const float* gamma, const float* beta, float* running_mean, float* running_var,
                       float* save_mean_rstd, float* a, int CtotA, int N, int H, int W, float eps, float momentum, int training, int relu,
                       float* a_amax, void* workspace) {
    (void)a_amax; (void)workspace;
    const size_t HW = (size_t)H * W;
    const double M = (double)N * HW;
    for (int c = 0; c < C; ++c) {
        double mean, var;
        if (training) {
            double s = 0.0, s2 = 0.0;
            for (int b = 0; b < N; ++b)
                for (size_t q = 0; q < HW; ++q) s += y[((size_t)b * CtotY + c) * HW + q];
            mean = s / M;
            for (int b = 0; b < N; ++b)
                for (size_t q = 0; q < HW; ++q) { const double d = y[((size_t)b * CtotY + c) * HW + q] - mean; s2 += d * d; }
            var = s2 / M;                                                     /* biased: normalisation */
            const double unbiased = M > 1 ? s2 / (M - 1) : var;               /* unbiased: running estimate */
            running_mean[c] = (float)((1.0 - momentum) * running_mean[c] + momentum * mean);
            running_var[c] = (float)((1.0 - momentum) * running_var[c] + momentum * unbiased);
        } else {
            mean = running_mean[c];
            var = running_var[c];
        }
        const double rstd = 1.0 / sqrt(var + eps);
        if (training && save_mean_rstd) { save_mean_rstd[c] = (float)mean; save_mean_rstd[C + c] = (float)rstd; }
        for (int b = 0; b < N; ++b)
            for (size_t q = 0; q < HW; ++q) {
                float v = (float)((y[((size_t)b * CtotY + c) * HW + q] - mean) * rstd * gamma[c] + beta[c]);
                if (relu && v < 0.f) v = 0.f;
                a[((size_t)b * CtotA + c) * HW + q] = v;
            }
    }
    return 0;
}

int uz_cpu_bn_relu_bwd(const float* da, int CtotDa, const float* y, int C, int CtotY, const float* gamma, const float* beta,
                       const float* save_mean_rstd, float* dy, int CtotDy, float* dgamma, float* dbeta, float* dbias,
                       int N, int H, int W, int relu, float* dy_amax, void* workspace) {
    (void)dy_amax; (void)workspace;
    const size_t HW = (size_t)H * W;
    const double M = (double)N * HW;
    for (int c = 0; c < C; ++c) {
        const double mean = save_mean_rstd[c], rstd = save_mean_rstd[C + c];
        double sg = 0.0, sb = 0.0;
        for (int b = 0; b < N; ++b)
            for (size_t q = 0; q < HW; ++q) {
                const double xh = (y[((size_t)b * CtotY + c) * HW + q] - mean) * rstd;
                double dz = da[((size_t)b * CtotDa + c) * HW + q];
                if (relu && !((float)(xh * gamma[c] + beta[c]) > 0.f)) dz = 0.0;             /* threshold_backward on the recomputed output */
                sg += dz * xh;
                sb += dz;
            }
        double sdy = 0.0;
        for (int b = 0; b < N; ++b)
            for (size_t q = 0; q < HW; ++q) {
                const double xh = (y[((size_t)b * CtotY + c) * HW + q] - mean) * rstd;
                double dz = da[((size_t)b * CtotDa + c) * HW + q];
                if (relu && !((float)(xh * gamma[c] + beta[c]) > 0.f)) dz = 0.0;
                const double g = gamma[c] * rstd * (dz - sb / M - xh * sg / M);
                dy[((size_t)b * CtotDy + c) * HW + q] = (float)g;
                sdy += g;
            }
        dgamma[c] = (float)sg;
        dbeta[c] = (float)sb;
        if (dbias) dbias[c] = (float)sdy;
    }
    return 0;
}

int uz_cpu_relu_bwd(const float* da, int CtotDa, const float* a, int C, int CtotA, float* dy, int CtotDy, float* dbias,
                    int N, int H, int W, float* dy_amax, void* workspace) {
    (void)dy_amax; (void)workspace;
    const size_t HW = (size_t)H * W;
    for (int c = 0; c < C; ++c) {
        double s = 0.0;
        for (int b = 0; b < N; ++b)
            for (size_t q = 0; q < HW; ++q) {
                const float g = a[((size_t)b * CtotA + c) * HW + q] > 0.f ? da[((size_t)b * CtotDa + c) * HW + q] : 0.f;
                dy[((size_t)b * CtotDy + c) * HW + q] = g;
                s += g;
            }
        if (dbias) dbias[c] = (float)s;
    }
    return 0;
}

/* ------------------------------------------------------------------ resampling */
int uz_cpu_avgpool2_fwd(const float* x, int C, int CtotX, float* y, int CtotY, int N, int H, int W, const float* x_amax, float* y_amax) {
    (void)x_amax; (void)y_amax;
    const int Ho = (H + 1) / 2, Wo = (W + 1) / 2;
    for (int b = 0; b < N; ++b)
        for (int c = 0; c < C; ++c)
            for (int oy = 0; oy < Ho; ++oy)
                for (int ox = 0; ox < Wo; ++ox) {
                    double s = 0.0;
                    int cnt = 0;
                    for (int yy = 2 * oy; yy < 2 * oy + 2 && yy < H; ++yy)
                        for (int xx = 2 * ox; xx < 2 * ox + 2 && xx < W; ++xx) { s += x[IDX4(b, c, yy, xx, CtotX, H, W)]; ++cnt; }
                    y[IDX4(b, c, oy, ox, CtotY, Ho, Wo)] = (float)(s / cnt);
                }
    return 0;
}
int uz_cpu_avgpool2_bwd(const float* dy, int C, int CtotDy, float* dx, int CtotDx, int N, int H, int W, int accumulate) {
    const int Ho = (H + 1) / 2, Wo = (W + 1) / 2;
    for (int b = 0; b < N; ++b)
        for (int c = 0; c < C; ++c)
            for (int yy = 0; yy < H; ++yy)
                for (int xx = 0; xx < W; ++xx) {
                    const int oy = yy / 2, ox = xx / 2;
                    const int cnt = ((2 * oy + 2 < H ? 2 * oy + 2 : H) - 2 * oy) * ((2 * ox + 2 < W ? 2 * ox + 2 : W) - 2 * ox);
                    const float v = dy[IDX4(b, c, oy, ox, CtotDy, Ho, Wo)] / (float)cnt;
                    float* d = dx + IDX4(b, c, yy, xx, CtotDx, H, W);
                    *d = accumulate ? *d + v : v;
                }
    return 0;
}

/* ATen area_pixel_compute_source_index + the lambda pair of upsample_bilinear2d (float arithmetic, as ATen) */
static void src_index(int o, float scale, int ac, int in, int* i0, int* ip, float* l0, float* l1) {
    float r;
    if (ac) r = scale * (float)o;
    else { r = scale * ((float)o + 0.5f) - 0.5f; if (r < 0.f) r = 0.f; }
    *i0 = (int)r;
    if (*i0 > in - 1) *i0 = in - 1;
    *ip = (*i0 < in - 1) ? 1 : 0;
    *l1 = r - (float)*i0;
    *l0 = 1.f - *l1;
}
static float bil_scale(int in, int out, int ac) { return ac ? (out > 1 ? (float)(in - 1) / (float)(out - 1) : 0.f) : 0.5f; }

int uz_cpu_bilinear2x_fwd(const float* x, int C, int CtotX, float* y, int CtotY, int N, int H, int W, int align_corners,
                          const float* x_amax, float* y_amax) {
    (void)x_amax; (void)y_amax;
    const int Ho = 2 * H, Wo = 2 * W;
    const float sh = bil_scale(H, Ho, align_corners), sw = bil_scale(W, Wo, align_corners);
    for (int b = 0; b < N; ++b)
        for (int c = 0; c < C; ++c)
            for (int oy = 0; oy < Ho; ++oy)
                for (int ox = 0; ox < Wo; ++ox) {
                    int h1, hp, w1, wp; float h0l, h1l, w0l, w1l;
                    src_index(oy, sh, align_corners, H, &h1, &hp, &h0l, &h1l);
                    src_index(ox, sw, align_corners, W, &w1, &wp, &w0l, &w1l);
                    const float* r0 = x + IDX4(b, c, h1, w1, CtotX, H, W);
                    const float* r1 = r0 + (size_t)hp * W;
                    y[IDX4(b, c, oy, ox, CtotY, Ho, Wo)] = h0l * (w0l * r0[0] + w1l * r0[wp]) + h1l * (w0l * r1[0] + w1l * r1[wp]);
                }
    return 0;
}
int uz_cpu_bilinear2x_bwd(const float* dy, int C, int CtotDy, float* dx, int CtotDx, int N, int H, int W, int align_corners, int accumulate) {
    const int Ho = 2 * H, Wo = 2 * W;
    const float sh = bil_scale(H, Ho, align_corners), sw = bil_scale(W, Wo, align_corners);
    double* tmp = (double*)malloc(sizeof(double) * (size_t)H * W);
    if (!tmp) return -1;
    for (int b = 0; b < N; ++b)
        for (int c = 0; c < C; ++c) {
            memset(tmp, 0, sizeof(double) * (size_t)H * W);
            for (int oy = 0; oy < Ho; ++oy)
                for (int ox = 0; ox < Wo; ++ox) {                              /* scatter form of upsample_bilinear2d_backward */
                    int h1, hp, w1, wp; float h0l, h1l, w0l, w1l;
                    src_index(oy, sh, align_corners, H, &h1, &hp, &h0l, &h1l);
                    src_index(ox, sw, align_corners, W, &w1, &wp, &w0l, &w1l);
                    const double g = dy[IDX4(b, c, oy, ox, CtotDy, Ho, Wo)];
                    tmp[(size_t)h1 * W + w1] += h0l * w0l * g;
                    tmp[(size_t)h1 * W + w1 + wp] += h0l * w1l * g;
                    tmp[(size_t)(h1 + hp) * W + w1] += h1l * w0l * g;
                    tmp[(size_t)(h1 + hp) * W + w1 + wp] += h1l * w1l * g;
                }
            for (int q = 0; q < H * W; ++q) {
                float* d = dx + ((size_t)b * CtotDx + c) * H * W + q;
                *d = accumulate ? *d + (float)tmp[q] : (float)tmp[q];
            }
        }
    free(tmp);
    return 0;
}
int uz_cpu_nearest_fwd(const float* x, int C, int CtotX, float* y, int CtotY, int N, int H, int W, int factor) {
    const int Ho = H * factor, Wo = W * factor;
    for (int b = 0; b < N; ++b)
        for (int c = 0; c < C; ++c)
            for (int oy = 0; oy < Ho; ++oy)
                for (int ox = 0; ox < Wo; ++ox) y[IDX4(b, c, oy, ox, CtotY, Ho, Wo)] = x[IDX4(b, c, oy / factor, ox / factor, CtotX, H, W)];
    return 0;
}
int uz_cpu_nearest_bwd(const float* dy, int C, int CtotDy, float* dx, int CtotDx, int N, int H, int W, int factor, int accumulate) {
    const int Ho = H * factor, Wo = W * factor;
    for (int b = 0; b < N; ++b)
        for (int c = 0; c < C; ++c)
            for (int yy = 0; yy < H; ++yy)
                for (int xx = 0; xx < W; ++xx) {
                    double s = 0.0;
                    for (int j = 0; j < factor; ++j)
                        for (int i = 0; i < factor; ++i) s += dy[IDX4(b, c, yy * factor + j, xx * factor + i, CtotDy, Ho, Wo)];
                    float* d = dx + IDX4(b, c, yy, xx, CtotDx, H, W);
                    *d = accumulate ? *d + (float)s : (float)s;
                }
    return 0;
}

/* ------------------------------------------------------------------ latent heads / losses */
int uz_cpu_posterior_input(const float* patch, int in_ch, const float* mask, int nlabels, float* out, int N, int H, int W) {
    const size_t HW = (size_t)H * W;
    for (int b = 0; b < N; ++b) {
        for (int c = 0; c < in_ch; ++c) memcpy(out + ((size_t)b * (in_ch + nlabels) + c) * HW, patch + ((size_t)b * in_ch + c) * HW, HW * sizeof(float));
        for (int l = 0; l < nlabels; ++l)
            for (size_t q = 0; q < HW; ++q) out[((size_t)b * (in_ch + nlabels) + in_ch + l) * HW + q] = (mask[(size_t)b * HW + q] == (float)l ? 1.f : 0.f) - 0.5f;
    }
    return 0;
}
static float softplus(float x) { return x > 20.f ? x : log1pf(expf(x)); }                 /* F.softplus(beta 1, threshold 20) */
int uz_cpu_latent_sample_fwd(const float* mu, const float* pre_sigma, const float* eps, float* sigma, float* z, size_t n, int act) {
    for (size_t i = 0; i < n; ++i) {
        const float s = act ? expf(pre_sigma[i]) : softplus(pre_sigma[i]);
        sigma[i] = s;
        if (z) z[i] = mu[i] + s * eps[i];
    }
    return 0;
}
int uz_cpu_latent_sample_bwd(const float* dmu, const float* dsigma, const float* dz, const float* eps, const float* sigma,
                             float* dmu_pre, float* dpre_sigma, size_t n, int act) {
    for (size_t i = 0; i < n; ++i) {
        const float gz = dz ? dz[i] : 0.f;
        dmu_pre[i] = (dmu ? dmu[i] : 0.f) + gz;
        const float ds = (dsigma ? dsigma[i] : 0.f) + gz * (eps ? eps[i] : 0.f);
        dpre_sigma[i] = ds * (act ? sigma[i] : 1.f - expf(-sigma[i]));
    }
    return 0;
}
/* The tail of a SampleZBlock as one call (phiseg.py:95-105: mu = mu_conv(h); sigma = softplus(sigma_conv(h)); z = mu + sigma * eps) and its
 * backward: twins of uz_latent_heads_* - by definition the composition of the 1x1 convolution twins and the sampling twins above. */
int uz_cpu_latent_heads_fwd(const float* h, int Cin, int CinTot, const float* w_mu, const float* b_mu, const float* w_sigma, const float* b_sigma,
                            const float* eps, float* mu, float* pre_sigma, float* sigma, float* z, int L, int N, int H, int W, int act) {
    uz_cpu_conv_fwd(h, Cin, CinTot, w_mu, b_mu, mu, L, L, N, H, W, 1, 0, NULL, NULL, NULL, NULL, 0);
    uz_cpu_conv_fwd(h, Cin, CinTot, w_sigma, b_sigma, pre_sigma, L, L, N, H, W, 1, 0, NULL, NULL, NULL, NULL, 0);
    return uz_cpu_latent_sample_fwd(mu, pre_sigma, eps, sigma, z, (size_t)N * L * H * W, act);
}
int uz_cpu_latent_heads_bwd_data(const float* dy_a, const float* dy_b, int L, const float* w_a, const float* w_b, float* dh, int Cin, int CinTot,
                                 int N, int H, int W, int accumulate) {
    uz_cpu_conv_bwd_data(dy_a, L, L, w_a, dh, Cin, CinTot, N, H, W, 1, accumulate, NULL, NULL, NULL, 0);
    return uz_cpu_conv_bwd_data(dy_b, L, L, w_b, dh, Cin, CinTot, N, H, W, 1, 1, NULL, NULL, NULL, 0);
}
int uz_cpu_latent_heads_bwd_weight(const float* h, int Cin, int CinTot, const float* dy_a, const float* dy_b, int L, float* dw_a, float* db_a,
                                   float* dw_b, float* db_b, int N, int H, int W, void* workspace, size_t workspace_bytes) {
    (void)workspace; (void)workspace_bytes;
    uz_cpu_conv_bwd_weight(h, Cin, CinTot, dy_a, L, L, dw_a, db_a, N, H, W, 1, NULL, NULL, NULL, 0);
    return uz_cpu_conv_bwd_weight(h, Cin, CinTot, dy_b, L, L, dw_b, db_b, N, H, W, 1, NULL, NULL, NULL, 0);
}
/* KL_two_gauss_with_diag_cov with sigma1_fs = sigma1 * sigma0 (phiseg.py:438-439) */
int uz_cpu_kl_fwd(const float* mu0, const float* s0, const float* mu1, const float* s1, int N, int per_sample, float weight, float* loss_out) {
    double tot = 0.0;
    for (size_t i = 0; i < (size_t)N * per_sample; ++i) {
        const double a = (double)s0[i] * s0[i], bq = (double)s1[i] * s0[i], d = (double)mu1[i] - mu0[i];
        tot += 0.5 * ((a + d * d) / (bq + 1e-10) + log(bq + 1e-10) - log(a + 1e-10) - 1.0);
    }
    loss_out[0] = (float)(weight * tot / N);
    return 0;
}
int uz_cpu_kl_bwd(const float* mu0, const float* s0, const float* mu1, const float* s1, int N, int per_sample, float weight,
                  const float* loss_scale, float* dmu0, float* ds0, float* dmu1, float* ds1) {
    const double k = 0.5 * weight * loss_scale[0] / N;
    for (size_t i = 0; i < (size_t)N * per_sample; ++i) {
        const double a0 = s0[i], a1 = s1[i], d = (double)mu1[i] - mu0[i];
        const double A = a0 * a0 + d * d, B = a1 * a0 + 1e-10;
        dmu0[i] = (float)(k * (-2.0 * d / B));
        dmu1[i] = (float)(k * (2.0 * d / B));
        ds0[i] = (float)(k * ((2.0 * a0 * B - A * a1) / (B * B) + a1 / B - 2.0 * a0 / (a0 * a0 + 1e-10)));
        ds1[i] = (float)(k * (-A * a0 / (B * B) + a0 / B));
    }
    return 0;
}
/* residual_multinoulli_loss: level l loss = mean_b sum_pix CE(sum_{j>=l} s_j, mask) */
int uz_cpu_residual_ce_fwd(const float* const* s_ptrs, int L, int K, const float* mask, int N, int H, int W, float* loss_out, void* workspace) {
    (void)workspace;
    const size_t HW = (size_t)H * W;
    double acc[8];
    for (int l = 0; l < L; ++l) loss_out[l] = 0.f;
    double* tot = (double*)calloc((size_t)L, sizeof(double));
    if (!tot) return -1;
    for (int b = 0; b < N; ++b)
        for (size_t q = 0; q < HW; ++q) {
            const int t = (int)mask[(size_t)b * HW + q];
            for (int k = 0; k < K; ++k) acc[k] = 0.0;
            for (int l = L - 1; l >= 0; --l) {
                double mx = -1e300, se = 0.0;
                for (int k = 0; k < K; ++k) { acc[k] += s_ptrs[l][((size_t)b * K + k) * HW + q]; if (acc[k] > mx) mx = acc[k]; }
                for (int k = 0; k < K; ++k) se += exp(acc[k] - mx);
                tot[l] += mx + log(se) - acc[t];
            }
        }
    for (int l = 0; l < L; ++l) loss_out[l] = (float)(tot[l] / N);
    free(tot);
    return 0;
}
int uz_cpu_residual_ce_bwd(const float* const* s_ptrs, float* const* ds_ptrs, int L, int K, const float* mask, int N, int H, int W, const float* loss_scale) {
    /* loss_l sees acc_l = sum_{j >= l} s_j, so d(sum_l loss_l) / d s_j = sum_{l <= j} (softmax(acc_l) - onehot) * scale / N */
    const size_t HW = (size_t)H * W;
    double acc[8], P[8][8];
    if (L > 8 || K > 8) return -1;
    for (int b = 0; b < N; ++b)
        for (size_t q = 0; q < HW; ++q) {
            const int t = (int)mask[(size_t)b * HW + q];
            for (int k = 0; k < K; ++k) acc[k] = 0.0;
            for (int l = L - 1; l >= 0; --l) {
                double mx = -1e300, se = 0.0;
                for (int k = 0; k < K; ++k) { acc[k] += s_ptrs[l][((size_t)b * K + k) * HW + q]; if (acc[k] > mx) mx = acc[k]; }
                for (int k = 0; k < K; ++k) se += exp(acc[k] - mx);
                for (int k = 0; k < K; ++k) P[l][k] = exp(acc[k] - mx) / se - (k == t ? 1.0 : 0.0);
            }
            for (int k = 0; k < K; ++k) {
                double run = 0.0;
                for (int j = 0; j < L; ++j) {
                    run += P[j][k];
                    ds_ptrs[j][((size_t)b * K + k) * HW + q] = (float)(run * loss_scale[0] / N);
                }
            }
        }
    return 0;
}
int uz_cpu_sum_terms(const float* terms, int n, float* total) {
    double s = 0.0;
    for (int i = 0; i < n; ++i) s += terms[i];
    total[0] = (float)s;
    return 0;
}

/* ------------------------------------------------------------------ optimiser: torch.optim.Adam with L2 weight decay added to the gradient */
int uz_cpu_adam_step(float* params, const float* grads, float* exp_avg, float* exp_avg_sq, size_t n, int64_t step, float lr, float beta1,
                     float beta2, float eps, float weight_decay, float grad_scale) {
    const double bc1 = 1.0 - pow(beta1, (double)step), bc2 = 1.0 - pow(beta2, (double)step);
    for (size_t i = 0; i < n; ++i) {
        const float g = grads[i] * grad_scale + weight_decay * params[i];
        exp_avg[i] = beta1 * exp_avg[i] + (1.f - beta1) * g;
        exp_avg_sq[i] = beta2 * exp_avg_sq[i] + (1.f - beta2) * g * g;
        const double denom = sqrt((double)exp_avg_sq[i]) / sqrt(bc2) + eps;
        params[i] = (float)(params[i] - lr / bc1 * exp_avg[i] / denom);
    }
    return 0;
}
int uz_cpu_add_views(const float* a, int CtotA, const float* b, int CtotB, float* y, int CtotY, int C, int N, int H, int W,
                     float alpha, int accumulate, const float* a_amax, const float* b_amax, float* y_amax) {
    (void)a_amax; (void)b_amax; (void)y_amax;
    const size_t HW = (size_t)H * W;
    for (int n = 0; n < N; ++n)
        for (int c = 0; c < C; ++c)
            for (size_t q = 0; q < HW; ++q) {
                float v = a[((size_t)n * CtotA + c) * HW + q] + (b ? alpha * b[((size_t)n * CtotB + c) * HW + q] : 0.f);
                float* d = y + ((size_t)n * CtotY + c) * HW + q;
                *d = accumulate ? *d + v : v;
            }
    return 0;
}
