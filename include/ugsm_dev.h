/*
 * ugsm_dev.h -- entry points of libugsm_dev.so that libugsm.so does not have.
 *
 * libugsm_dev.so is the product's sources plus ug_stereomatcher_amd/csrc/dev/ (csrc/Makefile): everything include/ugsm.h declares, plus what
 * only the tests and the measurement tools need -- kernel_path 1 (one kernel per reference stage: the A/B reference of the fused kernels),
 * ugsm_config.march_min_pixels < 0 (round 1's LDS-tiled K-cost), and the probes below.  A maintainer links libugsm.so.
 */
#ifndef UGSM_DEV_H
#define UGSM_DEV_H

#include "ugsm.h"

#ifdef __cplusplus
extern "C" {
#endif
#if defined(__GNUC__)
#pragma GCC visibility push(default)
#endif

/* The fused kernels' exact arithmetic shortcuts (f32 first quotient of PolyDisparity, x/3 by two
 * FMAs) evaluated on caller-supplied operands: delta/corr = PolyDisparity(c,l,r,thr)
 * (MatchLib.cu:805-836), third = c/3.0f for c >= 0.  Lets tests force the rare fallback branches. */
int ugsm_stage_poly_probe(ugsm_ctx *ctx, const float *d_c, const float *d_l, const float *d_r,
                          const float *d_thr, float *d_delta, float *d_corr, float *d_third, int n);

/* K-smooth's shared-reciprocal division (three weighted sums over one sumCorr, MatchLib.cu:1131-1139)
 * on caller-supplied operands, with the kernel's own range test and literal fallback:
 * q_f[i] must equal the IEEE binary32 quotient a_f[i] / s[i] bit for bit. */
int ugsm_stage_div3_probe(ugsm_ctx *ctx, const float *d_a0, const float *d_a1, const float *d_a2,
                          const float *d_s, float *d_q0, float *d_q1, float *d_q2, int n);

/* K-cost's range-guarded division (the compiler's binary32 division sequence without v_div_scale / v_div_fixup, used when
 * every pyramid value of the pair is 0 or in [2^-12, 2^9]; csrc/ugsm_exact.hpp) on caller-supplied operands:
 * q[i] must equal the IEEE binary32 quotient n[i] / d[i] bit for bit for operands that are 0 or in [2^-62, 2^37]. */
int ugsm_stage_div_probe(ugsm_ctx *ctx, const float *d_n, const float *d_d, float *d_q, int n);

#if defined(__GNUC__)
#pragma GCC visibility pop
#endif
#ifdef __cplusplus
}
#endif
#endif /* UGSM_DEV_H */
