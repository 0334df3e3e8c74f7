/*
 * ugsm_oracle.h -- CPU restatement of the ug_stereomatcher pyramidal matcher.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing in the product path (ug_stereomatcher_amd/,
 * include/, libugsm.so) may include, link or call this.  Only tests/, bench.py's
 * cpu_baseline leg and __graft_entry__.smoke() use it, as the checker.
 *
 * PARITY PIN STATUS: the reference ships no tests, golden vectors or CPU matcher
 * (SURVEY.md section 4 / 8c), and its CUDA path cannot be built here.  The only
 * reference fragment that compiles in this image is convolutionSeparable_gold.cpp
 * (oracle/_ref/libgold.so, built by oracle/Makefile from /root/reference in place);
 * the zero-padded blur of this oracle is pinned against it.  Everything else is
 * "parity unpinned" by the reference: pinned only by known-answer tests derived
 * from the reference source and by this restatement's own golden fixtures.
 *
 * Arithmetic contract (see DESIGN.md "Float contract"): IEEE-754 binary32 with the
 * reference's float->double promotions mirrored, no FMA contraction
 * (-ffp-contract=off), taps accumulated in the reference's order.
 */
#ifndef UGSM_ORACLE_H
#define UGSM_ORACLE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ORC_MAX_LEVELS 32

/* MatchGPULib.cpp:761-774 -- literal taps / sequential f32 sum. */
void orc_gauss_taps(float g[5]);
/* MatchGPULib.cpp:344-348 */
void orc_box_taps(float a[5]);

/* MatchGPULib.cpp:1224-1228 (w[i+1] = (int)(w[i] / 1.41421356)). Returns 0, or -1 if a level < 1 px. */
int orc_level_dims(int W, int H, int levels, int *w, int *h);
/* MatchGPULib.cpp:1741 / :1260 */
int orc_iterations_for_level(int i);
/* MatchGPULib.cpp:2257-2261 (level>11 <=> i<2 with 14 levels; generalised: i < 2) */
int orc_smooth_passes_for_level(int i);
/* MatchGPULib.cpp:2299-2306: thr consumed by iteration m=1..mi -> out[0..mi-1] */
void orc_threshold_schedule(int mi, float *out);

/* MatchGPULib.cpp:332-338 */
void orc_rgb_to_planes(const uint8_t *rgb, int W, int H, int stride, float *planes /*3*W*H*/);

/* MatchLib.cu:71-156,195-278 with the U2/U3 zero-padding interpretation
 * (== convolutionSeparable_gold.cpp:20-75). row then column, row result rounded to f32. */
void orc_conv_rows_zero(float *dst, const float *src, int W, int H, const float taps[5]);
void orc_conv_cols_zero(float *dst, const float *src, int W, int H, const float taps[5]);
/* MatchLib.cu:1461-1565 / 1593-1697 (texture fetch => clamp) */
void orc_conv_rows_clamp(float *dst, const float *src, int W, int H, const float taps[5]);
void orc_conv_cols_clamp(float *dst, const float *src, int W, int H, const float taps[5]);

/* MatchLib.cu:311-339 : dst[x,y] = src[floor((x+.5f)*sf), floor((y+.5f)*sf)] */
void orc_subsample(float *dst, int W2, int H2, const float *src, int W, int H, float sf);

/* MatchGPULib.cpp:1033-1125.  planes0: 3*W*H.  out[l] must hold 3*w[l]*h[l] floats. */
int orc_pyramid(const float *planes0, int W, int H, int levels, float **out);

/* MatchLib.cu:790-843 */
void orc_poly(float c, float l, float r, float thr, float *delta, float *corr);

/* MatchLib.cu:372-401 + MatchGPULib.cpp:1526-1590 : NN upsample x SCALE of 3 planes */
void orc_seed(float *dst, int W2, int H2, const float *src, int W, int H);
/* MatchGPULib.cpp:1595-1655 : upsample to (Wup,Hup) then crop (fovW,fovH) at (l,u) */
void orc_seed_fovea(float *dst, int fovW, int fovH, const float *src, int Wup, int Hup, int l, int u);

/* MatchGPULib.cpp:1662-2489 -- iterations m_from..m_to (1-based, inclusive) of a
 * level with mi total iterations.  L,R: 3 planes each; d: (dx,dy,conf) 3 planes in/out.
 * is_top: this is the coarsest level (conf blend skipped on m==1, :2223).
 * dbg (optional, may be NULL): 8 planes W*H written on the LAST iteration run:
 *   [0..4] Q for shifts (-1,0),(1,0),(0,-1),(0,1),(0,0); [5] dx', [6] dy', [7] kappa
 *   (pre-smoothing, after confidence blend). */
void orc_iterate_level(const float *L, const float *R, float *d, int W, int H,
                       int mi, int S, int is_top, int m_from, int m_to, float *dbg);

/* Single pieces of an iteration, for stage-level tests. */
void orc_smooth_pass(float *dst3, const float *src3, int W, int H);      /* MatchLib.cu:1092-1145, x3 */
void orc_box3(float *d3, int W, int H);                                    /* MatchGPULib.cpp:2361-2412 */

/* MatchGPULib.cpp:303-403 (fov==0).  out: 3 planes W*H (dx,dy,conf). */
int orc_match_full(const uint8_t *rgbL, const uint8_t *rgbR, int W, int H, int stride,
                   int levels, float *out);

/* MatchGPULib.cpp:429-700 (matchStack / matchStackPyramid).  fovea_levels = F (ref 7).
 * stack: 3 * F * fovH * fovW  as [plane c][level k finest first][row][col]
 * pyrL/pyrR (optional): F*3*fovH*fovW as [level k][channel][row][col] (UG_GPU_matcher.cpp:203-213).
 * off_x/off_y: window-centre offset from the image centre in level-0 pixels (0,0 = reference). */
int orc_match_foveated(const uint8_t *rgbL, const uint8_t *rgbR, int W, int H, int stride,
                       int levels, int fovea_levels, int off_x, int off_y,
                       float *stack, float *pyrL, float *pyrR, int *fovW, int *fovH);

/* Fovea geometry shared with orc_match_foveated (generalised window):
 * origin of the fovea crop at level lev (lev < F-1) and crop origin inside the upsampled seed. */
void orc_fovea_geometry(int W, int H, int levels, int F, int off_x, int off_y,
                        int *fovW, int *fovH, int *org_x, int *org_y /*[F-1]*/,
                        int *crop_x, int *crop_y /*[F-1], for transition lev+1 -> lev */);

/* SURVEY 8f row f-1 -- getPointCloud.cpp:886-949 (get3DPoint, non-foveated), P1/P2 row-major 3x4 doubles;
 * xyz: X, Y, Z planes of W*H floats. */
void orc_triangulate(const float *dispx, const float *dispy, int W, int H, const double *P1, const double *P2, float *xyz);

/* getPointCloud.cpp:431-484 (margins) and :387-421 (scale of mapXcoord / mapYcoord). Returns 0 or -1. */
int orc_fovea_mapping(int W, int H, int src_level, int dest_level, int *left_margin, int *upper_margin, float *scale);
/* getPointCloud.cpp:892-903 + the closed form: level src_level of the (F*fovH) x fovW stacks -> X, Y, Z planes fovH x fovW */
void orc_triangulate_fovea(const float *stackx, const float *stacky, int fovW, int fovH, int src_level, int left_margin,
                           int upper_margin, float scale, const double *P1, const double *P2, float *xyz);
/* SURVEY 8f row f-3 -- MatchGPULib.cpp:2589-2701 + MatchLib.cu:435-462.  stack3: 3 x F x fovH x fovW; out3: 3 x H x W. */
int orc_reconstruct_full(const float *stack3, int W, int H, int levels, int F, int off_x, int off_y, float *out3);

/* SURVEY 8f row f-4 -- MatchGPULib.cpp:1323-1437: confidence-weighted mean absolute change of dx (out2[0]) and dy (out2[1])
 * between two (dx, dy, conf) fields, weights = the new field's conf; fixed-order binary64 sums (see the .c file). */
void orc_weighted_difference(const float *newd3, const float *oldd3, int W, int H, float out2[2]);

/* LR-consistency check (north_star; no reference counterpart): zeroes the confidence of left pixels whose match in the
 * right-to-left field does not point back within tau (in x or in y); returns how many.  See the .c file for the definition. */
long orc_lr_check(float *left3, const float *right3, int W, int H, float tau);

int orc_num_threads(void);
void orc_set_num_threads(int n);

#ifdef __cplusplus
}
#endif
#endif
