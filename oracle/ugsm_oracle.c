/*
 * ugsm_oracle.c -- CPU restatement of gerac83/ug_stereomatcher's pyramidal matcher.
 *
 * TEST INFRASTRUCTURE ONLY (see ugsm_oracle.h).  "parity unpinned" by the reference
 * except for the zero-padded blur, which is pinned against the reference's own
 * convolutionSeparable_gold.cpp (oracle/_ref/libgold.so).
 *
 * Every function cites the reference file:line it restates (paths relative to
 * /root/reference/src/gpu_matcher/).  Written from the behaviour of that code, not
 * from its structure: the reference runs ~130 single-op CUDA launches per iteration
 * on textures; here each stage is a plain loop over rows.
 *
 * Build: gcc -O2 -fopenmp -ffp-contract=off -fno-fast-math (oracle/Makefile).
 * Float contract: binary32 everywhere, the CUDA source's implicit double
 * promotions mirrored, a*b+c never fused, taps added in source order.
 * Texture fetches (MatchLib.cu:56-60, never configured => point, clamp,
 * unnormalised) are restated as t[clamp(floor(y))][clamp(floor(x))].
 * Reference UB is resolved as SURVEY.md section 9 lists (U1 zero seed, U2/U3 zero
 * padding on all four sides).
 */
#include "ugsm_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define ORC_SCALE 1.41421356 /* MatchLib_common.h:15 (double literal) */

/* Thread policy.  The GPU box reports hundreds of logical CPUs but gives a job a share of
 * ~16; an OpenMP team sized by omp_get_max_threads() there spins itself to a standstill.
 * The team size is therefore explicit: UGSM_ORACLE_THREADS or 8 by default, and 1 for
 * images too small to amortise a fork/join (results never depend on the team size). */
static int g_threads = 0;
static int cap_threads(void)
{
    if (g_threads > 0) return g_threads;
    const char *e = getenv("UGSM_ORACLE_THREADS");
    int n = e ? atoi(e) : 8;
    return n > 0 ? n : 1;
}
int orc_num_threads(void)
{
#ifdef _OPENMP
    return cap_threads();
#else
    return 1;
#endif
}
void orc_set_num_threads(int n) { g_threads = n; }
static void pick_threads(size_t pixels)
{
#ifdef _OPENMP
    omp_set_dynamic(0);
    omp_set_num_threads(pixels < 40000 ? 1 : cap_threads());
#else
    (void)pixels;
#endif
}

/* ---- constants ------------------------------------------------------------------ */

/* MatchGPULib.cpp:761-774: the sigma=1.1 computation above it is overwritten by
 * literals, which are then renormalised by their sequential float sum. */
void orc_gauss_taps(float g[5])
{
    float k[5];
    k[0] = 0.0816475;
    k[1] = 0.218507;
    k[2] = 0.303281;
    k[3] = 0.218507;
    k[4] = 0.0816475;
    float kern = 0;
    for (int i = 0; i < 5; i++) kern = kern + k[i];
    for (int i = 0; i < 5; i++) g[i] = k[i] / kern;
}

/* MatchGPULib.cpp:344-348 */
void orc_box_taps(float a[5])
{
    a[0] = 0.0;
    a[1] = 0.3333;
    a[2] = 0.3333;
    a[3] = 0.3333;
    a[4] = 0.0;
}

/* MatchGPULib.cpp:1224-1228: int = int / double, truncated. */
int orc_level_dims(int W, int H, int levels, int *w, int *h)
{
    w[0] = W;
    h[0] = H;
    for (int i = 0; i < levels - 1; i++) {
        w[i + 1] = w[i] / ORC_SCALE;
        h[i + 1] = h[i] / ORC_SCALE;
    }
    for (int i = 0; i < levels; i++)
        if (w[i] < 1 || h[i] < 1) return -1;
    return 0;
}

/* MatchGPULib.cpp:1741: mi = ((13-level)>5) ? levelcutoff : ((13-level+1)*2), 13-level == i */
int orc_iterations_for_level(int i) { return (i > 5) ? 22 : ((i + 1) * 2); }

/* MatchGPULib.cpp:2257-2261: realSmoothtime = 5, or 10 when level>11 (i<2) */
int orc_smooth_passes_for_level(int i) { return (i < 2) ? 10 : 5; }

/* MatchGPULib.cpp:1673 (threshold=1.0 on entry) and :2299-2306 (update after the
 * iteration's smoothing, i.e. it affects iteration m+1). */
void orc_threshold_schedule(int mi, float *out)
{
    float threshold = 1.0;
    for (int m = 1; m <= mi; m++) {
        out[m - 1] = threshold;
        if (m % 2 == 0) {
            if ((mi / 2 - m / 2) < 7) {
                threshold = ((mi / 2 - m / 2) - 1) * ((1 - 0.1) / (mi / 2 - 1.0)) + 0.1;
            } else {
                threshold = 1.0;
            }
        }
    }
}

/* ---- input ---------------------------------------------------------------------- */

/* MatchGPULib.cpp:332-338 */
void orc_rgb_to_planes(const uint8_t *rgb, int W, int H, int stride, float *planes)
{
    pick_threads((size_t)W * H);
    for (int k = 0; k < 3; k++) {
        float *p = planes + (size_t)k * W * H;
#pragma omp parallel for schedule(static)
        for (int i = 0; i < H; i++)
            for (int j = 0; j < W; j++) p[(size_t)i * W + j] = (float)rgb[(size_t)i * stride + j * 3 + k];
    }
}

/* ---- texture fetch -------------------------------------------------------------- */

static inline int clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }

/* floor + clamp of an unnormalised texture coordinate; NaN -> 0 (never produced
 * by the algorithm; defined here so oracle and kernels agree). */
static inline int tex_index(float coord, int n)
{
    float f = floorf(coord);
    if (!(f >= 0.0f)) return 0;
    if (f > (float)(n - 1)) return n - 1;
    return (int)f;
}

/* ---- convolutions --------------------------------------------------------------- */

/* MatchLib.cu:127-134: sum=0; for j=-2..2: sum += c_Kernel[R-j]*s[x+j]; outside -> 0
 * (halo loads zero-guarded :105-116; U2: right edge also zero). Identical to
 * convolutionSeparable_gold.cpp:30-41 which skips out-of-range taps. */
void orc_conv_rows_zero(float *dst, const float *src, int W, int H, const float t[5])
{
    pick_threads((size_t)W * H);
#pragma omp parallel for schedule(static)
    for (int y = 0; y < H; y++) {
        const float *s = src + (size_t)y * W;
        float *o = dst + (size_t)y * W;
        for (int x = 0; x < W; x++) {
            float sum = 0;
            for (int j = -2; j <= 2; j++) {
                int xx = x + j;
                if (xx >= 0 && xx < W) sum += t[2 - j] * s[xx];
            }
            o[x] = sum;
        }
    }
}

/* MatchLib.cu:250-259 (+U3) == convolutionSeparable_gold.cpp:59-72 */
void orc_conv_cols_zero(float *dst, const float *src, int W, int H, const float t[5])
{
    pick_threads((size_t)W * H);
#pragma omp parallel for schedule(static)
    for (int y = 0; y < H; y++) {
        float *o = dst + (size_t)y * W;
        for (int x = 0; x < W; x++) {
            float sum = 0;
            for (int j = -2; j <= 2; j++) {
                int yy = y + j;
                if (yy >= 0 && yy < H) sum += t[2 - j] * src[(size_t)yy * W + x];
            }
            o[x] = sum;
        }
    }
}

/* MatchLib.cu:1478-1491 / 1610-1623: sum += tex(x+k, y) * taps[R-k], clamp addressing */
void orc_conv_rows_clamp(float *dst, const float *src, int W, int H, const float t[5])
{
    pick_threads((size_t)W * H);
#pragma omp parallel for schedule(static)
    for (int y = 0; y < H; y++) {
        const float *s = src + (size_t)y * W;
        float *o = dst + (size_t)y * W;
        for (int x = 0; x < W; x++) {
            float sum = 0;
            for (int k = -2; k <= 2; k++) sum += s[clampi(x + k, 0, W - 1)] * t[2 - k];
            o[x] = sum;
        }
    }
}

/* MatchLib.cu:1545-1558 / 1677-1690 */
void orc_conv_cols_clamp(float *dst, const float *src, int W, int H, const float t[5])
{
    pick_threads((size_t)W * H);
#pragma omp parallel for schedule(static)
    for (int y = 0; y < H; y++) {
        float *o = dst + (size_t)y * W;
        for (int x = 0; x < W; x++) {
            float sum = 0;
            for (int k = -2; k <= 2; k++) sum += src[(size_t)clampi(y + k, 0, H - 1) * W + x] * t[2 - k];
            o[x] = sum;
        }
    }
}

/* ---- pyramid -------------------------------------------------------------------- */

/* MatchLib.cu:320-332: x=(float)ix+0.5f; tex2D(src, x*sf, y*sf) */
void orc_subsample(float *dst, int W2, int H2, const float *src, int W, int H, float sf)
{
    pick_threads((size_t)W2 * H2);
#pragma omp parallel for schedule(static)
    for (int iy = 0; iy < H2; iy++) {
        float y = (float)iy + 0.5f;
        int sy = tex_index(y * sf, H);
        for (int ix = 0; ix < W2; ix++) {
            float x = (float)ix + 0.5f;
            int sx = tex_index(x * sf, W);
            dst[(size_t)iy * W2 + ix] = src[(size_t)sy * W + sx];
        }
    }
}

/* MatchGPULib.cpp:1033-1125: two interleaved octave chains.
 *   level 1   = blur(level 0) sampled with sf=(float)SCALE       (:1082-1087)
 *   level i+2 = blur(level i) sampled with sf=2.0f, i < levels-2 (:1088-1096)
 * blur = smem row conv then smem column conv, zero padded (:912-925). */
int orc_pyramid(const float *planes0, int W, int H, int levels, float **out)
{
    int w[ORC_MAX_LEVELS], h[ORC_MAX_LEVELS];
    if (levels < 1 || levels > ORC_MAX_LEVELS) return -1;
    if (orc_level_dims(W, H, levels, w, h)) return -1;
    float g[5];
    orc_gauss_taps(g);
    memcpy(out[0], planes0, sizeof(float) * 3 * (size_t)W * H);
    float *tmp = (float *)malloc(sizeof(float) * (size_t)W * H);
    float *blur = (float *)malloc(sizeof(float) * (size_t)W * H);
    if (!tmp || !blur) { free(tmp); free(blur); return -2; }
    for (int i = 0; i < levels; i++) {
        int need1 = (i == 0 && levels > 1);
        int need2 = (i + 2 < levels);
        if (!need1 && !need2) continue;
        size_t n = (size_t)w[i] * h[i];
        for (int k = 0; k < 3; k++) {
            orc_conv_rows_zero(tmp, out[i] + k * n, w[i], h[i], g);
            orc_conv_cols_zero(blur, tmp, w[i], h[i], g);
            if (need1) {
                float sf = ORC_SCALE; /* :1083 float scalefactors=SCALE */
                orc_subsample(out[1] + (size_t)k * w[1] * h[1], w[1], h[1], blur, w[i], h[i], sf);
            }
            if (need2) {
                float sf = 0.000 + (int)(ORC_SCALE * ORC_SCALE + 0.5); /* :1090 -> 2.0f */
                orc_subsample(out[i + 2] + (size_t)k * w[i + 2] * h[i + 2], w[i + 2], h[i + 2], blur, w[i], h[i], sf);
            }
        }
    }
    free(tmp);
    free(blur);
    return 0;
}

/* ---- parabola ------------------------------------------------------------------- */

/* MatchLib.cu:805-836.  c=centre, l/r = neighbours; thr = current clamp.
 * Double sub-expressions exactly where the CUDA source promotes:
 *   :813 (-b1 * 0.5)/c1         : float*double literal, float divisor -> double
 *   :814 min(thr, max(d, 0.0-thr)) : CUDA's (float,double) overloads -> fmax/fmin in double
 *   :819 cstar > 1.0, :821 d > 1e-10 : double compares
 *   :822 d_hor * ((1.0 - c) / d) : double
 *   :830 0.3 * cstar + 0.7      : double */
void orc_poly(float c, float l, float r, float thr, float *delta, float *corr)
{
    float b1 = (r - l) / 2;
    float c1 = r - (c + b1);
    if (c1 < 0) {
        float dh = (-b1 * 0.5) / c1;
        dh = fmin((double)thr, fmax((double)dh, (0.0 - thr)));
        float cstar = (c1 * dh + b1) * dh + c;
        if (cstar > 1.0) {
            float d = cstar - c;
            if (d > 1e-10) {
                dh = dh * ((1.0 - c) / d);
            }
            *delta = dh;
            *corr = 1.0;
        } else {
            *delta = dh;
            *corr = 0.3 * cstar + 0.7;
        }
    } else {
        *delta = 0.0;
        *corr = 0.4;
    }
}

/* ---- seeding -------------------------------------------------------------------- */

/* MatchLib.cu:381-394: dst = SCALE * tex(src, x*sf, y*sf), sf=(float)(1/SCALE)
 * (MatchGPULib.cpp:1222), SCALE*src in double then stored to float. 3 planes. */
void orc_seed(float *dst, int W2, int H2, const float *src, int W, int H)
{
    pick_threads((size_t)W2 * H2);
    const float sf = 1 / ORC_SCALE;
    for (int c = 0; c < 3; c++) {
        const float *s = src + (size_t)c * W * H;
        float *o = dst + (size_t)c * W2 * H2;
#pragma omp parallel for schedule(static)
        for (int iy = 0; iy < H2; iy++) {
            float y = (float)iy + 0.5f;
            int sy = tex_index(y * sf, H);
            for (int ix = 0; ix < W2; ix++) {
                float x = (float)ix + 0.5f;
                int sx = tex_index(x * sf, W);
                float v = s[(size_t)sy * W + sx];
                o[(size_t)iy * W2 + ix] = ORC_SCALE * v;
            }
        }
    }
}

/* MatchGPULib.cpp:1595-1655: src is fovW x fovH; upsample to Wup x Hup
 * (:1628-1636), then copy rows u..u+fovH-1, cols l..l+fovW-1 (:1642-1644). */
void orc_seed_fovea(float *dst, int fovW, int fovH, const float *src, int Wup, int Hup, int l, int u)
{
    pick_threads((size_t)fovW * fovH);
    const float sf = 1 / ORC_SCALE;
    for (int c = 0; c < 3; c++) {
        const float *s = src + (size_t)c * fovW * fovH;
        float *o = dst + (size_t)c * fovW * fovH;
#pragma omp parallel for schedule(static)
        for (int i = 0; i < fovH; i++) {
            float y = (float)(u + i) + 0.5f;
            int sy = tex_index(y * sf, fovH);
            for (int j = 0; j < fovW; j++) {
                float x = (float)(l + j) + 0.5f;
                int sx = tex_index(x * sf, fovW);
                float v = s[(size_t)sy * fovW + sx];
                o[(size_t)i * fovW + j] = ORC_SCALE * v;
            }
        }
    }
    (void)Hup;
    (void)Wup;
}

/* ---- one level ------------------------------------------------------------------ */

/* MatchLib.cu:1108-1133: confidence-weighted cross mean; ix>0 && iy>0 only (:1106),
 * other pixels keep their value (in-place destination, MatchGPULib.cpp:2269-2289).
 * All three planes use the PRE-pass confidence as weight (:2264-2266). */
void orc_smooth_pass(float *dst3, const float *src3, int W, int H)
{
    pick_threads((size_t)W * H);
    size_t n = (size_t)W * H;
    const float *conf = src3 + 2 * n;
    for (int p = 0; p < 3; p++) {
        const float *s = src3 + p * n;
        float *o = dst3 + p * n;
#pragma omp parallel for schedule(static)
        for (int iy = 0; iy < H; iy++) {
            for (int ix = 0; ix < W; ix++) {
                size_t at = (size_t)iy * W + ix;
                if (ix > 0 && iy > 0) {
                    int xm = ix - 1, xp = clampi(ix + 1, 0, W - 1);
                    int ym = iy - 1, yp = clampi(iy + 1, 0, H - 1);
                    float sumDisp = 0, sumCorr = 0;
                    float v, w;
                    v = s[at];                      w = conf[at];
                    sumDisp = v * w + sumDisp;      sumCorr = sumCorr + w;
                    v = s[(size_t)iy * W + xm];     w = conf[(size_t)iy * W + xm];
                    sumDisp = v * w + sumDisp;      sumCorr = sumCorr + w;
                    v = s[(size_t)iy * W + xp];     w = conf[(size_t)iy * W + xp];
                    sumDisp = v * w + sumDisp;      sumCorr = sumCorr + w;
                    v = s[(size_t)ym * W + ix];     w = conf[(size_t)ym * W + ix];
                    sumDisp = v * w + sumDisp;      sumCorr = sumCorr + w;
                    v = s[(size_t)yp * W + ix];     w = conf[(size_t)yp * W + ix];
                    sumDisp = v * w + sumDisp;      sumCorr = sumCorr + w;
                    o[at] = sumDisp / sumCorr;
                } else {
                    o[at] = s[at];
                }
            }
        }
    }
}

/* MatchGPULib.cpp:2361-2412: rows (Ta) then columns (Ta) on each of dx, dy, conf */
void orc_box3(float *d3, int W, int H)
{
    size_t n = (size_t)W * H;
    float a[5];
    orc_box_taps(a);
    float *tmp = (float *)malloc(sizeof(float) * n);
    for (int p = 0; p < 3; p++) {
        orc_conv_rows_clamp(tmp, d3 + p * n, W, H, a);
        orc_conv_cols_clamp(d3 + p * n, tmp, W, H, a);
    }
    free(tmp);
}

/* MatchLib.cu:686-687 / 1006-1007: if(v>1) v=1; if(v<0) v=0;  (NaN stays NaN) */
static inline float clamp01(float v)
{
    if (v > 1) v = 1.0;
    if (v < 0) v = 0.0;
    return v;
}

void orc_iterate_level(const float *L, const float *R, float *d, int W, int H, int mi, int S, int is_top,
                       int m_from, int m_to, float *dbg)
{
    pick_threads((size_t)W * H);
    const size_t n = (size_t)W * H;
    float g[5];
    orc_gauss_taps(g);
    /* MatchGPULib.cpp:1677 move[] built once from threshold=1.0 */
    static const int mvx[5] = {-1, 1, 0, 0, 0};
    static const int mvy[5] = {0, 0, -1, 1, 0};
    float *thr = (float *)malloc(sizeof(float) * (mi > 0 ? mi : 1));
    orc_threshold_schedule(mi, thr);

    float *Rw = (float *)malloc(sizeof(float) * n);
    float *t0 = (float *)malloc(sizeof(float) * n);
    float *t1 = (float *)malloc(sizeof(float) * n);
    float *A = (float *)malloc(sizeof(float) * n);
    float *B = (float *)malloc(sizeof(float) * n);
    float *Q = (float *)malloc(sizeof(float) * 5 * n);
    float *nd = (float *)malloc(sizeof(float) * 3 * n);
    float *sm = (float *)malloc(sizeof(float) * 3 * n);

    float *dx = d, *dy = d + n, *cf = d + 2 * n;

    for (int m = m_from; m <= m_to; m++) {
        for (int k = 0; k < 3; k++) {
            const float *Lk = L + k * n;
            const float *Rk = R + k * n;
            /* warp, MatchLib.cu:510-515: tex(R, x+dx, y+dy), x=(float)ix+0.5f */
#pragma omp parallel for schedule(static)
            for (int iy = 0; iy < H; iy++) {
                float y = (float)iy + 0.5f;
                for (int ix = 0; ix < W; ix++) {
                    float x = (float)ix + 0.5f;
                    size_t at = (size_t)iy * W + ix;
                    int sx = tex_index(x + dx[at], W);
                    int sy = tex_index(y + dy[at], H);
                    Rw[at] = Rk[(size_t)sy * W + sx];
                }
            }
            /* squares (MatchLib.cu:569-570) and their clamp-addressed blur
             * (MatchGPULib.cpp:1863-1901) */
#pragma omp parallel for schedule(static)
            for (size_t i = 0; i < n; i++) t0[i] = Lk[i] * Lk[i];
            orc_conv_rows_clamp(t1, t0, W, H, g);
            orc_conv_cols_clamp(A, t1, W, H, g);
#pragma omp parallel for schedule(static)
            for (size_t i = 0; i < n; i++) t0[i] = Rw[i] * Rw[i];
            orc_conv_rows_clamp(t1, t0, W, H, g);
            orc_conv_cols_clamp(B, t1, W, H, g);

            for (int s = 0; s < 5; s++) {
                const int sx = mvx[s], sy = mvy[s];
                /* CompareMove MatchLib.cu:622-624: L(x,y) * R'(x+sx, y+sy) clamp */
#pragma omp parallel for schedule(static)
                for (int iy = 0; iy < H; iy++) {
                    int yy = clampi(iy + sy, 0, H - 1);
                    for (int ix = 0; ix < W; ix++) {
                        int xx = clampi(ix + sx, 0, W - 1);
                        t0[(size_t)iy * W + ix] = Lk[(size_t)iy * W + ix] * Rw[(size_t)yy * W + xx];
                    }
                }
                /* smem blur of the product, zero padded (MatchGPULib.cpp:1932-1945) */
                orc_conv_rows_zero(t1, t0, W, H, g);
                orc_conv_cols_zero(t0, t1, W, H, g);
                /* MoveCorrelation MatchLib.cu:681-687 and channel accumulate
                 * MatchGPULib.cpp:2033-2070: q0 ; q1+q0 ; ((q0+q1)+q2)/3 */
                float *Qs = Q + s * n;
#pragma omp parallel for schedule(static)
                for (int iy = 0; iy < H; iy++) {
                    int yy = clampi(iy + sy, 0, H - 1);
                    for (int ix = 0; ix < W; ix++) {
                        int xx = clampi(ix + sx, 0, W - 1);
                        size_t at = (size_t)iy * W + ix;
                        float src = t0[at];
                        float warpl = A[at];
                        float warpr = B[(size_t)yy * W + xx];
                        float q = clamp01((src * src) / (warpl * warpr));
                        if (k == 0) Qs[at] = q;
                        else if (k == 1) Qs[at] = q + Qs[at];
                        else Qs[at] = (Qs[at] + q) / 3.0f;
                    }
                }
            }
        }

        /* parabola x/y (MatchGPULib.cpp:2129-2152), corr product (:2159), update
         * (:2206-2220), confidence blend (:2223-2250) */
        const float th = thr[m - 1];
        const int blend = !(is_top && m == 1);
#pragma omp parallel for schedule(static)
        for (size_t i = 0; i < n; i++) {
            float ddx, ddy, cx, cy;
            orc_poly(Q[4 * n + i], Q[0 * n + i], Q[1 * n + i], th, &ddx, &cx);
            orc_poly(Q[4 * n + i], Q[2 * n + i], Q[3 * n + i], th, &ddy, &cy);
            float kap = cy * cx;
            nd[i] = dx[i] + ddx;
            nd[n + i] = dy[i] + ddy;
            if (blend) {
                float v = 0.75 * cf[i] + 0.25 * kap;
                kap = clamp01(v);
            }
            nd[2 * n + i] = kap;
        }
        if (dbg && m == m_to) {
            memcpy(dbg, Q, sizeof(float) * 5 * n);
            memcpy(dbg + 5 * n, nd, sizeof(float) * 3 * n);
        }
        /* S Jacobi passes (MatchGPULib.cpp:2262-2292) */
        float *a = nd, *b = sm;
        for (int j = 0; j < S; j++) {
            orc_smooth_pass(b, a, W, H);
            float *t = a; a = b; b = t;
        }
        /* box (MatchGPULib.cpp:2361-2412), then keep on "device" (:2420-2426) */
        orc_box3(a, W, H);
        memcpy(d, a, sizeof(float) * 3 * n);
    }
    free(thr); free(Rw); free(t0); free(t1); free(A); free(B); free(Q); free(nd); free(sm);
}

/* ---- drivers -------------------------------------------------------------------- */

/* MatchGPULib.cpp:1196-1318 without fovea. pl/pr: per-level 3-plane images. */
static void matching_full(float **pl, float **pr, const int *w, const int *h, int levels, float *out)
{
    float *cur = (float *)calloc((size_t)3 * w[levels - 1] * h[levels - 1], sizeof(float)); /* U1: zeros */
    for (int i = levels - 1; i >= 0; i--) {
        int mi = orc_iterations_for_level(i);
        orc_iterate_level(pl[i], pr[i], cur, w[i], h[i], mi, orc_smooth_passes_for_level(i), i == levels - 1, 1, mi, NULL);
        if (i > 0) {
            float *nxt = (float *)malloc(sizeof(float) * 3 * (size_t)w[i - 1] * h[i - 1]);
            orc_seed(nxt, w[i - 1], h[i - 1], cur, w[i], h[i]);
            free(cur);
            cur = nxt;
        }
    }
    memcpy(out, cur, sizeof(float) * 3 * (size_t)w[0] * h[0]);
    free(cur);
}

static int build_pyramids(const uint8_t *rgbL, const uint8_t *rgbR, int W, int H, int stride, int levels,
                          const int *w, const int *h, float **pl, float **pr)
{
    float *p0 = (float *)malloc(sizeof(float) * 3 * (size_t)W * H);
    if (!p0) return -2;
    for (int i = 0; i < levels; i++) {
        pl[i] = (float *)malloc(sizeof(float) * 3 * (size_t)w[i] * h[i]);
        pr[i] = (float *)malloc(sizeof(float) * 3 * (size_t)w[i] * h[i]);
    }
    orc_rgb_to_planes(rgbL, W, H, stride, p0);
    orc_pyramid(p0, W, H, levels, pl);
    orc_rgb_to_planes(rgbR, W, H, stride, p0);
    orc_pyramid(p0, W, H, levels, pr);
    free(p0);
    return 0;
}

int orc_match_full(const uint8_t *rgbL, const uint8_t *rgbR, int W, int H, int stride, int levels, float *out)
{
    int w[ORC_MAX_LEVELS], h[ORC_MAX_LEVELS];
    float *pl[ORC_MAX_LEVELS], *pr[ORC_MAX_LEVELS];
    if (levels < 1 || levels > ORC_MAX_LEVELS) return -1;
    if (orc_level_dims(W, H, levels, w, h)) return -1;
    if (build_pyramids(rgbL, rgbR, W, H, stride, levels, w, h, pl, pr)) return -2;
    matching_full(pl, pr, w, h, levels, out);
    for (int i = 0; i < levels; i++) { free(pl[i]); free(pr[i]); }
    return 0;
}

/* Fovea geometry.  Reference (centred, off=0):
 *   fovW,fovH = w[F-1],h[F-1]                         MatchGPULib.cpp:1143-1144,1233-1234
 *   origin at level lev<F-1: (w/2 - fovW/2, h/2 - fovH/2)        :1145-1146,1173-1176
 *   seed crop origin: (w[F-2]/2 - fovW/2, h[F-2]/2 - fovH/2)     :1612-1615
 * Generalisation (not in the reference): the window centre may be offset by
 * (off_x, off_y) level-0 pixels; per level the offset is off/sqrt2^lev rounded,
 * clamped so the window stays inside the level; the seed crop moves by the part
 * of the child's offset that the parent's (x sqrt2) does not already carry. */
void orc_fovea_geometry(int W, int H, int levels, int F, int off_x, int off_y, int *fovW, int *fovH,
                        int *org_x, int *org_y, int *crop_x, int *crop_y)
{
    int w[ORC_MAX_LEVELS], h[ORC_MAX_LEVELS];
    orc_level_dims(W, H, levels, w, h);
    const int fw = w[F - 1], fh = h[F - 1];
    *fovW = fw;
    *fovH = fh;
    int ex[ORC_MAX_LEVELS], ey[ORC_MAX_LEVELS]; /* effective offsets after clamping */
    ex[F - 1] = 0;
    ey[F - 1] = 0;
    for (int lev = F - 2; lev >= 0; lev--) {
        int cx = w[lev] / 2 - fw / 2, cy = h[lev] / 2 - fh / 2;
        int ox = cx + (int)lrint(off_x / pow(ORC_SCALE, lev));
        int oy = cy + (int)lrint(off_y / pow(ORC_SCALE, lev));
        ox = clampi(ox, 0, w[lev] - fw);
        oy = clampi(oy, 0, h[lev] - fh);
        org_x[lev] = ox;
        org_y[lev] = oy;
        ex[lev] = ox - cx;
        ey[lev] = oy - cy;
    }
    const int Wup = w[F - 2 >= 0 ? F - 2 : 0], Hup = h[F - 2 >= 0 ? F - 2 : 0];
    for (int lev = F - 2; lev >= 0; lev--) {
        int lx = Wup / 2 - fw / 2 + ex[lev] - (int)lrint(ORC_SCALE * ex[lev + 1]);
        int ly = Hup / 2 - fh / 2 + ey[lev] - (int)lrint(ORC_SCALE * ey[lev + 1]);
        crop_x[lev] = clampi(lx, 0, Wup - fw);
        crop_y[lev] = clampi(ly, 0, Hup - fh);
    }
}

int orc_match_foveated(const uint8_t *rgbL, const uint8_t *rgbR, int W, int H, int stride, int levels,
                       int F, int off_x, int off_y, float *stack, float *pyrL, float *pyrR, int *fovW_out,
                       int *fovH_out)
{
    int w[ORC_MAX_LEVELS], h[ORC_MAX_LEVELS];
    float *pl[ORC_MAX_LEVELS], *pr[ORC_MAX_LEVELS];
    if (levels < 2 || levels > ORC_MAX_LEVELS || F < 2 || F > levels) return -1;
    if (orc_level_dims(W, H, levels, w, h)) return -1;
    int fw, fh, ox[ORC_MAX_LEVELS], oy[ORC_MAX_LEVELS], cx[ORC_MAX_LEVELS], cy[ORC_MAX_LEVELS];
    orc_fovea_geometry(W, H, levels, F, off_x, off_y, &fw, &fh, ox, oy, cx, cy);
    *fovW_out = fw;
    *fovH_out = fh;
    if (build_pyramids(rgbL, rgbR, W, H, stride, levels, w, h, pl, pr)) return -2;

    /* CreateFoveatedPyramid MatchGPULib.cpp:1171-1185: crop levels 0..F-2 */
    const size_t fn = (size_t)fw * fh;
    for (int lev = F - 2; lev >= 0; lev--) {
        float *cl = (float *)malloc(sizeof(float) * 3 * fn);
        float *cr = (float *)malloc(sizeof(float) * 3 * fn);
        for (int k = 0; k < 3; k++)
            for (int i = 0; i < fh; i++) {
                memcpy(cl + k * fn + (size_t)i * fw, pl[lev] + (size_t)k * w[lev] * h[lev] + (size_t)(oy[lev] + i) * w[lev] + ox[lev], sizeof(float) * fw);
                memcpy(cr + k * fn + (size_t)i * fw, pr[lev] + (size_t)k * w[lev] * h[lev] + (size_t)(oy[lev] + i) * w[lev] + ox[lev], sizeof(float) * fw);
            }
        free(pl[lev]);
        free(pr[lev]);
        pl[lev] = cl;
        pr[lev] = cr;
    }
    /* pyramid stacks as the node packs them (UG_GPU_matcher.cpp:203-213) */
    for (int k = 0; k < F; k++) {
        if (pyrL) memcpy(pyrL + (size_t)k * 3 * fn, pl[k], sizeof(float) * 3 * fn);
        if (pyrR) memcpy(pyrR + (size_t)k * 3 * fn, pr[k], sizeof(float) * 3 * fn);
    }

    /* matching() with foveatedmatching==1, MatchGPULib.cpp:1230-1294 */
    int mw[ORC_MAX_LEVELS], mh[ORC_MAX_LEVELS];
    for (int i = 0; i < levels; i++) { mw[i] = (i < F - 1) ? fw : w[i]; mh[i] = (i < F - 1) ? fh : h[i]; }
    float *cur = (float *)calloc((size_t)3 * mw[levels - 1] * mh[levels - 1], sizeof(float));
    for (int i = levels - 1; i >= 0; i--) {
        int mi = orc_iterations_for_level(i);
        orc_iterate_level(pl[i], pr[i], cur, mw[i], mh[i], mi, orc_smooth_passes_for_level(i), i == levels - 1, 1, mi, NULL);
        if (i < F) {
            /* node packing UG_GPU_matcher.cpp:293-303: level k at rows k*fovH.. */
            for (int c = 0; c < 3; c++)
                memcpy(stack + ((size_t)c * F + i) * fn, cur + c * fn, sizeof(float) * fn);
        }
        if (i > 0) {
            float *nxt = (float *)malloc(sizeof(float) * 3 * (size_t)mw[i - 1] * mh[i - 1]);
            if (i >= F) orc_seed(nxt, mw[i - 1], mh[i - 1], cur, mw[i], mh[i]);           /* :1283-1287 */
            else orc_seed_fovea(nxt, fw, fh, cur, w[F - 2], h[F - 2], cx[i - 1], cy[i - 1]); /* :1288-1292 */
            free(cur);
            cur = nxt;
        }
    }
    free(cur);
    for (int i = 0; i < levels; i++) { free(pl[i]); free(pr[i]); }
    return 0;
}

/* ---- triangulation of the full-resolution disparity -----------------------
 * src/pointcloud/getPointCloud.cpp:886-949 (CdynamicCalibration::get3DPoint, non-foveated branch
 * :909-914), called per pixel by the reconstruction loops (:640-660 with sampling, :778).
 * P1, P2: 3x4 projection matrices, row major, double (Mat_<double>).  The closed form mixes float
 * and double exactly as the C++ source does: a..j,x,y are floats; pow(v,2.0) promotes to double
 * (v*v is exact in binary64 for a binary32 v, so the product restates pow); literals 2.0 are double.
 * No contraction (host code built without -mfma). out: X, Y, Z planes. */
static inline double sq_d(float v) { return (double)v * (double)v; }

/* The closed form of get3DPoint (getPointCloud.cpp:908-948) for one left/right correspondence. */
static inline void tri_point(float x1, float y1, float x2, float y2, const double *P1, const double *P2, float *X, float *Y, float *Z)
{
    float a, b, c, d, e, f, g, h, i, j, x, y;
    a = (float)P1[0];
    b = (float)(P1[2] - x1);
    c = (float)P1[5];
    d = (float)(P1[6] - y1);
    e = (float)(P2[0] - x2 * P2[8]);
    f = (float)(P2[1] - x2 * P2[9]);
    g = (float)(P2[2] - x2 * P2[10]);
    h = (float)(P2[4] - y2 * P2[8]);
    i = (float)(P2[5] - y2 * P2[9]);
    j = (float)(P2[6] - y2 * P2[10]);
    x = (float)(x2 * P2[11] - P2[3]);
    y = (float)(y2 * P2[11] - P2[7]);
    float XUp = (d*f*h - c*g*h - d*e*i + c*e*j)*(-(d*i*x) + c*j*x + d*f*y - c*g*y) +
                sq_d(b)*((f*h - e*i)*(-(i*x) + f*y) + sq_d(c)*(e*x + h*y)) +
                a*b*((-(g*i) + f*j)*(i*x - f*y) + c*d*(f*x + i*y) - sq_d(c)*(g*x + j*y));
    float YUp = (sq_d(b)*(f*h - e*i) + d*(d*f*h - c*g*h - d*e*i + c*e*j))*(h*x - e*y) +
                a*b*((c*d*e + g*h*i - 2.0*f*h*j + e*i*j)*x + (c*d*h + f*g*h - 2.0*e*g*i + e*f*j)*y) +
                sq_d(a)*((g*i - f*j)*(-(j*x) + g*y) + sq_d(d)*(f*x + i*y) - c*d*(g*x + j*y));
    float ZUp = c*(-(d*f*h) + c*g*h + d*e*i - c*e*j)*(h*x - e*y) - a*b*((f*h - e*i)*(-(i*x) + f*y) +
                sq_d(c)*(e*x + h*y)) + sq_d(a)*((g*i - f*j)*(i*x - f*y) - c*d*(f*x + i*y) +
                sq_d(c)*(g*x + j*y));
    float divisor = sq_d(b)*(sq_d(c)*(sq_d(e) + sq_d(h)) + sq_d(f*h - e*i)) +
                    sq_d(d*f*h - c*g*h - d*e*i + c*e*j) - 2.0*a*b*(-(c*d*(e*f + h*i)) +
                    (f*h - e*i)*(-(g*i) + f*j) + sq_d(c)*(e*g + h*j)) + sq_d(a)*
                    (sq_d(d)*(sq_d(f) + sq_d(i)) + sq_d(g*i - f*j) - 2.0*c*d*(f*g + i*j) +
                    sq_d(c)*(sq_d(g) + sq_d(j)));
    *X = XUp / divisor;
    *Y = YUp / divisor;
    *Z = ZUp / divisor;
}

void orc_triangulate(const float *dispx, const float *dispy, int W, int H, const double *P1, const double *P2, float *xyz)
{
    const size_t n = (size_t)W * H;
    pick_threads(n);
#pragma omp parallel for schedule(static)
    for (int yy = 0; yy < H; yy++) {
        for (int xx = 0; xx < W; xx++) {
            const size_t at = (size_t)yy * W + xx;
            float x1, x2, y1, y2;
            x1 = xx;
            y1 = yy;
            x2 = xx + dispx[at];
            y2 = yy + dispy[at];
            tri_point(x1, y1, x2, y2, P1, P2, &xyz[at], &xyz[n + at], &xyz[2 * n + at]);
        }
    }
}

/* getPointCloud.cpp:387-484: where level `src_level` of the fovea stack sits in level `dest_level` of the
 * full pyramid (left_marginOf_in / upper_marginOf_in) and the coordinate scale of mapXcoord / mapYcoord
 * (pow(float, float) on the float-rounded M_SQRT2 / M_SQRT1_2). */
int orc_fovea_mapping(int W, int H, int src_level, int dest_level, int *left_margin, int *upper_margin, float *scale)
{
    int w[ORC_MAX_LEVELS], h[ORC_MAX_LEVELS];
    int scaled = 6 - src_level;
    if (src_level < dest_level) scaled = src_level + dest_level;
    if (scaled < 0 || scaled >= 15 || dest_level < 0 || dest_level >= 15) return -1;
    w[0] = W;
    h[0] = H;
    for (int i = 0; i < 14; i++) {
        w[i + 1] = (int)(w[i] / 1.41421356);
        h[i + 1] = (int)(h[i] / 1.41421356);
    }
    *left_margin = w[dest_level] / 2 - w[scaled] / 2;
    *upper_margin = h[dest_level] / 2 - h[scaled] / 2;
    float root = (src_level < dest_level) ? (float)0.70710678118654752440 : (float)1.41421356237309504880;
    *scale = powf(root, (float)abs(src_level - dest_level));
    return 0;
}

/* get3DPoint, foveated branch (getPointCloud.cpp:892-903) for every pixel of level `src_level` of the
 * (F*fovH) x fovW disparity stacks.  mapXcoord / mapYcoord take an int: the right-image coordinate
 * xx + disparity is truncated toward zero before it is scaled (as in the reference). */
void orc_triangulate_fovea(const float *stackx, const float *stacky, int fovW, int fovH, int src_level, int left_margin,
                           int upper_margin, float scale, const double *P1, const double *P2, float *xyz)
{
    const size_t n = (size_t)fovW * fovH;
    pick_threads(n);
#pragma omp parallel for schedule(static)
    for (int yy = 0; yy < fovH; yy++) {
        for (int xx = 0; xx < fovW; xx++) {
            const size_t at = (size_t)yy * fovW + xx;
            const size_t sat = ((size_t)yy + (size_t)fovH * src_level) * fovW + xx;
            const float x1 = (float)left_margin + (float)xx * scale;
            const float y1 = (float)upper_margin + (float)yy * scale;
            const int sx = (int)(xx + stackx[sat]);
            const int sy = (int)(yy + stacky[sat]);
            const float x2 = (float)left_margin + (float)sx * scale;
            const float y2 = (float)upper_margin + (float)sy * scale;
            tri_point(x1, y1, x2, y2, P1, P2, &xyz[at], &xyz[n + at], &xyz[2 * n + at]);
        }
    }
}

/* hierarchicalDisparity (MatchGPULib.cpp:2589-2701) + partsubsampleDispKernel (MatchLib.cu:435-462):
 * full-frame field from the foveated stack.  Level F-1 (whole frame at w[F-1] x h[F-1]) is upsampled level by
 * level, out[x,y] = s * in[tex((x+.5)/s), tex((y+.5)/s)] with s = (float)SCALE applied to every channel, and the
 * fovea of the finer level is pasted at its crop origin.  stack3: 3 planes of F x fovH x fovW (level 0 first). */
int orc_reconstruct_full(const float *stack3, int W, int H, int levels, int F, int off_x, int off_y, float *out3)
{
    int w[ORC_MAX_LEVELS], h[ORC_MAX_LEVELS];
    if (orc_level_dims(W, H, levels, w, h)) return -1;
    if (F < 1 || F > levels) return -1;
    int fovW, fovH, ox[ORC_MAX_LEVELS], oy[ORC_MAX_LEVELS], cx[ORC_MAX_LEVELS], cy[ORC_MAX_LEVELS];
    orc_fovea_geometry(W, H, levels, F, off_x, off_y, &fovW, &fovH, ox, oy, cx, cy);
    const size_t fl = (size_t)fovW * fovH;
    const float s = (float)1.41421356;
    float *cur = (float *)malloc(3 * fl * sizeof(float));
    for (int c = 0; c < 3; c++) memcpy(cur + c * fl, stack3 + ((size_t)c * F + (F - 1)) * fl, fl * sizeof(float));
    int cw = w[F - 1], ch = h[F - 1];
    for (int level = F - 1; level > 0; level--) {
        const int nw = w[level - 1], nh = h[level - 1];
        float *nxt = (level == 1) ? out3 : (float *)malloc(3 * (size_t)nw * nh * sizeof(float));
        pick_threads((size_t)nw * nh);
        for (int c = 0; c < 3; c++) {
            const float *src = cur + (size_t)c * cw * ch;
            const float *fov = stack3 + ((size_t)c * F + (level - 1)) * fl;
            float *dst = nxt + (size_t)c * nw * nh;
#pragma omp parallel for schedule(static)
            for (int iy = 0; iy < nh; iy++) {
                const int ty = tex_index(((float)iy + 0.5f) / s, ch);
                for (int ix = 0; ix < nw; ix++) {
                    const int fx = ix - ox[level - 1], fy = iy - oy[level - 1];
                    if (fx >= 0 && fx < fovW && fy >= 0 && fy < fovH) dst[(size_t)iy * nw + ix] = fov[(size_t)fy * fovW + fx];
                    else dst[(size_t)iy * nw + ix] = s * src[(size_t)ty * cw + tex_index(((float)ix + 0.5f) / s, cw)];
                }
            }
        }
        free(cur);
        cur = nxt;
        cw = nw;
        ch = nh;
    }
    if (F == 1) {
        memcpy(out3, cur, 3 * fl * sizeof(float));
        free(cur);
    }
    return 0;
}

/* ---- convergence measure of the reference's (never called) early exit -------------------------
 * MatchGPULib.cpp:1323-1437 (differenceIterations / weightedDifference) with kernels 17, 18 (MatchLib.cu:1174-1373):
 * theDif = sum(|D - OldD| * conf) / sum(conf); the iteration would stop when theDif of both disparities is below a threshold.
 * The per-pixel term is the reference's, in float (MatchLib.cu:1194-1199: abs(a - b), then times conf).  Its sum is NOT
 * defined by the reference: reduceGPU is called with the block count in place of the block size (:1363-1400) and the order of
 * a tree reduction is an implementation detail.  Defined interpretation of this build: binary64
 * sums in a fixed order -- per row, 64 column classes x = l (mod 64) each summed left to right, the classes added l = 0..63;
 * the rows likewise in 64 row classes -- and theDif = (float)(S / C). */
void orc_weighted_difference(const float *newd3, const float *oldd3, int W, int H, float out2[2])
{
    const size_t n = (size_t)W * H;
    const float *conf = newd3 + 2 * n;
    double ts[2][64], tc[64];
    for (int l = 0; l < 64; l++) ts[0][l] = ts[1][l] = tc[l] = 0.0;
    for (int y = 0; y < H; y++) {
        double ps[2][64], pc[64];
        for (int l = 0; l < 64; l++) ps[0][l] = ps[1][l] = pc[l] = 0.0;
        for (int x = 0; x < W; x++) {
            const size_t at = (size_t)y * W + x;
            const float c = conf[at];
            for (int k = 0; k < 2; k++) {
                float t = fabsf(newd3[k * n + at] - oldd3[k * n + at]);
                t = t * c;
                ps[k][x & 63] += (double)t;
            }
            pc[x & 63] += (double)c;
        }
        double rs[2] = {0.0, 0.0}, rc = 0.0;
        for (int l = 0; l < 64; l++) { rs[0] += ps[0][l]; rs[1] += ps[1][l]; rc += pc[l]; }
        ts[0][y & 63] += rs[0];
        ts[1][y & 63] += rs[1];
        tc[y & 63] += rc;
    }
    double S[2] = {0.0, 0.0}, C = 0.0;
    for (int l = 0; l < 64; l++) { S[0] += ts[0][l]; S[1] += ts[1][l]; C += tc[l]; }
    out2[0] = (float)(S[0] / C);
    out2[1] = (float)(S[1] / C);
}


/* LR-consistency check.  Named by BASELINE.json's north_star; the REFERENCE HAS NONE (no right-to-left pass anywhere in
 * MatchLib.cu / MatchGPULib.cpp), so this restates the build's own definition, opt-in and OFF in
 * every parity run.  left3 / right3: (dx, dy, conf) of the left-to-right match and of the match with the images exchanged.  The
 * match of left pixel (x, y) is right pixel (x + dx, y + dy) (getPointCloud.cpp:910-913); the right field is fetched there the way
 * the matcher fetches -- nearest neighbour, clamped: tex_index on the same float coordinate as the warp (MatchLib.cu:510-515) --
 * and should point back.  Where !(|dxL + dxR'| <= tau) or !(|dyL + dyR'| <= tau) (so also where either sum is a NaN) the left
 * confidence becomes 0; dx and dy stay.  Returns the number of pixels marked. */
long orc_lr_check(float *left3, const float *right3, int W, int H, float tau)
{
    const size_t n = (size_t)W * H;
    long marked = 0;
    for (int iy = 0; iy < H; iy++) {
        const float y = (float)iy + 0.5f;
        for (int ix = 0; ix < W; ix++) {
            const float x = (float)ix + 0.5f;
            const size_t at = (size_t)iy * W + ix;
            const float dxl = left3[at], dyl = left3[n + at];
            const int sx = tex_index(x + dxl, W), sy = tex_index(y + dyl, H);
            const size_t rt = (size_t)sy * W + sx;
            const float ex = fabsf(dxl + right3[rt]), ey = fabsf(dyl + right3[n + rt]);
            if (!(ex <= tau) || !(ey <= tau)) {
                left3[2 * n + at] = 0.0f;
                marked++;
            }
        }
    }
    return marked;
}
