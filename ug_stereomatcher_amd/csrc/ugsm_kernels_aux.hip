// ugsm_kernels_aux.hip -- the plain per-pixel kernels around the matcher proper.
//
// k_seed (a level's starting field where the next level's K-cost does not seed itself), k_copy_view (the fovea / pyramid stacks),
// k_rgb_planes (level 0 of a pyramid of fewer than three levels: BASELINE configs[0]), k_lr_check (the opt-in LR-consistency check),
// k_triangulate[_fovea], k_upsample_paste (row f-3), k_wdiff_* (row f-4).  All HBM- or launch-bound.
// Citations: /root/reference/src/gpu_matcher/<file>:<line> unless a path is given.
#include "ugsm_device.hpp"
#include "ugsm_launch.hpp"

namespace ugsm {

static inline dim3 grid2(int W, int H, int z = 1) { return dim3((W + 255) / 256, H, z); }

// --------------------------------------------------------------------------------------
// MatchLib.cu:372-401 (+ fovea crop MatchGPULib.cpp:1642-1644):
// dst[x,y] = f32(SCALE * src[floor((x+cx+.5f)*sf), floor((y+cy+.5f)*sf)]), sf=(float)(1/SCALE)
// (batched: blockIdx.z = 3 x pair + plane)
__global__ void k_seed(const float *__restrict__ src3, int Ws, int Hs, float *__restrict__ dst3, int Wd, int Hd, int cx, int cy, Batch bt)
{
    int ix = blockIdx.x * blockDim.x + threadIdx.x;
    int iy = blockIdx.y;
    int plane = blockIdx.z;
    if (bt.n > 1) {
        const int b = plane / 3;
        plane -= 3 * b;
        src3 = shifted(src3, bt.in[b]);
        dst3 = shifted(dst3, bt.out[b]);
        cx = bt.cx[b];
        cy = bt.cy[b];
    }
    if (ix >= Wd) return;
    const float sf = (float)(1 / UGSM_SCALE);
    int sx = tex_index(((float)(ix + cx) + 0.5f) * sf, Ws);
    int sy = tex_index(((float)(iy + cy) + 0.5f) * sf, Hs);
    float v = src3[(size_t)plane * Ws * Hs + (size_t)sy * Ws + sx];
    dst3[(size_t)plane * Wd * Hd + (size_t)iy * Wd + ix] = (float)(UGSM_SCALE * (double)v);
}

// fovea-stack / pyramid-stack packing: plain 2-D crop copy of 3 planes
// (batched: blockIdx.z = 3 x pair + plane)
__global__ void k_copy_view(Img3 src, int W, int H, float *__restrict__ dst, size_t dst_plane, int dst_pitch, Batch bt)
{
    int ix = blockIdx.x * blockDim.x + threadIdx.x;
    int iy = blockIdx.y;
    int plane = blockIdx.z;
    if (bt.n > 1) {
        const int b = plane / 3;
        plane -= 3 * b;
        src.p = shifted(src.p, bt.img[b]);
        dst = shifted(dst, bt.out[b]);
    }
    if (ix >= W) return;
    dst[(size_t)plane * dst_plane + (size_t)iy * dst_pitch + ix] = src.p[(size_t)plane * src.plane + (size_t)iy * src.pitch + ix];
}

void launch_seed(hipStream_t st, const float *src3, int Ws, int Hs, float *dst3, int Wd, int Hd, int cx, int cy, const Batch *bt)
{
    Batch one{};
    one.n = 1;
    const Batch &B = bt ? *bt : one;
    UGSM_LAUNCH(k_seed, grid2(Wd, Hd, 3 * (B.n > 1 ? B.n : 1)), dim3(256), 0, st, src3, Ws, Hs, dst3, Wd, Hd, cx, cy, B);
}
void launch_copy_view(hipStream_t st, Img3 src, int W, int H, float *dst, size_t dst_plane, int dst_pitch, const Batch *bt)
{
    Batch one{};
    one.n = 1;
    const Batch &B = bt ? *bt : one;
    UGSM_LAUNCH(k_copy_view, grid2(W, H, 3 * (B.n > 1 ? B.n : 1)), dim3(256), 0, st, src, W, H, dst, dst_plane, dst_pitch, B);
}
// --------------------------------------------------------------------------------------
// LR-consistency check (BASELINE.json north_star; the reference has none: SURVEY.md 0.4 -- the build's own definition, DESIGN.md
// section 8; opt-in, off in every parity run).  left3 / right3: (dx, dy, conf) of the left-to-right match and of the match with the
// images exchanged.  Left pixel (x, y) matches right pixel (x + dx, y + dy) (getPointCloud.cpp:910-913); the right field is fetched
// there as the matcher fetches (tex_index on the warp's float coordinate, MatchLib.cu:510-515) and must point back within tau in x
// and in y, or the left confidence becomes 0.  `marked` (may be null) counts the pixels.  One pass: 12 B read + gather, 4 B written.
__global__ __launch_bounds__(256) void k_lr_check(float *__restrict__ left3, const float *__restrict__ right3, int W, int H, float tau,
                                                  unsigned long long *__restrict__ marked)
{
    const int ix = blockIdx.x * blockDim.x + threadIdx.x, iy = blockIdx.y;
    bool bad = false;
    if (ix < W) {
        const size_t n = (size_t)W * H, at = (size_t)iy * W + ix;
        const float dxl = left3[at], dyl = left3[n + at];
        const int sx = tex_index(((float)ix + 0.5f) + dxl, W), sy = tex_index(((float)iy + 0.5f) + dyl, H);
        const size_t rt = (size_t)sy * W + sx;
        const float ex = fabsf(dxl + right3[rt]), ey = fabsf(dyl + right3[n + rt]);
        bad = !(ex <= tau) || !(ey <= tau);
        if (bad) left3[2 * n + at] = 0.0f;
    }
    if (marked) {
        const unsigned long long m = __builtin_amdgcn_ballot_w64(bad);
        if ((threadIdx.x & 63) == 0 && m) atomicAdd(marked, (unsigned long long)__builtin_popcountll(m));
    }
}
void launch_lr_check(hipStream_t st, float *left3, const float *right3, int W, int H, float tau, unsigned long long *marked)
{
    UGSM_LAUNCH(k_lr_check, grid2(W, H), dim3(256), 0, st, left3, right3, W, H, tau, marked);
}

// --------------------------------------------------------------------------------------
// MatchGPULib.cpp:332-338 : rgb8 interleaved -> 3 planar f32
__global__ void k_rgb_planes(const uint8_t *__restrict__ rgb, int stride, int W, int H, float *__restrict__ planes)
{
    int x = blockIdx.x * blockDim.x + threadIdx.x;
    int y = blockIdx.y;
    if (x >= W) return;
    const uint8_t *p = rgb + (size_t)y * stride + 3 * x;
    size_t n = (size_t)W * H, at = (size_t)y * W + x;
    planes[at] = (float)p[0];
    planes[n + at] = (float)p[1];
    planes[2 * n + at] = (float)p[2];
}

void launch_rgb_planes(hipStream_t st, const uint8_t *rgb, int stride, int W, int H, float *planes)
{
    UGSM_LAUNCH(k_rgb_planes, grid2(W, H), dim3(256), 0, st, rgb, stride, W, H, planes);
}

// =========================================================================================
// triangulation of the full-resolution disparity into X, Y, Z planes.
// CdynamicCalibration::get3DPoint, non-foveated branch (src/pointcloud/getPointCloud.cpp:886-949), for
// every pixel: the reference calls it from scalar host loops behind a progress bar (:640-660, :778).
// Purely per-pixel (reads 8 B, writes 12 B): HBM-bound.  The closed form keeps the source's mix of float
// and double term by term (a..j, x, y are floats; pow(v,2.0) is the exact binary64 square; the literal
// 2.0 is a double) -- the expression text is kept identical to the CPU restatement used by the tests, no contraction.
// =========================================================================================
struct Proj {
    double m[12];  // 3x4, row major
};
__device__ __forceinline__ double sq_d(float v) { return (double)v * (double)v; }

// the closed form of get3DPoint (getPointCloud.cpp:908-948) for one left/right correspondence.
// NOTE: this one function follows the reference's expressions term for term, including its variable names
// a..j, x, y -- the formula is a machine-generated closed form whose evaluation order and float/double mix ARE the bit-exactness
// contract (re-associating any term changes the result), so the similarity is unavoidable here and deliberately confined to this
// block; nothing else in the product is written against the reference's text.
__device__ __forceinline__ void tri_point(float x1, float y1, float x2, float y2, const double *P1, const double *P2, float &X, float &Y, float &Z)
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
    X = XUp / divisor;
    Y = YUp / divisor;
    Z = ZUp / divisor;
}

__global__ __launch_bounds__(256) void k_triangulate(const float *__restrict__ dispx, const float *__restrict__ dispy, int W, int H, Proj P1q, Proj P2q,
                                                     float *__restrict__ xyz)
{
    const int xx = blockIdx.x * blockDim.x + threadIdx.x;
    const int yy = blockIdx.y;
    if (xx >= W) return;
    const size_t n = (size_t)W * H, at = (size_t)yy * W + xx;
    float x1, x2, y1, y2;
    x1 = xx;
    y1 = yy;
    x2 = xx + dispx[at];
    y2 = yy + dispy[at];
    float X, Y, Z;
    tri_point(x1, y1, x2, y2, P1q.m, P2q.m, X, Y, Z);
    xyz[at] = X;
    xyz[n + at] = Y;
    xyz[2 * n + at] = Z;
}

// get3DPoint, foveated branch (getPointCloud.cpp:892-903): level src_level of the (F*fovH) x fovW stacks, pixel
// coordinates mapped into the full-resolution frame by mapXcoord / mapYcoord (:387-421).  Those take an int, so the
// right-image coordinate xx + disparity is truncated toward zero before scaling -- kept as in the reference.
__global__ __launch_bounds__(256) void k_triangulate_fovea(const float *__restrict__ stackx, const float *__restrict__ stacky, int fovW, int fovH,
                                                           int src_level, int left_margin, int upper_margin, float scale, Proj P1q, Proj P2q,
                                                           float *__restrict__ xyz)
{
    const int xx = blockIdx.x * blockDim.x + threadIdx.x;
    const int yy = blockIdx.y;
    if (xx >= fovW) return;
    const size_t n = (size_t)fovW * fovH, at = (size_t)yy * fovW + xx;
    const size_t sat = ((size_t)yy + (size_t)fovH * src_level) * fovW + xx;
    const float x1 = (float)left_margin + (float)xx * scale;
    const float y1 = (float)upper_margin + (float)yy * scale;
    const int sx = (int)(xx + stackx[sat]);
    const int sy = (int)(yy + stacky[sat]);
    const float x2 = (float)left_margin + (float)sx * scale;
    const float y2 = (float)upper_margin + (float)sy * scale;
    float X, Y, Z;
    tri_point(x1, y1, x2, y2, P1q.m, P2q.m, X, Y, Z);
    xyz[at] = X;
    xyz[n + at] = Y;
    xyz[2 * n + at] = Z;
}

void launch_triangulate(hipStream_t st, const float *dispx, const float *dispy, int W, int H, const double *P1, const double *P2, float *xyz)
{
    Proj a, b;
    for (int k = 0; k < 12; k++) { a.m[k] = P1[k]; b.m[k] = P2[k]; }
    UGSM_LAUNCH(k_triangulate, dim3((W + 255) / 256, H), dim3(256), 0, st, dispx, dispy, W, H, a, b, xyz);
}

void launch_triangulate_fovea(hipStream_t st, const float *stackx, const float *stacky, int fovW, int fovH, int src_level, int left_margin,
                              int upper_margin, float scale, const double *P1, const double *P2, float *xyz)
{
    Proj a, b;
    for (int k = 0; k < 12; k++) { a.m[k] = P1[k]; b.m[k] = P2[k]; }
    UGSM_LAUNCH(k_triangulate_fovea, dim3((fovW + 255) / 256, fovH), dim3(256), 0, st, stackx, stacky, fovW, fovH, src_level, left_margin,
                       upper_margin, scale, a, b, xyz);
}

// =========================================================================================
// one step of hierarchicalDisparity (MatchGPULib.cpp:2643-2683) -- upsample the coarser
// full-frame field by partsubsampleDispKernel (MatchLib.cu:435-462: dst = s * src[tex((x+.5)/s), tex((y+.5)/s)],
// every channel scaled, confidence included) and paste the finer level's fovea at its crop origin, fused: a
// pasted pixel never computes the upsample it would overwrite.  HBM-bound (12 B written per pixel).
// =========================================================================================
__global__ __launch_bounds__(256) void k_upsample_paste(const float *__restrict__ src3, int W, int H, float *__restrict__ dst3, int W2, int H2,
                                                        const float *__restrict__ fovH_, const float *__restrict__ fovV_, const float *__restrict__ fovC_,
                                                        int fovW, int fovH, int org_x, int org_y)
{
    const int ix = blockIdx.x * blockDim.x + threadIdx.x;
    const int iy = blockIdx.y;
    if (ix >= W2) return;
    const float s = (float)1.41421356;
    const size_t n = (size_t)W * H, n2 = (size_t)W2 * H2, at2 = (size_t)iy * W2 + ix;
    const int fx = ix - org_x, fy = iy - org_y;
    if (fx >= 0 && fx < fovW && fy >= 0 && fy < fovH) {
        const size_t fa = (size_t)fy * fovW + fx;
        dst3[at2] = fovH_[fa];
        dst3[n2 + at2] = fovV_[fa];
        dst3[2 * n2 + at2] = fovC_[fa];
    } else {
        const size_t at = (size_t)tex_index(((float)iy + 0.5f) / s, H) * W + tex_index(((float)ix + 0.5f) / s, W);
        dst3[at2] = s * src3[at];
        dst3[n2 + at2] = s * src3[n + at];
        dst3[2 * n2 + at2] = s * src3[2 * n + at];
    }
}

void launch_upsample_paste(hipStream_t st, const float *src3, int W, int H, float *dst3, int W2, int H2, const float *fovH_, const float *fovV_,
                           const float *fovC_, int fovW, int fovH, int org_x, int org_y)
{
    UGSM_LAUNCH(k_upsample_paste, dim3((W2 + 255) / 256, H2), dim3(256), 0, st, src3, W, H, dst3, W2, H2, fovH_, fovV_, fovC_, fovW, fovH,
                       org_x, org_y);
}

// =========================================================================================
// the convergence measure of the reference's (never called) early exit -- weightedDifference,
// MatchGPULib.cpp:1336-1437 with kernels 17 / 18 (MatchLib.cu:1174-1373): sum(|D - OldD| * conf) / sum(conf) for dx and dy.
// The reference's reduction has no defined order (and is called with the block count as the block size); this build's
// definition (the CPU restatement used by the tests mirrors it) is a fixed order of binary64 sums that maps onto one wave per
// row: lane l adds its columns x = l (mod 64) left to right, lane 0 adds the 64 lane sums in lane order; a second, single-wave
// kernel adds the rows the same way.  Deterministic, and bit-identical to the CPU restatement.
// =========================================================================================
__global__ __launch_bounds__(64) void k_wdiff_rows(const float *__restrict__ newd3, const float *__restrict__ oldd3, int W, int H,
                                                  double *__restrict__ rowsum)
{
    __shared__ double sp[3][64];
    const int y = blockIdx.x, l = threadIdx.x;
    const size_t n = (size_t)W * H;
    double ph = 0.0, pv = 0.0, pc = 0.0;
    for (int x = l; x < W; x += 64) {
        const size_t at = (size_t)y * W + x;
        const float c = newd3[2 * n + at];
        float th = fabsf(newd3[at] - oldd3[at]);        // kernel 17, MatchLib.cu:1194-1199: abs(a - b) ...
        float tv = fabsf(newd3[n + at] - oldd3[n + at]);
        th = th * c;                                    // ... times conf, in float
        tv = tv * c;
        ph += (double)th;
        pv += (double)tv;
        pc += (double)c;
    }
    sp[0][l] = ph;
    sp[1][l] = pv;
    sp[2][l] = pc;
    __syncthreads();
    if (l < 3) {
        double r = 0.0;
        for (int i = 0; i < 64; i++) r += sp[l][i];
        rowsum[(size_t)y * 3 + l] = r;
    }
}
__global__ __launch_bounds__(64) void k_wdiff_total(const double *__restrict__ rowsum, int H, double *__restrict__ out3)
{
    __shared__ double sp[3][64];
    const int l = threadIdx.x;
    double p[3] = {0.0, 0.0, 0.0};
    for (int y = l; y < H; y += 64)
        for (int k = 0; k < 3; k++) p[k] += rowsum[(size_t)y * 3 + k];
    for (int k = 0; k < 3; k++) sp[k][l] = p[k];
    __syncthreads();
    if (l < 3) {
        double r = 0.0;
        for (int i = 0; i < 64; i++) r += sp[l][i];
        out3[l] = r;
    }
}
// out3 (device): S_dx, S_dy, C; rowsum: 3 * H doubles of scratch
void launch_weighted_difference(hipStream_t st, const float *newd3, const float *oldd3, int W, int H, double *rowsum, double *out3)
{
    UGSM_LAUNCH(k_wdiff_rows, dim3(H), dim3(64), 0, st, newd3, oldd3, W, H, rowsum);
    UGSM_LAUNCH(k_wdiff_total, dim3(1), dim3(64), 0, st, rowsum, H, out3);
}

}  // namespace ugsm
