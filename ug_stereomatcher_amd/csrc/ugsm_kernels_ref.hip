// ugsm_kernels_ref.hip -- the plain per-pixel kernels.
//
// In the product (libugsm.so): k_seed (a level's starting field where the next level's K-cost does not seed itself), k_copy_view
// (the fovea / pyramid stacks), k_lr_check and k_rgb_planes (level 0 of a pyramid of fewer than three levels: BASELINE configs[0]).
// In libugsm_dev.so only (UGSM_DEV_LIB; round 4): kernel_path 1, one stage per kernel, global memory only -- the
// plainest possible gfx950 statement of each stage (one thread per output pixel, every neighbourhood re-read through L1/L2).  They
// exist to (a) get a first correct HIP path, (b) expose per-stage intermediates to the parity tests and (c) A/B the fused kernels of
// ugsm_kernels_fused.hip, which must match them bit for bit.  They are NOT a production path.
//
// Citations: /root/reference/src/gpu_matcher/<file>:<line>.
#include "ugsm_device.hpp"
#include "ugsm_launch.hpp"

namespace ugsm {

// ---- in the product: seeding where the next level does not march, the fovea / pyramid stacks, the LR check ----------

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

#ifdef UGSM_DEV_LIB  // kernel_path 1, one kernel per reference stage: the A/B reference of the fused kernels -- in libugsm_dev.so only

// --------------------------------------------------------------------------------------
// MatchGPULib.cpp:1071-1096 + MatchLib.cu:71-156,195-278,311-339.
// dst[x,y] = blur(src)[floor((x+.5f)*sf), floor((y+.5f)*sf)], blur = zero-padded row conv
// (rounded to f32) then zero-padded column conv.  The reference blurs the whole level and
// then samples it; only the sampled sites are evaluated here.
__global__ void k_blur_decimate(const float *__restrict__ src3, int W, int H, float *__restrict__ dst3, int W2, int H2, float sf)
{
    int ix = blockIdx.x * blockDim.x + threadIdx.x;
    int iy = blockIdx.y;
    if (ix >= W2) return;
    const float *src = src3 + (size_t)blockIdx.z * W * H;
    int sx = tex_index(((float)ix + 0.5f) * sf, W);
    int sy = tex_index(((float)iy + 0.5f) * sf, H);
    float r[5];
#pragma unroll
    for (int j = -2; j <= 2; j++) {
        int yy = sy + j;
        float v[5];
#pragma unroll
        for (int i = -2; i <= 2; i++) {
            int xx = sx + i;
            v[i + 2] = (yy >= 0 && yy < H && xx >= 0 && xx < W) ? src[(size_t)yy * W + xx] : 0.0f;
        }
        r[j + 2] = tap5(v[0], v[1], v[2], v[3], v[4]);
    }
    dst3[(size_t)blockIdx.z * W2 * H2 + (size_t)iy * W2 + ix] = tap5(r[0], r[1], r[2], r[3], r[4]);
}

// --------------------------------------------------------------------------------------
// MatchLib.cu:556-578 + 1461-1565: dst = colconv_clamp(rowconv_clamp(src^2))
__global__ void k_sqblur_clamp(Img3 src, int W, int H, float *__restrict__ dst3)
{
    int ix = blockIdx.x * blockDim.x + threadIdx.x;
    int iy = blockIdx.y;
    if (ix >= W) return;
    const float *s = src.p + (size_t)blockIdx.z * src.plane;
    float r[5];
#pragma unroll
    for (int j = -2; j <= 2; j++) {
        int yy = clampi(iy + j, 0, H - 1);
        float v[5];
#pragma unroll
        for (int i = -2; i <= 2; i++) {
            float t = s[(size_t)yy * src.pitch + clampi(ix + i, 0, W - 1)];
            v[i + 2] = t * t;
        }
        r[j + 2] = tap5(v[0], v[1], v[2], v[3], v[4]);
    }
    dst3[(size_t)blockIdx.z * W * H + (size_t)iy * W + ix] = tap5(r[0], r[1], r[2], r[3], r[4]);
}

// --------------------------------------------------------------------------------------
// warpAbyB, MatchLib.cu:499-520
__global__ void k_warp(Img3 R, const float *__restrict__ d3, int W, int H, float *__restrict__ Rw3)
{
    int ix = blockIdx.x * blockDim.x + threadIdx.x;
    int iy = blockIdx.y;
    if (ix >= W) return;
    size_t n = (size_t)W * H, at = (size_t)iy * W + ix;
    int sx = tex_index(((float)ix + 0.5f) + d3[at], W);
    int sy = tex_index(((float)iy + 0.5f) + d3[n + at], H);
    size_t from = (size_t)sy * R.pitch + sx;
#pragma unroll
    for (int k = 0; k < 3; k++) Rw3[k * n + at] = R.p[k * R.plane + from];
}

// --------------------------------------------------------------------------------------
// One iteration's cost + update for one pixel (MatchGPULib.cpp:1745-2250), naive form:
//   P_s(x',y') = L(x',y') * R'(clamp(x'+sx), clamp(y'+sy))         CompareMove   MatchLib.cu:622-624
//   N_s        = colconv_zero(rowconv_zero(P_s))                    smem convs    :71-278
//   q_s,k      = clamp01(N_s^2 / (A * B(clamp(x+sx),clamp(y+sy))))  MoveCorrelation :681-687
//   Q_s        = ((q_s,0 + q_s,1) + q_s,2) / 3.0f                   MatchGPULib.cpp:2033-2070
//   parabola x (Q0,Q4,Q1), y (Q2,Q4,Q3); kappa = rho_y*rho_x; d += delta; conf blend.
__global__ void k_cost_ref(Img3 L, const float *__restrict__ Rw3, const float *__restrict__ A3, const float *__restrict__ B3,
                           const float *__restrict__ d3, float *__restrict__ nd3, int W, int H, float thr, int blend,
                           float *__restrict__ dbg8)
{
    int ix = blockIdx.x * blockDim.x + threadIdx.x;
    int iy = blockIdx.y;
    if (ix >= W) return;
    const size_t n = (size_t)W * H, at = (size_t)iy * W + ix;
    const int mvx[5] = {-1, 1, 0, 0, 0};
    const int mvy[5] = {0, 0, -1, 1, 0};
    float Q[5];
    for (int k = 0; k < 3; k++) {
        const float *Lk = L.p + (size_t)k * L.plane;
        const float *Rk = Rw3 + k * n;
        const float a = A3[k * n + at];
#pragma unroll
        for (int s = 0; s < 5; s++) {
            const int sx = mvx[s], sy = mvy[s];
            float r[5];
#pragma unroll
            for (int j = -2; j <= 2; j++) {
                int yy = iy + j;
                float v[5];
#pragma unroll
                for (int i = -2; i <= 2; i++) {
                    int xx = ix + i;
                    float p = 0.0f;
                    if (yy >= 0 && yy < H && xx >= 0 && xx < W) {
                        float lv = Lk[(size_t)yy * L.pitch + xx];
                        float rv = Rk[(size_t)clampi(yy + sy, 0, H - 1) * W + clampi(xx + sx, 0, W - 1)];
                        p = lv * rv;
                    }
                    v[i + 2] = p;
                }
                r[j + 2] = tap5(v[0], v[1], v[2], v[3], v[4]);
            }
            float N = tap5(r[0], r[1], r[2], r[3], r[4]);
            float b = B3[k * n + (size_t)clampi(iy + sy, 0, H - 1) * W + clampi(ix + sx, 0, W - 1)];
            float q = ncc2(N, a, b);
            if (k == 0) Q[s] = q;
            else if (k == 1) Q[s] = q + Q[s];
            else Q[s] = (Q[s] + q) / 3.0f;
        }
    }
    float ddx, ddy, cx, cy;
    poly(Q[4], Q[0], Q[1], thr, ddx, cx);
    poly(Q[4], Q[2], Q[3], thr, ddy, cy);
    float kap = cy * cx;
    float ndx = d3[at] + ddx;
    float ndy = d3[n + at] + ddy;
    if (blend) kap = blend_conf(d3[2 * n + at], kap);
    nd3[at] = ndx;
    nd3[n + at] = ndy;
    nd3[2 * n + at] = kap;
    if (dbg8) {
#pragma unroll
        for (int s = 0; s < 5; s++) dbg8[s * n + at] = Q[s];
        dbg8[5 * n + at] = ndx;
        dbg8[6 * n + at] = ndy;
        dbg8[7 * n + at] = kap;
    }
}

// --------------------------------------------------------------------------------------
// smoothKernel, MatchLib.cu:1092-1145, for dx, dy and conf in one launch.  Row 0 and
// column 0 pass through (ix>0 && iy>0 guard, :1106).  Weight = pre-pass conf for all three.
__global__ void k_smooth_pass(const float *__restrict__ s3, float *__restrict__ o3, int W, int H)
{
    int ix = blockIdx.x * blockDim.x + threadIdx.x;
    int iy = blockIdx.y;
    if (ix >= W) return;
    const size_t n = (size_t)W * H, at = (size_t)iy * W + ix;
    if (ix > 0 && iy > 0) {
        size_t aw = at - 1, ae = (size_t)iy * W + clampi(ix + 1, 0, W - 1);
        size_t an = at - W, as = (size_t)clampi(iy + 1, 0, H - 1) * W + ix;
        const float *cf = s3 + 2 * n;
        float wc = cf[at], ww = cf[aw], we = cf[ae], wn = cf[an], ws = cf[as];
        float sumCorr = 0.0f;
        sumCorr = sumCorr + wc;
        sumCorr = sumCorr + ww;
        sumCorr = sumCorr + we;
        sumCorr = sumCorr + wn;
        sumCorr = sumCorr + ws;
#pragma unroll
        for (int p = 0; p < 3; p++) {
            const float *s = s3 + p * n;
            float sumDisp = 0.0f;
            sumDisp = s[at] * wc + sumDisp;
            sumDisp = s[aw] * ww + sumDisp;
            sumDisp = s[ae] * we + sumDisp;
            sumDisp = s[an] * wn + sumDisp;
            sumDisp = s[as] * ws + sumDisp;
            o3[p * n + at] = sumDisp / sumCorr;
        }
    } else {
#pragma unroll
        for (int p = 0; p < 3; p++) o3[p * n + at] = s3[p * n + at];
    }
}

// --------------------------------------------------------------------------------------
// convolutionRows/ColumnsKernelTa on dx, dy, conf (MatchGPULib.cpp:2361-2412), clamp
// addressing, taps {0,.3333,.3333,.3333,0}; row pass rounded to f32 before the column pass.
__global__ void k_box(const float *__restrict__ s3, float *__restrict__ o3, int W, int H)
{
    int ix = blockIdx.x * blockDim.x + threadIdx.x;
    int iy = blockIdx.y;
    if (ix >= W) return;
    const size_t n = (size_t)W * H;
    const float *s = s3 + (size_t)blockIdx.z * n;
    float r[5];
#pragma unroll
    for (int j = -2; j <= 2; j++) {
        const float *row = s + (size_t)clampi(iy + j, 0, H - 1) * W;
        r[j + 2] = box5(row[clampi(ix - 2, 0, W - 1)], row[clampi(ix - 1, 0, W - 1)], row[ix],
                        row[clampi(ix + 1, 0, W - 1)], row[clampi(ix + 2, 0, W - 1)]);
    }
    o3[(size_t)blockIdx.z * n + (size_t)iy * W + ix] = box5(r[0], r[1], r[2], r[3], r[4]);
}

// ---- launchers -------------------------------------------------------------------------

void launch_blur_decimate_ref(hipStream_t st, const float *src3, int W, int H, float *dst3, int W2, int H2, float sf)
{
    UGSM_LAUNCH(k_blur_decimate, grid2(W2, H2, 3), dim3(256), 0, st, src3, W, H, dst3, W2, H2, sf);
}
void launch_sqblur_clamp_ref(hipStream_t st, Img3 src, int W, int H, float *dst3)
{
    UGSM_LAUNCH(k_sqblur_clamp, grid2(W, H, 3), dim3(256), 0, st, src, W, H, dst3);
}
void launch_warp_ref(hipStream_t st, Img3 R, const float *d3, int W, int H, float *Rw3)
{
    UGSM_LAUNCH(k_warp, grid2(W, H), dim3(256), 0, st, R, d3, W, H, Rw3);
}
void launch_cost_ref(hipStream_t st, Img3 L, const float *Rw3, const float *A3, const float *B3, const float *d3, float *nd3,
                     int W, int H, float thr, int blend, float *dbg8)
{
    UGSM_LAUNCH(k_cost_ref, grid2(W, H), dim3(256), 0, st, L, Rw3, A3, B3, d3, nd3, W, H, thr, blend, dbg8);
}
void launch_smooth_pass_ref(hipStream_t st, const float *s3, float *o3, int W, int H)
{
    UGSM_LAUNCH(k_smooth_pass, grid2(W, H), dim3(256), 0, st, s3, o3, W, H);
}
void launch_box_ref(hipStream_t st, const float *s3, float *o3, int W, int H)
{
    UGSM_LAUNCH(k_box, grid2(W, H, 3), dim3(256), 0, st, s3, o3, W, H);
}
#endif  // UGSM_DEV_LIB


}  // namespace ugsm
