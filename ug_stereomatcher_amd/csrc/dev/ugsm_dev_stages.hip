// dev/ugsm_dev_stages.hip -- libugsm_dev.so only: kernel_path 1, one kernel per reference stage, global memory only -- the plainest
// possible gfx950 statement of each stage (one thread per output pixel, every neighbourhood re-read through L1/L2).  They exist to
// (a) get a first correct HIP path, (b) expose per-stage intermediates to the parity tests and (c) A/B the product's fused kernels, which
// must match them bit for bit.  They are NOT a production path; a maintainer links libugsm.so, which does not contain this file.
//
// Citations: /root/reference/src/gpu_matcher/<file>:<line>.
#include "../ugsm_device.hpp"
#include "../ugsm_launch.hpp"

namespace ugsm {

static inline dim3 grid2(int W, int H, int z = 1) { return dim3((W + 255) / 256, H, z); }


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

}  // namespace ugsm
