// dev/ugsm_dev_probes.hip -- libugsm_dev.so only: test hooks that run the product kernels' exact arithmetic shortcuts (ugsm_exact.hpp: the
// parabola fast path, x / 3, the shared-reciprocal division, the range-guarded division) on arbitrary operands, so that their rarely
// taken fallbacks and the special values are exercised against the oracle's literal forms (include/ugsm_dev.h).
#include "../ugsm_exact.hpp"
#include "../ugsm_launch.hpp"

namespace ugsm {

// test hook (tests only): poly_fast on arbitrary operands, so that its rarely taken f64 fallback and the
// special values are exercised against the oracle's literal PolyDisparity
__global__ void k_poly_probe(const float *__restrict__ c, const float *__restrict__ l, const float *__restrict__ r, const float *__restrict__ thr,
                             float *__restrict__ delta, float *__restrict__ corr, float *__restrict__ third, int n)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) {
        poly_fast(c[i], l[i], r[i], thr[i], delta[i], corr[i]);
        third[i] = (c[i] >= 0.0f || c[i] != c[i]) ? div3_nonneg(c[i]) : 0.0f;
    }
}
void launch_poly_probe(hipStream_t st, const float *c, const float *l, const float *r, const float *thr, float *delta, float *corr, float *third, int n)
{
    UGSM_LAUNCH(k_poly_probe, dim3((n + 255) / 256), dim3(256), 0, st, c, l, r, thr, delta, corr, third, n);
}

// test hook (tests only): the shared-reciprocal division exactly as k_smooth_fused applies it (fast form,
// range test, literal redo)
__global__ void k_div3_probe(const float *__restrict__ a0, const float *__restrict__ a1, const float *__restrict__ a2, const float *__restrict__ s,
                             float *__restrict__ q0, float *__restrict__ q1, float *__restrict__ q2, int n)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) {
        float x, y, z;
        div3_shared(a0[i], a1[i], a2[i], s[i], x, y, z);
        if (!div3_shared_ok(s[i])) {
            x = a0[i] / s[i];
            y = a1[i] / s[i];
            z = a2[i] / s[i];
        }
        q0[i] = x;
        q1[i] = y;
        q2[i] = z;
    }
}
void launch_div3_probe(hipStream_t st, const float *a0, const float *a1, const float *a2, const float *s, float *q0, float *q1, float *q2, int n)
{
    UGSM_LAUNCH(k_div3_probe, dim3((n + 255) / 256), dim3(256), 0, st, a0, a1, a2, s, q0, q1, q2, n);
}

// test hook: the range-guarded division on arbitrary operands
__global__ void k_div_probe(const float *__restrict__ n, const float *__restrict__ d, float *__restrict__ q, int count)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < count) q[i] = div_inrange(n[i], d[i]);
}
void launch_div_probe(hipStream_t st, const float *n, const float *d, float *q, int count)
{
    UGSM_LAUNCH(k_div_probe, dim3((count + 255) / 256), dim3(256), 0, st, n, d, q, count);
}

}  // namespace ugsm
