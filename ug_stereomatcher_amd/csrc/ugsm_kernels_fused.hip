// placeholder: fused kernels land here next; until then kernel_path 0 routes the two
// simple stages to the reference-structured kernels and refuses the two fused ones.
#include "ugsm_device.hpp"
#include "ugsm_launch.hpp"
#include <cstdio>
#include <cstdlib>
namespace ugsm {
void launch_blur_decimate(hipStream_t st, const float *src3, int W, int H, float *dst3, int W2, int H2, float sf)
{
    launch_blur_decimate_ref(st, src3, W, H, dst3, W2, H2, sf);
}
void launch_sqblur_clamp(hipStream_t st, Img3 src, int W, int H, float *dst3) { launch_sqblur_clamp_ref(st, src, W, H, dst3); }
void launch_cost_fused(hipStream_t, Img3, Img3, const float *, const float *, float *, int, int, float, int)
{
    fprintf(stderr, "ugsm: fused cost kernel not built\n");
    abort();
}
void launch_smooth_fused(hipStream_t, const float *, float *, int, int, int, int)
{
    fprintf(stderr, "ugsm: fused smooth kernel not built\n");
    abort();
}
}  // namespace ugsm
