// ugsm_device.hpp -- per-pixel arithmetic shared by every gfx950 kernel.
//
// Float contract: the matcher is an iterated nearest-neighbour warp, so one
// ulp anywhere moves the result by ~0.2 px.  Every helper here performs the same
// IEEE-754 operations in the same order as the reference source line it cites
// (paths relative to /root/reference/src/gpu_matcher/); the translation unit is built
// with -ffp-contract=off so a*b+c is never fused, divisions are the correctly rounded
// v_div_scale/v_div_fmas/v_div_fixup sequence, and denormals are kept.
#pragma once

#include <hip/hip_runtime.h>

namespace ugsm {

// MatchGPULib.cpp:761-774 -- {0.0816475, 0.218507, 0.303281, ...} / their f32 sum.
// Bit patterns 0x3db90e25, 0x3e779fea, 0x3eabd904; ugsm_create recomputes them the reference's way
// and refuses to start (UGSM_ERR_STATE) if these literals ever disagree.
#define UGSM_G0 0.09035900980234146f
#define UGSM_G1 0.24182096123695374f
#define UGSM_G2 0.3356400728225708f
// MatchGPULib.cpp:344-348
#define UGSM_BOX 0.3333f
// MatchLib_common.h:15
#define UGSM_SCALE 1.41421356

// Batched launches (Batch, ugsm_launch.hpp): a pointer of the launch (pair 0's) moved to this workgroup's pair
template <class T>
__device__ __forceinline__ T *shifted(T *p, long long bytes)
{
    return reinterpret_cast<T *>(reinterpret_cast<unsigned long long>(p) + (unsigned long long)bytes);
}

__device__ __forceinline__ int clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }

// Default-configured texture reference (MatchLib.cu:56-60): point filter, clamp,
// unnormalised -> t[clamp(floor(coord))].  NaN -> 0 (never produced; defined so that
// the kernels and the CPU oracle agree).
__device__ __forceinline__ int tex_index(float coord, int n)
{
    float f = floorf(coord);
    if (!(f >= 0.0f)) return 0;
    if (f > (float)(n - 1)) return n - 1;
    return (int)f;
}

// MatchLib.cu:686-687, 1006-1007: if(v>1) v=1; if(v<0) v=0;  NaN stays NaN.
__device__ __forceinline__ float clamp01(float v)
{
    if (v > 1.0f) v = 1.0f;
    if (v < 0.0f) v = 0.0f;
    return v;
}

// 5-tap accumulate in the reference's order: sum=0; sum += t[-2]*g0 ... (MatchLib.cu:127-134,
// 1484-1487).  0 + x == x for the non-negative finite inputs these taps see, and the
// first add is kept anyway so that the sequence is literally the reference's.
__device__ __forceinline__ float tap5(float a, float b, float c, float d, float e)
{
    float sum = 0.0f;
    sum += a * UGSM_G0;
    sum += b * UGSM_G1;
    sum += c * UGSM_G2;
    sum += d * UGSM_G1;
    sum += e * UGSM_G0;
    return sum;
}

// Same with the "average" taps {0, .3333, .3333, .3333, 0} (MatchLib.cu:1616-1619).
__device__ __forceinline__ float box5(float a, float b, float c, float d, float e)
{
    float sum = 0.0f;
    sum += a * 0.0f;
    sum += b * UGSM_BOX;
    sum += c * UGSM_BOX;
    sum += d * UGSM_BOX;
    sum += e * 0.0f;
    return sum;
}

// PolyDisparity, MatchLib.cu:805-836.  Double sub-expressions exactly where the CUDA
// source promotes (literal 0.5, 0.0, 1.0, 1e-10, 0.3, 0.7 are doubles there).
__device__ __forceinline__ void poly(float c, float l, float r, float thr, float &delta, float &corr)
{
    float b1 = (r - l) / 2.0f;
    float c1 = r - (c + b1);
    if (c1 < 0.0f) {
        float dh = (float)(((double)(-b1) * 0.5) / (double)c1);
        dh = (float)fmin((double)thr, fmax((double)dh, 0.0 - (double)thr));
        float cstar = (c1 * dh + b1) * dh + c;
        if ((double)cstar > 1.0) {
            float d = cstar - c;
            if ((double)d > 1e-10) dh = (float)((double)dh * ((1.0 - (double)c) / (double)d));
            delta = dh;
            corr = 1.0f;
        } else {
            delta = dh;
            corr = (float)(0.3 * (double)cstar + 0.7);
        }
    } else {
        delta = 0.0f;
        corr = 0.4f;
    }
}

// TrueConfidence, MatchLib.cu:1003-1007 (double arithmetic, stored to float, clamped)
__device__ __forceinline__ float blend_conf(float old_c, float new_c)
{
    float v = (float)(0.75 * (double)old_c + 0.25 * (double)new_c);
    return clamp01(v);
}

// MoveCorrelation, MatchLib.cu:681-687
__device__ __forceinline__ float ncc2(float n, float a, float b) { return clamp01((n * n) / (a * b)); }

}  // namespace ugsm
