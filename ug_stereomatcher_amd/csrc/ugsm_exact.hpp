// ugsm_exact.hpp -- helpers shared by the production kernels (ugsm_kernels_march*.hip, ugsm_kernels_small.hip, ugsm_kernels_smooth.hip):
// uniform-base global accesses, the XCD-aware tile order, the exact arithmetic shortcuts
// and the DPP lane-neighbour moves.
#pragma once
#include "ugsm_device.hpp"

namespace ugsm {

struct f4 {
    float v[4];
};
// XCD-aware tile order.  Workgroups are dealt round-robin over the 8 XCDs (b and b+8 share an XCD and
// its L2 -- speed only, never correctness).  With the natural order x-neighbouring tiles land on
// different XCDs and every halo line is fetched from HBM once per XCD (measured: 118 B read per
// pixel-iteration against 68 requested).  Remap so that each XCD walks a contiguous band of tile rows;
// bijective for any tile count (cdna_hip_programming.md T1).
// (orig: the workgroup's index in dispatch order: blockIdx.x, or its index inside a group of workgroups that starts at a multiple
// of 8, which lands on the same XCD)
__device__ __forceinline__ void xcd_tile_at(int orig, int n_tiles, int tiles_x, int &tx, int &ty)
{
    const int q = n_tiles >> 3, r = n_tiles & 7, xcd = orig & 7;
    const int t = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (orig >> 3);
    ty = t / tiles_x;
    tx = t - ty * tiles_x;
}
__device__ __forceinline__ void xcd_tile(int n_tiles, int tiles_x, int &tx, int &ty) { xcd_tile_at((int)blockIdx.x, n_tiles, tiles_x, tx, ty); }

// Global accesses as scalar base + 32-bit lane byte offset: the base is pinned to SGPRs (readfirstlane) and the
// access is made in the global address space explicitly, so that the plane offset does not migrate into a 64-bit vector
// add per access and the access does not degrade to a flat one.
typedef __attribute__((address_space(1))) const char gchar_c;
typedef __attribute__((address_space(1))) char gchar;
typedef __attribute__((address_space(1))) const float gfloat_c;
typedef __attribute__((address_space(1))) float gfloat;
__device__ __forceinline__ gchar_c *uniform_base(const float *p)
{
    const unsigned long long v = reinterpret_cast<unsigned long long>(p);
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)v), hi = __builtin_amdgcn_readfirstlane((unsigned)(v >> 32));
    return (gchar_c *)(((unsigned long long)hi << 32) | lo);
}
__device__ __forceinline__ float ld_at(gchar_c *base, unsigned off) { return *(gfloat_c *)(base + off); }
__device__ __forceinline__ void st_at(gchar_c *base, unsigned off, float v) { *(gfloat *)((gchar *)base + off) = v; }

__device__ __forceinline__ void ld4(const float *p, float *o)
{
    const float4 t = *reinterpret_cast<const float4 *>(p);
    o[0] = t.x; o[1] = t.y; o[2] = t.z; o[3] = t.w;
}
__device__ __forceinline__ void ld2(const float *p, float *o)
{
    const float2 t = *reinterpret_cast<const float2 *>(p);
    o[0] = t.x; o[1] = t.y;
}
__device__ __forceinline__ void st4(float *p, const float *o)
{
    *reinterpret_cast<float4 *>(p) = make_float4(o[0], o[1], o[2], o[3]);
}

// products are >= +0, so "0 + x" is x exactly and the first add of tap5 can be dropped
__device__ __forceinline__ float tap5p(float a, float b, float c, float d, float e)
{
    float sum = a * UGSM_G0;
    sum += b * UGSM_G1;
    sum += c * UGSM_G2;
    sum += d * UGSM_G1;
    sum += e * UGSM_G0;
    return sum;
}

// box5 (ugsm_device.hpp) with its two zero taps folded into FMAs: a*0.0f is exact (+-0, or NaN for a = NaN / Inf), so
// fma(a, 0, s) rounds the same sum the two-step form rounds.  8 operations instead of 10.
__device__ __forceinline__ float box5f(float a, float b, float c, float d, float e)
{
    float sum = __builtin_fmaf(a, 0.0f, 0.0f);
    sum += b * UGSM_BOX;
    sum += c * UGSM_BOX;
    sum += d * UGSM_BOX;
    return __builtin_fmaf(e, 0.0f, sum);
}

// ---- exact shortcuts used by the fused kernels only (the one-stage-per-kernel path and the CPU
// oracle keep the literal forms; the parity tests compare the two) ------------------------------
//
// x / 3.0f for x >= 0 (finite or NaN): q = RN(x*c), r = x - 3q exactly (fma), q' = RN(q + r*c) with
// c = RN(1/3).  Checked exhaustively against the IEEE quotient for all 2^31-2^23 non-negative
// finite floats, subnormals included; 3 instructions instead of 11.
__device__ __forceinline__ float div3_nonneg(float x)
{
    const float c = 0x1.555556p-2f;
    const float q = x * c;
    const float r = __builtin_fmaf(-3.0f, q, x);
    return __builtin_fmaf(r, c, q);
}
// MoveCorrelation (MatchLib.cu:681-687): N*N >= +0 and A*B >= +0 (or NaN), so the quotient is never
// negative (and never -0) and the "< 0" arm of the clamp is dead; NaN (0/0) still passes through.
__device__ __forceinline__ float ncc2_nn(float n, float a, float b)
{
    // (IEEE-754-2019 minimum, one v_minimum3_f32 on gfx950, has the same NaN-keeping semantics as this compare +
    // select pair, but measured 4 % slower on the whole kernel: 550 vs 527 us at 16 MP)
    float v = (n * n) / (a * b);
    if (v > 1.0f) v = 1.0f;
    return v;
}
// ---- range-guarded division ---------------------------------------------------------------------------------
// hipcc expands a binary32 `n / d` to v_div_scale x2, v_rcp, six FMA/MUL, v_div_fmas, v_div_fixup (LowerFDIV32).  When
// both operands are zero or lie in [2^-62, 2^37] -- no operand, reciprocal, quotient or remainder of the sequence leaves
// the normal range -- v_div_scale returns its operand unscaled with VCC = 0, v_div_fmas is a plain FMA and
// v_div_fixup passes the quotient through (its special cases are zero / Inf / NaN operands: 0/d gives +0 and 0/0 NaN
// here as well, through rcp(0) = Inf and 0 * Inf = NaN).  The remaining eight instructions are these, operation for
// operation, so the result is the same correctly rounded quotient.  15 divisions per pixel-iteration: ~26 issue
// cycles each instead of ~41 (tools/valubench.hip).
// The guarantee comes from the data: every pyramid value v is checked once, when it is produced, for
// v == 0 or 2^-12 <= v <= 2^9 (range_ok below); then R'^2, L*R' lie in {0} U [2^-24, 2^18], their 5x5 Gaussian sums
// (every tap >= 0.09) in {0} U [2^-31, 2^18.1], and N^2, A*B in {0} U [2^-62, 2^36.2].  A pair with any value outside
// (a NaN, or the blurred fringe of an isolated bright pixel on black) takes the compiler's full sequence instead.
// tests: test_division_in_range_is_ieee.
__device__ __forceinline__ bool range_ok(const float v) { return v == 0.0f || (v >= 0x1p-12f && v <= 0x1p9f); }
__device__ __forceinline__ float div_inrange(const float n, const float d)
{
    float r = __builtin_amdgcn_rcpf(d);
    const float e = __builtin_fmaf(-d, r, 1.0f);
    r = __builtin_fmaf(e, r, r);
    float q = n * r;
    const float e0 = __builtin_fmaf(-d, q, n);
    q = __builtin_fmaf(e0, r, q);
    const float e1 = __builtin_fmaf(-d, q, n);
    return __builtin_fmaf(e1, r, q);
}
template <bool FAST>
__device__ __forceinline__ float ncc2_t(const float n, const float a, const float b)
{
    const float v = FAST ? div_inrange(n * n, a * b) : (n * n) / (a * b);
    // "if (v > 1) v = 1" with a NaN kept (MatchLib.cu:686; the "< 0" arm is dead, see ncc2_nn): IEEE-754-2019 minimum, one
    // v_minimum3_f32 (4.4 issue cycles) instead of a compare + select (6.7)
    return __builtin_elementwise_minimum(v, 1.0f);
}

// PolyDisparity (MatchLib.cu:805-836) with the first quotient in binary32 when that is provably the
// same number: (-b1*0.5) is exact in f32 unless it underflows, and rounding a binary64 quotient of two
// binary32 numbers to binary32 equals the correctly rounded binary32 quotient (53 >= 2*24+2, double
// rounding is innocuous for division).  Operands outside [2^-100, ...) take the literal f64 route.
__device__ __forceinline__ void poly_fast(float c, float l, float r, float thr, float &delta, float &corr)
{
    float b1 = (r - l) / 2.0f;
    float c1 = r - (c + b1);
    if (c1 < 0.0f) {
        // The binary32 quotient for every lane; the binary64 route only when some lane of the wave needs it, behind a
        // wave-uniform branch: written as `cond ? f32 : f64` the compiler evaluates BOTH divisions for every pixel and selects
        // (the binary64 one costs ~80 issue cycles per parabola).
        float dh = (-b1 * 0.5f) / c1;
        const bool f32_ok = (fabsf(b1) >= 0x1p-100f || b1 == 0.0f) && c1 <= -0x1p-100f;
        if (__builtin_amdgcn_ballot_w64(!f32_ok) != 0) {
            asm volatile("; binary64 route of PolyDisparity's first quotient" ::: "memory");  // (keeps the block from being speculated)
            if (!f32_ok) dh = (float)(((double)(-b1) * 0.5) / (double)c1);
        }
        // fmin(thr, fmax(dh, -thr)) as one v_med3_f32: dh is never a NaN here (b1, c1 finite, c1 < 0) and thr > 0
        dh = __builtin_amdgcn_fmed3f(dh, -thr, thr);
        float cstar = (c1 * dh + b1) * dh + c;
        if (cstar > 1.0f) {
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

// The value the neighbouring lane holds in `v` (DPP wave shifts; lane 0 / lane 63 read 0: with bound_ctrl and no `old` operand the
// compiler needs no copy in front of the DPP move and can fold it into the consuming instruction).
__device__ __forceinline__ float lane_below(float v)  // from lane - 1
{
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x138, 0xf, 0xf, true));
}
__device__ __forceinline__ float lane_above(float v)  // from lane + 1
{
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x130, 0xf, 0xf, true));
}

// smoothKernel divides the three weighted sums of a pixel by the same sumCorr (MatchLib.cu:1131-1139).
// Exact shortcut: one reciprocal in binary64, refined to <= 2^-53 relative error (one third-order step from
// v_rcp_f32: r0 (1 + e + e^2), three FMAs), then q_f = RN32(RN64(a_f * r)).  The binary64 product is within 2^-51.4 (relative) of the
// true quotient, and a quotient of two binary32 numbers is never closer than 2^-49 (relative) to a
// binary32 rounding boundary (x/y - m = (X*2^k - M*Y)*2^(b+c)/y with X, Y < 2^24, M < 2^25 odd: a nonzero
// integer over Y), so the final rounding equals that of the IEEE binary32 quotient, overflow and
// subnormal results included.  15 VALU operations for three quotients instead of 36.  A quad with any
// denominator outside [2^-64, 2^64] (zero, negative, NaN, Inf, tiny) redoes those pixels with the literal
// division.  tests: test_smooth_division_*.
__device__ __forceinline__ bool div3_shared_ok(const float s) { return s >= 0x1p-64f && s <= 0x1p64f; }
__device__ __forceinline__ void div3_shared(const float a0, const float a1, const float a2, const float s, float &q0, float &q1, float &q2)
{
    const double sd = (double)s;
    const double r0 = (double)__builtin_amdgcn_rcpf(s);  // relative error e, |e| <= 2^-22
    const double e = __builtin_fma(-sd, r0, 1.0);         // e = 1 - s*r0, exact up to 2^-75
    const double t = __builtin_fma(e, e, e);              // e + e^2
    const double r = __builtin_fma(r0, t, r0);            // r0 (1 + e + e^2) = (1 - e^3) / s
    q0 = (float)((double)a0 * r);
    q1 = (float)((double)a1 * r);
    q2 = (float)((double)a2 * r);
}

}  // namespace ugsm
