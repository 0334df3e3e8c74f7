// valubench -- issue cost of the VALU instructions the fused kernels are made of, relative to v_fma_f32
// (development tool).  One workgroup of 256 threads per CU x 4 (one..four waves per SIMD), each wave runs
// ITER x 32 independent instructions of one kind.  Costs are reported in ACTUAL shader cycles: every wave stamps
// s_memtime (shader clock) and s_memrealtime (constant 100 MHz) around its loop; cycles per wave-instruction per SIMD =
// median over waves of (delta s_memtime) / (instructions x waves per SIMD), and the clock the chip held during the
// loop = delta s_memtime / delta s_memrealtime x 100 MHz (MI355X_MICROARCH.md, DVFS give-back item 6).  The wall-clock
// figure at the NOMINAL clock is printed beside it (round 1 quoted only that one, which overstates the cycle costs by
// nominal / actual clock).
//   hipcc -O3 --offload-arch=gfx950 tools/valubench.hip -o tools/valubench && ./tools/valubench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)

__device__ long long *g_stamps = nullptr;  // per wave: delta s_memtime, delta s_memrealtime
#define STAMP_BEGIN const long long t0__ = (long long)__builtin_amdgcn_s_memtime(), r0__ = (long long)__builtin_amdgcn_s_memrealtime();
#define STAMP_END                                                                                       \
    {                                                                                                   \
        const long long t1__ = (long long)__builtin_amdgcn_s_memtime(), r1__ = (long long)__builtin_amdgcn_s_memrealtime(); \
        if ((threadIdx.x & 63) == 0 && g_stamps) {                                                      \
            const size_t w__ = (size_t)blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64;               \
            g_stamps[2 * w__] = t1__ - t0__;                                                            \
            g_stamps[2 * w__ + 1] = r1__ - r0__;                                                        \
        }                                                                                               \
    }
#define REP8(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7)
#define KERNEL(name, decl, body, sink)                                                   \
    __global__ __launch_bounds__(256) void name(float *out, int iters)                  \
    {                                                                                    \
        decl;                                                                            \
        STAMP_BEGIN                                                                      \
        for (int it = 0; it < iters; it++) {                                             \
            REP8(body) REP8(body) REP8(body) REP8(body)                                  \
        }                                                                                \
        STAMP_END                                                                        \
        out[blockIdx.x * 256 + threadIdx.x] = sink;                                      \
    }

#define DECLF float a[8], b = threadIdx.x * 1e-3f + 1.0f, c = 0.5f; for (int i = 0; i < 8; i++) a[i] = threadIdx.x + i
#define DECLD double a[8], b = threadIdx.x * 1e-3 + 1.0, c = 0.5; for (int i = 0; i < 8; i++) a[i] = threadIdx.x + i
#define DECL2 float2 a[8]; float2 b = make_float2(threadIdx.x * 1e-3f + 1.0f, 1.5f), c = make_float2(0.5f, 0.25f); for (int i = 0; i < 8; i++) a[i] = make_float2(threadIdx.x + i, i)

#define B_FMA(i) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));
#define B_MUL(i) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(a[i]) : "v"(b));
#define B_ADD(i) asm volatile("v_add_f32 %0, %0, %1" : "+v"(a[i]) : "v"(b));
#define B_PKMUL(i) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(a[i]) : "v"(b));
#define B_PKADD(i) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(a[i]) : "v"(b));
#define B_PKFMA(i) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));
#define B_RCP(i) asm volatile("v_rcp_f32 %0, %0" : "+v"(a[i]));
#define B_DSCALE(i) asm volatile("v_div_scale_f32 %0, vcc, %0, %1, %0" : "+v"(a[i]) : "v"(b) : "vcc");
#define B_DFMAS(i) asm volatile("v_div_fmas_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));
#define B_DFIX(i) asm volatile("v_div_fixup_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));
#define B_CND(i) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(a[i]) : "v"(b));  // vcc is only read: no clobber, or the compiler pads every statement with s_nop
#define B_CND64(i) asm volatile("v_cndmask_b32_e64 %0, %0, %1, s[20:21]" : "+v"(a[i]) : "v"(b));
#define B_CNDK(i) asm volatile("v_cndmask_b32 %0, %1, %2, vcc" : "=v"(a[i]) : "v"(b), "v"(c));
#define B_MAX(i) asm volatile("v_max_f32 %0, %0, %1" : "+v"(a[i]) : "v"(b));
#define B_MED3(i) asm volatile("v_med3_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));
#define B_CMP(i) asm volatile("v_cmp_lt_f32 vcc, %0, %1" : : "v"(a[i]), "v"(b) : "vcc");
#define B_CMPS(i) asm volatile("v_cmp_lt_f32_e64 s[20:21], %0, %1" : : "v"(a[i]), "v"(b) : "s20", "s21");
#define B_MOV(i) asm volatile("v_mov_b32 %0, %1" : "=v"(a[i]) : "v"(b));
#define B_CMPCND(i) asm volatile("v_cmp_lt_f32 vcc, %0, %1\n\tv_cndmask_b32 %0, %0, %2, vcc" : "+v"(a[i]) : "v"(b), "v"(c) : "vcc");
#define B_CMPCND64(i) asm volatile("v_cmp_lt_f32_e64 s[20:21], %0, %1\n\tv_cndmask_b32_e64 %0, %0, %2, s[20:21]" : "+v"(a[i]) : "v"(b), "v"(c) : "s20", "s21");
#define B_MINIMUM3(i) asm volatile("v_minimum3_f32 %0, %0, %1, %1" : "+v"(a[i]) : "v"(b));
#define B_MIN3(i) asm volatile("v_min3_f32 %0, %0, %1, %1" : "+v"(a[i]) : "v"(b));
#define B_FMA64(i) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));
#define B_MUL64(i) asm volatile("v_mul_f64 %0, %0, %1" : "+v"(a[i]) : "v"(b));
#define B_ADD64(i) asm volatile("v_add_f64 %0, %0, %1" : "+v"(a[i]) : "v"(b));
#define B_RCP64(i) asm volatile("v_rcp_f64 %0, %0" : "+v"(a[i]));
#define B_DPP(i) asm volatile("v_mov_b32_dpp %0, %0 wave_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(a[i]));

#define B_ADDDPP(i) asm volatile("v_add_f32_dpp %0, %1, %0 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1" : "+v"(a[i]) : "v"(b));
#define B_MULDPP(i) asm volatile("v_mul_f32_dpp %0, %1, %0 wave_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:1" : "+v"(a[i]) : "v"(b));
#define B_ADDDPPROW(i) asm volatile("v_add_f32_dpp %0, %1, %0 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1" : "+v"(a[i]) : "v"(b));
#define B_ADDDPPQUAD(i) asm volatile("v_add_f32_dpp %0, %1, %0 quad_perm:[1,2,3,0] row_mask:0xf bank_mask:0xf bound_ctrl:1" : "+v"(a[i]) : "v"(b));
#define B_MULLIT(i) asm volatile("v_mul_f32 %0, 0x3e779fea, %0" : "+v"(a[i]));
#define B_FMA3(i) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(a[i]) : "v"(b), "v"(c));
#define B_FMAC(i) asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));
#define B_CVTI(i) asm volatile("v_cvt_i32_f32 %0, %0" : "+v"(a[i]));
#define B_FLOOR(i) asm volatile("v_floor_f32 %0, %0" : "+v"(a[i]));
#define B_IADD(i) asm volatile("v_add_u32 %0, %0, %1" : "+v"(a[i]) : "v"(b));
#define B_IMAD24(i) asm volatile("v_mad_u32_u24 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));
#define B_ILSHL(i) asm volatile("v_lshlrev_b32 %0, 2, %0" : "+v"(a[i]));
#define B_IADDLSHL(i) asm volatile("v_add_lshl_u32 %0, %0, %1, 2" : "+v"(a[i]) : "v"(b));
#define B_IMED3(i) asm volatile("v_med3_i32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));
KERNEL(k_add_u32, DECLF, B_IADD, a[0] + a[1] + a[2] + a[3] + a[4] + a[5] + a[6] + a[7])
KERNEL(k_mad_u32_u24, DECLF, B_IMAD24, a[0] + a[1] + a[2] + a[3] + a[4] + a[5] + a[6] + a[7])
KERNEL(k_lshlrev_b32, DECLF, B_ILSHL, a[0] + a[1] + a[2] + a[3] + a[4] + a[5] + a[6] + a[7])
KERNEL(k_add_lshl_u32, DECLF, B_IADDLSHL, a[0] + a[1] + a[2] + a[3] + a[4] + a[5] + a[6] + a[7])
KERNEL(k_med3_i32, DECLF, B_IMED3, a[0] + a[1] + a[2] + a[3] + a[4] + a[5] + a[6] + a[7])
KERNEL(k_add_dpp, DECLF, B_ADDDPP, a[0] + a[1] + a[2] + a[3] + a[4] + a[5] + a[6] + a[7])
KERNEL(k_mul_dpp, DECLF, B_MULDPP, a[0] + a[1] + a[2] + a[3] + a[4] + a[5] + a[6] + a[7])
KERNEL(k_add_dpp_row, DECLF, B_ADDDPPROW, a[0] + a[1] + a[2] + a[3] + a[4] + a[5] + a[6] + a[7])
KERNEL(k_add_dpp_quad, DECLF, B_ADDDPPQUAD, a[0] + a[1] + a[2] + a[3] + a[4] + a[5] + a[6] + a[7])
KERNEL(k_mul_literal, DECLF, B_MULLIT, a[0] + a[1] + a[2] + a[3] + a[4] + a[5] + a[6] + a[7])
KERNEL(k_fma_3src, DECLF, B_FMA3, a[0] + a[1] + a[2] + a[3] + a[4] + a[5] + a[6] + a[7])
KERNEL(k_fmac, DECLF, B_FMAC, a[0] + a[1] + a[2] + a[3] + a[4] + a[5] + a[6] + a[7])
KERNEL(k_cvt_i32, DECLF, B_CVTI, a[0] + a[1] + a[2] + a[3] + a[4] + a[5] + a[6] + a[7])
KERNEL(k_floor, DECLF, B_FLOOR, a[0] + a[1] + a[2] + a[3] + a[4] + a[5] + a[6] + a[7])
KERNEL(k_fma, DECLF, B_FMA, a[0] + a[1] + a[2] + a[3] + a[4] + a[5] + a[6] + a[7])
KERNEL(k_mul, DECLF, B_MUL, a[0] + a[1] + a[2] + a[3] + a[4] + a[5] + a[6] + a[7])
KERNEL(k_add, DECLF, B_ADD, a[0] + a[1] + a[2] + a[3] + a[4] + a[5] + a[6] + a[7])
KERNEL(k_pkmul, DECL2, B_PKMUL, a[0].x + a[1].y + a[2].x + a[3].y + a[4].x + a[5].y + a[6].x + a[7].y)
KERNEL(k_pkadd, DECL2, B_PKADD, a[0].x + a[1].y + a[2].x + a[3].y + a[4].x + a[5].y + a[6].x + a[7].y)
KERNEL(k_pkfma, DECL2, B_PKFMA, a[0].x + a[1].y + a[2].x + a[3].y + a[4].x + a[5].y + a[6].x + a[7].y)
KERNEL(k_rcp, DECLF, B_RCP, a[0] + a[1] + a[2] + a[3] + a[4] + a[5] + a[6] + a[7])
KERNEL(k_dscale, DECLF, B_DSCALE, a[0] + a[1] + a[2] + a[3] + a[4] + a[5] + a[6] + a[7])
KERNEL(k_dfmas, DECLF, B_DFMAS, a[0] + a[1] + a[2] + a[3] + a[4] + a[5] + a[6] + a[7])
KERNEL(k_dfix, DECLF, B_DFIX, a[0] + a[1] + a[2] + a[3] + a[4] + a[5] + a[6] + a[7])
KERNEL(k_cnd, DECLF, B_CND, a[0] + a[1] + a[2] + a[3] + a[4] + a[5] + a[6] + a[7])
KERNEL(k_cnd64, DECLF, B_CND64, a[0] + a[1] + a[2] + a[3] + a[4] + a[5] + a[6] + a[7])
KERNEL(k_cnd_nodep, DECLF, B_CNDK, a[0] + a[1] + a[2] + a[3] + a[4] + a[5] + a[6] + a[7])
KERNEL(k_max, DECLF, B_MAX, a[0] + a[1] + a[2] + a[3] + a[4] + a[5] + a[6] + a[7])
KERNEL(k_med3, DECLF, B_MED3, a[0] + a[1] + a[2] + a[3] + a[4] + a[5] + a[6] + a[7])
KERNEL(k_cmp_vcc, DECLF, B_CMP, a[0] + a[1] + a[2] + a[3] + a[4] + a[5] + a[6] + a[7])
KERNEL(k_cmp_sgpr, DECLF, B_CMPS, a[0] + a[1] + a[2] + a[3] + a[4] + a[5] + a[6] + a[7])
KERNEL(k_mov, DECLF, B_MOV, a[0] + a[1] + a[2] + a[3] + a[4] + a[5] + a[6] + a[7])
KERNEL(k_cmp_cnd_vcc, DECLF, B_CMPCND, a[0] + a[1] + a[2] + a[3] + a[4] + a[5] + a[6] + a[7])
KERNEL(k_cmp_cnd_sgpr, DECLF, B_CMPCND64, a[0] + a[1] + a[2] + a[3] + a[4] + a[5] + a[6] + a[7])
__global__ __launch_bounds__(256) void k_cnd_vccinit(float *out, int iters)
{
    DECLF;
    asm volatile("s_mov_b64 vcc, exec" ::: "vcc");
    STAMP_BEGIN
    for (int it = 0; it < iters; it++) {
        REP8(B_CND) REP8(B_CND) REP8(B_CND) REP8(B_CND)
    }
    STAMP_END
    out[blockIdx.x * 256 + threadIdx.x] = a[0] + a[1] + a[2] + a[3] + a[4] + a[5] + a[6] + a[7];
}
KERNEL(k_minimum3, DECLF, B_MINIMUM3, a[0] + a[1] + a[2] + a[3] + a[4] + a[5] + a[6] + a[7])
KERNEL(k_min3, DECLF, B_MIN3, a[0] + a[1] + a[2] + a[3] + a[4] + a[5] + a[6] + a[7])
KERNEL(k_dpp, DECLF, B_DPP, a[0] + a[1] + a[2] + a[3] + a[4] + a[5] + a[6] + a[7])
KERNEL(k_fma64, DECLD, B_FMA64, (float)(a[0] + a[1] + a[2] + a[3] + a[4] + a[5] + a[6] + a[7]))
KERNEL(k_mul64, DECLD, B_MUL64, (float)(a[0] + a[1] + a[2] + a[3] + a[4] + a[5] + a[6] + a[7]))
KERNEL(k_add64, DECLD, B_ADD64, (float)(a[0] + a[1] + a[2] + a[3] + a[4] + a[5] + a[6] + a[7]))
KERNEL(k_rcp64, DECLD, B_RCP64, (float)(a[0] + a[1] + a[2] + a[3] + a[4] + a[5] + a[6] + a[7]))

// conversions need two register classes: keep both sides live
__global__ __launch_bounds__(256) void k_cvt_f64_f32(float *out, int iters)
{
    float a[8]; double d[8];
    for (int i = 0; i < 8; i++) a[i] = threadIdx.x + i;
    STAMP_BEGIN
    for (int it = 0; it < iters; it++) {
#define B(i) asm volatile("v_cvt_f64_f32 %0, %1" : "=v"(d[i]) : "v"(a[i]));
        REP8(B) REP8(B) REP8(B) REP8(B)
#undef B
    }
    STAMP_END
    out[blockIdx.x * 256 + threadIdx.x] = (float)(d[0] + d[1] + d[2] + d[3] + d[4] + d[5] + d[6] + d[7]);
}
__global__ __launch_bounds__(256) void k_cvt_f32_f64(float *out, int iters)
{
    float a[8]; double d[8];
    for (int i = 0; i < 8; i++) d[i] = threadIdx.x + i;
    STAMP_BEGIN
    for (int it = 0; it < iters; it++) {
#define B(i) asm volatile("v_cvt_f32_f64 %0, %1" : "=v"(a[i]) : "v"(d[i]));
        REP8(B) REP8(B) REP8(B) REP8(B)
#undef B
    }
    STAMP_END
    out[blockIdx.x * 256 + threadIdx.x] = a[0] + a[1] + a[2] + a[3] + a[4] + a[5] + a[6] + a[7];
}

#include <algorithm>
#include <vector>
int main()
{
    hipDeviceProp_t pr; CK(hipGetDeviceProperties(&pr, 0));
    const int cus = pr.multiProcessorCount;
    const double ghz = pr.clockRate * 1e-6;
    printf("%s: %d CUs, %.2f GHz nominal\n", pr.name, cus, ghz);
    float *out; CK(hipMalloc(&out, sizeof(float) * 256 * cus * 8));
    long long *stamps; CK(hipMalloc(&stamps, sizeof(long long) * 2 * 4 * cus * 8));
    CK(hipMemcpyToSymbol(HIP_SYMBOL(g_stamps), &stamps, sizeof(stamps)));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int iters = 4096;
    printf("%-16s %-12s %-22s %-12s %s\n", "kernel", "waves/SIMD", "cycles/wave-instr/SIMD", "clock GHz", "(wall time at the nominal clock)");
    auto run = [&](const char *name, void (*k)(float *, int), int wg_per_cu) {
        hipLaunchKernelGGL(k, dim3(cus * wg_per_cu), dim3(256), 0, 0, out, 64);
        CK(hipDeviceSynchronize());
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL(k, dim3(cus * wg_per_cu), dim3(256), 0, 0, out, iters);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        const int nw = cus * wg_per_cu * 4;
        std::vector<long long> hs(2 * (size_t)nw);
        CK(hipMemcpy(hs.data(), stamps, sizeof(long long) * 2 * nw, hipMemcpyDeviceToHost));
        std::vector<double> cyc(nw), clk(nw);
        for (int w = 0; w < nw; w++) {
            cyc[w] = (double)hs[2 * w] / ((double)iters * 32 * wg_per_cu);
            clk[w] = hs[2 * w + 1] > 0 ? (double)hs[2 * w] / (double)hs[2 * w + 1] * 0.1 : 0.0;
        }
        std::nth_element(cyc.begin(), cyc.begin() + nw / 2, cyc.end());
        std::nth_element(clk.begin(), clk.begin() + nw / 2, clk.end());
        // per SIMD: wg_per_cu waves, each iters*32 instructions
        const double nom = ms * 1e-3 * ghz * 1e9 / ((double)iters * 32 * wg_per_cu);
        printf("%-16s %-12d %-22.2f %-12.2f (%.2f)\n", name, wg_per_cu, cyc[nw / 2], clk[nw / 2], nom);
    };
#define RUN(k) run(#k, k, 1); run(#k, k, 2); run(#k, k, 3); run(#k, k, 4);
    RUN(k_fma) RUN(k_fma_3src) RUN(k_fmac) RUN(k_mul) RUN(k_mul_literal) RUN(k_add) RUN(k_add_dpp) RUN(k_add_dpp_row) RUN(k_add_dpp_quad) RUN(k_mul_dpp) RUN(k_dpp) RUN(k_mov) RUN(k_pkmul) RUN(k_pkadd) RUN(k_pkfma)
    RUN(k_rcp) RUN(k_dscale) RUN(k_dfmas) RUN(k_dfix) RUN(k_minimum3) RUN(k_min3) RUN(k_cmp_cnd_vcc) RUN(k_cmp_cnd_sgpr) RUN(k_cnd64) RUN(k_max) RUN(k_med3)
    RUN(k_cmp_vcc) RUN(k_cmp_sgpr) RUN(k_floor) RUN(k_cvt_i32) RUN(k_add_u32) RUN(k_mad_u32_u24) RUN(k_lshlrev_b32) RUN(k_add_lshl_u32) RUN(k_med3_i32)
    RUN(k_fma64) RUN(k_mul64) RUN(k_add64) RUN(k_rcp64) RUN(k_cvt_f64_f32) RUN(k_cvt_f32_f64)
    return 0;
}
