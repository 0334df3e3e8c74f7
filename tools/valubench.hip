// valubench -- issue cost of the VALU instructions the fused kernels are made of, relative to v_fma_f32
// (development tool).  One workgroup of 256 threads per CU x 4 (one..four waves per SIMD), each wave runs
// ITER x 32 independent instructions of one kind; time per instruction per wave comes out in cycles
// assuming the clock the chip reports.
//   hipcc -O3 --offload-arch=gfx950 tools/valubench.hip -o tools/valubench && ./tools/valubench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)

#define REP8(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7)
#define KERNEL(name, decl, body, sink)                                                   \
    __global__ __launch_bounds__(256) void name(float *out, int iters)                  \
    {                                                                                    \
        decl;                                                                            \
        for (int it = 0; it < iters; it++) {                                             \
            REP8(body) REP8(body) REP8(body) REP8(body)                                  \
        }                                                                                \
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
    for (int it = 0; it < iters; it++) {
        REP8(B_CND) REP8(B_CND) REP8(B_CND) REP8(B_CND)
    }
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
    for (int it = 0; it < iters; it++) {
#define B(i) asm volatile("v_cvt_f64_f32 %0, %1" : "=v"(d[i]) : "v"(a[i]));
        REP8(B) REP8(B) REP8(B) REP8(B)
#undef B
    }
    out[blockIdx.x * 256 + threadIdx.x] = (float)(d[0] + d[1] + d[2] + d[3] + d[4] + d[5] + d[6] + d[7]);
}
__global__ __launch_bounds__(256) void k_cvt_f32_f64(float *out, int iters)
{
    float a[8]; double d[8];
    for (int i = 0; i < 8; i++) d[i] = threadIdx.x + i;
    for (int it = 0; it < iters; it++) {
#define B(i) asm volatile("v_cvt_f32_f64 %0, %1" : "=v"(a[i]) : "v"(d[i]));
        REP8(B) REP8(B) REP8(B) REP8(B)
#undef B
    }
    out[blockIdx.x * 256 + threadIdx.x] = a[0] + a[1] + a[2] + a[3] + a[4] + a[5] + a[6] + a[7];
}

int main()
{
    hipDeviceProp_t pr; CK(hipGetDeviceProperties(&pr, 0));
    const int cus = pr.multiProcessorCount;
    const double ghz = pr.clockRate * 1e-6;
    printf("%s: %d CUs, %.2f GHz nominal\n", pr.name, cus, ghz);
    float *out; CK(hipMalloc(&out, sizeof(float) * 256 * cus * 8));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int iters = 4096;
    auto run = [&](const char *name, void (*k)(float *, int), int wg_per_cu) {
        hipLaunchKernelGGL(k, dim3(cus * wg_per_cu), dim3(256), 0, 0, out, 64);
        CK(hipDeviceSynchronize());
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL(k, dim3(cus * wg_per_cu), dim3(256), 0, 0, out, iters);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        // per SIMD: wg_per_cu waves, each iters*32 instructions
        const double cyc = ms * 1e-3 * ghz * 1e9 / ((double)iters * 32 * wg_per_cu);
        printf("%-16s %d wave/SIMD  %6.2f cycles per wave-instruction (at nominal clock)\n", name, wg_per_cu, cyc);
    };
#define RUN(k) run(#k, k, 1); run(#k, k, 4);
    RUN(k_fma) RUN(k_mul) RUN(k_add) RUN(k_pkmul) RUN(k_pkadd) RUN(k_pkfma) RUN(k_rcp) RUN(k_dscale) RUN(k_dfmas) RUN(k_dfix) RUN(k_minimum3) RUN(k_min3) RUN(k_cnd) RUN(k_cnd_vccinit) RUN(k_cmp_cnd_vcc) RUN(k_cmp_cnd_sgpr) RUN(k_cnd64) RUN(k_cnd_nodep) RUN(k_max) RUN(k_med3) RUN(k_cmp_vcc) RUN(k_cmp_sgpr) RUN(k_mov) RUN(k_dpp)
    RUN(k_fma64) RUN(k_mul64) RUN(k_add64) RUN(k_rcp64) RUN(k_cvt_f64_f32) RUN(k_cvt_f32_f64)
    return 0;
}
