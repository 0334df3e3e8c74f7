// valubench -- issue cost of the VALU instructions the fused kernels are made of (development tool).
//
// Round 3 rewrite.  Round 2's loops were 32 instructions long: the loop latch (s_add, s_cmp, a taken
// branch) then costs a wave ~25 % on top of every instruction, which is where the "2.5 cycles, not 2.0" of round 2 came from.
// Here a loop body is BODY = 512 instructions (latch < 2 %), the accumulators are 16 (dependent distance 16), every kernel runs
// for >= 2 s of back-to-back launches before the stamped launch (DVFS settles), and the guide's calibration points are in the
// table: s_nop 0 (4 cycles for one wave), v_exp_f32 (8), v_fma_f32 (4 for one wave alone, 2 per SIMD at >= 2 waves;
// MI355X_MICROARCH.md, "Per-instruction cycle constants").
//
// Two figures per row, both in shader cycles per wave-instruction per SIMD:
//   stamped = median over waves of (delta s_memtime of the wave's loop) / (instructions x waves per SIMD) -- valid when all waves
//             of a SIMD run concurrently for the whole loop (they are launched together and run the same loop);
//   wall    = (kernel wall time by HIP events) x (in-kernel clock = delta s_memtime / delta s_memrealtime x 100 MHz)
//             / (instructions x waves per SIMD) -- includes launch ramp and tail.
// Mixed rows interleave two instruction kinds (e.g. one DPP add per three plain adds) to see whether their costs add.
//   hipcc -O3 --offload-arch=gfx950 tools/valubench.hip -o tools/valubench && ./tools/valubench [seconds_of_warmup_per_row]
#include <hip/hip_runtime.h>
#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)

__device__ long long *g_stamps = nullptr;  // per wave: delta s_memtime, delta s_memrealtime (never read by the kernels)
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
// 16 accumulators; X(i) is one instruction (or a group) on accumulator i
#define REP16(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7) X(8) X(9) X(10) X(11) X(12) X(13) X(14) X(15)
#define REP512(X) REP16(X) REP16(X) REP16(X) REP16(X) REP16(X) REP16(X) REP16(X) REP16(X) REP16(X) REP16(X) REP16(X) REP16(X) REP16(X) REP16(X) REP16(X) REP16(X) \
                  REP16(X) REP16(X) REP16(X) REP16(X) REP16(X) REP16(X) REP16(X) REP16(X) REP16(X) REP16(X) REP16(X) REP16(X) REP16(X) REP16(X) REP16(X) REP16(X)
constexpr int BODY = 512;
#define SINKF (a[0] + a[1] + a[2] + a[3] + a[4] + a[5] + a[6] + a[7] + a[8] + a[9] + a[10] + a[11] + a[12] + a[13] + a[14] + a[15])
#define KERNEL(name, decl, body, sink)                                                   \
    __global__ __launch_bounds__(256) void name(float *out, int iters)                  \
    {                                                                                    \
        decl;                                                                            \
        STAMP_BEGIN                                                                      \
        for (int it = 0; it < iters; it++) {                                             \
            REP512(body)                                                                 \
        }                                                                                \
        STAMP_END                                                                        \
        out[blockIdx.x * 256 + threadIdx.x] = sink;                                      \
    }
#define DECLF float a[16], b = threadIdx.x * 1e-3f + 1.0f, c = 0.5f; for (int i = 0; i < 16; i++) a[i] = threadIdx.x + i
#define DECLD double a[16], b = threadIdx.x * 1e-3 + 1.0, c = 0.5; for (int i = 0; i < 16; i++) a[i] = threadIdx.x + i

#define B_FMA(i) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));
#define B_FMAC(i) asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));
#define B_MUL(i) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(a[i]) : "v"(b));
#define B_ADD(i) asm volatile("v_add_f32 %0, %0, %1" : "+v"(a[i]) : "v"(b));
#define B_SNOP(i) asm volatile("s_nop 0");
#define B_EXP(i) asm volatile("v_exp_f32 %0, %0" : "+v"(a[i]));
#define B_RCP(i) asm volatile("v_rcp_f32 %0, %0" : "+v"(a[i]));
#define B_MAX(i) asm volatile("v_max_f32 %0, %0, %1" : "+v"(a[i]) : "v"(b));
#define B_MIN3(i) asm volatile("v_min3_f32 %0, %0, %1, %1" : "+v"(a[i]) : "v"(b));
#define B_MINIMUM3(i) asm volatile("v_minimum3_f32 %0, %0, %1, %1" : "+v"(a[i]) : "v"(b));
#define B_MED3(i) asm volatile("v_med3_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));
#define B_MOV(i) asm volatile("v_mov_b32 %0, %1" : "=v"(a[i]) : "v"(b));
#define B_CND(i) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(a[i]) : "v"(b));
#define B_CMP(i) asm volatile("v_cmp_lt_f32 vcc, %0, %1" : : "v"(a[i]), "v"(b) : "vcc");
#define B_FLOOR(i) asm volatile("v_floor_f32 %0, %0" : "+v"(a[i]));
#define B_CVTI(i) asm volatile("v_cvt_i32_f32 %0, %0" : "+v"(a[i]));
#define B_IADD(i) asm volatile("v_add_u32 %0, %0, %1" : "+v"(a[i]) : "v"(b));
#define B_ILSHL(i) asm volatile("v_lshlrev_b32 %0, 2, %0" : "+v"(a[i]));
#define B_IADDLSHL(i) asm volatile("v_add_lshl_u32 %0, %0, %1, 2" : "+v"(a[i]) : "v"(b));
#define B_IMAD24(i) asm volatile("v_mad_u32_u24 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));
#define B_ADDDPP(i) asm volatile("v_add_f32_dpp %0, %1, %0 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1" : "+v"(a[i]) : "v"(b));
#define B_ADDDPPROW(i) asm volatile("v_add_f32_dpp %0, %1, %0 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1" : "+v"(a[i]) : "v"(b));
#define B_MOVDPP(i) asm volatile("v_mov_b32_dpp %0, %0 wave_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(a[i]));
#define B_ADDCLAMP(i) asm volatile("v_add_f32_e64 %0, %0, %1 clamp" : "+v"(a[i]) : "v"(b));
#define B_FMACLAMP(i) asm volatile("v_fma_f32 %0, %0, %1, %2 clamp" : "+v"(a[i]) : "v"(b), "v"(c));
#define B_MULABS(i) asm volatile("v_mul_f32_e64 %0, |%0|, %1" : "+v"(a[i]) : "v"(b));
#define B_DSCALE(i) asm volatile("v_div_scale_f32 %0, vcc, %0, %1, %0" : "+v"(a[i]) : "v"(b) : "vcc");
#define B_DFMAS(i) asm volatile("v_div_fmas_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));
#define B_DFIX(i) asm volatile("v_div_fixup_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));
#define B_FMA64(i) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));
#define B_MUL64(i) asm volatile("v_mul_f64 %0, %0, %1" : "+v"(a[i]) : "v"(b));
#define B_ADD64(i) asm volatile("v_add_f64 %0, %0, %1" : "+v"(a[i]) : "v"(b));
#define B_PKFMA(i) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(a2[i]) : "v"(b2), "v"(c2));
#define B_PKMUL(i) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(a2[i]) : "v"(b2));
#define B_PKADD(i) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(a2[i]) : "v"(b2));
// mixes: four INDEPENDENT instructions per slot, each on a different accumulator (dependent distance stays 16 groups), so that BODY
// counts groups of 4 for these rows.  (The first version of these rows chained the four on one accumulator and measured the
// dependent-issue latency instead: 7.0 cycles per instruction for one wave = 1.66 x the independent cost, as the guide says.)
#define ACC(i, k) a[((i) + (k)) % 16]
#define M_ADD(i, k) asm volatile("v_add_f32 %0, %0, %1" : "+v"(ACC(i, k)) : "v"(b));
#define M_MUL(i, k) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(ACC(i, k)) : "v"(b));
#define M_FMA(i, k) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(ACC(i, k)) : "v"(b), "v"(c));
#define B_MIX_DPP1_ADD3(i) B_ADDDPP(i) M_ADD(i, 4) M_MUL(i, 8) M_ADD(i, 12)
#define B_MIX_RCP1_FMA3(i) B_RCP(i) M_FMA(i, 4) M_FMA(i, 8) M_FMA(i, 12)
#define B_MIX_MAX1_ADD3(i) B_MAX(i) M_ADD(i, 4) M_MUL(i, 8) M_ADD(i, 12)
#define B_MIX_SNOP1_ADD3(i) B_SNOP(i) M_ADD(i, 4) M_MUL(i, 8) M_ADD(i, 12)
#define B_MIX_SALU1_ADD3(i) asm volatile("s_add_u32 s20, s20, 1" ::: "s20", "scc");  /* (s_add writes SCC: without the clobber the loop latch read it and the kernel never ended) */ M_ADD(i, 4) M_MUL(i, 8) M_ADD(i, 12)
#define B_MIX_F64_FMA3(i) asm volatile("v_cvt_i32_f32 %0, %0" : "+v"(ACC(i, 0))); M_FMA(i, 4) M_FMA(i, 8) M_FMA(i, 12)
// dependent chains (latency, not throughput): four instructions in a row on ONE accumulator
#define B_DEP_ADD4(i) B_ADD(i) B_MUL(i) B_ADD(i) B_MUL(i)
#define B_DEP_DPP4(i) B_ADDDPP(i) B_ADDDPP(i) B_ADDDPP(i) B_ADDDPP(i)

// grouped mixes (round 5): the same 1 : 3 proportion as the rows above, but the slow instructions in runs of 4 or 16 followed by their 12 or 48
// plain ones -- does the ORDER of a mix matter, i.e. is the cost of a slow instruction among plain ones a switching cost?
#define RUN16(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7) X(8) X(9) X(10) X(11) X(12) X(13) X(14) X(15)  // (REP16 cannot expand inside itself)
#define P3(i, k) M_ADD(i, k) M_MUL(i, (k) + 1) M_ADD(i, (k) + 2)
#define B_G4_DPP(i) if constexpr ((i) % 4 == 0) { B_ADDDPP(i) B_ADDDPP((i) + 1) B_ADDDPP((i) + 2) B_ADDDPP((i) + 3) P3(i, 4) P3(i, 7) P3(i, 10) P3(i, 13) }
#define B_G16_DPP(i) if constexpr ((i) == 0) { RUN16(B_ADDDPP) RUN16(B_ADD) RUN16(B_MUL) RUN16(B_ADD) }
#define F3(i, k) M_FMA(i, k) M_FMA(i, (k) + 1) M_FMA(i, (k) + 2)
#define B_G4_RCP(i) if constexpr ((i) % 4 == 0) { B_RCP(i) B_RCP((i) + 1) B_RCP((i) + 2) B_RCP((i) + 3) F3(i, 4) F3(i, 7) F3(i, 10) F3(i, 13) }
#define B_G16_RCP(i) if constexpr ((i) == 0) { RUN16(B_RCP) RUN16(B_FMA) RUN16(B_FMA) RUN16(B_FMA) }
#define B_CVT(i) asm volatile("v_cvt_i32_f32 %0, %0" : "+v"(a[i]));
#define B_G4_CVT(i) if constexpr ((i) % 4 == 0) { B_CVT(i) B_CVT((i) + 1) B_CVT((i) + 2) B_CVT((i) + 3) F3(i, 4) F3(i, 7) F3(i, 10) F3(i, 13) }
#define B_G16_CVT(i) if constexpr ((i) == 0) { RUN16(B_CVT) RUN16(B_FMA) RUN16(B_FMA) RUN16(B_FMA) }
#define B_MIX_CMP1_ADD3(i) B_CMP(i) M_ADD(i, 4) M_MUL(i, 8) M_ADD(i, 12)
#define B_G16_CMP(i) if constexpr ((i) == 0) { RUN16(B_CMP) RUN16(B_ADD) RUN16(B_MUL) RUN16(B_ADD) }
// alternating 1 : 1
#define B_ALT_DPP_ADD(i) if constexpr ((i) % 2 == 0) { B_ADDDPP(i) } else { B_ADD(i) }
#define B_ALT_FMA_ADD(i) if constexpr ((i) % 2 == 0) { B_FMA(i) } else { B_ADD(i) }
// the neighbouring lane's value through the LDS crossbar instead of DPP (round 5): ds_bpermute_b32 + the plain consumer.  a[i] comes back
// asynchronously (lgkmcnt, in order); its next use is 12 slots later, so at most 11 newer ones may be outstanding there.
#define B_BPERM(i) asm volatile("ds_bpermute_b32 %0, %1, %0" : "+v"(a[i]) : "v"(baddr));
#define B_MIX_BPERM1_ADD3(i) B_BPERM(i) asm volatile("s_waitcnt lgkmcnt(11)"); M_ADD(i, 4) M_MUL(i, 8) M_ADD(i, 12)
#define B_MIX_BPERM1_ADD4(i) B_BPERM(i) asm volatile("s_waitcnt lgkmcnt(11)"); M_ADD(i, 4) M_MUL(i, 8) M_ADD(i, 12) M_ADD(i, 5)
#define B_MIX_DPP1_ADD4(i) B_MOVDPP(i) M_ADD(i, 4) M_MUL(i, 8) M_ADD(i, 12) M_ADD(i, 5)
#define DECLFB DECLF; const int baddr = ((threadIdx.x + 63) & 63) * 4
// which DPP form is the expensive one among plain instructions? (round 5)
#define B_ADDDPPSELF(i) asm volatile("v_add_f32_dpp %0, %0, %1 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1" : "+v"(a[i]) : "v"(b));
#define B_MOVDPPT(i) asm volatile("v_mov_b32_dpp %0, %1 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1" : "=v"(t) : "v"(a[i]));
#define B_ADDT(i) asm volatile("v_add_f32 %0, %1, %2" : "=v"(a[i]) : "v"(t), "v"(b));
#define B_MIX_ADDDPP1_ADD4(i) B_ADDDPP(i) M_ADD(i, 4) M_MUL(i, 8) M_ADD(i, 12) M_ADD(i, 5)
#define B_MIX_ADDDPPSELF1_ADD4(i) B_ADDDPPSELF(i) M_ADD(i, 4) M_MUL(i, 8) M_ADD(i, 12) M_ADD(i, 5)
#define B_MIX_ADDDPPSELF1_ADD3(i) B_ADDDPPSELF(i) M_ADD(i, 4) M_MUL(i, 8) M_ADD(i, 12)
#define B_MIX_MOVDPP1_ADD3(i) B_MOVDPP(i) M_ADD(i, 4) M_MUL(i, 8) M_ADD(i, 12)
#define B_MIX_MOVADD_ADD3(i) B_MOVDPPT(i) B_ADDT(i) M_ADD(i, 4) M_MUL(i, 8) M_ADD(i, 12)
#define DECLFT DECLF; float t = 0.0f
KERNEL(k_fma, DECLF, B_FMA, SINKF)
KERNEL(k_fmac, DECLF, B_FMAC, SINKF)
KERNEL(k_mul, DECLF, B_MUL, SINKF)
KERNEL(k_add, DECLF, B_ADD, SINKF)
KERNEL(k_s_nop0, DECLF, B_SNOP, SINKF)
KERNEL(k_exp, DECLF, B_EXP, SINKF)
KERNEL(k_rcp, DECLF, B_RCP, SINKF)
KERNEL(k_max, DECLF, B_MAX, SINKF)
KERNEL(k_min3, DECLF, B_MIN3, SINKF)
KERNEL(k_minimum3, DECLF, B_MINIMUM3, SINKF)
KERNEL(k_med3, DECLF, B_MED3, SINKF)
KERNEL(k_mov, DECLF, B_MOV, SINKF)
KERNEL(k_cmp_vcc, DECLF, B_CMP, SINKF)
KERNEL(k_floor, DECLF, B_FLOOR, SINKF)
KERNEL(k_cvt_i32, DECLF, B_CVTI, SINKF)
KERNEL(k_add_u32, DECLF, B_IADD, SINKF)
KERNEL(k_lshlrev_b32, DECLF, B_ILSHL, SINKF)
KERNEL(k_add_lshl_u32, DECLF, B_IADDLSHL, SINKF)
KERNEL(k_mad_u32_u24, DECLF, B_IMAD24, SINKF)
KERNEL(k_add_dpp_wave_shr, DECLF, B_ADDDPP, SINKF)
KERNEL(k_add_dpp_row_shr, DECLF, B_ADDDPPROW, SINKF)
KERNEL(k_mov_dpp, DECLF, B_MOVDPP, SINKF)
KERNEL(k_add_clamp, DECLF, B_ADDCLAMP, SINKF)
KERNEL(k_fma_clamp, DECLF, B_FMACLAMP, SINKF)
KERNEL(k_mul_abs, DECLF, B_MULABS, SINKF)
KERNEL(k_div_scale, DECLF, B_DSCALE, SINKF)
KERNEL(k_div_fmas, DECLF, B_DFMAS, SINKF)
KERNEL(k_div_fixup, DECLF, B_DFIX, SINKF)
KERNEL(k_fma64, DECLD, B_FMA64, (float)SINKF)
KERNEL(k_mul64, DECLD, B_MUL64, (float)SINKF)
KERNEL(k_add64, DECLD, B_ADD64, (float)SINKF)
KERNEL(k_mix_dpp1_add3, DECLF, B_MIX_DPP1_ADD3, SINKF)
KERNEL(k_mix_rcp1_fma3, DECLF, B_MIX_RCP1_FMA3, SINKF)
KERNEL(k_mix_max1_add3, DECLF, B_MIX_MAX1_ADD3, SINKF)
KERNEL(k_mix_snop1_add3, DECLF, B_MIX_SNOP1_ADD3, SINKF)
KERNEL(k_mix_salu1_add3, DECLF, B_MIX_SALU1_ADD3, SINKF)
KERNEL(k_mix_cvt1_fma3, DECLF, B_MIX_F64_FMA3, SINKF)
KERNEL(k_dep_add4, DECLF, B_DEP_ADD4, SINKF)
KERNEL(k_dep_dpp4, DECLF, B_DEP_DPP4, SINKF)
KERNEL(k_g4_dpp4_add12, DECLF, B_G4_DPP, SINKF)
KERNEL(k_g16_dpp16_add48, DECLF, B_G16_DPP, SINKF)
KERNEL(k_g4_rcp4_fma12, DECLF, B_G4_RCP, SINKF)
KERNEL(k_g16_rcp16_fma48, DECLF, B_G16_RCP, SINKF)
KERNEL(k_g4_cvt4_fma12, DECLF, B_G4_CVT, SINKF)
KERNEL(k_g16_cvt16_fma48, DECLF, B_G16_CVT, SINKF)
KERNEL(k_mix_cmp1_add3, DECLF, B_MIX_CMP1_ADD3, SINKF)
KERNEL(k_g16_cmp16_add48, DECLF, B_G16_CMP, SINKF)
KERNEL(k_alt_dpp_add, DECLF, B_ALT_DPP_ADD, SINKF)
KERNEL(k_alt_fma_add, DECLF, B_ALT_FMA_ADD, SINKF)
KERNEL(k_mix_bperm1_add3, DECLFB, B_MIX_BPERM1_ADD3, SINKF)
KERNEL(k_mix_bperm1_add4, DECLFB, B_MIX_BPERM1_ADD4, SINKF)
KERNEL(k_mix_movdpp1_add4, DECLF, B_MIX_DPP1_ADD4, SINKF)
KERNEL(k_bperm, DECLFB, B_BPERM, SINKF)
KERNEL(k_mix_adddpp1_add4, DECLF, B_MIX_ADDDPP1_ADD4, SINKF)
KERNEL(k_mix_adddppself1_add4, DECLF, B_MIX_ADDDPPSELF1_ADD4, SINKF)
KERNEL(k_mix_adddppself1_add3, DECLF, B_MIX_ADDDPPSELF1_ADD3, SINKF)
KERNEL(k_mix_movdpp1_add3, DECLF, B_MIX_MOVDPP1_ADD3, SINKF)
KERNEL(k_mix_movadd_add3, DECLFT, B_MIX_MOVADD_ADD3, SINKF + t)
__global__ __launch_bounds__(256) void k_cndmask(float *out, int iters)
{
    DECLF;
    asm volatile("s_mov_b64 vcc, exec" ::: "vcc");
    STAMP_BEGIN
    for (int it = 0; it < iters; it++) { REP512(B_CND) }
    STAMP_END
    out[blockIdx.x * 256 + threadIdx.x] = SINKF;
}
#define DECL2 float2 a2[16]; float2 b2 = make_float2(threadIdx.x * 1e-3f + 1.0f, 1.5f), c2 = make_float2(0.5f, 0.25f); for (int i = 0; i < 16; i++) a2[i] = make_float2(threadIdx.x + i, i)
#define SINK2 (a2[0].x + a2[1].y + a2[2].x + a2[3].y + a2[4].x + a2[5].y + a2[6].x + a2[7].y + a2[8].x + a2[9].y + a2[10].x + a2[11].y + a2[12].x + a2[13].y + a2[14].x + a2[15].y)
KERNEL(k_pk_fma, DECL2, B_PKFMA, SINK2)
KERNEL(k_pk_mul, DECL2, B_PKMUL, SINK2)
KERNEL(k_pk_add, DECL2, B_PKADD, SINK2)
// conversions need two register classes
__global__ __launch_bounds__(256) void k_cvt_f64_f32(float *out, int iters)
{
    float a[16]; double d[16];
    for (int i = 0; i < 16; i++) a[i] = threadIdx.x + i;
    STAMP_BEGIN
    for (int it = 0; it < iters; it++) {
#define BX(i) asm volatile("v_cvt_f64_f32 %0, %1" : "=v"(d[i]) : "v"(a[i]));
        REP512(BX)
#undef BX
    }
    STAMP_END
    double s = 0; for (int i = 0; i < 16; i++) s += d[i];
    out[blockIdx.x * 256 + threadIdx.x] = (float)s;
}
__global__ __launch_bounds__(256) void k_cvt_f32_f64(float *out, int iters)
{
    float a[16]; double d[16];
    for (int i = 0; i < 16; i++) d[i] = threadIdx.x + i;
    STAMP_BEGIN
    for (int it = 0; it < iters; it++) {
#define BX(i) asm volatile("v_cvt_f32_f64 %0, %1" : "=v"(a[i]) : "v"(d[i]));
        REP512(BX)
#undef BX
    }
    STAMP_END
    out[blockIdx.x * 256 + threadIdx.x] = SINKF;
}

int main(int argc, char **argv)
{
    const double warm_s = argc > 1 ? atof(argv[1]) : 2.0;
    hipDeviceProp_t pr; CK(hipGetDeviceProperties(&pr, 0));
    const int cus = pr.multiProcessorCount;
    printf("%s: %d CUs, %.2f GHz nominal; loop body %d instructions, 16 accumulators, %.1f s of back-to-back launches before each stamped launch\n",
           pr.name, cus, pr.clockRate * 1e-6, BODY, warm_s);
    float *out; CK(hipMalloc(&out, sizeof(float) * 256 * cus * 8));
    long long *stamps; CK(hipMalloc(&stamps, sizeof(long long) * 2 * 4 * cus * 8));
    CK(hipMemcpyToSymbol(HIP_SYMBOL(g_stamps), &stamps, sizeof(stamps)));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int iters = 256;  // 131 072 instructions per wave per launch
    printf("%-22s %-6s %-10s %-10s %-9s\n", "kernel", "w/SIMD", "stamped", "wall", "clock GHz");
    auto run = [&](const char *name, void (*k)(float *, int), int wg_per_cu, int per_slot, double warm) {
        const auto t_start = std::chrono::steady_clock::now();
        do {
            for (int i = 0; i < 20; i++) hipLaunchKernelGGL(k, dim3(cus * wg_per_cu), dim3(256), 0, 0, out, iters);
            CK(hipDeviceSynchronize());
        } while (std::chrono::duration<double>(std::chrono::steady_clock::now() - t_start).count() < warm);
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL(k, dim3(cus * wg_per_cu), dim3(256), 0, 0, out, iters);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        const int nw = cus * wg_per_cu * 4;
        std::vector<long long> hs(2 * (size_t)nw);
        CK(hipMemcpy(hs.data(), stamps, sizeof(long long) * 2 * nw, hipMemcpyDeviceToHost));
        const double instr = (double)iters * BODY * per_slot;
        std::vector<double> cyc(nw), clk(nw);
        for (int w = 0; w < nw; w++) {
            cyc[w] = (double)hs[2 * w] / (instr * wg_per_cu);
            clk[w] = hs[2 * w + 1] > 0 ? (double)hs[2 * w] / (double)hs[2 * w + 1] * 0.1 : 0.0;
        }
        std::nth_element(cyc.begin(), cyc.begin() + nw / 2, cyc.end());
        std::nth_element(clk.begin(), clk.begin() + nw / 2, clk.end());
        const double wall = ms * 1e-3 * clk[nw / 2] * 1e9 / (instr * wg_per_cu);
        printf("%-22s %-6d %-10.2f %-10.2f %-9.2f\n", name, wg_per_cu, cyc[nw / 2], wall, clk[nw / 2]);
        fflush(stdout);
    };
#define RUN(k) run(#k, k, 1, 1, warm_s); run(#k, k, 2, 1, warm_s); run(#k, k, 3, 1, 0.3); run(#k, k, 4, 1, 0.3);
#define RUN4(k) run(#k, k, 1, 4, warm_s); run(#k, k, 2, 4, warm_s); run(#k, k, 3, 4, 0.3); run(#k, k, 4, 4, 0.3);
    const bool only_mix = argc > 2;
    if (!only_mix) {
    printf("# calibration points (MI355X_MICROARCH.md: s_nop 0 = 4, v_exp_f32 = 8, v_fma_f32 = 4 for one wave alone, 2 per SIMD at >= 2 waves)\n");
    RUN(k_s_nop0) RUN(k_exp) RUN(k_fma)
    printf("# full-rate candidates\n");
    RUN(k_fmac) RUN(k_mul) RUN(k_add) RUN(k_add_u32) RUN(k_mul_abs) RUN(k_add_clamp) RUN(k_fma_clamp)
    printf("# others\n");
    RUN(k_mov) RUN(k_max) RUN(k_min3) RUN(k_minimum3) RUN(k_med3) RUN(k_cmp_vcc) RUN(k_cndmask) RUN(k_floor) RUN(k_cvt_i32) RUN(k_lshlrev_b32) RUN(k_add_lshl_u32) RUN(k_mad_u32_u24)
    RUN(k_add_dpp_wave_shr) RUN(k_add_dpp_row_shr) RUN(k_mov_dpp) RUN(k_rcp) RUN(k_div_scale) RUN(k_div_fmas) RUN(k_div_fixup)
    RUN(k_fma64) RUN(k_mul64) RUN(k_add64) RUN(k_cvt_f64_f32) RUN(k_cvt_f32_f64) RUN(k_pk_fma) RUN(k_pk_mul) RUN(k_pk_add)
    }
    const bool only_lds = argc > 2 && argv[2][0] == 'l';
    if (!only_lds) {
    printf("# mixes: cycles per INSTRUCTION of the group of four (if costs add: (c1 + 3 x c_plain) / 4)\n");
    RUN4(k_mix_dpp1_add3) RUN4(k_mix_rcp1_fma3) RUN4(k_mix_max1_add3) RUN4(k_mix_snop1_add3) RUN4(k_mix_salu1_add3) RUN4(k_mix_cvt1_fma3)
    printf("# grouped mixes: the slow instructions of a 1 : 3 mix in runs of 4 / 16 (cycles per instruction, as above); 1 : 1 alternations\n");
    RUN4(k_g4_dpp4_add12) RUN4(k_g16_dpp16_add48) RUN4(k_g4_rcp4_fma12) RUN4(k_g16_rcp16_fma48) RUN4(k_g4_cvt4_fma12) RUN4(k_g16_cvt16_fma48)
    RUN4(k_mix_cmp1_add3) RUN4(k_g16_cmp16_add48) RUN(k_alt_dpp_add) RUN(k_alt_fma_add)
    }
    printf("# lane shift through the LDS crossbar: ds_bpermute_b32 among plain instructions (cycles per instruction of the group; the 1 + 4 rows count 5 per slot)\n");
    RUN(k_bperm) RUN4(k_mix_bperm1_add3)
    run("k_mix_bperm1_add4", k_mix_bperm1_add4, 1, 5, warm_s); run("k_mix_bperm1_add4", k_mix_bperm1_add4, 2, 5, warm_s); run("k_mix_bperm1_add4", k_mix_bperm1_add4, 3, 5, 0.3); run("k_mix_bperm1_add4", k_mix_bperm1_add4, 4, 5, 0.3);
    run("k_mix_movdpp1_add4", k_mix_movdpp1_add4, 1, 5, warm_s); run("k_mix_movdpp1_add4", k_mix_movdpp1_add4, 2, 5, warm_s); run("k_mix_movdpp1_add4", k_mix_movdpp1_add4, 3, 5, 0.3); run("k_mix_movdpp1_add4", k_mix_movdpp1_add4, 4, 5, 0.3);
    printf("# DPP forms among plain instructions (per instruction; rows _add4 and movadd count 5 per slot)\n");
#define RUN5(k) run(#k, k, 1, 5, warm_s); run(#k, k, 2, 5, warm_s); run(#k, k, 3, 5, 0.3); run(#k, k, 4, 5, 0.3);
    RUN5(k_mix_adddpp1_add4) RUN5(k_mix_adddppself1_add4) RUN4(k_mix_adddppself1_add3) RUN4(k_mix_movdpp1_add3) RUN5(k_mix_movadd_add3) RUN4(k_mix_dpp1_add3)
    printf("# dependent chains of four on one accumulator (issue-to-issue latency of dependent instructions)\n");
    RUN4(k_dep_add4) RUN4(k_dep_dpp4)
    return 0;
}
