// kbench -- stand-alone timing harness for the fused kernels (development tool, not product).
//   hipcc -O3 --offload-arch=gfx950 -ffp-contract=off tools/kbench.hip -o tools/kbench
//   ./kbench [W H reps]
// Times k_cost_fused and k_smooth_fused on one level-sized random problem with HIP events.
#include "../ug_stereomatcher_amd/csrc/ugsm_kernels_ref.hip"
#include "../ug_stereomatcher_amd/csrc/ugsm_kernels_fused.hip"
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>
using namespace ugsm;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)
int main(int argc, char **argv)
{
    int W = argc > 1 ? atoi(argv[1]) : 4928, H = argc > 2 ? atoi(argv[2]) : 3264, reps = argc > 3 ? atoi(argv[3]) : 10;
    size_t n = (size_t)W * H;
    std::vector<float> hL(3 * n), hR(3 * n), hd(3 * n);
    unsigned s = 12345;
    auto rnd = [&]() { s = s * 1664525u + 1013904223u; return (s >> 8) * (1.0f / 16777216.0f); };
    for (size_t i = 0; i < 3 * n; i++) { hL[i] = 1 + 254 * rnd(); hR[i] = 1 + 254 * rnd(); }
    for (size_t i = 0; i < n; i++) { int x = i % W, y = i / W; hd[i] = 30.0f * sinf(x * 0.002f) * cosf(y * 0.003f) + 0.3f * (rnd() - 0.5f); hd[n + i] = 0.75f * sinf(y * 0.002f) + 0.3f * (rnd() - 0.5f); hd[2 * n + i] = 0.3f + 0.7f * rnd(); }
    float *L, *R, *A, *d, *o;
    CK(hipMalloc(&L, 12 * n)); CK(hipMalloc(&R, 12 * n)); CK(hipMalloc(&A, 12 * n)); CK(hipMalloc(&d, 12 * n)); CK(hipMalloc(&o, 12 * n));
    CK(hipMemcpy(L, hL.data(), 12 * n, hipMemcpyHostToDevice)); CK(hipMemcpy(R, hR.data(), 12 * n, hipMemcpyHostToDevice));
    CK(hipMemcpy(d, hd.data(), 12 * n, hipMemcpyHostToDevice));
    hipStream_t st; CK(hipStreamCreate(&st));
    Img3 iL{L, W, n}, iR{R, W, n};
    launch_sqblur_clamp(st, iL, W, H, A);
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    auto timeit = [&](const char *name, auto fn) {
        fn(); CK(hipStreamSynchronize(st));
        CK(hipEventRecord(a, st));
        for (int i = 0; i < reps; i++) fn();
        CK(hipEventRecord(b, st)); CK(hipEventSynchronize(b));
        float ms; CK(hipEventElapsedTime(&ms, a, b));
        printf("%-28s %4dx%-4d  %8.1f us/launch  %7.1f Gpx/s\n", name, W, H, 1e3 * ms / reps, n / (ms / reps) / 1e6);
    };
    dim3 grid((W + TX - 1) / TX, (H + TY - 1) / TY);
    const int ctx = (W + TX - 1) / TX, cnt = ctx * ((H + TY - 1) / TY);
#define COST(ABL) timeit("k_cost_fused<" #ABL ">", [&]() { hipLaunchKernelGGL(k_cost_fused<ABL>, dim3(cnt), dim3(256), 0, st, iL, iR, A, d, o, W, H, 1.0f, 1, ctx, cnt); })
#define SPLIT_PLACEHOLDER
#define SPLIT(ABL) timeit("k_cost_split<" #ABL ">", [&]() { hipLaunchKernelGGL(k_cost_split<ABL>, dim3(cnt), dim3(512), 0, st, iL, iR, A, d, o, W, H, 1.0f, 1, ctx, cnt); })
    for (int round = 0; round < 2; round++) {  // interleaved rounds, one process (A/B rule)
        COST(0); SPLIT(0);
    }
    {
        auto run = [&](auto kern, int stx, int sty, int nt, const char *nm) {
            const size_t bytes = 3 * (size_t)(sty + 14) * (stx + 20) * sizeof(float);
            (void)hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
            const int stxn = (W + stx - 1) / stx, stn = stxn * ((H + sty - 1) / sty);
            timeit(nm, [&]() { hipLaunchKernelGGL(kern, dim3(stn), dim3(nt), bytes, st, d, o, W, H, 5, 1, stxn, stn); });
        };
        for (int round = 0; round < 2; round++) {
            run(k_smooth_fused<128, 64, 1024>, 128, 64, 1024, "smooth<128,64,1024> p5+box");
            run(k_smooth_fused<64, 64, 512>, 64, 64, 512, "smooth<64,64,512> p5+box");
        }
    }
    timeit("k_smooth_fused p5", [&]() { launch_smooth_fused(st, d, o, W, H, 5, 0); });
    timeit("k_smooth_fused p5+box", [&]() { launch_smooth_fused(st, d, o, W, H, 5, 1); });
    CK(hipGetLastError());
    return 0;
}
