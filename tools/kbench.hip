// kbench -- stand-alone timing harness for the fused kernels (development tool, not product).
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -ffp-contract=off -fno-fast-math -fno-slp-vectorize -Iinclude tools/kbench.hip -o tools/kbench
//   ./kbench [W H reps]
// Times k_cost_fused and k_smooth_fused on one level-sized random problem with HIP events.
#define UGSM_DEV_KERNELS 1  // the development forms of the marching kernels (two pixels per lane, FMA contract): not in libugsm.so
#define UGSM_DEV_LIB 1      // ... and what libugsm_dev.so has over libugsm.so: kernel_path 1, k_smooth_march, k_iter_small, the probes
#include "../ug_stereomatcher_amd/csrc/ugsm_kernels_ref.hip"
#include "../ug_stereomatcher_amd/csrc/ugsm_kernels_fused.hip"
#include "../ug_stereomatcher_amd/csrc/ugsm_kernels_march.hip"
#include "../ug_stereomatcher_amd/csrc/ugsm_kernels_march4.hip"
#include "../ug_stereomatcher_amd/csrc/ugsm_kernels_small.hip"
#include <algorithm>
#include <array>
#include <cmath>
#include <cstdio>
#include <chrono>
#include <cstdlib>
#include <cstring>
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
    const char *data_file = getenv("KBENCH_DATA") ? getenv("KBENCH_DATA") : ((argc > 5 && atoi(argv[4]) < 2) ? argv[5] : nullptr);
    if (data_file) {  // real data: 9 planes of W*H floats (L0 L1 L2 R0 R1 R2 dx dy conf), e.g. from tools/kbench_real.py (any mode: KBENCH_DATA=file)
        FILE *f = fopen(data_file, "rb");
        if (!f || fread(hL.data(), 4, 3 * n, f) != 3 * n || fread(hR.data(), 4, 3 * n, f) != 3 * n || fread(hd.data(), 4, 3 * n, f) != 3 * n) { printf("cannot read %s\n", data_file); return 1; }
        fclose(f);
        printf("inputs from %s\n", data_file);
    }
    float *L, *R, *A, *d, *o;
    CK(hipMalloc(&L, 12 * n)); CK(hipMalloc(&R, 12 * n)); CK(hipMalloc(&A, 12 * n)); CK(hipMalloc(&d, 12 * n)); CK(hipMalloc(&o, 12 * n));
    CK(hipMemcpy(L, hL.data(), 12 * n, hipMemcpyHostToDevice)); CK(hipMemcpy(R, hR.data(), 12 * n, hipMemcpyHostToDevice));
    CK(hipMemcpy(d, hd.data(), 12 * n, hipMemcpyHostToDevice));
    hipStream_t st; CK(hipStreamCreate(&st));
    unsigned *rb; CK(hipMalloc(&rb, 64)); CK(hipMemset(rb, 0, 64));  // range flag: inputs are in [1, 255]
    const unsigned *rbs[2] = {nullptr, rb};
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
    if (argc > 4 && atoi(argv[4]) == 1) {  // counter runs: the two production kernels only
        for (int i = 0; i < reps; i++) launch_cost_fused(st, iL, iR, A, d, o, W, H, 1.0f, 1);
        for (int i = 0; i < reps; i++) launch_smooth_fused(st, d, o, W, H, 5, 0);
        for (int i = 0; i < reps; i++) launch_smooth_fused(st, d, o, W, H, 5, 1);
        CK(hipStreamSynchronize(st));
        return 0;
    }
#ifdef UGSM_MARCH_STAMP
    if (argc > 4 && atoi(argv[4]) == 6) {  // in-kernel clock of the marching K-cost under sustained launches + where and when its waves ran: kbench_stamp W H reps 6
        const int nb = 16384;
        if (argc > 6) { march_age_permille[0] = atoi(argv[5]); march_age_permille[1] = atoi(argv[6]); printf("strips by age class: %d/%d/%d per mille\n", march_age_permille[0], march_age_permille[1], 1000 - march_age_permille[0] - march_age_permille[1]); }
        long long *dst; CK(hipMalloc(&dst, sizeof(long long) * 4 * nb)); CK(hipMemset(dst, 0, sizeof(long long) * 4 * nb));
        CK(hipMemcpyToSymbol(HIP_SYMBOL(g_march_stamps), &dst, sizeof(dst)));
        const auto t_start = std::chrono::steady_clock::now();
        int launches = 0;
        while (std::chrono::duration<double>(std::chrono::steady_clock::now() - t_start).count() < 2.5) {  // >= 2 s of back-to-back launches
            for (int i = 0; i < 50; i++) launch_cost_march(st, iL, iR, A, d, o, W, H, 0.55f, 1, 0, 1, 0, rb);
            CK(hipStreamSynchronize(st));
            launches += 50;
        }
        CK(hipMemset(dst, 0, sizeof(long long) * 4 * nb));
        launch_cost_march(st, iL, iR, A, d, o, W, H, 0.55f, 1, 0, 1, 0, rb);  // the launch that is analysed
        CK(hipStreamSynchronize(st));
        std::vector<long long> hs(4 * (size_t)nb);
        CK(hipMemcpy(hs.data(), dst, sizeof(long long) * 4 * nb, hipMemcpyDeviceToHost));
        std::vector<double> clk, cyc;
        struct Wv { long long key; long long start; double cycles; int wave; };
        std::vector<Wv> wv;
        long long t_min = -1;
        for (int b = 0; b < nb; b++)
            if (hs[4 * b + 1] > 0) {
                clk.push_back((double)hs[4 * b] / (double)hs[4 * b + 1] * 0.1);
                cyc.push_back((double)hs[4 * b]);
                const unsigned hw = (unsigned)(hs[4 * b + 3] & 0xffffffff), xcc = (unsigned)(hs[4 * b + 3] >> 32) & 0xf;
                const long long key = ((long long)xcc << 16) | (hw & 0xfff0);  // XCC, SE/SH/CU (bits 15:8), SIMD (bits 5:4)
                wv.push_back({key, hs[4 * b + 2], (double)hs[4 * b], b});
                if (t_min < 0 || hs[4 * b + 2] < t_min) t_min = hs[4 * b + 2];
            }
        std::sort(clk.begin(), clk.end()); std::sort(cyc.begin(), cyc.end());
        printf("k_cost_march np=1 %dx%d after %d back-to-back launches: in-kernel clock median %.3f GHz (p10 %.3f, p90 %.3f) over %zu waves; "
               "wave lifetime cycles p10 %.0f, median %.0f, p90 %.0f, p99 %.0f, max %.0f\n", W, H, launches, clk[clk.size() / 2], clk[clk.size() / 10], clk[clk.size() * 9 / 10], clk.size(),
               cyc[cyc.size() / 10], cyc[cyc.size() / 2], cyc[cyc.size() * 9 / 10], cyc[cyc.size() * 99 / 100], cyc.back());
        // census: the waves of every physical SIMD in the order they started
        std::sort(wv.begin(), wv.end(), [](const Wv &a, const Wv &b) { return a.key != b.key ? a.key < b.key : a.start < b.start; });
        std::vector<double> by_rank[8], start_by_rank[8];
        std::vector<int> per_simd;
        long long dwave_hist[6] = {0, 0, 0, 0, 0, 0};
        size_t i0 = 0;
        int printed = 0;
        while (i0 < wv.size()) {
            size_t i1 = i0;
            while (i1 < wv.size() && wv[i1].key == wv[i0].key) i1++;
            per_simd.push_back((int)(i1 - i0));
            for (size_t i = i0; i < i1 && i - i0 < 8; i++) {
                by_rank[i - i0].push_back(wv[i].cycles);
                start_by_rank[i - i0].push_back((double)(wv[i].start - t_min) * 0.01);  // us
                if (i > i0) {
                    const long long db = llabs((long long)wv[i].wave / 4 - (long long)wv[i - 1].wave / 4);  // distance in workgroups
                    dwave_hist[db == 0 ? 0 : db == 1 ? 1 : db < 8 ? 2 : db < 200 ? 3 : db < 300 ? 4 : 5]++;
                }
            }
            if (printed < 6) {
                printf("  SIMD %05llx:", (unsigned long long)wv[i0].key);
                for (size_t i = i0; i < i1; i++) printf("  [wg %d wave %d start %.2f us, %.0f kcycles]", wv[i].wave / 4, wv[i].wave % 4, (wv[i].start - t_min) * 0.01, wv[i].cycles / 1e3);
                printf("\n");
                printed++;
            }
            i0 = i1;
        }
        std::sort(per_simd.begin(), per_simd.end());
        printf("  %zu SIMDs hold waves: waves per SIMD min %d, median %d, max %d\n", per_simd.size(), per_simd.front(), per_simd[per_simd.size() / 2], per_simd.back());
        for (int r = 0; r < 8; r++)
            if (!by_rank[r].empty()) {
                std::sort(by_rank[r].begin(), by_rank[r].end()); std::sort(start_by_rank[r].begin(), start_by_rank[r].end());
                printf("  start rank %d on its SIMD: %zu waves, start median %.2f us (p90 %.2f), lifetime median %.0f kcycles (p10 %.0f, p90 %.0f)\n", r, by_rank[r].size(),
                       start_by_rank[r][start_by_rank[r].size() / 2], start_by_rank[r][start_by_rank[r].size() * 9 / 10],
                       by_rank[r][by_rank[r].size() / 2] / 1e3, by_rank[r][by_rank[r].size() / 10] / 1e3, by_rank[r][by_rank[r].size() * 9 / 10] / 1e3);
            }
        printf("  workgroup-index distance between consecutive starters of one SIMD: same wg %lld, 1: %lld, 2-7: %lld, 8-199: %lld, 200-299: %lld, >=300: %lld\n",
               dwave_hist[0], dwave_hist[1], dwave_hist[2], dwave_hist[3], dwave_hist[4], dwave_hist[5]);
        timeit("k_cost_march np=1 (stamped build)", [&]() { launch_cost_march(st, iL, iR, A, d, o, W, H, 0.55f, 1, 0, 1, 0, rb); });
        return 0;
    }
#endif
    if (argc > 4 && atoi(argv[4]) == 15) {  // tile heights of the 112-column K-smooth tile: bits against 36 rows, timing, the policy's choice: kbench W H reps 15
        float *o2; CK(hipMalloc(&o2, 12 * n));
        std::vector<float> ha(3 * n), hb(3 * n);
        for (int P : {5, 2, 0})
            for (int box = 0; box < 2; box++) {
                if (P == 0 && !box) continue;
                launch_smooth_fused(st, d, o, W, H, P, box, 36);
                CK(hipStreamSynchronize(st)); CK(hipMemcpy(ha.data(), o, 12 * n, hipMemcpyDeviceToHost));
                for (int rows : {35, 33, 29, 24, 18, 16}) {
                    CK(hipMemset(o2, 0xee, 12 * n));
                    launch_smooth_fused(st, d, o2, W, H, P, box, rows);
                    CK(hipStreamSynchronize(st)); CK(hipGetLastError());
                    CK(hipMemcpy(hb.data(), o2, 12 * n, hipMemcpyDeviceToHost));
                    size_t bad = 0;
                    for (size_t i = 0; i < 3 * n; i++) bad += memcmp(&ha[i], &hb[i], 4) != 0;
                    if (bad) printf("k_smooth_fused P=%d box=%d rows=%d vs rows=36: %zu of %zu values differ\n", P, box, rows, bad, 3 * n);
                }
            }
        printf("k_smooth_fused tile heights 16..35 against 36, P = 5 / 2 / 0, with and without the box: compared\n");
        printf("policy: latency %d rows, throughput %d rows\n", smooth_tile_rows(W, H, 1), smooth_tile_rows(W, H, 0));
        for (int round = 0; round < 2; round++)
            for (int rows = 36; rows >= 16; rows--) {
                const int tiles = ((W + 111) / 112) * ((H + rows - 1) / rows);
                char nm[64]; snprintf(nm, sizeof nm, "smooth p5+box rows=%d (%d tiles)", rows, tiles);
                timeit(nm, [&]() { launch_smooth_fused(st, d, o2, W, H, 5, 1, rows); });
            }
        CK(hipGetLastError());
        return 0;
    }
    if (argc > 4 && atoi(argv[4]) == 14) {  // channel-parallel marching K-cost (k_cost_march4) against k_cost_march and k_cost_split: bits + timing over strip heights: kbench W H reps 14
        float *o2; CK(hipMalloc(&o2, 12 * n));
        float *cz; CK(hipMalloc(&cz, 12 * n));  // a "coarser" field for the seeded form: same size / sqrt(2)
        const int Wc = (int)(W / 1.41421356), Hc = (int)(H / 1.41421356);
        std::vector<float> ha(3 * n), hb(3 * n);
        const SeedMap none{0, 0, 0, 0}, smap{Wc, Hc, 0, 0};
        for (int seeded = 0; seeded < 2; seeded++)
            for (int blend = 0; blend < 2; blend++) {
                CK(hipMemset(o, 0xff, 12 * n)); CK(hipMemset(o2, 0xee, 12 * n));
                if (seeded) { launch_cost_march_seeded(st, iL, iR, A, d, smap, o, W, H, 0.55f, blend, 0, rb); launch_cost_march4(st, iL, iR, A, d, o2, W, H, 0.55f, blend, 0, rb, smap); }
                else { launch_cost_march(st, iL, iR, A, d, o, W, H, 0.55f, blend, 0, 1, 0, rb); launch_cost_march4(st, iL, iR, A, d, o2, W, H, 0.55f, blend, 0, rb, none); }
                CK(hipStreamSynchronize(st)); CK(hipGetLastError());
                CK(hipMemcpy(ha.data(), o, 12 * n, hipMemcpyDeviceToHost)); CK(hipMemcpy(hb.data(), o2, 12 * n, hipMemcpyDeviceToHost));
                size_t bad = 0, first = 0;
                for (size_t i = 0; i < 3 * n; i++)
                    if (memcmp(&ha[i], &hb[i], 4) != 0 && !(ha[i] != ha[i] && hb[i] != hb[i])) { if (!bad) first = i; bad++; }
                printf("k_cost_march4 vs k_cost_march seeded=%d blend=%d: %zu of %zu values differ%s\n", seeded, blend, bad, 3 * n, bad ? "" : " (bit-exact)");
                if (bad) printf("  first at plane %zu y %zu x %zu: %g vs %g\n", first / n, (first % n) / W, first % W, ha[first], hb[first]);
            }
        for (int round = 0; round < 2; round++) {
            timeit("k_cost_split", [&]() { launch_cost_fused(st, iL, iR, A, d, o, W, H, 0.55f, 1); });
            timeit("k_cost_march rows=0", [&]() { launch_cost_march(st, iL, iR, A, d, o, W, H, 0.55f, 1, 0, 1, 0, rb); });
            if (n <= 300000) timeit("k_cost_small", [&]() { launch_cost_small(st, iL, iR, A, d, o, W, H, 0.55f, 1); });
            for (int rows : {0, 6, 8, 10, 12, 16, 20, 24, 32, 48}) {
                char nm[64]; snprintf(nm, sizeof nm, "k_cost_march4 rows=%d", rows);
                timeit(nm, [&]() { launch_cost_march4(st, iL, iR, A, d, o2, W, H, 0.55f, 1, rows, rb, none); });
            }
            timeit("k_cost_march4 seeded rows=0", [&]() { launch_cost_march4(st, iL, iR, A, d, o2, W, H, 0.55f, 1, 0, rb, smap); });
        }
        CK(hipGetLastError());
        return 0;
    }
    if (argc > 4 && atoi(argv[4]) == 13) {  // a coarse level's 22 iterations (k_cost_small + k_smooth_small, 44 dependent launches): eager launches against one HIP graph replay: kbench W H reps 13
        float *o2; CK(hipMalloc(&o2, 12 * n));
        auto level = [&]() {
            float *a = d, *b = o;
            for (int m = 0; m < 22; m++) {
                launch_cost_small(st, iL, iR, A, a, o2, W, H, 0.55f, 1);
                launch_smooth_small(st, o2, b, W, H, 5, 1, 32);
                std::swap(a, b);
            }
        };
        hipGraph_t graph; hipGraphExec_t exec;
        CK(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
        level();
        CK(hipStreamEndCapture(st, &graph));
        CK(hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0));
        for (int round = 0; round < 3; round++) {
            timeit("22 iterations, 44 eager launches", level);
            timeit("22 iterations, one graph replay", [&]() { CK(hipGraphLaunch(exec, st)); });
        }
        // host-side cost of issuing them (the stream is kept busy, so this is what the host pays, not what the GPU takes)
        for (int round = 0; round < 2; round++) {
            CK(hipStreamSynchronize(st));
            auto t0 = std::chrono::steady_clock::now();
            for (int i = 0; i < reps; i++) level();
            auto t1 = std::chrono::steady_clock::now();
            CK(hipStreamSynchronize(st));
            auto t2 = std::chrono::steady_clock::now();
            for (int i = 0; i < reps; i++) CK(hipGraphLaunch(exec, st));
            auto t3 = std::chrono::steady_clock::now();
            CK(hipStreamSynchronize(st));
            printf("host time to issue one level: eager %.1f us (%.2f us per launch), graph %.1f us\n", std::chrono::duration<double, std::micro>(t1 - t0).count() / reps,
                   std::chrono::duration<double, std::micro>(t1 - t0).count() / reps / 44, std::chrono::duration<double, std::micro>(t3 - t2).count() / reps);
        }
        CK(hipGetLastError());
        return 0;
    }
    if (argc > 4 && atoi(argv[4]) == 12) {  // marching K-cost, strips by age class (round 3): bits against uniform strips + timing over the shares: kbench W H reps 12
        float *o2; CK(hipMalloc(&o2, 12 * n));
        std::vector<float> ha(3 * n), hb(3 * n);
        march_age_permille[0] = march_age_permille[1] = 0;
        launch_cost_march(st, iL, iR, A, d, o, W, H, 0.55f, 1, 0, 1, 0, rb);
        CK(hipStreamSynchronize(st));
        CK(hipMemcpy(ha.data(), o, 12 * n, hipMemcpyDeviceToHost));
        std::vector<std::array<int, 2>> shares = {{0, 0}, {420, 350}, {450, 340}, {440, 360}, {460, 350}, {470, 340}, {480, 340}, {450, 360}, {470, 360}, {500, 320}, {480, 330}, {460, 330}, {430, 370}};
        if (argc > 6) {  // kbench W H reps 12 a0 b0 a1 b1 ...: the shares to sweep
            shares.clear();
            for (int i = 5; i + 1 < argc; i += 2) shares.push_back({atoi(argv[i]), atoi(argv[i + 1])});
        }
        for (auto &sh : shares) {
            march_age_permille[0] = sh[0]; march_age_permille[1] = sh[1];
            CK(hipMemset(o2, 0xff, 12 * n));
            launch_cost_march(st, iL, iR, A, d, o2, W, H, 0.55f, 1, 0, 1, 0, rb);
            CK(hipStreamSynchronize(st));
            CK(hipGetLastError());
            CK(hipMemcpy(hb.data(), o2, 12 * n, hipMemcpyDeviceToHost));
            size_t bad = 0, first = 0;
            for (size_t i = 0; i < 3 * n; i++)
                if (memcmp(&ha[i], &hb[i], 4) != 0 && !(ha[i] != ha[i] && hb[i] != hb[i])) { if (!bad) first = i; bad++; }
            printf("age shares %d/%d/%d vs uniform strips: %zu of %zu values differ%s\n", sh[0], sh[1], 1000 - sh[0] - sh[1], bad, 3 * n, bad ? "" : " (bit-exact)");
            if (bad) printf("  first at plane %zu y %zu x %zu: %g vs %g\n", first / n, (first % n) / W, first % W, ha[first], hb[first]);
        }
        for (int round = 0; round < 3; round++)
            for (auto &sh : shares) {
                march_age_permille[0] = sh[0]; march_age_permille[1] = sh[1];
                char nm[64]; snprintf(nm, sizeof nm, "k_cost_march age %d/%d/%d", sh[0], sh[1], 1000 - sh[0] - sh[1]);
                timeit(nm, [&]() { launch_cost_march(st, iL, iR, A, d, o2, W, H, 0.55f, 1, 0, 1, 0, rb); });
            }
        march_age_permille[0] = march_age_permille[1] = 0;
        CK(hipGetLastError());
        return 0;
    }
    if (argc > 4 && atoi(argv[4]) == 3) {  // counter runs of the marching K-cost: kbench W H reps 3 np rows
        const int np = argc > 5 ? atoi(argv[5]) : 1, rows = argc > 6 ? atoi(argv[6]) : 0;
        for (int i = 0; i < reps; i++) launch_cost_march(st, iL, iR, A, d, o, W, H, 0.55f, 1, 0, np, rows, rb);
        CK(hipStreamSynchronize(st));
        return 0;
    }
    if (argc > 4 && atoi(argv[4]) == 10) {  // strip heights of the marching K-cost on a mid level: kbench W H reps 10
        float *o2; CK(hipMalloc(&o2, 12 * n));
        for (int round = 0; round < 2; round++) {
            timeit("k_cost_split", [&]() { launch_cost_fused(st, iL, iR, A, d, o, W, H, 0.55f, 1); });
            for (int rows : {0, 8, 10, 12, 14, 16, 18, 20, 24, 28, 32, 40, 48, 64}) {
                char nm[64]; snprintf(nm, sizeof nm, "k_cost_march rows=%d", rows);
                timeit(nm, [&]() { launch_cost_march(st, iL, iR, A, d, o2, W, H, 0.55f, 1, 0, 1, rows, rb); });
            }
        }
        CK(hipGetLastError());
        return 0;
    }
    if (argc > 4 && atoi(argv[4]) == 9) {  // pyramid base: whole kernel against its level-0 part alone (no level-1 / level-2 sites): kbench W H reps 9
        uint8_t *rgb; CK(hipMalloc(&rgb, 3 * n)); CK(hipMemset(rgb, 77, 3 * n));
        const int W1 = (int)(W / 1.41421356), H1 = (int)(H / 1.41421356), W2 = W / 2, H2 = H / 2;
        float *l1 = R, *l2 = A;
        for (int round = 0; round < 3; round++) {
            timeit("k_pyr_base", [&]() { launch_pyr_base(st, rgb, 3 * W, W, H, L, l1, W1, H1, l2, W2, H2, rb); });
#define PYRV(ABL) timeit("k_pyr_base<" #ABL ">", [&]() { hipLaunchKernelGGL(k_pyr_base<ABL>, dim3(((W + BTX - 1) / BTX) * ((H + BTY - 1) / BTY)), dim3(256), 0, st, rgb, 3 * W, W, H, L, l1, W1, H1, l2, W2, H2, rb, (W + BTX - 1) / BTX, ((W + BTX - 1) / BTX) * ((H + BTY - 1) / BTY), Batch{1}, PyrWindow{0, 0, 0, 0}); })
            PYRV(1); PYRV(2); PYRV(6); PYRV(8); PYRV(9); PYRV(14); PYRV(15);
            timeit("k_pyr_base level 0 only", [&]() { launch_pyr_base(st, rgb, 3 * W, W, H, L, l1, 0, 0, l2, 0, 0, rb); });
            timeit("blur_decimate sqrt2", [&]() { launch_blur_decimate(st, L, W, H, o, W1, H1, 1.41421356f, nullptr); });
            timeit("blur_decimate 2", [&]() { launch_blur_decimate(st, L, W, H, o, W2, H2, 2.0f, nullptr); });
            timeit("seed 8M -> 16M", [&]() { launch_seed(st, d, W1, H1, o, W, H, 0, 0); });
            timeit("sqblur", [&]() { launch_sqblur_clamp(st, iL, W, H, o); });
        }
        CK(hipGetLastError());
        return 0;
    }
    if (argc > 4 && atoi(argv[4]) == 16) {  // k_iter_small (smoothing of iteration m + cost step of iteration m+1, one launch) against k_smooth_small + k_cost_small: kbench W H reps 16
        float *o2, *f1; CK(hipMalloc(&o2, 12 * n)); CK(hipMalloc(&f1, 12 * n));
        std::vector<float> ha(3 * n), hb(3 * n);
        for (int P : {5, 3, 1})
            for (int blend = 0; blend < 2; blend++) {
                CK(hipMemset(o, 0xff, 12 * n)); CK(hipMemset(o2, 0xee, 12 * n));
                launch_smooth_small(st, d, f1, W, H, P, 1, 32);
                launch_cost_small(st, iL, iR, A, f1, o, W, H, 0.55f, blend);
                launch_iter_small(st, iL, iR, A, d, o2, W, H, 0.55f, blend, P);
                CK(hipStreamSynchronize(st)); CK(hipGetLastError());
                CK(hipMemcpy(ha.data(), o, 12 * n, hipMemcpyDeviceToHost)); CK(hipMemcpy(hb.data(), o2, 12 * n, hipMemcpyDeviceToHost));
                size_t bad = 0, first = 0;
                for (size_t i = 0; i < 3 * n; i++)
                    if (memcmp(&ha[i], &hb[i], 4) != 0 && !(ha[i] != ha[i] && hb[i] != hb[i])) { if (!bad) first = i; bad++; }
                printf("k_iter_small vs k_smooth_small + k_cost_small P=%d blend=%d: %zu of %zu values differ%s\n", P, blend, bad, 3 * n, bad ? "" : " (bit-exact)");
                if (bad) printf("  first at plane %zu y %zu x %zu: %g vs %g\n", first / n, (first % n) / W, first % W, ha[first], hb[first]);
            }
        for (int round = 0; round < 2; round++) {
            for (int rh : {18, 24, 32}) {
                char nm[96]; snprintf(nm, sizeof nm, "smooth_small rh=%d + cost_small", rh);
                timeit(nm, [&]() { launch_smooth_small(st, d, f1, W, H, 5, 1, rh); launch_cost_small(st, iL, iR, A, f1, o, W, H, 0.55f, 1); });
            }
            timeit("k_iter_small", [&]() { launch_iter_small(st, iL, iR, A, d, o2, W, H, 0.55f, 1, 5); });
            for (int P = 0; P <= 4; P++) { char nm[64]; snprintf(nm, sizeof nm, "k_iter_small P=%d", P); timeit(nm, [&]() { launch_iter_small(st, iL, iR, A, d, o2, W, H, 0.55f, 1, P); }); }
            timeit("k_cost_small", [&]() { launch_cost_small(st, iL, iR, A, d, o, W, H, 0.55f, 1); });
            timeit("k_smooth_small rh=18", [&]() { launch_smooth_small(st, d, f1, W, H, 5, 1, 18); });
        }
        CK(hipGetLastError());
        return 0;
    }
    if (argc > 4 && atoi(argv[4]) == 7) {  // latency kernels of the coarse levels against the LDS-tiled ones: kbench W H reps 7
        float *o2; CK(hipMalloc(&o2, 12 * n));
        std::vector<float> ha(3 * n), hb(3 * n);
        auto cmp = [&](const char *what) {
            CK(hipStreamSynchronize(st));
            CK(hipMemcpy(ha.data(), o, 12 * n, hipMemcpyDeviceToHost)); CK(hipMemcpy(hb.data(), o2, 12 * n, hipMemcpyDeviceToHost));
            size_t bad = 0, first = 0;
            for (size_t i = 0; i < 3 * n; i++)
                if (memcmp(&ha[i], &hb[i], 4) != 0) { if (!bad) first = i; bad++; }
            printf("%s: %zu of %zu values differ%s\n", what, bad, 3 * n, bad ? "" : " (bit-exact)");
            if (bad) printf("  first at plane %zu y %zu x %zu: %g vs %g\n", first / n, (first % n) / W, first % W, ha[first], hb[first]);
        };
        for (int blend = 0; blend < 2; blend++) {
            CK(hipMemset(o, 0xff, 12 * n)); CK(hipMemset(o2, 0xee, 12 * n));
            launch_cost_fused(st, iL, iR, A, d, o, W, H, 0.55f, blend);
            launch_cost_small(st, iL, iR, A, d, o2, W, H, 0.55f, blend);
            cmp(blend ? "k_cost_small vs k_cost_split (blend)" : "k_cost_small vs k_cost_split");
        }
        for (int P = 0; P <= 5; P++)
            for (int box = 0; box < 2; box++) {
                if (P == 0 && !box) continue;
                for (int rh : {18, 24, 32}) {
                    CK(hipMemset(o, 0xff, 12 * n)); CK(hipMemset(o2, 0xee, 12 * n));
                    launch_smooth_fused(st, d, o, W, H, P, box);
                    launch_smooth_small(st, d, o2, W, H, P, box, rh);
                    char nm[96]; snprintf(nm, sizeof nm, "k_smooth_small rh=%d P=%d box=%d vs k_smooth_fused", rh, P, box);
                    cmp(nm);
                }
            }
        for (int round = 0; round < 2; round++) {
            for (int rh : {18, 24, 32}) {
                char nm[96]; snprintf(nm, sizeof nm, "k_smooth_small rh=%d p5+box", rh);
                timeit(nm, [&]() { launch_smooth_small(st, o, o2, W, H, 5, 1, rh); });
                snprintf(nm, sizeof nm, "small + smooth_small rh=%d", rh);
                timeit(nm, [&]() { launch_cost_small(st, iL, iR, A, d, o, W, H, 0.55f, 1); launch_smooth_small(st, o, o2, W, H, 5, 1, rh); });
            }
            timeit("k_cost_split", [&]() { launch_cost_fused(st, iL, iR, A, d, o, W, H, 0.55f, 1); });
            timeit("k_cost_small", [&]() { launch_cost_small(st, iL, iR, A, d, o2, W, H, 0.55f, 1); });
            timeit("k_smooth_fused p5+box", [&]() { launch_smooth_fused(st, o, o2, W, H, 5, 1); });
            timeit("split + smooth_fused", [&]() { launch_cost_fused(st, iL, iR, A, d, o, W, H, 0.55f, 1); launch_smooth_fused(st, o, o2, W, H, 5, 1); });
            timeit("small + smooth_fused", [&]() { launch_cost_small(st, iL, iR, A, d, o, W, H, 0.55f, 1); launch_smooth_fused(st, o, o2, W, H, 5, 1); });
        }
        CK(hipGetLastError());
        return 0;
    }
    if (argc > 4 && atoi(argv[4]) == 5) {  // marching K-smooth against the LDS-tiled one: bit comparison + timing
        float *o2; CK(hipMalloc(&o2, 12 * n));
        std::vector<float> ha(3 * n), hb(3 * n);
        for (int box = 0; box < 2; box++) {
            launch_smooth_fused(st, d, o, W, H, 5, box);
            CK(hipStreamSynchronize(st));
            CK(hipMemcpy(ha.data(), o, 12 * n, hipMemcpyDeviceToHost));
            for (int np = 1; np <= 2; np++) {
                CK(hipMemset(o2, 0xff, 12 * n));
                launch_smooth_march(st, d, o2, W, H, box, np, 0);
                CK(hipStreamSynchronize(st));
                CK(hipMemcpy(hb.data(), o2, 12 * n, hipMemcpyDeviceToHost));
                size_t bad = 0, first = 0;
                for (size_t i = 0; i < 3 * n; i++)
                    if (memcmp(&ha[i], &hb[i], 4) != 0 && !(ha[i] != ha[i] && hb[i] != hb[i])) { if (!bad) first = i; bad++; }
                printf("smooth march np=%d box=%d vs k_smooth_fused: %zu of %zu values differ%s\n", np, box, bad, 3 * n, bad ? "" : " (bit-exact)");
                if (bad) printf("  first at plane %zu y %zu x %zu: %g vs %g\n", first / n, (first % n) / W, first % W, ha[first], hb[first]);
            }
        }
        for (int round = 0; round < 2; round++)
            for (int box = 0; box < 2; box++) {
                char nm[64]; snprintf(nm, sizeof nm, "k_smooth_fused p5 box=%d", box);
                timeit(nm, [&]() { launch_smooth_fused(st, d, o, W, H, 5, box); });
                for (int np = 1; np <= 2; np++)
                    for (int rows : {0, 32, 48, 64, 96, 128}) {
                        snprintf(nm, sizeof nm, "k_smooth_march np=%d box=%d rows=%d", np, box, rows);
                        timeit(nm, [&]() { launch_smooth_march(st, d, o2, W, H, box, np, rows); });
                    }
            }
        CK(hipGetLastError());
        return 0;
    }
    if (argc > 4 && atoi(argv[4]) == 4) {  // timing of the marching K-cost, both float contracts: kbench W H reps 4
        for (int round = 0; round < 2; round++)
            for (int fm = 0; fm < 4; fm++)
                for (int rows : {0, 48, 96}) {
                    char nm[64]; snprintf(nm, sizeof nm, "march np=1 fmad=%d fastdiv=%d rows=%d", fm & 1, fm >> 1, rows);
                    timeit(nm, [&]() { launch_cost_march(st, iL, iR, A, d, o, W, H, 0.55f, 1, fm & 1, 1, rows, rbs[fm >> 1]); });
                }
        return 0;
    }
    if (argc > 4 && atoi(argv[4]) == 20) {  // K-smooth phase shift (round 5): the second workgroup of every CU starts late, once per launch: kbench W H reps 20
        auto run = [&](auto kern, int stx, int sty, int nt, const char *nm, int P = 5, int box = 1) {
            const size_t bytes = 3 * (size_t)(sty + 14) * (stx + 16 + UGSM_SMOOTH_PAD(stx)) * sizeof(float);
            (void)hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
            const int stxn = (W + stx - 1) / stx, stn = stxn * ((H + sty - 1) / sty);
            timeit(nm, [&]() { hipLaunchKernelGGL(kern, dim3(stn), dim3(nt), bytes, st, d, o, W, H, P, box, stxn, stn, sty, Batch{1}); });
        };
        for (int round = 0; round < 3; round++) {
            run((k_smooth_fused<112, 36, 512, 0, true>), 112, 36, 512, "smooth<112,36,512> p5+box");
            for (int sl : {0, 1, 2, 3, 4, 6, 8}) {
                CK(hipMemcpyToSymbol(HIP_SYMBOL(smooth_phase_sleep), &sl, sizeof sl));
                char nm[96]; snprintf(nm, sizeof nm, "  phase shift %d x 3.5 us p5+box", sl);
                run((k_smooth_fused<112, 36, 512, 16, true>), 112, 36, 512, nm);
            }
            run((k_smooth_fused<112, 36, 512, 0, true>), 112, 36, 512, "smooth<112,36,512> p5", 5, 0);
            for (int sl : {0, 2, 4}) {
                CK(hipMemcpyToSymbol(HIP_SYMBOL(smooth_phase_sleep), &sl, sizeof sl));
                char nm[96]; snprintf(nm, sizeof nm, "  phase shift %d x 3.5 us p5", sl);
                run((k_smooth_fused<112, 36, 512, 16, true>), 112, 36, 512, nm, 5, 0);
            }
            run((k_smooth_fused<112, 36, 512, 8, true>), 112, 36, 512, "smooth<112,36,512> p5, the box's halo, no box", 5, 0);
            run((k_smooth_fused<112, 36, 512, 0, true>), 112, 36, 512, "smooth<112,36,512> p0+box", 0, 1);
            run((k_smooth_fused<112, 36, 512, 0, true>), 112, 36, 512, "smooth<112,36,512> p0", 0, 0);
            for (int P = 1; P <= 7; P++) {   // the cost of a launch over its halo: P passes without the box (halo P), with the box (halo P + 2), and with the box's halo alone
                char nm[96];
                snprintf(nm, sizeof nm, "  P=%d no box (halo %d)", P, P);
                run((k_smooth_fused<112, 36, 512, 0, true>), 112, 36, 512, nm, P, 0);
                if (P <= 5) {
                    snprintf(nm, sizeof nm, "  P=%d + box (halo %d)", P, P + 2);
                    run((k_smooth_fused<112, 36, 512, 0, true>), 112, 36, 512, nm, P, 1);
                    snprintf(nm, sizeof nm, "  P=%d, halo %d, no box", P, P + 2);
                    run((k_smooth_fused<112, 36, 512, 8, true>), 112, 36, 512, nm, P, 0);
                }
            }
            // occupancy: three workgroups per CU (LDS 53.8 KB each) -- eight waves each at <= 80 VGPRs, or four waves each
            run((k_smooth_fused<112, 21, 512, 0, true, 6>), 112, 21, 512, "smooth<112,21,512> occ 6 (3 WG/CU) p5+box");
            run((k_smooth_fused<112, 21, 512, 0, true, 4>), 112, 21, 512, "smooth<112,21,512> occ 4 (2 WG/CU) p5+box");
            run((k_smooth_fused<112, 21, 256, 0, true, 3>), 112, 21, 256, "smooth<112,21,256> occ 3 (3 WG/CU) p5+box");
            run((k_smooth_fused<112, 21, 384, 0, true, 5>), 112, 21, 384, "smooth<112,21,384> occ 5 (3 WG/CU x 6 waves) p5+box");
            run((k_smooth_fused<112, 28, 512, 0, true, 5>), 112, 28, 512, "smooth<112,28,512> occ 5 p5+box");
            run((k_smooth_fused<112, 36, 512, 0, true, 5>), 112, 36, 512, "smooth<112,36,512> occ 5 p5+box");
        }
        CK(hipGetLastError());
        return 0;
    }
    if (argc > 4 && atoi(argv[4]) == 19) {  // K-cost issue-slot study: the production launch (np = 1, strips by age class) of THIS build
        // (-DMARCH_ILV=0|1|2 ...) -- bits against k_cost_split, then `reps` back-to-back launches, three rounds: kbench W H reps 19
        float *o2; CK(hipMalloc(&o2, 12 * n));
        std::vector<float> ha(3 * n), hb(3 * n);
        launch_cost_fused(st, iL, iR, A, d, o, W, H, 0.55f, 1);
        CK(hipMemset(o2, 0xff, 12 * n));
        launch_cost_march(st, iL, iR, A, d, o2, W, H, 0.55f, 1, 0, 1, 0, rb);
        CK(hipStreamSynchronize(st));
        CK(hipMemcpy(ha.data(), o, 12 * n, hipMemcpyDeviceToHost));
        CK(hipMemcpy(hb.data(), o2, 12 * n, hipMemcpyDeviceToHost));
        size_t bad = 0;
        for (size_t i = 0; i < 3 * n; i++)
            if (memcmp(&ha[i], &hb[i], 4) != 0 && !(ha[i] != ha[i] && hb[i] != hb[i])) bad++;
#ifndef MARCH_VARIANT
#define MARCH_VARIANT "base"
#endif
        printf("variant %s: march np=1 vs k_cost_split: %zu of %zu values differ%s\n", MARCH_VARIANT, bad, 3 * n, bad ? "" : " (bit-exact)");
        for (int round = 0; round < 3; round++) timeit("k_cost_march np=1 rows=0 [" MARCH_VARIANT "]", [&]() { launch_cost_march(st, iL, iR, A, d, o2, W, H, 0.55f, 1, 0, 1, 0, rb); });
        CK(hipGetLastError());
        return bad ? 1 : 0;
    }
    if (argc > 4 && atoi(argv[4]) == 2) {  // marching K-cost against the LDS-tiled one: bit comparison + timing
        float *o2; CK(hipMalloc(&o2, 12 * n));
        std::vector<float> ha(3 * n), hb(3 * n);
        launch_cost_fused(st, iL, iR, A, d, o, W, H, 0.55f, 1);
        CK(hipStreamSynchronize(st));
        CK(hipMemcpy(ha.data(), o, 12 * n, hipMemcpyDeviceToHost));
        for (int np = 1; np <= 2; np++) {
            CK(hipMemset(o2, 0xff, 12 * n));
            launch_cost_march(st, iL, iR, A, d, o2, W, H, 0.55f, 1, 0, np, 0, rb);
            CK(hipStreamSynchronize(st));
            CK(hipMemcpy(hb.data(), o2, 12 * n, hipMemcpyDeviceToHost));
            size_t bad = 0, first = 0;
            for (size_t i = 0; i < 3 * n; i++)
                if (memcmp(&ha[i], &hb[i], 4) != 0 && !(ha[i] != ha[i] && hb[i] != hb[i])) { if (!bad) first = i; bad++; }
            printf("march np=%d vs k_cost_split: %zu of %zu values differ%s\n", np, bad, 3 * n, bad ? "" : " (bit-exact)");
            if (bad) printf("  first at plane %zu y %zu x %zu: %g vs %g\n", first / n, (first % n) / W, first % W, ha[first], hb[first]);
        }
        const int rows_list[] = {0, 24, 32, 48, 64, 96, 128};
        for (int round = 0; round < 2; round++) {
            SPLIT(0);
            for (int np = 1; np <= 2; np++)
                for (int rows : rows_list) {
                    char nm[64]; snprintf(nm, sizeof nm, "k_cost_march np=%d rows=%d", np, rows);
                    timeit(nm, [&]() { launch_cost_march(st, iL, iR, A, d, o2, W, H, 0.55f, 1, 0, np, rows, rb); });
                }
        }
        CK(hipGetLastError());
        return 0;
    }
    for (int round = 0; round < 2; round++) {  // interleaved rounds, one process (A/B rule)
        COST(0); SPLIT(0);
        timeit("k_cost_split<0, 6 waves>", [&]() { hipLaunchKernelGGL((k_cost_split<0, 6>), dim3(cnt), dim3(512), 0, st, iL, iR, A, d, o, W, H, 1.0f, 1, ctx, cnt); });
    }
    {   // phase stamps of k_cost_split<256>
        const int nb = std::min(cnt, 8192);
        long long *dst; CK(hipMalloc(&dst, sizeof(long long) * 48 * nb)); CK(hipMemset(dst, 0, sizeof(long long) * 48 * nb));
        CK(hipMemcpyToSymbol(HIP_SYMBOL(g_cost_stamps), &dst, sizeof(dst)));
        hipLaunchKernelGGL(k_cost_split<256>, dim3(cnt), dim3(512), 0, st, iL, iR, A, d, o, W, H, 1.0f, 1, ctx, cnt);
        CK(hipStreamSynchronize(st));
        std::vector<long long> hs(48 * (size_t)nb);
        CK(hipMemcpy(hs.data(), dst, sizeof(long long) * 48 * nb, hipMemcpyDeviceToHost));
        const char *names[16] = {"start", "P0 issued", "c0 P1+bar (data in LDS)", "c0 P2+bar", "c0 P2.5+bar", "c0 P3", "c1 P1+bar", "c1 P2+bar", "c1 P2.5+bar", "c1 P3",
                                 "c2 P1+bar", "c2 P2+bar", "c2 P2.5+bar", "c2 P3", "bar after P3", "end"};
        for (int role = 0; role < 2; role++) {
            printf("k_cost_split phase stamps, role %d (mean cycles since previous stamp over %d workgroups; shader clock)\n", role, nb);
            double tot = 0;
            {   // P0 detail: start -> 16 (ridx computed: d3 arrived) -> 17 (L loads issued) -> 18 (R gathers issued) -> 1
                const int seq[5] = {0, 16, 17, 18, 1};
                const char *nm[5] = {"", "  P0: d3 in, ridx done", "  P0: L loads issued", "  P0: R gathers issued", "  P0: A/d loads issued"};
                for (int j = 1; j < 5; j++) {
                    double sum = 0; int m = 0;
                    for (int b = 0; b < nb; b++) {
                        const long long *t = &hs[((size_t)b * 2 + role) * 24];
                        if (t[seq[j]] && t[seq[j - 1]]) { sum += (double)(t[seq[j]] - t[seq[j - 1]]); m++; }
                    }
                    printf("  %-26s %9.0f\n", nm[j], m ? sum / m : 0.0);
                }
            }
            for (int i = 1; i < 16; i++) {
                double sum = 0; int m = 0;
                for (int b = 0; b < nb; b++) {
                    const long long *t = &hs[((size_t)b * 2 + role) * 24];
                    if (t[i] && t[i - 1]) { sum += (double)(t[i] - t[i - 1]); m++; }
                }
                printf("  %-26s %9.0f\n", names[i], m ? sum / m : 0.0);
                tot += m ? sum / m : 0.0;
            }
            printf("  %-26s %9.0f\n", "total", tot);
        }
        long long t0 = hs[0], t1 = 0;
        for (int b = 0; b < nb; b++) { t0 = std::min(t0, hs[(size_t)b * 48]); t1 = std::max(t1, hs[(size_t)b * 48 + 15]); }
        printf("  first start -> last end of those workgroups: %lld ticks\n", t1 - t0);
        dst = nullptr; CK(hipMemcpyToSymbol(HIP_SYMBOL(g_cost_stamps), &dst, sizeof(dst)));
    }
    {
        auto run = [&](auto kern, int stx, int sty, int nt, const char *nm, int P = 5, int box = 1) {
            const size_t bytes = 3 * (size_t)(sty + 14) * (stx + 16 + UGSM_SMOOTH_PAD(stx)) * sizeof(float);
            (void)hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
            const int stxn = (W + stx - 1) / stx, stn = stxn * ((H + sty - 1) / sty);
            timeit(nm, [&]() { hipLaunchKernelGGL(kern, dim3(stn), dim3(nt), bytes, st, d, o, W, H, P, box, stxn, stn, sty, Batch{1}); });
        };
        for (int round = 0; round < 2; round++) {
            run((k_smooth_fused<112, 36, 512, 0, true>), 112, 36, 512, "smooth<112,36,512> p5+box");
            run((k_smooth_fused<112, 36, 512, 0, true>), 112, 36, 512, "smooth<112,36,512> p5", 5, 0);
            run((k_smooth_fused<112, 36, 512, 2, true>), 112, 36, 512, "smooth<112,36,512> selects p5+box");
            run((k_smooth_fused<112, 36, 512, 2, true>), 112, 36, 512, "smooth<112,36,512> selects p5", 5, 0);
            run((k_smooth_fused<112, 36, 512, 0, true>), 112, 36, 512, "smooth<112,36,512> p0", 0, 0);
            run((k_smooth_fused<112, 36, 512, 0, true>), 112, 36, 512, "smooth<112,36,512> p1", 1, 0);
        }
    }
    if (argc > 4 && atoi(argv[4]) == 17) {  // k_smooth_pipe against k_smooth_fused (112-column tile): bit-exactness and time
        unsigned *queue; CK(hipMalloc(&queue, 64));
        float *o2; CK(hipMalloc(&o2, 12 * n));
        std::vector<float> ha(3 * n), hb(3 * n);
        for (int wgs : {256, 512, 128}) {
            smooth_pipe_workgroups = wgs;
            for (int box = 0; box < 2; box++)
                for (int P : {5, 1, 0, 3}) {
                    if (P == 0 && !box) continue;
                    for (int rows : {36, 29}) {
                        launch_smooth_fused(st, d, o, W, H, P, box, rows);
                        CK(hipMemsetAsync(queue, 0, 64, st));
                        launch_smooth_pipe(st, d, o2, W, H, P, box, rows, queue);
                        CK(hipStreamSynchronize(st)); CK(hipGetLastError());
                        CK(hipMemcpy(ha.data(), o, 12 * n, hipMemcpyDeviceToHost)); CK(hipMemcpy(hb.data(), o2, 12 * n, hipMemcpyDeviceToHost));
                        size_t bad = 0;
                        for (size_t i = 0; i < 3 * n; i++) bad += memcmp(&ha[i], &hb[i], 4) != 0;
                        printf("wgs %d P=%d box=%d rows=%d: %zu of %zu values differ%s\n", wgs, P, box, rows, bad, 3 * n, bad ? "  <-- MISMATCH" : "");
                    }
                }
            for (int round = 0; round < 2; round++) {
                char nm[64];
                timeit("k_smooth_fused p5+box", [&]() { launch_smooth_fused(st, d, o, W, H, 5, 1, 36); });
                snprintf(nm, sizeof nm, "k_smooth_pipe p5+box wgs%d", wgs);
                timeit(nm, [&]() { CK(hipMemsetAsync(queue, 0, 64, st)); launch_smooth_pipe(st, d, o2, W, H, 5, 1, 36, queue); });
                timeit("k_smooth_fused p5", [&]() { launch_smooth_fused(st, d, o, W, H, 5, 0, 36); });
                snprintf(nm, sizeof nm, "k_smooth_pipe p5 wgs%d", wgs);
                timeit(nm, [&]() { CK(hipMemsetAsync(queue, 0, 64, st)); launch_smooth_pipe(st, d, o2, W, H, 5, 0, 36, queue); });
                snprintf(nm, sizeof nm, "k_smooth_pipe p0 wgs%d", wgs);
                timeit(nm, [&]() { CK(hipMemsetAsync(queue, 0, 64, st)); launch_smooth_pipe(st, d, o2, W, H, 0, 0, 36, queue); });
            }
        }
        return 0;
    }
    if (argc > 4 && atoi(argv[4]) == 18) {  // how two kernels on two streams share the chip: alone, and side by side
        hipStream_t s2; int lo = 0, hi = 0;
        (void)hipDeviceGetStreamPriorityRange(&lo, &hi);
        hipStream_t s1; CK(hipStreamCreateWithPriority(&s1, hipStreamNonBlocking, hi)); CK(hipStreamCreateWithPriority(&s2, hipStreamNonBlocking, hi));
        float *o2; CK(hipMalloc(&o2, 12 * n));
        auto kc = [&](hipStream_t q, float *dst) { launch_cost_march(q, iL, iR, A, d, dst, W, H, 1.0f, 1, 0, 1, 0, rb); };
        auto ks = [&](hipStream_t q, float *dst) { launch_smooth_fused(q, d, dst, W, H, 5, 1, 36); };
        auto k4 = [&](hipStream_t q, float *dst) { launch_cost_march4(q, iL, iR, A, d, dst, W, H, 1.0f, 1, 0, rb, SeedMap{0, 0, 0, 0}); };
        auto wall = [&](auto fa, auto fb, bool both) {
            for (int i = 0; i < 2; i++) { fa(s1, o); if (both) fb(s2, o2); }
            CK(hipDeviceSynchronize());
            auto t0 = std::chrono::steady_clock::now();
            for (int i = 0; i < reps; i++) { fa(s1, o); if (both) fb(s2, o2); }
            CK(hipDeviceSynchronize());
            return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / reps;
        };
        for (int round = 0; round < 3; round++) {
            march4_small_registers = round == 2;  // third round: k_cost_march4 compiled for 8 waves per SIMD (<= 64 VGPRs): it fits beside two K-smooth workgroups
            if (round == 2) printf("k_cost_march4 at <= 64 VGPRs:\n");
            const double c = wall(kc, kc, false), sm = wall(ks, ks, false), m4 = wall(k4, k4, false);
            printf("alone (us per launch): K-cost march %.1f   K-smooth p5+box %.1f   K-cost march4 %.1f\n", c, sm, m4);
            const double cs = wall(kc, ks, true), cc = wall(kc, kc, true), ss = wall(ks, ks, true), s4 = wall(k4, ks, true);
            printf("side by side on two streams (us per pair of launches; the sum of the two alone in brackets):\n");
            printf("  K-cost march  + K-smooth  %.1f (%.1f)  -> %.2f of the sum\n", cs, c + sm, cs / (c + sm));
            printf("  K-cost march4 + K-smooth  %.1f (%.1f)  -> %.2f of the sum\n", s4, m4 + sm, s4 / (m4 + sm));
            printf("  K-cost march  + K-cost    %.1f (%.1f)  -> %.2f\n", cc, 2 * c, cc / (2 * c));
            printf("  K-smooth      + K-smooth  %.1f (%.1f)  -> %.2f\n", ss, 2 * sm, ss / (2 * sm));
        }
        return 0;
    }
    timeit("k_smooth_fused p5", [&]() { launch_smooth_fused(st, d, o, W, H, 5, 0); });
    timeit("k_smooth_fused p5+box", [&]() { launch_smooth_fused(st, d, o, W, H, 5, 1); });
    {
        const int W1 = (int)(W / 1.41421356), H1 = (int)(H / 1.41421356), W2 = W / 2, H2 = H / 2;
        for (int round = 0; round < 2; round++) {
            timeit("blur_decimate sqrt2", [&]() { launch_blur_decimate(st, L, W, H, o, W1, H1, 1.41421356f, nullptr); });
            timeit("blur_decimate 2", [&]() { launch_blur_decimate(st, L, W, H, o, W2, H2, 2.0f, nullptr); });
            timeit("sqblur", [&]() { launch_sqblur_clamp(st, iL, W, H, o); });
        }
    }
    CK(hipGetLastError());
    return 0;
}
