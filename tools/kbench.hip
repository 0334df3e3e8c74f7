// kbench -- stand-alone timing harness for the kernels (development tool, not product).
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -ffp-contract=off -fno-fast-math -fno-slp-vectorize -Iinclude tools/kbench.hip -o tools/kbench
//   ./kbench W H reps mode [args]            KBENCH_DATA=file: real planes (tools/kbench_real.py) instead of random ones
// The kernels are the product's own sources (+ libugsm_dev.so's LDS-tiled K-cost as the second opinion), compiled into this binary:
// the product files hold no development switch, so what is timed here is what libugsm.so runs.  The variants that rounds 2-5 measured
// and did not keep (two pixels per lane, FMA contraction, the pipelined / marching / product-form / Newton K-smooth, k_iter_small, ...)
// are in the git history; their results are in profiles/r0[2-5]_kbench_* and docs/HISTORY.md.
//   2   k_cost_march against k_cost_split: bits + timing over strip heights
//   5   k_smooth_fused: time over the passes, with and without the box
//   7   the coarse levels' latency kernels against the LDS-tiled ones: bits + timing
//   9   pyramid, seeding, A planes: timing
//   10  strip heights of the marching K-cost on a mid level
//   12  strips by age class: bits against uniform strips + timing over the shares
//   13  a coarse level's 44 dependent launches: eager against one HIP graph replay
//   14  k_cost_march4 against k_cost_march / k_cost_split / k_cost_small: bits + timing over strip heights
//   15  tile heights of the 112-column K-smooth tile: bits against 36 rows, timing
//   18  two kernels on two streams: alone and side by side
#define UGSM_DEV_LIB 1  // (the launch header's declarations of the dev kernels)
#include "../ug_stereomatcher_amd/csrc/ugsm_kernels_aux.hip"
#include "../ug_stereomatcher_amd/csrc/ugsm_kernels_pyr.hip"
#include "../ug_stereomatcher_amd/csrc/ugsm_kernels_smooth.hip"
#include "../ug_stereomatcher_amd/csrc/ugsm_kernels_march.hip"
#include "../ug_stereomatcher_amd/csrc/ugsm_kernels_march4.hip"
#include "../ug_stereomatcher_amd/csrc/ugsm_kernels_small.hip"
#include "../ug_stereomatcher_amd/csrc/dev/ugsm_dev_cost_tiled.hip"
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
    const int mode = argc > 4 ? atoi(argv[4]) : 0;
    size_t n = (size_t)W * H;
    std::vector<float> hL(3 * n), hR(3 * n), hd(3 * n);
    unsigned s = 12345;
    auto rnd = [&]() { s = s * 1664525u + 1013904223u; return (s >> 8) * (1.0f / 16777216.0f); };
    for (size_t i = 0; i < 3 * n; i++) { hL[i] = 1 + 254 * rnd(); hR[i] = 1 + 254 * rnd(); }
    for (size_t i = 0; i < n; i++) { int x = i % W, y = i / W; hd[i] = 30.0f * sinf(x * 0.002f) * cosf(y * 0.003f) + 0.3f * (rnd() - 0.5f); hd[n + i] = 0.75f * sinf(y * 0.002f) + 0.3f * (rnd() - 0.5f); hd[2 * n + i] = 0.3f + 0.7f * rnd(); }
    if (const char *data_file = getenv("KBENCH_DATA")) {  // real data: 9 planes of W*H floats (L0 L1 L2 R0 R1 R2 dx dy conf), e.g. from tools/kbench_real.py
        FILE *f = fopen(data_file, "rb");
        if (!f || fread(hL.data(), 4, 3 * n, f) != 3 * n || fread(hR.data(), 4, 3 * n, f) != 3 * n || fread(hd.data(), 4, 3 * n, f) != 3 * n) { printf("cannot read %s\n", data_file); return 1; }
        fclose(f);
        printf("inputs from %s\n", data_file);
    }
    float *L, *R, *A, *d, *o, *o2;
    CK(hipMalloc(&L, 12 * n)); CK(hipMalloc(&R, 12 * n)); CK(hipMalloc(&A, 12 * n)); CK(hipMalloc(&d, 12 * n)); CK(hipMalloc(&o, 12 * n)); CK(hipMalloc(&o2, 12 * n));
    CK(hipMemcpy(L, hL.data(), 12 * n, hipMemcpyHostToDevice)); CK(hipMemcpy(R, hR.data(), 12 * n, hipMemcpyHostToDevice));
    CK(hipMemcpy(d, hd.data(), 12 * n, hipMemcpyHostToDevice));
    hipStream_t st; CK(hipStreamCreate(&st));
    unsigned *rb; CK(hipMalloc(&rb, 64)); CK(hipMemset(rb, 0, 64));  // range flag: inputs are in [1, 255]
    Img3 iL{L, W, n}, iR{R, W, n};
    launch_sqblur_clamp(st, iL, W, H, A);
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    auto timeit = [&](const char *name, auto fn) {
        fn(); CK(hipStreamSynchronize(st));
        CK(hipEventRecord(a, st));
        for (int i = 0; i < reps; i++) fn();
        CK(hipEventRecord(b, st)); CK(hipEventSynchronize(b));
        float ms; CK(hipEventElapsedTime(&ms, a, b));
        printf("%-36s %4dx%-4d  %8.1f us/launch  %7.1f Gpx/s\n", name, W, H, 1e3 * ms / reps, n / (ms / reps) / 1e6);
    };
    std::vector<float> ha(3 * n), hb(3 * n);
    auto cmp = [&](const char *what) {  // o against o2, bit for bit (NaN == NaN)
        CK(hipStreamSynchronize(st)); CK(hipGetLastError());
        CK(hipMemcpy(ha.data(), o, 12 * n, hipMemcpyDeviceToHost)); CK(hipMemcpy(hb.data(), o2, 12 * n, hipMemcpyDeviceToHost));
        size_t bad = 0, first = 0;
        for (size_t i = 0; i < 3 * n; i++)
            if (memcmp(&ha[i], &hb[i], 4) != 0 && !(ha[i] != ha[i] && hb[i] != hb[i])) { if (!bad) first = i; bad++; }
        printf("%s: %zu of %zu values differ%s\n", what, bad, 3 * n, bad ? "" : " (bit-exact)");
        if (bad) printf("  first at plane %zu y %zu x %zu: %g vs %g\n", first / n, (first % n) / W, first % W, ha[first], hb[first]);
    };
    const SeedMap none{0, 0, 0, 0};
    auto march = [&](float *dst, int blend = 1, int rows = 0, hipStream_t q = nullptr) { launch_cost_march(q ? q : st, iL, iR, A, d, dst, W, H, 0.55f, blend, rows, rb); };
    if (mode == 2) {
        launch_cost_fused(st, iL, iR, A, d, o, W, H, 0.55f, 1);
        CK(hipMemset(o2, 0xff, 12 * n));
        march(o2);
        cmp("k_cost_march vs k_cost_split");
        for (int round = 0; round < 2; round++) {
            timeit("k_cost_split", [&]() { launch_cost_fused(st, iL, iR, A, d, o, W, H, 0.55f, 1); });
            for (int rows : {0, 24, 32, 48, 64, 96, 128}) {
                char nm[64]; snprintf(nm, sizeof nm, "k_cost_march rows=%d", rows);
                timeit(nm, [&]() { march(o2, 1, rows); });
            }
        }
    } else if (mode == 5) {
        for (int round = 0; round < 2; round++)
            for (int box = 0; box < 2; box++)
                for (int P = 0; P <= 5; P++) {
                    char nm[64]; snprintf(nm, sizeof nm, "k_smooth_fused P=%d box=%d", P, box);
                    timeit(nm, [&]() { launch_smooth_fused(st, d, o, W, H, P, box, 36); });
                }
    } else if (mode == 7) {
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
            timeit("small + smooth_fused", [&]() { launch_cost_small(st, iL, iR, A, d, o, W, H, 0.55f, 1); launch_smooth_fused(st, o, o2, W, H, 5, 1); });
        }
    } else if (mode == 9) {
        uint8_t *rgb; CK(hipMalloc(&rgb, 3 * n)); CK(hipMemset(rgb, 77, 3 * n));
        const int W1 = (int)(W / 1.41421356), H1 = (int)(H / 1.41421356), W2 = W / 2, H2 = H / 2;
        float *l1 = R, *l2 = A;
        for (int round = 0; round < 3; round++) {
            timeit("k_pyr_base", [&]() { launch_pyr_base(st, rgb, 3 * W, W, H, L, l1, W1, H1, l2, W2, H2, rb); });
            timeit("k_pyr_base_march (fovea window)", [&]() { launch_pyr_base(st, rgb, 3 * W, W, H, L, l1, W1, H1, l2, W2, H2, rb, nullptr, PyrWindow{W / 2 - 307, H / 2 - 203, 615, 407}); });
            timeit("blur_decimate sqrt2", [&]() { launch_blur_decimate(st, L, W, H, o, W1, H1, 1.41421356f, nullptr); });
            timeit("blur_decimate 2", [&]() { launch_blur_decimate(st, L, W, H, o, W2, H2, 2.0f, nullptr); });
            timeit("seed 8M -> 16M", [&]() { launch_seed(st, d, W1, H1, o, W, H, 0, 0); });
            timeit("sqblur", [&]() { launch_sqblur_clamp(st, iL, W, H, o); });
        }
    } else if (mode == 10) {
        for (int round = 0; round < 2; round++) {
            timeit("k_cost_split", [&]() { launch_cost_fused(st, iL, iR, A, d, o, W, H, 0.55f, 1); });
            for (int rows : {0, 8, 10, 12, 14, 16, 18, 20, 24, 28, 32, 40, 48, 64}) {
                char nm[64]; snprintf(nm, sizeof nm, "k_cost_march rows=%d", rows);
                timeit(nm, [&]() { march(o2, 1, rows); });
            }
        }
    } else if (mode == 12) {
        march_age_permille[0] = march_age_permille[1] = 0;
        march(o);
        std::vector<std::array<int, 2>> shares = {{0, 0}, {420, 350}, {450, 340}, {440, 360}, {460, 350}, {470, 340}, {480, 340}, {450, 360}, {470, 360}, {500, 320}, {480, 330}, {460, 330}, {430, 370}};
        if (argc > 6) {  // kbench W H reps 12 a0 b0 a1 b1 ...: the shares to sweep
            shares.clear();
            for (int i = 5; i + 1 < argc; i += 2) shares.push_back({atoi(argv[i]), atoi(argv[i + 1])});
        }
        for (auto &sh : shares) {
            march_age_permille[0] = sh[0]; march_age_permille[1] = sh[1];
            CK(hipMemset(o2, 0xff, 12 * n));
            march(o2);
            char nm[96]; snprintf(nm, sizeof nm, "age shares %d/%d/%d vs uniform strips", sh[0], sh[1], 1000 - sh[0] - sh[1]);
            cmp(nm);
        }
        for (int round = 0; round < 3; round++)
            for (auto &sh : shares) {
                march_age_permille[0] = sh[0]; march_age_permille[1] = sh[1];
                char nm[64]; snprintf(nm, sizeof nm, "k_cost_march age %d/%d/%d", sh[0], sh[1], 1000 - sh[0] - sh[1]);
                timeit(nm, [&]() { march(o2); });
            }
    } else if (mode == 13) {
        auto level = [&]() {
            float *x = d, *y = o;
            for (int m = 0; m < 22; m++) {
                launch_cost_small(st, iL, iR, A, x, o2, W, H, 0.55f, 1);
                launch_smooth_small(st, o2, y, W, H, 5, 1, 32);
                std::swap(x, y);
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
        for (int round = 0; round < 2; round++) {  // host-side cost of issuing them (the stream is kept busy: what the host pays, not what the GPU takes)
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
    } else if (mode == 14) {
        const int Wc = (int)(W / 1.41421356), Hc = (int)(H / 1.41421356);  // a "coarser" field for the seeded form: the same buffer read at size / sqrt(2)
        const SeedMap smap{Wc, Hc, 0, 0};
        for (int seeded = 0; seeded < 2; seeded++)
            for (int blend = 0; blend < 2; blend++) {
                CK(hipMemset(o, 0xff, 12 * n)); CK(hipMemset(o2, 0xee, 12 * n));
                if (seeded) { launch_cost_march_seeded(st, iL, iR, A, d, smap, o, W, H, 0.55f, blend, 0, rb); launch_cost_march4(st, iL, iR, A, d, o2, W, H, 0.55f, blend, 0, rb, smap); }
                else { march(o, blend); launch_cost_march4(st, iL, iR, A, d, o2, W, H, 0.55f, blend, 0, rb, none); }
                char nm[96]; snprintf(nm, sizeof nm, "k_cost_march4 vs k_cost_march seeded=%d blend=%d", seeded, blend);
                cmp(nm);
            }
        for (int round = 0; round < 2; round++) {
            timeit("k_cost_split", [&]() { launch_cost_fused(st, iL, iR, A, d, o, W, H, 0.55f, 1); });
            timeit("k_cost_march rows=0", [&]() { march(o); });
            if (n <= 300000) timeit("k_cost_small", [&]() { launch_cost_small(st, iL, iR, A, d, o, W, H, 0.55f, 1); });
            for (int rows : {0, 6, 8, 10, 12, 16, 20, 24, 32, 48}) {
                char nm[64]; snprintf(nm, sizeof nm, "k_cost_march4 rows=%d", rows);
                timeit(nm, [&]() { launch_cost_march4(st, iL, iR, A, d, o2, W, H, 0.55f, 1, rows, rb, none); });
            }
            timeit("k_cost_march4 seeded rows=0", [&]() { launch_cost_march4(st, iL, iR, A, d, o2, W, H, 0.55f, 1, 0, rb, smap); });
        }
    } else if (mode == 15) {
        for (int P : {5, 2, 0})
            for (int box = 0; box < 2; box++) {
                if (P == 0 && !box) continue;
                launch_smooth_fused(st, d, o, W, H, P, box, 36);
                for (int rows : {35, 33, 29, 24, 18, 16}) {
                    CK(hipMemset(o2, 0xee, 12 * n));
                    launch_smooth_fused(st, d, o2, W, H, P, box, rows);
                    char nm[96]; snprintf(nm, sizeof nm, "k_smooth_fused P=%d box=%d rows=%d vs rows=36", P, box, rows);
                    cmp(nm);
                }
            }
        printf("tile height picked: a call alone %d rows, a call that shares the chip %d rows\n", smooth_tile_rows(W, H, 1), smooth_tile_rows(W, H, 0));
        for (int round = 0; round < 2; round++)
            for (int rows = 36; rows >= 16; rows--) {
                const int tiles = ((W + 111) / 112) * ((H + rows - 1) / rows);
                char nm[64]; snprintf(nm, sizeof nm, "smooth p5+box rows=%d (%d tiles)", rows, tiles);
                timeit(nm, [&]() { launch_smooth_fused(st, d, o2, W, H, 5, 1, rows); });
            }
    } else if (mode == 18) {
        int lo = 0, hi = 0;
        (void)hipDeviceGetStreamPriorityRange(&lo, &hi);
        hipStream_t s1, s2; CK(hipStreamCreateWithPriority(&s1, hipStreamNonBlocking, hi)); CK(hipStreamCreateWithPriority(&s2, hipStreamNonBlocking, hi));
        auto kc = [&](hipStream_t q, float *dst) { march(dst, 1, 0, q); };
        auto ks = [&](hipStream_t q, float *dst) { launch_smooth_fused(q, d, dst, W, H, 5, 1, 36); };
        auto k4 = [&](hipStream_t q, float *dst) { launch_cost_march4(q, iL, iR, A, d, dst, W, H, 1.0f, 1, 0, rb, none); };
        auto wall = [&](auto fa, auto fb, bool both) {
            for (int i = 0; i < 2; i++) { fa(s1, o); if (both) fb(s2, o2); }
            CK(hipDeviceSynchronize());
            auto t0 = std::chrono::steady_clock::now();
            for (int i = 0; i < reps; i++) { fa(s1, o); if (both) fb(s2, o2); }
            CK(hipDeviceSynchronize());
            return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / reps;
        };
        for (int round = 0; round < 2; round++) {
            const double c = wall(kc, kc, false), sm = wall(ks, ks, false), m4 = wall(k4, k4, false);
            printf("alone (us per launch): K-cost march %.1f   K-smooth p5+box %.1f   K-cost march4 %.1f\n", c, sm, m4);
            const double cs = wall(kc, ks, true), cc = wall(kc, kc, true), ss = wall(ks, ks, true), s4 = wall(k4, ks, true);
            printf("side by side on two streams (us per pair of launches; the sum of the two alone in brackets):\n");
            printf("  K-cost march  + K-smooth  %.1f (%.1f)  -> %.2f of the sum\n", cs, c + sm, cs / (c + sm));
            printf("  K-cost march4 + K-smooth  %.1f (%.1f)  -> %.2f of the sum\n", s4, m4 + sm, s4 / (m4 + sm));
            printf("  K-cost march  + K-cost    %.1f (%.1f)  -> %.2f\n", cc, 2 * c, cc / (2 * c));
            printf("  K-smooth      + K-smooth  %.1f (%.1f)  -> %.2f\n", ss, 2 * sm, ss / (2 * sm));
        }
    } else {
        timeit("k_cost_march", [&]() { march(o); });
        timeit("k_smooth_fused p5", [&]() { launch_smooth_fused(st, d, o, W, H, 5, 0); });
        timeit("k_smooth_fused p5+box", [&]() { launch_smooth_fused(st, d, o, W, H, 5, 1); });
    }
    CK(hipGetLastError());
    return 0;
}
