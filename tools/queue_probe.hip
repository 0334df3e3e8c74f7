// queue_probe -- which HIP streams of a process share a hardware queue (development tool, not product).
//   hipcc -O2 --offload-arch=gfx950 tools/queue_probe.hip -o tools/queue_probe
//   ./queue_probe [n_streams] [use_null_stream_first] [priority pattern, e.g. hhhhnnnn: h = greatest priority, n = default, l = least]
// Two kernels on streams that share a hardware queue run one after the other (AQL packets of an in-order stream carry the barrier
// bit, and the bit orders the whole queue); on different queues they overlap.  The probe launches a one-wave kernel that spins for
// a fixed time on every pair of streams and prints the matrix of "serialised" pairs, then the groups.
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(1); } } while (0)

__global__ void spin(long long ticks, int *sink)
{
    const long long t0 = wall_clock64();
    int n = 0;
    while (wall_clock64() - t0 < ticks && n < (1 << 26)) n++;  // (bounded: the kernel ends even if the clock does not move)
    if (sink && n == -1) *sink = n;
}

int main(int argc, char **argv)
{
    const int n = argc > 1 ? atoi(argv[1]) : 8;
    const int use_null = argc > 2 ? atoi(argv[2]) : 1;
    int rate_khz = 100000;
    CK(hipDeviceGetAttribute(&rate_khz, hipDeviceAttributeWallClockRate, 0));
    const long long ticks = (long long)rate_khz * 300 / 1000;  // 300 us
    if (use_null) { void *p; CK(hipMalloc(&p, 1 << 20)); CK(hipMemset(p, 0, 1 << 20)); CK(hipDeviceSynchronize()); }
    std::vector<hipStream_t> st(n);
    const char *pat = argc > 3 ? argv[3] : "";
    int least = 0, greatest = 0;
    CK(hipDeviceGetStreamPriorityRange(&least, &greatest));
    printf("stream priorities: least %d, greatest %d; pattern '%s'\n", least, greatest, pat);
    for (int i = 0; i < n; i++) {
        const char c = i < (int)strlen(pat) ? pat[i] : 'n';
        if (c == 'n') CK(hipStreamCreateWithFlags(&st[i], hipStreamNonBlocking));
        else CK(hipStreamCreateWithPriority(&st[i], hipStreamNonBlocking, c == 'h' ? greatest : least));
    }
    for (auto &s : st) { hipLaunchKernelGGL(spin, dim3(1), dim3(64), 0, s, 1000, nullptr); }
    CK(hipDeviceSynchronize());
    auto both = [&](int i, int j) {
        auto t0 = std::chrono::steady_clock::now();
        hipLaunchKernelGGL(spin, dim3(1), dim3(64), 0, st[i], ticks, nullptr);
        if (j >= 0) hipLaunchKernelGGL(spin, dim3(1), dim3(64), 0, st[j], ticks, nullptr);
        CK(hipStreamSynchronize(st[i]));
        if (j >= 0) CK(hipStreamSynchronize(st[j]));
        return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
    };
    const double one = both(0, -1);
    printf("%d streams (null stream used first: %d), one spin kernel: %.0f us\n", n, use_null, one);
    std::vector<int> group(n, -1);
    int ng = 0;
    for (int i = 0; i < n; i++) {
        printf("stream %d:", i);
        for (int j = 0; j < n; j++) {
            if (i == j) { printf("  ."); continue; }
            const bool serial = both(i, j) > 1.6 * one;
            printf("  %c", serial ? 'S' : '-');
            if (serial && j < i && group[i] < 0) group[i] = group[j];
        }
        if (group[i] < 0) group[i] = ng++;
        printf("   -> queue group %d\n", group[i]);
    }
    // all at once: the makespan tells how many run concurrently
    auto t0 = std::chrono::steady_clock::now();
    for (auto &s : st) hipLaunchKernelGGL(spin, dim3(1), dim3(64), 0, s, ticks, nullptr);
    CK(hipDeviceSynchronize());
    printf("all %d at once: %.0f us (= %.1f kernels deep)\n", n, std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count(),
           std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / one);
    return 0;
}
