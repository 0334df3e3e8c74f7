// pcie_probe -- how fast do the three result planes of a 16 MP pair (3 x 64.3 MB) reach page-locked host memory?
//   one hipMemcpyAsync of all three; three hipMemcpyAsync on one stream; three on three streams; a kernel that stores straight into the
//   (device-mapped) host buffer; the same for the way up (two 48 MB images).  Development probe (profiles/r06_pcie_probe.txt); not product.
//   hipcc --offload-arch=gfx950 -O3 tools/pcie_probe.hip -o tools/pcie_probe
#include <hip/hip_runtime.h>

#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHK(x)                                                                              \
    do {                                                                                    \
        hipError_t e__ = (x);                                                               \
        if (e__ != hipSuccess) {                                                            \
            fprintf(stderr, "%s: %s (line %d)\n", #x, hipGetErrorString(e__), __LINE__);     \
            exit(1);                                                                        \
        }                                                                                   \
    } while (0)

__global__ void k_copy16(const float4 *__restrict__ src, float4 *__restrict__ dst, size_t n16)
{
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n16; i += (size_t)gridDim.x * blockDim.x) dst[i] = src[i];
}

static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

template <class F>
static double time_ms(F &&f, int reps = 7)
{
    std::vector<double> t;
    for (int r = 0; r < reps + 1; r++) {
        CHK(hipDeviceSynchronize());
        const double t0 = now();
        f();
        CHK(hipDeviceSynchronize());
        if (r) t.push_back((now() - t0) * 1e3);
    }
    std::sort(t.begin(), t.end());
    return t[t.size() / 2];
}

int main()
{
    const size_t plane = (size_t)4928 * 3264 * sizeof(float), down = 3 * plane, img = (size_t)4928 * 3264 * 3, up = 2 * img;
    char *d = nullptr, *d2 = nullptr, *h = nullptr, *h2 = nullptr, *hd = nullptr;
    CHK(hipMalloc((void **)&d, down));
    CHK(hipMalloc((void **)&d2, up));
    CHK(hipHostMalloc((void **)&h2, up, hipHostMallocDefault));
    CHK(hipHostMalloc((void **)&h, down, hipHostMallocDefault));
    CHK(hipHostGetDevicePointer((void **)&hd, h, 0));
    CHK(hipMemset(d, 1, down));
    hipStream_t st[3];
    for (auto &s : st) CHK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    auto report = [&](const char *what, size_t bytes, double ms) { printf("%-78s %7.3f ms  %6.1f GB/s\n", what, ms, bytes / ms * 1e-6); };

    report("down, 193 MB: one hipMemcpyAsync", down, time_ms([&] { CHK(hipMemcpyAsync(h, d, down, hipMemcpyDeviceToHost, st[0])); }));
    report("down: three hipMemcpyAsync (one per plane) on one stream", down, time_ms([&] {
               for (int k = 0; k < 3; k++) CHK(hipMemcpyAsync(h + k * plane, d + k * plane, plane, hipMemcpyDeviceToHost, st[0]));
           }));
    report("down: three hipMemcpyAsync on three streams", down, time_ms([&] {
               for (int k = 0; k < 3; k++) CHK(hipMemcpyAsync(h + k * plane, d + k * plane, plane, hipMemcpyDeviceToHost, st[k]));
           }));
    report("down: six half-plane copies on three streams", down, time_ms([&] {
               for (int k = 0; k < 6; k++) CHK(hipMemcpyAsync(h + k * plane / 2, d + k * plane / 2, plane / 2, hipMemcpyDeviceToHost, st[k % 3]));
           }));
    for (int blocks : {2, 4, 8, 16, 32, 64, 256, 1024, 4096})
        for (int threads : {256, 1024}) {
            char name[128];
            snprintf(name, sizeof name, "down: a kernel storing into the mapped host buffer, %d x %d threads", blocks, threads);
            report(name, down, time_ms([&] { hipLaunchKernelGGL(k_copy16, dim3(blocks), dim3(threads), 0, st[0], (const float4 *)d, (float4 *)hd, down / 16); }));
        }
    report("down: the kernel on the first two planes + hipMemcpyAsync of the third, two streams", down, time_ms([&] {
               hipLaunchKernelGGL(k_copy16, dim3(1024), dim3(256), 0, st[0], (const float4 *)d, (float4 *)hd, 2 * plane / 16);
               CHK(hipMemcpyAsync(h + 2 * plane, d + 2 * plane, plane, hipMemcpyDeviceToHost, st[1]));
           }));
    report("up, 96 MB: one hipMemcpyAsync", up, time_ms([&] { CHK(hipMemcpyAsync(d, h, up, hipMemcpyHostToDevice, st[0])); }));
    report("up: two hipMemcpyAsync (one per image) on two streams", up, time_ms([&] {
               for (int k = 0; k < 2; k++) CHK(hipMemcpyAsync(d + k * img, h + k * img, img, hipMemcpyHostToDevice, st[k]));
           }));
    report("up: a kernel loading from the mapped host buffer, 1024 x 256 threads", up,
           time_ms([&] { hipLaunchKernelGGL(k_copy16, dim3(1024), dim3(256), 0, st[0], (const float4 *)hd, (float4 *)d, up / 16); }));
    report("both ways at once: 193 MB down on one stream, 96 MB up on another", down + up, time_ms([&] {
               CHK(hipMemcpyAsync(h, d, down, hipMemcpyDeviceToHost, st[0]));
               CHK(hipMemcpyAsync(d2, h2, up, hipMemcpyHostToDevice, st[1]));
           }));
    char *hd2 = nullptr;
    CHK(hipHostGetDevicePointer((void **)&hd2, h2, 0));
    report("both ways at once: hipMemcpyAsync down, a KERNEL loading the images up", down + up, time_ms([&] {
               CHK(hipMemcpyAsync(h, d, down, hipMemcpyDeviceToHost, st[0]));
               hipLaunchKernelGGL(k_copy16, dim3(1024), dim3(256), 0, st[1], (const float4 *)hd2, (float4 *)d2, up / 16);
           }));
    report("both ways at once: a KERNEL storing the planes down, hipMemcpyAsync up", down + up, time_ms([&] {
               hipLaunchKernelGGL(k_copy16, dim3(1024), dim3(256), 0, st[0], (const float4 *)d, (float4 *)hd, down / 16);
               CHK(hipMemcpyAsync(d2, h2, up, hipMemcpyHostToDevice, st[1]));
           }));
    report("both ways at once: kernels both ways", down + up, time_ms([&] {
               hipLaunchKernelGGL(k_copy16, dim3(1024), dim3(256), 0, st[0], (const float4 *)d, (float4 *)hd, down / 16);
               hipLaunchKernelGGL(k_copy16, dim3(1024), dim3(256), 0, st[1], (const float4 *)hd2, (float4 *)d2, up / 16);
           }));
    report("both ways at once, equal sizes: 96 MB down and 96 MB up, hipMemcpyAsync both", 2 * up, time_ms([&] {
               CHK(hipMemcpyAsync(h, d, up, hipMemcpyDeviceToHost, st[0]));
               CHK(hipMemcpyAsync(d2, h2, up, hipMemcpyHostToDevice, st[1]));
           }));
    return 0;
}
