// gridbar -- what a grid-wide barrier inside one persistent kernel costs on this chip against a chain of dependent launches:
// the question behind a persistent kernel for the coarse pyramid levels (a coarse level is 44 dependent launches of 2-6 us of work each).
// Every phase each workgroup writes a line of data, passes the barrier and checks the line another workgroup (another XCD) wrote in that
// phase, so the barrier is measured with the visibility it has to provide.  Spins are bounded: a barrier that does not complete sets an
// error flag and every wave leaves.
// build: hipcc -O3 --offload-arch=gfx950 tools/gridbar.hip -o tools/gridbar ; run: tools/gridbar [phases]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

struct Bar {
    unsigned *count;  // arrivals, monotonically increasing
    unsigned *error;
};

// One counter, agent scope.  Thread 0 of each workgroup arrives (release: the workgroup's stores are written back beyond its XCD's L2)
// and polls (acquire: stale lines of the L1 / L2 are dropped); the workgroup barriers on either side extend both to every wave.
__device__ __forceinline__ bool grid_barrier(const Bar b, const unsigned target)
{
    __syncthreads();
    bool ok = true;
    if (threadIdx.x == 0) {
        __hip_atomic_fetch_add(b.count, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
        unsigned spins = 0;
        while (__hip_atomic_load(b.count, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < target) {
            __builtin_amdgcn_s_sleep(1);
            if (++spins > (1u << 22)) {
                __hip_atomic_store(b.error, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                ok = false;
                break;
            }
        }
    }
    __syncthreads();
    return ok;
}

template <int WORK>
__global__ void k_persistent(Bar b, float *data, int phases, unsigned *mismatch)
{
    const int G = gridDim.x, g = blockIdx.x, t = threadIdx.x, T = blockDim.x;
    unsigned bad = 0;
    for (int p = 0; p < phases; ++p) {
        float v = (float)(p * 131 + g);
        for (int i = 0; i < WORK; ++i) v = v * 1.0000001f + 0.0f;  // (a stand-in for the phase's work)
        if (WORK == 0) v = (float)(p * 131 + g);
        data[(size_t)g * T + t] = (float)(p * 131 + g) + (WORK ? 0.0f * v : 0.0f);
        if (!grid_barrier(b, (unsigned)(2 * p + 1) * G)) return;
        const int o = (g + 3) % G;  // (b and b + 3 sit on different XCDs)
        if (data[(size_t)o * T + t] != (float)(p * 131 + o)) ++bad;
        if (!grid_barrier(b, (unsigned)(2 * p + 2) * G)) return;  // (nobody overwrites a line before its reader is done)
    }
    if (bad) atomicAdd(mismatch, bad);
}

__global__ void k_phase_write(float *data, int p)
{
    data[(size_t)blockIdx.x * blockDim.x + threadIdx.x] = (float)(p * 131 + blockIdx.x);
}
__global__ void k_phase_check(const float *data, int p, unsigned *mismatch)
{
    const int o = (blockIdx.x + 3) % gridDim.x;
    if (data[(size_t)o * blockDim.x + threadIdx.x] != (float)(p * 131 + o)) atomicAdd(mismatch, 1u);
}

int main(int argc, char **argv)
{
    const int phases = argc > 1 ? atoi(argv[1]) : 500;
    hipStream_t st;
    CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    unsigned *d_ctl;
    CK(hipMalloc(&d_ctl, 256));
    float *d_data;
    CK(hipMalloc(&d_data, (size_t)2048 * 1024 * 4));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    printf("workgroups x threads : persistent kernel, us per barrier (2 per phase)  |  chain of dependent launches, us per launch   [mismatches]\n");
    for (int T : {256, 1024})
        for (int G : {8, 32, 64, 128, 256, 512}) {
            if (T == 1024 && G > 256) continue;  // (every workgroup must be resident)
            Bar b{d_ctl, d_ctl + 1};
            float best_p = 1e30f, best_c = 1e30f;
            unsigned mm[4] = {0, 0, 0, 0};
            for (int rep = 0; rep < 3; ++rep) {
                CK(hipMemsetAsync(d_ctl, 0, 256, st));
                CK(hipEventRecord(e0, st));
                hipLaunchKernelGGL(k_persistent<0>, dim3(G), dim3(T), 0, st, b, d_data, phases, d_ctl + 2);
                CK(hipEventRecord(e1, st));
                CK(hipStreamSynchronize(st));
                float ms;
                CK(hipEventElapsedTime(&ms, e0, e1));
                if (ms < best_p) best_p = ms;
                unsigned h[4];
                CK(hipMemcpy(h, d_ctl, 16, hipMemcpyDeviceToHost));
                if (h[1]) { printf("  barrier timed out (G=%d T=%d)\n", G, T); return 2; }
                mm[0] += h[2];
                CK(hipMemsetAsync(d_ctl, 0, 256, st));
                CK(hipEventRecord(e0, st));
                for (int p = 0; p < phases; ++p) {
                    hipLaunchKernelGGL(k_phase_write, dim3(G), dim3(T), 0, st, d_data, p);
                    hipLaunchKernelGGL(k_phase_check, dim3(G), dim3(T), 0, st, d_data, p, d_ctl + 2);
                }
                CK(hipEventRecord(e1, st));
                CK(hipStreamSynchronize(st));
                CK(hipEventElapsedTime(&ms, e0, e1));
                if (ms < best_c) best_c = ms;
                CK(hipMemcpy(h, d_ctl, 16, hipMemcpyDeviceToHost));
                mm[1] += h[2];
            }
            printf("%4d x %4d : %7.2f  |  %7.2f   [%u %u]\n", G, T, best_p * 1e3f / (2 * phases), best_c * 1e3f / (2 * phases), mm[0], mm[1]);
            fflush(stdout);
        }
    return 0;
}
