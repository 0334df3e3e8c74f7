// ldsbench -- which (lane mapping, row stride) makes ds_read_b128 conflict-free on gfx950?  (development tool)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)
__global__ __launch_bounds__(256) void k(float *out, int S, int QW, int colmajor, int iters)
{
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int tid = threadIdx.x;
    for (int i = tid; i < 16000; i += 256) lds[i] = (float)i;
    __syncthreads();
    int row, q;
    if (colmajor) { row = tid & 31; q = tid >> 5; }
    else { row = tid / QW; q = tid % QW; }
    const float4 *p = reinterpret_cast<const float4 *>(lds + row * S + q * 4);
    float acc = 0.f;
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int u = 0; u < 8; u++) {
            float4 v = p[((it * 5 + u * 3) & 31)];  // uniform shift per access: same bank pattern, new address
            asm volatile("" : "+v"(v.x), "+v"(v.y), "+v"(v.z), "+v"(v.w));
            acc += v.x;
        }
    }
    out[blockIdx.x * 256 + tid] = acc;
}
int main()
{
    float *o; CK(hipMalloc(&o, 4 * 256 * 4096));
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    CK(hipFuncSetAttribute(reinterpret_cast<const void *>(&k), hipFuncAttributeMaxDynamicSharedMemorySize, 65536));
    const int iters = 4000;
    for (int cm = 0; cm < 2; cm++)
        for (int QW : {8, 20, 16}) {
            if (cm && QW != 8) continue;
            for (int S = 32; S <= 100; S += 4) {
                if (!cm && S < QW * 4) continue;
                hipLaunchKernelGGL(k, dim3(1024), dim3(256), 64000, 0, o, S, QW, cm, 10);
                CK(hipEventRecord(a));
                hipLaunchKernelGGL(k, dim3(1024), dim3(256), 64000, 0, o, S, QW, cm, iters);
                CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
                float ms; CK(hipEventElapsedTime(&ms, a, b));
                printf("%s QW=%2d S=%3d (%2d quads): %7.3f ms\n", cm ? "col-major" : "row-major", QW, S, S / 4, ms);
            }
        }
    return 0;
}
