// Micro-benchmark: event-to-event time of a trivial kernel vs grid size and block size.
#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
__global__ void tiny(float* out, int lds_touch) {
    extern __shared__ float sm[];
    if (lds_touch) sm[threadIdx.x] = 1.f;
    if (threadIdx.x == 0 && blockIdx.x == 0) out[0] = 1.f;
}
int main() {
    float* out; CK(hipMalloc(&out, 1024));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int threads : {256, 512}) for (size_t lds : {(size_t)0, (size_t)25 * 1024, (size_t)82 * 1024})
    for (int blocks : {8, 64, 216, 640, 1312, 4096, 16384}) {
        CK(hipFuncSetAttribute((const void*)tiny, hipFuncAttributeMaxDynamicSharedMemorySize, 100 * 1024));
        float best = 1e9, sum = 0;
        for (int rep = 0; rep < 12; ++rep) {
            CK(hipDeviceSynchronize());
            CK(hipEventRecord(e0));
            hipLaunchKernelGGL(tiny, dim3(blocks), dim3(threads), lds, 0, out, lds ? 1 : 0);
            CK(hipEventRecord(e1));
            CK(hipDeviceSynchronize());
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            if (rep >= 2) { best = ms < best ? ms : best; sum += ms; }
        }
        printf("threads=%d lds=%zuKB blocks=%5d : min %.1f us mean %.1f us\n", threads, lds / 1024, blocks, best * 1e3, sum * 100);
    }
    // back-to-back launches (no sync between): per-launch cost in a stream
    for (int blocks : {8, 216, 1312}) {
        CK(hipDeviceSynchronize());
        CK(hipEventRecord(e0));
        for (int i = 0; i < 100; ++i) hipLaunchKernelGGL(tiny, dim3(blocks), dim3(256), 0, 0, out, 0);
        CK(hipEventRecord(e1));
        CK(hipDeviceSynchronize());
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        printf("100 back-to-back launches, blocks=%d: %.2f us per launch\n", blocks, ms * 10);
    }
    return 0;
}
