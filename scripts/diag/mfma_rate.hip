// Diagnostic: fp32 MFMA issue rate on gfx950, one and two waves per SIMD, 32x32x2 vs 16x16x4.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
template <int MODE>
__global__ __launch_bounds__(256) void k(float* out, int iters, float seed) {
    const float a = seed + threadIdx.x, b = seed * 0.5f;
    if (MODE == 0) {
        f32x16 acc[6];
        for (int i = 0; i < 6; ++i) for (int e = 0; e < 16; ++e) acc[i][e] = 0.f;
        for (int it = 0; it < iters; ++it)
#pragma unroll
            for (int i = 0; i < 6; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i], 0, 0, 0);
        float s = 0.f;
        for (int i = 0; i < 6; ++i) for (int e = 0; e < 16; ++e) s += acc[i][e];
        if (s == 1234.5f) out[0] = s;
    } else {
        f32x4 acc[6];
        for (int i = 0; i < 6; ++i) for (int e = 0; e < 4; ++e) acc[i][e] = 0.f;
        for (int it = 0; it < iters; ++it)
#pragma unroll
            for (int i = 0; i < 6; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[i], 0, 0, 0);
        float s = 0.f;
        for (int i = 0; i < 6; ++i) for (int e = 0; e < 4; ++e) s += acc[i][e];
        if (s == 1234.5f) out[0] = s;
    }
}
template <int MODE> int run(const char* name, float* out, double flop_per) {
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int iters = 2048;
    for (int bpc = 1; bpc <= 2; ++bpc) for (int cus : {64, 140, 256}) {
        float best = 1e9f;
        for (int rep = 0; rep < 5; ++rep) {
            CK(hipEventRecord(e0));
            hipLaunchKernelGGL(k<MODE>, dim3(cus * bpc), dim3(256), 0, 0, out, iters, 1.f);
            CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (ms < best) best = ms;
        }
        const double n = (double)bpc * iters * 6;   // MFMAs per SIMD (if blocks spread one per CU)
        printf("%-10s blocks %3d (x%d): %.1f us -> %.1f ns per MFMA per SIMD = %.0f cyc @2.4GHz; %.1f TFLOP/s\n", name, cus * bpc, bpc,
               best * 1e3, best * 1e6 / n, best * 1e6 / n * 2.4, flop_per * n * 4 * cus / (best * 1e-3) / 1e12);
    }
    return 0;
}
int main() {
    float* out; CK(hipMalloc(&out, 4));
    run<0>("32x32x2", out, 4096.0);
    run<1>("16x16x4", out, 2048.0);
    return 0;
}
