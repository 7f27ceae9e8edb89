// Diagnostic: VALU issue rate on gfx950 with 1..4 waves per SIMD: plain v_fma_f32 vs v_pk_fma_f32 vs v_exp_f32.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x2 __attribute__((ext_vector_type(2)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
template <int MODE>
__global__ __launch_bounds__(256) void rate_kernel(float* out, int iters, float seed) {
    float a[8];
    f32x2 p[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) { a[i] = seed + i + threadIdx.x; p[i] = {a[i], a[i] + 1.f}; }
    const float m = 1.0000001f;
    const f32x2 m2 = {m, m};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            if (MODE == 0) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(a[i]) : "v"(m));
            if (MODE == 1) asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(p[i]) : "v"(m2));
            if (MODE == 2) asm volatile("v_exp_f32 %0, %0" : "+v"(a[i]));
            if (MODE == 3) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(p[i]) : "v"(m2));
            if (MODE == 4) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(a[i]) : "v"(m));
        }
    }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i) s += a[i] + p[i][0] + p[i][1];
    if (s == 12345.678f) out[0] = s;
}
template <int MODE>
int run(const char* name, float* out) {
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int iters = 4096;
    for (int bpc = 1; bpc <= 8; bpc *= 2) {
        float best = 1e9f;
        for (int rep = 0; rep < 5; ++rep) {
            CK(hipEventRecord(e0));
            hipLaunchKernelGGL(rate_kernel<MODE>, dim3(256 * bpc), dim3(256), 0, 0, out, iters, 1.f);
            CK(hipEventRecord(e1));
            CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            if (ms < best) best = ms;
        }
        // per SIMD: bpc waves, each iters*8 instructions
        const double instr_per_simd = (double)bpc * iters * 8;
        printf("%-14s waves/SIMD %d: %.1f us -> %.2f ns per wave-instruction per SIMD (%.2f cyc at 2.4 GHz)\n", name, bpc,
               best * 1e3, best * 1e6 / instr_per_simd, best * 1e6 / instr_per_simd * 2.4);
    }
    return 0;
}
int main() {
    float* out; CK(hipMalloc(&out, 4));
    run<0>("v_fma_f32", out);
    run<4>("v_mul_f32", out);
    run<1>("v_pk_fma_f32", out);
    run<3>("v_pk_mul_f32", out);
    run<2>("v_exp_f32", out);
    return 0;
}
