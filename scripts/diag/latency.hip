// Calibration micro-benchmarks: dependent-load latency, clock, f32 MFMA rate, LDS atomic cost.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <numeric>
#include <algorithm>
#include <random>
typedef float f32x16 __attribute__((ext_vector_type(16)));

__global__ void chase(const unsigned* __restrict__ next, int steps, unsigned* out, unsigned long long* cyc, unsigned long long* rt) {
    unsigned i = threadIdx.x + blockIdx.x * 977u;
    unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int s = 0; s < steps; ++s) i = next[i];
    unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    out[blockIdx.x * blockDim.x + threadIdx.x] = i;
    if (threadIdx.x == 0) { cyc[blockIdx.x] = t1 - t0; rt[blockIdx.x] = r1 - r0; }
}

__global__ void mfma_loop(int iters, float* out, unsigned long long* cyc, unsigned long long* rt) {
    f32x16 a0 = {0}, a1 = {0};
    float x = threadIdx.x * 0.001f, y = 1.0f;
    unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int i = 0; i < iters; ++i) {
        a0 = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, a0, 0, 0, 0);
        a1 = __builtin_amdgcn_mfma_f32_32x32x2f32(y, x, a1, 0, 0, 0);
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0[0] + a1[3];
    if (threadIdx.x == 0) { cyc[blockIdx.x] = t1 - t0; rt[blockIdx.x] = r1 - r0; }
}

__global__ void lds_atomic_loop(int iters, float* out, unsigned long long* cyc) {
    __shared__ float acc[4096];
    for (int i = threadIdx.x; i < 4096; i += blockDim.x) acc[i] = 0.f;
    __syncthreads();
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < iters; ++i) atomicAdd(&acc[(threadIdx.x + i * 64) & 4095], 1.0f);
    __syncthreads();
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc[threadIdx.x];
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

__global__ void valu_exp_loop(int iters, float* out, unsigned long long* cyc) {
    float v = threadIdx.x * 1e-3f, s = 0.f;
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < iters; ++i) { s += __expf(v); v += 1e-6f; }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

__global__ void empty_kernel() {}

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

int main() {
    const size_t n = 64u << 20;   // 256 MiB of indices: beyond L2, around MALL size
    std::vector<unsigned> h(n);
    std::iota(h.begin(), h.end(), 0u);
    std::mt19937 rng(1);
    // random cyclic permutation (Sattolo)
    for (size_t i = n - 1; i > 0; --i) { size_t j = rng() % i; std::swap(h[i], h[j]); }
    unsigned *d_next, *d_out; unsigned long long *d_cyc, *d_rt; float* d_f;
    CK(hipMalloc(&d_next, n * 4)); CK(hipMalloc(&d_out, 1 << 22)); CK(hipMalloc(&d_cyc, 4096 * 8)); CK(hipMalloc(&d_rt, 4096 * 8));
    CK(hipMalloc(&d_f, 1 << 24));
    CK(hipMemcpy(d_next, h.data(), n * 4, hipMemcpyHostToDevice));
    unsigned long long cyc[4096], rt[4096];
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    float ms;
    // 1. dependent load latency: 1 wave, 1 lane effectively (64 lanes random)
    for (int rep = 0; rep < 2; ++rep) {
        const int steps = 200;
        CK(hipEventRecord(e0)); chase<<<1, 64>>>(d_next, steps, d_out, d_cyc, d_rt); CK(hipEventRecord(e1)); CK(hipDeviceSynchronize());
        CK(hipEventElapsedTime(&ms, e0, e1));
        CK(hipMemcpy(cyc, d_cyc, 8, hipMemcpyDeviceToHost)); CK(hipMemcpy(rt, d_rt, 8, hipMemcpyDeviceToHost));
        printf("chase 256MiB footprint (64 random lanes): %.1f ns/step  (%.0f cycles/step, clock %.2f GHz) kernel %.1f us\n",
               rt[0] * 10.0 / steps, (double)cyc[0] / steps, (double)cyc[0] / (rt[0] * 10.0), ms * 1e3);
    }
    // small footprint: L2-resident (first 1 MiB as its own cycle)
    {
        const size_t m = 256 << 10; std::vector<unsigned> s(m); std::iota(s.begin(), s.end(), 0u);
        for (size_t i = m - 1; i > 0; --i) { size_t j = rng() % i; std::swap(s[i], s[j]); }
        CK(hipMemcpy(d_next, s.data(), m * 4, hipMemcpyHostToDevice));
        for (int rep = 0; rep < 3; ++rep) {
            const int steps = 2000;
            chase<<<1, 64>>>(d_next, steps, d_out, d_cyc, d_rt); CK(hipDeviceSynchronize());
            CK(hipMemcpy(cyc, d_cyc, 8, hipMemcpyDeviceToHost)); CK(hipMemcpy(rt, d_rt, 8, hipMemcpyDeviceToHost));
            printf("chase 1MiB footprint: %.1f ns/step (%.0f cycles/step, clock %.2f GHz)\n", rt[0] * 10.0 / steps,
                   (double)cyc[0] / steps, (double)cyc[0] / (rt[0] * 10.0));
        }
    }
    // 2. MFMA f32 rate, 1 wave/SIMD on all CUs
    for (int rep = 0; rep < 3; ++rep) {
        const int iters = 2000;
        CK(hipEventRecord(e0)); mfma_loop<<<256, 256>>>(iters, d_f, d_cyc, d_rt); CK(hipEventRecord(e1)); CK(hipDeviceSynchronize());
        CK(hipEventElapsedTime(&ms, e0, e1));
        CK(hipMemcpy(cyc, d_cyc, 8, hipMemcpyDeviceToHost)); CK(hipMemcpy(rt, d_rt, 8, hipMemcpyDeviceToHost));
        printf("mfma_f32_32x32x2: %.1f cycles/mfma, clock %.2f GHz, kernel %.1f us\n", (double)cyc[0] / (2.0 * iters),
               (double)cyc[0] / (rt[0] * 10.0), ms * 1e3);
    }
    // 3. LDS float atomics
    for (int rep = 0; rep < 2; ++rep) {
        const int iters = 1000;
        lds_atomic_loop<<<1, 64>>>(iters, d_f, d_cyc); CK(hipDeviceSynchronize());
        CK(hipMemcpy(cyc, d_cyc, 8, hipMemcpyDeviceToHost));
        printf("LDS atomicAdd f32 (1 wave, distinct addresses): %.1f cycles per wave-instruction\n", (double)cyc[0] / iters);
    }
    // 4. exp throughput, 1 wave
    {
        const int iters = 4000;
        valu_exp_loop<<<1, 64>>>(iters, d_f, d_cyc); CK(hipDeviceSynchronize());
        CK(hipMemcpy(cyc, d_cyc, 8, hipMemcpyDeviceToHost));
        printf("__expf + add loop (1 wave): %.1f cycles/iter\n", (double)cyc[0] / iters);
    }
    // 5. empty kernel launch spacing
    {
        for (int i = 0; i < 10; ++i) empty_kernel<<<1, 64>>>();
        CK(hipDeviceSynchronize());
        CK(hipEventRecord(e0));
        for (int i = 0; i < 100; ++i) empty_kernel<<<256, 256>>>();
        CK(hipEventRecord(e1)); CK(hipDeviceSynchronize());
        CK(hipEventElapsedTime(&ms, e0, e1));
        printf("100 back-to-back empty kernels: %.2f us each\n", ms * 10.0);
    }
    return 0;
}
