// Micro-benchmark: how long does a block-per-CU burst of strided row-segment loads take?
// Mimics classify_fwd's Fq access: block (tile, n) reads C rows x 64 positions of an [N][C][P] array.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef float f32x2_u __attribute__((ext_vector_type(2), aligned(4)));
typedef float f32x4_u __attribute__((ext_vector_type(4), aligned(4)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

// MODE 0: as classify_fwd (lane -> 2 positions of row c + (lane>>5)), 32 loads per wave, 4 waves split C
// MODE 1: each wave-load = ONE row, 64 lanes x 4 B (256 B), 64 loads per wave
// MODE 2: 16-B loads: 16 lanes per row (64 positions), 4 rows per wave-load, 16 loads per wave
template <int MODE>
__global__ __launch_bounds__(256) void burst(const float* __restrict__ x, int C, int P, float* out, unsigned long long* st) {
    const int n = blockIdx.y, p0 = blockIdx.x * 64;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const float* base = x + (int64_t)n * C * P + p0;
    unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    float acc = 0.f;
    const int cbeg = wave * (C / 4);
    if (MODE == 0) {
        float a[32], b[32];
#pragma unroll
        for (int u = 0; u < 32; ++u) {
            f32x2_u v = *reinterpret_cast<const f32x2_u*>(base + (int64_t)(cbeg + 2 * u + (lane >> 5)) * P + 2 * (lane & 31));
            a[u] = v[0]; b[u] = v[1];
        }
#pragma unroll
        for (int u = 0; u < 32; ++u) acc += a[u] * b[u];
    } else if (MODE == 1) {
        float a[64];
#pragma unroll
        for (int u = 0; u < 64; ++u) a[u] = base[(int64_t)(cbeg + u) * P + lane];
#pragma unroll
        for (int u = 0; u < 64; ++u) acc += a[u];
    } else {
        f32x4_u a[16];
#pragma unroll
        for (int u = 0; u < 16; ++u)
            a[u] = *reinterpret_cast<const f32x4_u*>(base + (int64_t)(cbeg + 4 * u + (lane >> 4)) * P + 4 * (lane & 15));
#pragma unroll
        for (int u = 0; u < 16; ++u) acc += a[u][0] + a[u][1] + a[u][2] + a[u][3];
    }
    unsigned long long t1 = __builtin_amdgcn_s_memrealtime();
    out[(blockIdx.y * gridDim.x + blockIdx.x) * 256 + threadIdx.x] = acc;
    if (threadIdx.x == 0) st[blockIdx.y * gridDim.x + blockIdx.x] = t1 - t0;
}

template <int MODE>
int run(const char* name, const float* x, int N, int C, int P, int tiles, float* out, unsigned long long* st) {
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int rep = 0; rep < 3; ++rep) {
        CK(hipDeviceSynchronize());
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL(burst<MODE>, dim3(tiles, N), dim3(256), 0, 0, x, C, P, out, st);
        CK(hipEventRecord(e1));
        CK(hipDeviceSynchronize());
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        unsigned long long h[512]; CK(hipMemcpy(h, st, sizeof(unsigned long long) * tiles * N, hipMemcpyDeviceToHost));
        unsigned long long mx = 0, sum = 0; for (int i = 0; i < tiles * N; ++i) { mx = h[i] > mx ? h[i] : mx; sum += h[i]; }
        printf("%-28s P=%d rep%d kernel %.1f us; in-block load phase: mean %.2f us max %.2f us\n", name, P, rep, ms * 1e3,
               sum * 0.01 / (tiles * N), mx * 0.01);
    }
    return 0;
}

int main() {
    const int N = 8, C = 256;
    float* x; CK(hipMalloc(&x, (size_t)N * C * 2048 * 4)); CK(hipMemset(x, 0, (size_t)N * C * 2048 * 4));
    float* out; CK(hipMalloc(&out, 512 * 256 * 4));
    unsigned long long* st; CK(hipMalloc(&st, 512 * 8));
    for (int P : {1681, 1664, 2048}) {
        const int tiles = P / 64;      // full tiles only
        if (run<0>("dwordx2, 2 rows/load", x, N, C, P, tiles, out, st)) return 1;
        if (run<1>("dword, 1 row/load", x, N, C, P, tiles, out, st)) return 1;
        if (run<2>("dwordx4, 4 rows/load", x, N, C, P, tiles, out, st)) return 1;
    }
    return 0;
}
