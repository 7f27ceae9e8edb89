// Diagnostic: what a plain read-only stream reaches on this box (the ceiling for the pooling kernel's Fs stream):
// grid-stride 16-byte loads, U loads in flight per lane, several grid sizes, 1 GiB and 344 MB buffers.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
typedef float f32x4 __attribute__((ext_vector_type(4)));
template <int U, bool NT>
__global__ __launch_bounds__(256) void rd(const f32x4* __restrict__ p, size_t n, float* out) {
    float acc = 0.f;
    const size_t stride = (size_t)gridDim.x * 256;
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    for (; i + (U - 1) * stride < n; i += U * stride) {
        f32x4 v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) v[u] = NT ? __builtin_nontemporal_load(p + i + u * stride) : p[i + u * stride];
#pragma unroll
        for (int u = 0; u < U; ++u) acc += v[u][0] + v[u][1] + v[u][2] + v[u][3];
    }
    for (; i < n; i += stride) { const f32x4 v = p[i]; acc += v[0] + v[1] + v[2] + v[3]; }
    if (acc == 1234.5f) out[0] = acc;
}
// contiguous chunk per block (like one block per 107 KB span): block b reads [b*chunk, (b+1)*chunk)
template <int U>
__global__ __launch_bounds__(256) void rd_chunk(const f32x4* __restrict__ p, size_t chunk, float* out) {
    float acc = 0.f;
    const f32x4* q = p + (size_t)blockIdx.x * chunk;
    size_t i = threadIdx.x;
    for (; i + (U - 1) * 256 < chunk; i += U * 256) {
        f32x4 v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) v[u] = q[i + u * 256];
#pragma unroll
        for (int u = 0; u < U; ++u) acc += v[u][0] + v[u][1] + v[u][2] + v[u][3];
    }
    for (; i < chunk; i += 256) { const f32x4 v = q[i]; acc += v[0] + v[1] + v[2] + v[3]; }
    if (acc == 1234.5f) out[0] = acc;
}
template <typename F> float timeit(F f) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    f(); f();
    float best = 1e9f;
    for (int r = 0; r < 5; ++r) { hipEventRecord(e0); f(); hipEventRecord(e1); hipEventSynchronize(e1); float ms; hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms; }
    return best;
}
int main() {
    float* out; CK(hipMalloc(&out, 4));
    for (size_t mb : {344, 1024}) {
        const size_t bytes = mb << 20, n = bytes / 16;
        f32x4 *a, *b; CK(hipMalloc(&a, bytes)); CK(hipMalloc(&b, bytes)); CK(hipMemset(a, 0, bytes)); CK(hipMemset(b, 1, bytes));
        printf("%zu MB (two buffers read alternately, so nothing is served by the 256 MB cache):\n", mb);
        for (int bpc : {2, 4, 8, 16}) {
            int flip = 0;
            const float t4 = timeit([&] { hipLaunchKernelGGL((rd<4, false>), dim3(256 * bpc), dim3(256), 0, 0, (flip++ & 1) ? a : b, n, out); });
            const float t8 = timeit([&] { hipLaunchKernelGGL((rd<8, false>), dim3(256 * bpc), dim3(256), 0, 0, (flip++ & 1) ? a : b, n, out); });
            const float t8n = timeit([&] { hipLaunchKernelGGL((rd<8, true>), dim3(256 * bpc), dim3(256), 0, 0, (flip++ & 1) ? a : b, n, out); });
            printf("  grid-stride, %2d blocks/CU: U=4 %.2f TB/s, U=8 %.2f TB/s, U=8 non-temporal %.2f TB/s\n", bpc, bytes / (t4 * 1e-3) / 1e12,
                   bytes / (t8 * 1e-3) / 1e12, bytes / (t8n * 1e-3) / 1e12);
        }
        for (size_t kb : {107, 428}) {
            const size_t chunk = kb * 1024 / 16, nb = n / chunk;
            int flip = 0;
            const float t = timeit([&] { hipLaunchKernelGGL((rd_chunk<8>), dim3((unsigned)nb), dim3(256), 0, 0, (flip++ & 1) ? a : b, chunk, out); });
            printf("  one block per %zu KB span (%zu blocks): %.2f TB/s\n", kb, nb, nb * chunk * 16 / (t * 1e-3) / 1e12);
        }
        CK(hipFree(a)); CK(hipFree(b));
    }
    return 0;
}
