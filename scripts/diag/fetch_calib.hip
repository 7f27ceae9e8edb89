// Calibration of rocprofv3's FETCH_SIZE on gfx950 by load width: streams a buffer far beyond the 256 MiB
// Infinity Cache with 4-, 8- and 16-byte-per-lane loads (known byte count), so that FETCH_SIZE * 1024 / bytes gives
// the factor to apply to kernels that stream with that width.  Run under `rocprofv3 --pmc FETCH_SIZE`.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
template <typename V>
__global__ __launch_bounds__(256) void stream_read(const V* __restrict__ p, size_t n, float* out) {
    float s = 0.f;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        const V v = p[i];
        if constexpr (sizeof(V) == 4) s += v;
        else if constexpr (sizeof(V) == 8) s += v[0] + v[1];
        else s += v[0] + v[1] + v[2] + v[3];
    }
    if (s == 1234.5f) out[0] = s;
}
int main() {
    const size_t bytes = (size_t)1 << 30;
    float* buf; float* out;
    CK(hipMalloc(&buf, bytes)); CK(hipMalloc(&out, 4)); CK(hipMemset(buf, 0, bytes));
    for (int rep = 0; rep < 3; ++rep) {
        hipLaunchKernelGGL(stream_read<float>, dim3(4096), dim3(256), 0, 0, buf, bytes / 4, out);
        hipLaunchKernelGGL(stream_read<f32x2>, dim3(4096), dim3(256), 0, 0, (const f32x2*)buf, bytes / 8, out);
        hipLaunchKernelGGL(stream_read<f32x4>, dim3(4096), dim3(256), 0, 0, (const f32x4*)buf, bytes / 16, out);
    }
    CK(hipDeviceSynchronize());
    printf("streamed %zu bytes per launch with 4/8/16-byte loads\n", bytes);
    return 0;
}
