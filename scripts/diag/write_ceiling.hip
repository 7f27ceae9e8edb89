// Diagnostic: what write pattern reaches the box's fill rate?  usage: write_ceiling <MB>
//   span  : block b writes its own contiguous span (16 rows x P floats = 107,584 B at C2/C3), 4 KB per trip
//   stride: persistent blocks, trip t of block b writes chunk t * nblk + b (the chip writes one moving window)
// each with cached and non-temporal 16-byte stores.  build: hipcc -O3 --offload-arch=gfx950 write_ceiling.hip -o write_ceiling
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f32x4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

template <bool NT>
__global__ __launch_bounds__(256) void span_kernel(f32x4* out, size_t span_vec, float v) {
    f32x4* o = out + (size_t)blockIdx.x * span_vec;
    const f32x4 x = {v, v, v, v};
    for (size_t i = threadIdx.x; i < span_vec; i += 256) {
        if (NT) __builtin_nontemporal_store(x, o + i); else o[i] = x;
    }
}
template <bool NT, int CH>     // CH vectors per thread and chunk
__global__ __launch_bounds__(256) void stride_kernel(f32x4* out, size_t nvec, float v) {
    const f32x4 x = {v, v, v, v};
    const size_t chunk = 256 * CH;
    for (size_t c = blockIdx.x; c * chunk < nvec; c += gridDim.x) {
#pragma unroll
        for (int u = 0; u < CH; ++u) {
            const size_t i = c * chunk + u * 256 + threadIdx.x;
            if (i < nvec) { if (NT) __builtin_nontemporal_store(x, out + i); else out[i] = x; }
        }
    }
}
// thread-contiguous: every thread writes V adjacent 16-byte vectors (a wave store instruction = 64 pieces V*16 B apart)
template <bool NT, int V, int T>
__global__ __launch_bounds__(T) void tc_kernel(f32x4* out, size_t nvec, float v) {
    const f32x4 x = {v, v, v, v};
    const size_t chunk = (size_t)T * V;
    for (size_t c = blockIdx.x; c * chunk < nvec; c += gridDim.x) {
        const size_t i0 = c * chunk + (size_t)threadIdx.x * V;
#pragma unroll
        for (int u = 0; u < V; ++u)
            if (i0 + u < nvec) { if (NT) __builtin_nontemporal_store(x, out + i0 + u); else out[i0 + u] = x; }
    }
}
int main(int argc, char** argv) {
    const size_t mb = argc > 1 ? atoi(argv[1]) : 344;
    const size_t span_vec = 16 * 1681 / 4 * 4 / 4 * 1;   // 16 rows x 1681 floats = 26896 floats = 6724 vectors
    const size_t sv = 16 * 1681 / 4;
    size_t nspan = mb * 1000000 / (sv * 16);
    const size_t nvec = nspan * sv;
    f32x4* buf; CK(hipMalloc(&buf, nvec * 16 + 4096));
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    auto time = [&](const char* name, auto launch) {
        for (int i = 0; i < 3; ++i) launch();
        CK(hipEventRecord(a));
        for (int i = 0; i < 20; ++i) launch();
        CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
        float ms; CK(hipEventElapsedTime(&ms, a, b));
        printf("%-28s %7.1f us  %5.2f TB/s\n", name, ms / 20 * 1e3, nvec * 16.0 / (ms / 20 * 1e-3) / 1e12);
    };
    (void)span_vec;
    printf("%zu MB, %zu spans of %zu B\n", nvec * 16 / 1000000, nspan, sv * 16);
    time("hipMemsetD32", [&] { CK(hipMemsetD32Async((hipDeviceptr_t)buf, 0x3f800000, nvec * 4, 0)); });
    time("span cached", [&] { hipLaunchKernelGGL(span_kernel<false>, dim3(nspan), dim3(256), 0, 0, buf, sv, 1.f); });
    time("span nt", [&] { hipLaunchKernelGGL(span_kernel<true>, dim3(nspan), dim3(256), 0, 0, buf, sv, 1.f); });
    for (int nb : {512, 1024, 2048, 4096}) {
        char nm[64];
        snprintf(nm, 64, "stride cached 4KB x%d", nb);
        time(nm, [&] { hipLaunchKernelGGL((stride_kernel<false, 1>), dim3(nb), dim3(256), 0, 0, buf, nvec, 1.f); });
        snprintf(nm, 64, "stride nt 4KB x%d", nb);
        time(nm, [&] { hipLaunchKernelGGL((stride_kernel<true, 1>), dim3(nb), dim3(256), 0, 0, buf, nvec, 1.f); });
        snprintf(nm, 64, "stride cached 16KB x%d", nb);
        time(nm, [&] { hipLaunchKernelGGL((stride_kernel<false, 4>), dim3(nb), dim3(256), 0, 0, buf, nvec, 1.f); });
        snprintf(nm, 64, "stride nt 16KB x%d", nb);
        time(nm, [&] { hipLaunchKernelGGL((stride_kernel<true, 4>), dim3(nb), dim3(256), 0, 0, buf, nvec, 1.f); });
    }
    for (int nb : {256, 512, 1024, 2048, 8192}) {
        char nm[64];
        snprintf(nm, 64, "tc cached 64B/thr x%d", nb);
        time(nm, [&] { hipLaunchKernelGGL((tc_kernel<false, 4, 256>), dim3(nb), dim3(256), 0, 0, buf, nvec, 1.f); });
        snprintf(nm, 64, "tc nt 64B/thr x%d", nb);
        time(nm, [&] { hipLaunchKernelGGL((tc_kernel<true, 4, 256>), dim3(nb), dim3(256), 0, 0, buf, nvec, 1.f); });
        snprintf(nm, 64, "tc cached 64B/thr T1024 x%d", nb);
        time(nm, [&] { hipLaunchKernelGGL((tc_kernel<false, 4, 1024>), dim3(nb), dim3(1024), 0, 0, buf, nvec, 1.f); });
        snprintf(nm, 64, "stride cached 4KB T1024 x%d", nb);
        time(nm, [&] { hipLaunchKernelGGL((tc_kernel<false, 1, 1024>), dim3(nb), dim3(1024), 0, 0, buf, nvec, 1.f); });
    }
    {   // one block per chunk, no loop (what a fill kernel does): grid = nvec / (T * V)
        time("oneshot cached 16B/thr T256", [&] { hipLaunchKernelGGL((tc_kernel<false, 1, 256>), dim3((nvec + 255) / 256), dim3(256), 0, 0, buf, nvec, 1.f); });
        time("oneshot nt 16B/thr T256", [&] { hipLaunchKernelGGL((tc_kernel<true, 1, 256>), dim3((nvec + 255) / 256), dim3(256), 0, 0, buf, nvec, 1.f); });
        time("oneshot cached 64B/thr T256", [&] { hipLaunchKernelGGL((tc_kernel<false, 4, 256>), dim3((nvec + 1023) / 1024), dim3(256), 0, 0, buf, nvec, 1.f); });
        time("oneshot cached 16B/thr T1024", [&] { hipLaunchKernelGGL((tc_kernel<false, 1, 1024>), dim3((nvec + 1023) / 1024), dim3(1024), 0, 0, buf, nvec, 1.f); });
        time("oneshot cached 64B/thr T64", [&] { hipLaunchKernelGGL((tc_kernel<false, 4, 64>), dim3((nvec + 255) / 256), dim3(64), 0, 0, buf, nvec, 1.f); });
    }
    return 0;
}
