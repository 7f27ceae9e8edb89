#define DENET_STAMPS 1
#include "../../denet_amd/csrc/classify.hip"
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
#include <cstdlib>
int denet_ensure_dynamic_lds(const void* kernel, size_t bytes) {
    return hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes) == hipSuccess ? 0 : -1;
}
int main(int argc, char** argv) {
    const int N = argc > 1 ? atoi(argv[1]) : 8, way = argc > 2 ? atoi(argv[2]) : 1, K = argc > 3 ? atoi(argv[3]) : 21, C = 256, P = argc > 4 ? atoi(argv[4]) : 1681;
    const int dt = argc > 5 && atoi(argv[5]) ? DENET_BF16 : DENET_F32;
    float *Fq, *Wn, *logits, *bin; int64_t* labels;
    CK(hipMalloc(&Fq, (size_t)N * C * P * 4)); CK(hipMalloc(&Wn, N * K * C * 4)); CK(hipMalloc(&logits, N * K * P * 4));
    CK(hipMalloc(&bin, (size_t)N * way * 2 * P * 4)); CK(hipMalloc(&labels, N * way * 8));
    CK(hipMemset(Fq, 0, (size_t)N * C * P * 4)); CK(hipMemset(Wn, 0, N * K * C * 4));
    std::vector<int64_t> l(N * way, 3); for (int i = 0; i < N * way; ++i) l[i] = 3 + (i % way); CK(hipMemcpy(labels, l.data(), N * way * 8, hipMemcpyHostToDevice));
    unsigned long long* d_st; CK(hipMalloc(&d_st, 64 * 8));
    CK(hipMemcpyToSymbol(HIP_SYMBOL(g_stamps), &d_st, sizeof(d_st)));
    float* junk; CK(hipMalloc(&junk, 512u << 20));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int rep = 0; rep < 4; ++rep) {
        if (rep < 2) CK(hipMemset(junk, rep, 512u << 20));
        CK(hipMemset(d_st, 0, 64 * 8));
        CK(hipDeviceSynchronize());
        CK(hipEventRecord(e0));
        int rc = denet_classify_fwd(Fq, dt, Wn, labels, N, way, K, C, P, 1, logits, bin, nullptr);
        CK(hipEventRecord(e1));
        CK(hipDeviceSynchronize());
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        unsigned long long st[64]; CK(hipMemcpy(st, d_st, 64 * 8, hipMemcpyDeviceToHost));
        printf("classify_fwd rc=%d %s total %.1f us; block0 stamps(ns):", rc, rep < 2 ? "cold" : "warm", ms * 1e3);
        for (int i = 1; i < 8 && st[i]; ++i) printf(" %llu", (st[i] - st[0]) * 10);
        printf("\n");
    }
    return 0;
}
