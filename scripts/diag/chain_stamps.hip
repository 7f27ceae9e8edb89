// Diagnostic build: phase stamps inside the K2b chain kernel (never shipped).
#define DENET_STAMPS 1
#include "../../denet_amd/csrc/proto_extend.hip"
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
int main() {
    const int N = 8, way = 1, shot = 5, K = 21, C = 256, P = 1681, B = N * way * shot, n_part = 27;
    float *S, *cnt, *W1, *W2, *Wl, *hidden, *gate, *dWn, *u1, *u2, *u3, *coef;
    int64_t* labels;
    CK(hipMalloc(&S, B * C * 4)); CK(hipMalloc(&cnt, B * 4)); CK(hipMalloc(&W1, C * C * 4));
    CK(hipMalloc(&W2, C * C * 4)); CK(hipMalloc(&Wl, C * C * 4)); CK(hipMalloc(&hidden, B * C * 4)); CK(hipMalloc(&gate, B * C * 4));
    CK(hipMalloc(&dWn, (size_t)N * n_part * K * C * 4)); CK(hipMalloc(&labels, N * 8));
    CK(hipMalloc(&u1, B * C * 4)); CK(hipMalloc(&u2, B * C * 4)); CK(hipMalloc(&u3, B * C * 4)); CK(hipMalloc(&coef, B * C * 4));
    std::vector<float> h(C * C, 0.01f);
    for (float* p : {W1, W2, Wl}) CK(hipMemcpy(p, h.data(), C * C * 4, hipMemcpyHostToDevice));
    std::vector<float> s(B * C, 1.0f);
    for (float* p : {S, hidden, gate}) CK(hipMemcpy(p, s.data(), B * C * 4, hipMemcpyHostToDevice));
    CK(hipMemset(dWn, 0, (size_t)N * n_part * K * C * 4));
    std::vector<float> c(B, 100.f); CK(hipMemcpy(cnt, c.data(), B * 4, hipMemcpyHostToDevice));
    std::vector<int64_t> l(N, 3); CK(hipMemcpy(labels, l.data(), N * 8, hipMemcpyHostToDevice));
    unsigned long long* d_st; CK(hipMalloc(&d_st, 64 * 8));
    CK(hipMemcpyToSymbol(HIP_SYMBOL(g_stamps), &d_st, sizeof(d_st)));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int np : {27, 1}) for (int rep = 0; rep < 3; ++rep) {
        CK(hipMemset(d_st, 0, 64 * 8));
        CK(hipDeviceSynchronize());
        CK(hipEventRecord(e0));
        int rc = denet_pe_bwd_chain(dWn, np, S, cnt, hidden, gate, W1, W2, Wl, labels, nullptr, N, way, shot, K, C, P, u1, u2, u3, coef, nullptr);
        CK(hipEventRecord(e1));
        CK(hipDeviceSynchronize());
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        unsigned long long st[64]; CK(hipMemcpy(st, d_st, 64 * 8, hipMemcpyDeviceToHost));
        printf("chain n_part=%d rc=%d total %.1f us; stamps(ns from start):", np, rc, ms * 1e3);
        for (int i = 1; i < 9; ++i) printf(" %llu", st[i] ? (st[i] - st[0]) * 10 : 0ull);
        printf("\n");
    }
    return 0;
}
