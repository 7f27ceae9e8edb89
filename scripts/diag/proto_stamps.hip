// Diagnostic build: phase stamps inside the proto kernels (never shipped).
#define DENET_STAMPS 1
#include "../../denet_amd/csrc/proto_extend.hip"
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
int main() {
    const int N = 8, way = 1, shot = 5, K = 21, C = 256, P = 1681, B = N * way * shot;
    float *S, *cnt, *W1, *b1, *W2, *b2, *Wl, *bl, *weight, *hidden, *gate, *pool, *Wn;
    int64_t* labels;
    CK(hipMalloc(&S, B * C * 4)); CK(hipMalloc(&cnt, B * 4)); CK(hipMalloc(&W1, C * C * 4)); CK(hipMalloc(&b1, C * 4));
    CK(hipMalloc(&W2, C * C * 4)); CK(hipMalloc(&b2, C * 4)); CK(hipMalloc(&Wl, C * C * 4)); CK(hipMalloc(&bl, C * 4));
    CK(hipMalloc(&weight, K * C * 4)); CK(hipMalloc(&hidden, B * C * 4)); CK(hipMalloc(&gate, B * C * 4));
    CK(hipMalloc(&pool, B * C * 4)); CK(hipMalloc(&Wn, N * K * C * 4)); CK(hipMalloc(&labels, N * 8));
    std::vector<float> h(C * C, 0.01f);
    for (float* p : {W1, W2, Wl}) CK(hipMemcpy(p, h.data(), C * C * 4, hipMemcpyHostToDevice));
    for (float* p : {b1, b2, bl}) CK(hipMemcpy(p, h.data(), C * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(weight, h.data(), K * C * 4, hipMemcpyHostToDevice));
    std::vector<float> s(B * C, 1.0f); CK(hipMemcpy(S, s.data(), B * C * 4, hipMemcpyHostToDevice));
    std::vector<float> c(B, 100.f); CK(hipMemcpy(cnt, c.data(), B * 4, hipMemcpyHostToDevice));
    std::vector<int64_t> l(N, 3); CK(hipMemcpy(labels, l.data(), N * 8, hipMemcpyHostToDevice));
    unsigned long long* d_st; CK(hipMalloc(&d_st, 64 * 8));
    CK(hipMemcpyToSymbol(HIP_SYMBOL(g_stamps), &d_st, sizeof(d_st)));
    // big buffer to flush caches between runs
    float* junk; CK(hipMalloc(&junk, 512u << 20));
    for (int rep = 0; rep < 3; ++rep) {
        CK(hipMemset(junk, rep, 512u << 20));
        CK(hipMemset(d_st, 0, 64 * 8));
        int rc = denet_proto_extend_fwd(S, cnt, W1, b1, W2, b2, Wl, bl, weight, labels, nullptr, nullptr, N, way, shot, K, C, P,
                                        hidden, gate, pool, Wn, nullptr);
        CK(hipDeviceSynchronize());
        unsigned long long st[64]; CK(hipMemcpy(st, d_st, 64 * 8, hipMemcpyDeviceToHost));
        printf("fwd rc=%d stamps(ns from start):", rc);
        for (int i = 1; i < 12 && st[i]; ++i) printf(" %llu", (st[i] - st[0]) * 10);
        printf("\n");
    }
    return 0;
}
