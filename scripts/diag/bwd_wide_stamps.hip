// Diagnostic: phase stamps of one block of the persistent K > 32 classifier backward (first backward launch of
// the fused step), at C3 / C5 shapes, synthetic zeros.  Build twice: -DDENET_STAMP_ROLE_DWN=1 (dWn role) / =0 (dFq).
#define DENET_STAMPS 1
#define DENET_BLOCK_STAMPS 1
#include <vector>
#include <algorithm>
#include "../../denet_amd/csrc/dev_classify.hpp"
// the launch of head_step.hip without the loss reduction of block (0, 0)
template <typename T, int RT>
__global__ __launch_bounds__(64 * CLW_BWD_NW) void head_bwd_dwn_wide_kernel(ClsBwdArgs g, ClsBwdArgs g2, const float*, int, int,
                                                                            float*) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    const int ngrp = g.n_part;
    if ((int)blockIdx.x < ngrp) classify_bwd_wide_body<T, RT, true>(g, blockIdx.y, blockIdx.x, sm);
    else classify_bwd_wide_body<T, RT, false>(g2, blockIdx.y, blockIdx.x - ngrp, sm);
}
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
int main(int argc, char** argv) {
    const int N = argc > 1 ? atoi(argv[1]) : 20, K = argc > 2 ? atoi(argv[2]) : 81, P = argc > 3 ? atoi(argv[3]) : 1681, C = 256;
    const int only = argc > 4 ? atoi(argv[4]) : 0;     // 1: launch only the stamped role's blocks
    const int ngrp = cls_bwd_wide_groups(N, P, RT);
    float *Fq, *dFq, *Wn, *da, *db, *slab;
    CK(hipMalloc(&Fq, (size_t)N * C * P * 4)); CK(hipMalloc(&dFq, (size_t)N * C * P * 4)); CK(hipMalloc(&Wn, (size_t)N * K * C * 4));
    CK(hipMalloc(&da, (size_t)N * K * P * 4)); CK(hipMalloc(&db, (size_t)N * K * P * 4)); CK(hipMalloc(&slab, (size_t)N * ngrp * K * C * 4));
    CK(hipMemset(Fq, 0, (size_t)N * C * P * 4)); CK(hipMemset(Wn, 0, (size_t)N * K * C * 4));
    CK(hipMemset(da, 0, (size_t)N * K * P * 4)); CK(hipMemset(db, 0, (size_t)N * K * P * 4));
    unsigned long long* d_st; CK(hipMalloc(&d_st, 64 * 8));
    CK(hipMemcpyToSymbol(HIP_SYMBOL(g_stamps), &d_st, sizeof(d_st)));
    const int nblk = 2 * ngrp * N;
    unsigned long long* d_bs; CK(hipMalloc(&d_bs, (size_t)nblk * 3 * 8));
    CK(hipMemcpyToSymbol(HIP_SYMBOL(g_bstamps), &d_bs, sizeof(d_bs)));
    ClsBwdArgs g{Fq, Wn, da, db, nullptr, nullptr, nullptr, 0, N, K, C, P, ngrp, nullptr, slab, nullptr};
    ClsBwdArgs g2 = g; g2.dFq = dFq; g2.dWn_part = nullptr;
    const size_t lds = CLW_BWD_LDS(3);
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(head_bwd_dwn_wide_kernel<float, 3>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    printf("N %d K %d P %d groups/episode %d blocks %d lds %zu role %s\n", N, K, P, ngrp, 2 * ngrp * N, lds, DENET_STAMP_ROLE_DWN ? "dWn" : "dFq");
    for (int rep = 0; rep < 4; ++rep) {
        CK(hipMemset(d_st, 0, 64 * 8));
        CK(hipDeviceSynchronize());
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL((head_bwd_dwn_wide_kernel<float, 3>), dim3(2 * ngrp, N), dim3(512), lds, 0, g, g2, (const float*)nullptr, 0, 0, (float*)nullptr);
        CK(hipEventRecord(e1));
        CK(hipDeviceSynchronize());
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        unsigned long long st[64]; CK(hipMemcpy(st, d_st, 64 * 8, hipMemcpyDeviceToHost));
        printf("total %.1f us; stamps (ns from block start): first barrier %llu |", ms * 1e3, (st[1] - st[0]) * 10);
        for (int t = 0; t < 4; ++t)
            if (st[2 + 4 * t]) printf(" tile%d: turn@%llu issued@%llu mfma-done@%llu barrier@%llu |", t, (st[2 + 4 * t] - st[0]) * 10,
                                      (st[3 + 4 * t] - st[0]) * 10, (st[4 + 4 * t] - st[0]) * 10, (st[5 + 4 * t] - st[0]) * 10);
        printf(" loop end %llu, block end %llu\n", (st[30] - st[0]) * 10, (st[31] - st[0]) * 10);
        std::vector<unsigned long long> bs((size_t)nblk * 3);
        CK(hipMemcpy(bs.data(), d_bs, bs.size() * 8, hipMemcpyDeviceToHost));
        unsigned long long s_min = ~0ull, s_max = 0, e_min = ~0ull, e_max = 0;
        for (int b = 0; b < nblk; ++b) { s_min = std::min(s_min, bs[3 * b]); s_max = std::max(s_max, bs[3 * b]); e_min = std::min(e_min, bs[3 * b + 1]); e_max = std::max(e_max, bs[3 * b + 1]); }
        std::vector<unsigned long long> dur[2];
        for (int b = 0; b < nblk; ++b) dur[(b % (2 * ngrp)) >= ngrp].push_back(bs[3 * b + 1] - bs[3 * b]);
        for (auto& d : dur) std::sort(d.begin(), d.end());
        if (rep == 3) {
            std::vector<std::pair<unsigned long long, int>> o;
            for (int b = 0; b < nblk; ++b) o.push_back({bs[3 * b + 1] - bs[3 * b], b});
            std::sort(o.begin(), o.end());
            printf("   slowest / fastest blocks (dur ns, start ns, n, x, xcc, hw_id):\n");
            for (int q = 0; q < 12; ++q) {
                const int b = q < 8 ? o[o.size() - 1 - q].second : o[q - 8].second;
                printf("     %llu %llu n=%d x=%d xcc=%llu hwid=%llx\n", (bs[3 * b + 1] - bs[3 * b]) * 10, (bs[3 * b] - s_min) * 10, b / (2 * ngrp), b % (2 * ngrp),
                       bs[3 * b + 2] >> 32, bs[3 * b + 2] & 0xffffffffull);
            }
        }
        printf("   blocks: starts spread %llu ns, first end %llu, last end %llu (from first start); duration dWn min/med/max %llu/%llu/%llu, dFq %llu/%llu/%llu ns\n",
               (s_max - s_min) * 10, (e_min - s_min) * 10, (e_max - s_min) * 10, dur[0].front() * 10, dur[0][dur[0].size() / 2] * 10, dur[0].back() * 10,
               dur[1].front() * 10, dur[1][dur[1].size() / 2] * 10, dur[1].back() * 10);
    }
    return 0;
}
