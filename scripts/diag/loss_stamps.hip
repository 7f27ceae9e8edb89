// Diagnostic build: phase stamps inside the fused up-sampling + loss kernel (never shipped).
#define DENET_STAMPS 1
#define DENET_BLOCK_STAMPS 1
#include "../../denet_amd/csrc/upsample_loss.hip"
#include <cstdio>
#include <cstdlib>
#include <vector>
int denet_ensure_dynamic_lds(const void* kernel, size_t bytes) {
    return hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes) == hipSuccess ? 0 : -1;
}
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
int main(int argc, char** argv) {
    const int N = argc > 1 ? atoi(argv[1]) : 8, K = 21, h = 41, w = 41, H = 321, W = 321, P = h * w, Q = H * W;
    float *logits, *binl, *wgt, *pk, *pp, *da, *db;
    int64_t *Y, *Yb;
    CK(hipMalloc(&logits, (size_t)N * K * P * 4)); CK(hipMalloc(&binl, (size_t)N * 2 * P * 4));
    CK(hipMalloc(&wgt, N * K * 4)); CK(hipMalloc(&Y, (size_t)N * Q * 8)); CK(hipMalloc(&Yb, (size_t)N * Q * 8));
    CK(hipMalloc(&da, (size_t)N * K * P * 4)); CK(hipMalloc(&db, (size_t)N * K * P * 4));
    const int64_t parts_k = denet_upsample_loss_parts(K, h, w, H, W, 1), parts_p = denet_upsample_loss_parts(2, h, w, H, W, 0);
    CK(hipMalloc(&pk, N * parts_k * 8)); CK(hipMalloc(&pp, N * parts_p * 8));
    std::vector<float> hl((size_t)N * K * P);
    for (size_t i = 0; i < hl.size(); ++i) hl[i] = (float)((i * 2654435761u) % 1000) / 100.f;
    CK(hipMemcpy(logits, hl.data(), hl.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(binl, hl.data(), (size_t)N * 2 * P * 4, hipMemcpyHostToDevice));
    std::vector<float> hw(N * K, 1.f); CK(hipMemcpy(wgt, hw.data(), hw.size() * 4, hipMemcpyHostToDevice));
    std::vector<int64_t> hy((size_t)N * Q);
    for (size_t i = 0; i < hy.size(); ++i) hy[i] = (int64_t)((i * 40503u) % K);
    CK(hipMemcpy(Y, hy.data(), hy.size() * 8, hipMemcpyHostToDevice));
    for (size_t i = 0; i < hy.size(); ++i) hy[i] &= 1;
    CK(hipMemcpy(Yb, hy.data(), hy.size() * 8, hipMemcpyHostToDevice));
    unsigned long long* d_st; CK(hipMalloc(&d_st, 64 * 8));
    CK(hipMemcpyToSymbol(HIP_SYMBOL(g_stamps), &d_st, sizeof(d_st)));
    const size_t nblk = (size_t)parts_k * 2 * N;
    unsigned long long* d_bs; CK(hipMalloc(&d_bs, nblk * 24)); CK(hipMemset(d_bs, 0, nblk * 24));
    CK(hipMemcpyToSymbol(HIP_SYMBOL(g_bstamps), &d_bs, sizeof(d_bs)));
    for (size_t l : {(size_t)0, (size_t)8192, (size_t)16384, (size_t)20000, (size_t)23400, (size_t)30000, (size_t)33000, (size_t)40000}) {
        int nb = -1;
        CK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, wce_chain_pair_kernel<true, 8, 1>, 256, l));
        printf("occupancy(pair<true>, lds=%zu) = %d blocks/CU\n", l, nb);
    }
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int rep = 0; rep < 4; ++rep) {
        CK(hipMemset(d_st, 0, 64 * 8));
        CK(hipDeviceSynchronize());
        CK(hipEventRecord(e0));
        int rc = denet_loss_pair_fwd(logits, Y, wgt, binl, Yb, N, N, K, h, w, H, W, 0.25f, 2.f, pk, pp, da, db, nullptr);
        CK(hipEventRecord(e1));
        CK(hipDeviceSynchronize());
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        unsigned long long st[64]; CK(hipMemcpy(st, d_st, 64 * 8, hipMemcpyDeviceToHost));
        printf("loss_pair N=%d rc=%d total %.1f us; stamps(ns from start):", N, rc, ms * 1e3);
        for (int i = 1; i < 7; ++i) printf(" %llu", st[i] ? (st[i] - st[0]) * 10 : 0ull);
        printf("\n");
        if (rep == 3) {
            std::vector<unsigned long long> bs(nblk * 3); CK(hipMemcpy(bs.data(), d_bs, nblk * 24, hipMemcpyDeviceToHost));
            unsigned long long t0 = ~0ull;
            for (size_t b = 0; b < nblk; ++b) if (bs[3 * b] && bs[3 * b] < t0) t0 = bs[3 * b];
            // histogram of start / end times in 2 us bins, CE blocks (first half) and focal blocks apart
            for (int half = 0; half < 2; ++half) {
                int hs[32] = {0}, he[32] = {0}; double dur = 0; size_t cntb = 0;
                for (size_t b = half * nblk / 2; b < (half + 1) * nblk / 2; ++b) {
                    if (!bs[3 * b]) continue;
                    const double s0 = (bs[3 * b] - t0) * 0.01, e0 = (bs[3 * b + 1] - t0) * 0.01;
                    hs[(int)(s0 / 2) < 31 ? (int)(s0 / 2) : 31]++; he[(int)(e0 / 2) < 31 ? (int)(e0 / 2) : 31]++;
                    dur += e0 - s0; ++cntb;
                }
                printf("%s blocks=%zu mean dur %.1f us\n  start hist (2us bins):", half ? "focal" : "CE", cntb, dur / cntb);
                for (int i = 0; i < 20; ++i) printf(" %d", hs[i]);
                printf("\n  end hist:");
                for (int i = 0; i < 20; ++i) printf(" %d", he[i]);
                printf("\n");
            }
            // CE blocks per CU (xcc, se, sh, cu) and the mean end time of the blocks by that count
            std::vector<int> percu(8 * 8 * 2 * 16, 0);
            auto cuof = [&](size_t b) { const unsigned long long v = bs[3 * b + 2]; const unsigned id = (unsigned)v, xc = (unsigned)(v >> 32) & 7;
                                        return (int)((((xc * 8 + ((id >> 13) & 7)) * 2 + ((id >> 12) & 1)) * 16) + ((id >> 8) & 15)); };
            for (size_t b = 0; b < nblk / 2; ++b) percu[cuof(b)]++;
            int hist[12] = {0}; double endsum[12] = {0}; int endn[12] = {0};
            for (int c : percu) hist[c < 11 ? c : 11]++;
            for (size_t b = 0; b < nblk / 2; ++b) { const int c = percu[cuof(b)]; endsum[c < 11 ? c : 11] += (bs[3 * b + 1] - t0) * 0.01; endn[c < 11 ? c : 11]++; }
            printf("CUs by number of CE blocks:");
            for (int i = 0; i < 12; ++i) if (hist[i]) printf(" %d:%d", i, hist[i]);
            printf("\nmean end time of CE blocks by their CU's count:");
            for (int i = 1; i < 12; ++i) if (endn[i]) printf(" %d:%.1f", i, endsum[i] / endn[i]);
            printf("\n");
        }
    }
    return 0;
}
