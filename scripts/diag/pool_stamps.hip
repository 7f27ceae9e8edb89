// Diagnostic build: start / end time and CU of every block of the pooling kernel (never shipped).
#define DENET_BLOCK_STAMPS 1
#include "../../denet_amd/csrc/masked_sum.hip"
#include <cstdio>
#include <cstdlib>
#include <vector>
int denet_ensure_dynamic_lds(const void* kernel, size_t bytes) {
    return hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes) == hipSuccess ? 0 : -1;
}
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
int main(int argc, char** argv) {
    // usage: pool_stamps [B=40] [chain=1] [bin us=1] [h=41] [H=321] [bf16=0]
    const int B = argc > 1 ? atoi(argv[1]) : 40, C = 256, h = argc > 4 ? atoi(argv[4]) : 41, w = h,
              H = argc > 5 ? atoi(argv[5]) : 321, W = H, P = h * w;
    const int chain = argc > 2 ? atoi(argv[2]) : 1;
    const double binw = argc > 3 ? atof(argv[3]) : 1.0;   // histogram bin width, us
    const int bf16 = argc > 6 ? atoi(argv[6]) : 0;
    const int fdt = bf16 ? DENET_BF16 : DENET_F32;
    float *Fs, *mask, *cnt, *S, *W1, *pre1;
    int64_t* Ys;
    CK(hipMalloc(&Fs, (size_t)B * C * P * 4)); CK(hipMemset(Fs, 0, (size_t)B * C * P * 4));
    CK(hipMalloc(&Ys, (size_t)B * H * W * 8)); CK(hipMemset(Ys, 0, (size_t)B * H * W * 8));
    CK(hipMalloc(&mask, (size_t)B * P * 4)); CK(hipMalloc(&cnt, B * 4)); CK(hipMalloc(&S, B * C * 4));
    CK(hipMalloc(&W1, C * C * 4)); CK(hipMemset(W1, 0, C * C * 4)); CK(hipMalloc(&pre1, (size_t)B * 16 * C * 4));
    const size_t nblk = (size_t)B * 16;
    unsigned long long* d_bs; CK(hipMalloc(&d_bs, nblk * 24));
    CK(hipMemcpyToSymbol(HIP_SYMBOL(g_bstamps), &d_bs, sizeof(d_bs)));
    if (denet_mask_resize(Ys, B, H, W, h, w, mask, cnt, nullptr)) return 1;
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int rep = 0; rep < 4; ++rep) {
        CK(hipMemset(d_bs, 0, nblk * 24));
        CK(hipDeviceSynchronize());
        CK(hipEventRecord(e0));
        // chain: 1 = fused mask gather + split-K tail (the fused step's launch), 0 = gather, no tail,
        //        2 = mask resized beforehand, pooling only (denet_masked_sum_fwd)
        int rc = chain == 2 ? denet_masked_sum_fwd(Fs, fdt, mask, B, C, h * w, S, nullptr)
                            : denet_masked_pool_fwd_chain(Fs, fdt, Ys, B, C, H, W, h, w, mask, cnt, S, chain ? W1 : nullptr, chain ? pre1 : nullptr, nullptr);
        CK(hipEventRecord(e1));
        CK(hipDeviceSynchronize());
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        printf("pool B=%d %dx%d %s chain=%d rc=%d total %.1f us = %.2f TB/s\n", B, h, w, bf16 ? "bf16" : "f32", chain, rc, ms * 1e3, (double)B * C * P * (bf16 ? 2 : 4) / (ms * 1e-3) / 1e12);
        if (rep == 3) {
            std::vector<unsigned long long> bs(nblk * 3); CK(hipMemcpy(bs.data(), d_bs, nblk * 24, hipMemcpyDeviceToHost));
            unsigned long long t0 = ~0ull;
            for (size_t b = 0; b < nblk; ++b) if (bs[3 * b] && bs[3 * b] < t0) t0 = bs[3 * b];
            int hs[32] = {0}, he[32] = {0}; double dur = 0;
            for (size_t b = 0; b < nblk; ++b) {
                const double s0 = (bs[3 * b] - t0) * 0.01, e = (bs[3 * b + 1] - t0) * 0.01;
                hs[(int)(s0 / binw) < 31 ? (int)(s0 / binw) : 31]++; he[(int)(e / binw) < 31 ? (int)(e / binw) : 31]++; dur += e - s0;
            }
            printf("blocks=%zu mean dur %.1f us -> mean resident blocks %.0f\n  start hist (%.1f us bins):", nblk, dur / nblk, dur / (ms * 1e3), binw);
            for (int i = 0; i < 24; ++i) printf(" %d", hs[i]);
            printf("\n  end hist:");
            for (int i = 0; i < 24; ++i) printf(" %d", he[i]);
            printf("\n");
            std::vector<int> percu(8 * 8 * 2 * 16, 0);
            for (size_t b = 0; b < nblk; ++b) { const unsigned long long v = bs[3 * b + 2]; const unsigned id = (unsigned)v, xc = (unsigned)(v >> 32) & 7;
                percu[(((xc * 8 + ((id >> 13) & 7)) * 2 + ((id >> 12) & 1)) * 16) + ((id >> 8) & 15)]++; }
            int hist[12] = {0};
            for (int c : percu) hist[c < 11 ? c : 11]++;
            printf("CUs by number of blocks:");
            for (int i = 0; i < 12; ++i) if (hist[i]) printf(" %d:%d", i, hist[i]);
            printf("\n");
        }
    }
    return 0;
}
