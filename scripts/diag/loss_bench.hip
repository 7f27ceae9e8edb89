// Diagnostic: the loss kernels alone (pair, CE only, focal only), event-timed; run under rocprofv3 --pmc for SQ counters.
// usage: loss_bench [N=8] [K=21] [h=41] [H=321] [reps=20] [logit scale=10]
#include "../../denet_amd/csrc/upsample_loss.hip"
#include <cstdio>
#include <cstdlib>
#include <vector>
int denet_ensure_dynamic_lds(const void* kernel, size_t bytes) {
    return hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes) == hipSuccess ? 0 : -1;
}
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
int main(int argc, char** argv) {
    const int N = argc > 1 ? atoi(argv[1]) : 8, K = argc > 2 ? atoi(argv[2]) : 21, h = argc > 3 ? atoi(argv[3]) : 41,
              H = argc > 4 ? atoi(argv[4]) : 321, reps = argc > 5 ? atoi(argv[5]) : 20;
    const float amp = argc > 6 ? atof(argv[6]) : 10.f;
    const int w = h, W = H, P = h * w, Q = H * W;
    float *logits, *binl, *wgt, *pk, *pp, *da, *db;
    int64_t *Y, *Yb;
    CK(hipMalloc(&logits, (size_t)N * K * P * 4)); CK(hipMalloc(&binl, (size_t)N * 2 * P * 4));
    CK(hipMalloc(&wgt, N * K * 4)); CK(hipMalloc(&Y, (size_t)N * Q * 8)); CK(hipMalloc(&Yb, (size_t)N * Q * 8));
    CK(hipMalloc(&da, (size_t)N * K * P * 4)); CK(hipMalloc(&db, (size_t)N * K * P * 4));
    const int64_t parts_k = denet_upsample_loss_parts(K, h, w, H, W, 1), parts_p = denet_upsample_loss_parts(2, h, w, H, W, 0);
    if (parts_k < 0 || parts_p < 0) { printf("unsupported geometry\n"); return 1; }
    CK(hipMalloc(&pk, N * parts_k * 8)); CK(hipMalloc(&pp, N * parts_p * 8));
    std::vector<float> hl((size_t)N * K * P);
    for (size_t i = 0; i < hl.size(); ++i) hl[i] = (float)((i * 2654435761u) % 1000) / 1000.f * amp;
    CK(hipMemcpy(logits, hl.data(), hl.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(binl, hl.data(), (size_t)N * 2 * P * 4, hipMemcpyHostToDevice));
    std::vector<float> hw(N * K, 1.f); CK(hipMemcpy(wgt, hw.data(), hw.size() * 4, hipMemcpyHostToDevice));
    std::vector<int64_t> hy((size_t)N * Q);
    for (size_t i = 0; i < hy.size(); ++i) hy[i] = (int64_t)((i * 40503u) % K);
    CK(hipMemcpy(Y, hy.data(), hy.size() * 8, hipMemcpyHostToDevice));
    for (size_t i = 0; i < hy.size(); ++i) hy[i] &= 1;
    CK(hipMemcpy(Yb, hy.data(), hy.size() * 8, hipMemcpyHostToDevice));
    LossGeom g;
    loss_geom(K, h, w, H, W, true, &g);
    printf("geom: xsplit %d rows_max %d emax %d xw_max %d ncmax %d lds %zu | chain em %d kchunk %d lds %zu\n", g.xsplit,
           g.rows_max, g.emax, g.xw_max, g.ncmax, g.lds, g.em, g.kchunk, g.lds_chain);
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int mode = 0; mode < 3; ++mode) {
        float best = 1e9f, sum = 0.f;
        for (int rep = 0; rep < reps + 3; ++rep) {
            CK(hipEventRecord(e0));
            int rc = 0;
            if (mode == 0) rc = denet_loss_pair_fwd(logits, Y, wgt, binl, Yb, N, N, K, h, w, H, W, 0.25f, 2.f, pk, pp, da, db, nullptr);
            if (mode == 1) rc = denet_upsample_wce_fwd(logits, Y, wgt, N, K, h, w, H, W, pk, da, db, nullptr);
            if (mode == 2) rc = denet_upsample_focal_fwd(binl, Yb, N, h, w, H, W, 0.25f, 2.f, pp, nullptr);
            CK(hipEventRecord(e1));
            CK(hipEventSynchronize(e1));
            if (rc) { printf("rc=%d\n", rc); return 1; }
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            if (rep >= 3) { sum += ms; if (ms < best) best = ms; }
        }
        printf("%s N=%d K=%d %dx%d -> %dx%d: mean %.1f us, best %.1f us\n", mode == 0 ? "pair" : mode == 1 ? "wce+grad" : "focal", N, K,
               h, w, H, W, sum / reps * 1e3, best * 1e3);
    }
    return 0;
}
