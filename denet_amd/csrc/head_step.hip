// The whole head as two enqueue-only entry points: everything
// DENetTrainer.step does between the feature producer and optimizer.step()
// (trainer/trainers.py:951-975), i.e. K0..K5 forward and K4b..K1b backward, on
// one stream, with a caller-provided workspace and no host synchronisation
// (the reference syncs N*way + 3 times per step: SURVEY.md §1).
#include "denet_common.hpp"
#include "dev_classify.hpp"
#include "dev_masked_sum.hpp"
#include "dev_proto_extend.hpp"
#include "dev_upsample_loss.hpp"

struct HeadWs {
    float *mask, *cnt, *S, *hidden, *gate, *pool, *Wn, *part_k, *part_p, *scale, *dlog_a, *dlog_b, *dWn_part,
        *coef, *u;
    int64_t total;   // bytes
};

static inline int64_t al(int64_t n) { return (n + 63) & ~(int64_t)63; }   // floats, 256-B granules

static HeadWs carve(void* base, int64_t N, int64_t way, int64_t shot, int64_t K, int64_t C, int64_t h, int64_t w,
                    int64_t parts_k, int64_t parts_p) {
    const int64_t B = N * way * shot, P = h * w, NB = N * way;
    const int64_t n_part = denet_classify_bwd_parts(P);
    float* p = (float*)base;
    HeadWs s;
    int64_t off = 0;
#define TAKE(field, n) s.field = p + off; off += al(n)
    TAKE(mask, B * P);
    TAKE(cnt, B);
    TAKE(S, B * C);
    TAKE(hidden, B * C);
    TAKE(gate, B * C);
    TAKE(pool, B * C);
    TAKE(Wn, N * K * C);
    TAKE(part_k, N * parts_k * 2);
    TAKE(part_p, NB * parts_p * 2);
    TAKE(scale, N);
    TAKE(dlog_a, N * K * P);
    TAKE(dlog_b, N * K * P);
    TAKE(dWn_part, N * n_part * K * C);
    TAKE(coef, B * C);
    TAKE(u, 3 * B * C + N * K * C);
#undef TAKE
    s.total = off * 4;
    return s;
}

// ---------------------------------------------------------------------------
// Fused launches of the backward half.  At batch 8 every kernel of the step is latency-bound
// (5-30 us for a few MB), so the step time is the LENGTH of the dependency chain; independent
// pieces therefore share a launch, block roles picked by blockIdx:
//   B1  dWn slabs (K3b, first half)   [+ the loss reduction, in block (0,0), when deferred]
//   B2  K2b chain (N*way blocks)  ||  dFq (K3b, second half: N*T blocks)      512 threads
//   B3  K2b parameter gradients   ||  dFs = mask * coef (K1b)                  256 threads
// ---------------------------------------------------------------------------
template <typename T, int RT>
__global__ __launch_bounds__(256) void head_bwd_dwn_kernel(ClsBwdArgs g, const float* __restrict__ part_p, int NB,
                                                          int parts_p, float* __restrict__ losses) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    classify_bwd_body<T, RT, 4>(g, blockIdx.y, blockIdx.x, sm);
    if (losses != nullptr && blockIdx.x == 0 && blockIdx.y == 0) {
        __syncthreads();
        loss_finalize_body(g.part_k, part_p, g.N, NB, g.parts_k, parts_p, 0, losses, nullptr, nullptr, sm);
    }
}

template <typename T, int RT, int C, int CH>
__global__ __launch_bounds__(512) void head_bwd_mid_kernel(PeChainArgs ch, ClsBwdArgs g, int groups, int n_tiles) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    const int bx = blockIdx.x;
    if (bx < groups) {
        proto_extend_bwd_chain_body<C, CH>(ch, bx, sm);
    } else {
        const int r = bx - groups;
        classify_bwd_body<T, RT, 8>(g, r / n_tiles, r % n_tiles, sm);
    }
}

template <typename T>
__global__ __launch_bounds__(256) void head_bwd_tail_kernel(PeParamsArgs pa, const float* __restrict__ mask,
                                                           const float* __restrict__ coef, T* __restrict__ dFs, int C,
                                                           int P, int rpb, int ncb, int gx) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    const int bx = blockIdx.x, nms = gridDim.x - 4 * gx;
    if (bx < nms) {
        masked_sum_bwd_body<T>(mask, coef, dFs, C, P, rpb, bx / ncb, bx % ncb, sm);
    } else {
        const int r = bx - nms;
        proto_extend_bwd_params_body(pa, r / gx, r % gx, sm);
    }
}

template <typename T, int RT>
static int launch_bwd_dwn(const ClsBwdArgs& g, const float* part_p, int NB, int parts_p, float* losses, int n_part,
                          int N, hipStream_t st) {
    hipLaunchKernelGGL((head_bwd_dwn_kernel<T, RT>), dim3((unsigned)n_part, (unsigned)N), dim3(256), CLS_BWD_LDS(g.K),
                       st, g, part_p, NB, parts_p, losses);
    return DENET_OK;
}

template <typename T, int RT>
static int launch_bwd_mid(const PeChainArgs& ch, const ClsBwdArgs& g, int groups, int n_tiles, int N, hipStream_t st) {
    const size_t l0 = PE_CHAIN_LDS(g.C), l1 = CLS_BWD_LDS(g.K);
    size_t lds = l0 > l1 ? l0 : l1;
    // When the whole grid fits on the chip one block per CU, ask for more than half a CU's LDS so that
    // no two blocks share a CU: the chain blocks (the long pole: 768 KB of weights from L2 through ONE
    // CU each) then never compete with a dFq block for their CU's memory pipeline.
    if (groups + N * n_tiles <= 256 && lds < 82 * 1024) lds = 82 * 1024;
    dim3 grid((unsigned)(groups + N * n_tiles)), block(512);
    if (g.C == 256) {
        if (denet_ensure_dynamic_lds(reinterpret_cast<const void*>(head_bwd_mid_kernel<T, RT, 256, 16>), lds))
            return DENET_E_LAUNCH;
        hipLaunchKernelGGL((head_bwd_mid_kernel<T, RT, 256, 16>), grid, block, lds, st, ch, g, groups, n_tiles);
    } else if (g.C == 64) {
        if (denet_ensure_dynamic_lds(reinterpret_cast<const void*>(head_bwd_mid_kernel<T, RT, 64, 8>), lds))
            return DENET_E_LAUNCH;
        hipLaunchKernelGGL((head_bwd_mid_kernel<T, RT, 64, 8>), grid, block, lds, st, ch, g, groups, n_tiles);
    } else {
        return DENET_E_UNSUPPORTED;
    }
    return DENET_OK;
}

#define RT_DISPATCH(fn, T, RT, ...)                                \
    ((RT) == 1 ? fn<T, 1>(__VA_ARGS__) : (RT) == 2 ? fn<T, 2>(__VA_ARGS__) \
     : (RT) == 3 ? fn<T, 3>(__VA_ARGS__) : (RT) == 4 ? fn<T, 4>(__VA_ARGS__) : DENET_E_UNSUPPORTED)

extern "C" int64_t denet_head_step_workspace(int64_t N, int64_t way, int64_t shot, int64_t K, int64_t C, int64_t h,
                                             int64_t w, int64_t H, int64_t W) {
    if (N <= 0 || way <= 0 || shot <= 0 || K <= 0 || C <= 0 || h <= 0 || w <= 0 || H <= 0 || W <= 0)
        return DENET_E_BADARG;
    const int64_t pk = denet_upsample_loss_parts(K, h, w, H, W, 1), pp = denet_upsample_loss_parts(2, h, w, H, W, 0);
    if (pk < 0 || pp < 0) return DENET_E_UNSUPPORTED;
    return carve(nullptr, N, way, shot, K, C, h, w, pk, pp).total;
}

#define TRY(x)                 \
    do {                       \
        int rc__ = (x);        \
        if (rc__) return rc__; \
    } while (0)

// Optional probe: bracket ONE of the step's kernels with caller-owned HIP events on the
// launch stream, so a harness can time that kernel inside its own timed region.
#define PROBED(id, x)                                                                                  \
    do {                                                                                               \
        const bool on__ = probe && probe->which == (id) && probe->start && probe->stop;                \
        if (on__ && hipEventRecord((hipEvent_t)probe->start, (hipStream_t)stream) != hipSuccess)       \
            return DENET_E_LAUNCH;                                                                     \
        TRY(x);                                                                                        \
        if (on__ && hipEventRecord((hipEvent_t)probe->stop, (hipStream_t)stream) != hipSuccess)        \
            return DENET_E_LAUNCH;                                                                     \
    } while (0)

extern "C" int denet_head_step_fwd(const void* Fq, const void* Fs, int feat_dtype, const int64_t* Ys,
                                   const int64_t* labels, const int64_t* Yq, const int64_t* Yq_full,
                                   const float* wgt, const float* weight, const float* W1, const float* b1,
                                   const float* W2, const float* b2, const float* Wl, const float* bl,
                                   const float* gate_mask, const float* weight_mask, int64_t N, int64_t way,
                                   int64_t shot, int64_t K, int64_t C, int64_t h, int64_t w, int64_t H, int64_t W,
                                   int with_grad, float* logits_full, float* logits_binary,
                                   float* losses /*[2]: loss_k, loss_p*/, void* ws, int64_t ws_bytes,
                                   const denet_probe_t* probe, void* stream) {
    if (!ws || !logits_full || !logits_binary || !losses) return DENET_E_BADARG;
    if (N <= 0 || way <= 0 || shot <= 0 || K <= 0 || C <= 0 || h <= 0 || w <= 0) return DENET_E_BADARG;
    const int64_t pk = denet_upsample_loss_parts(K, h, w, H, W, 1), pp = denet_upsample_loss_parts(2, h, w, H, W, 0);
    if (pk < 0 || pp < 0) return DENET_E_UNSUPPORTED;
    HeadWs s = carve(ws, N, way, shot, K, C, h, w, pk, pp);
    if (ws_bytes < s.total) return DENET_E_WORKSPACE;
    const int64_t B = N * way * shot, P = h * w, NB = N * way;
    // the grad and no-grad variants of the CE kernel may tile the columns differently
    const int64_t pk_run = with_grad ? pk : denet_upsample_loss_parts(K, h, w, H, W, 0);
    PROBED(DENET_K_MASKED_SUM_FWD,
           denet_masked_pool_fwd(Fs, feat_dtype, Ys, B, C, H, W, h, w, s.mask, s.cnt, s.S, stream));
    PROBED(DENET_K_PROTO_EXTEND_FWD,
           denet_proto_extend_fwd(s.S, s.cnt, W1, b1, W2, b2, Wl, bl, weight, labels, gate_mask, weight_mask, N, way,
                                  shot, K, C, P, s.hidden, s.gate, s.pool, s.Wn, nullptr, stream));
    PROBED(DENET_K_CLASSIFY_FWD,
           denet_classify_fwd(Fq, feat_dtype, s.Wn, labels, N, way, K, C, P, 1, logits_full, logits_binary, stream));
    if (Yq_full && Yq) {
        PROBED(DENET_K_WCE_FWD,
               denet_loss_pair_fwd(logits_full, Yq_full, wgt, logits_binary, Yq, N, NB, K, h, w, H, W, 0.25f, 2.f,
                                   s.part_k, s.part_p, with_grad ? s.dlog_a : nullptr, with_grad ? s.dlog_b : nullptr,
                                   stream));
    } else if (Yq_full) {
        PROBED(DENET_K_WCE_FWD,
               denet_upsample_wce_fwd(logits_full, Yq_full, wgt, N, K, h, w, H, W, s.part_k,
                                      with_grad ? s.dlog_a : nullptr, with_grad ? s.dlog_b : nullptr, stream));
    } else if (Yq) {
        PROBED(DENET_K_FOCAL_FWD,
               denet_upsample_focal_fwd(logits_binary, Yq, NB, h, w, H, W, 0.25f, 2.f, s.part_p, stream));
    }
    if ((Yq_full || Yq) && with_grad != DENET_GRAD_DEFER_LOSSES)
        PROBED(DENET_K_LOSS_FINALIZE,
               denet_loss_finalize(Yq_full ? s.part_k : nullptr, Yq ? s.part_p : nullptr, N, NB, pk_run, pp, 0,
                                   losses, nullptr, s.scale, stream));
    return DENET_OK;
}

extern "C" int denet_head_step_bwd(const void* Fq, int feat_dtype, const int64_t* labels, const float* W1,
                                   const float* W2, const float* Wl, const float* gate_mask,
                                   const float* weight_mask, const float* grad_loss_k /*device scalar or NULL*/,
                                   int64_t N, int64_t way, int64_t shot, int64_t K, int64_t C, int64_t h, int64_t w,
                                   int64_t H, int64_t W, void* dFq, void* dFs, float* dweight, float* dW1, float* db1, float* dW2,
                                   float* db2, float* dWl, float* dbl, float* losses /*NULL unless deferred*/,
                                   int have_focal, void* ws, int64_t ws_bytes, const denet_probe_t* probe,
                                   void* stream) {
    if (!ws || !dFq || !dFs || !Fq) return DENET_E_BADARG;
    if (N <= 0 || way <= 0 || shot <= 0 || K <= 0 || C <= 0 || h <= 0 || w <= 0) return DENET_E_BADARG;
    if (!dW1 || !db1 || !dW2 || !db2 || !dWl || !dbl || !dweight || !labels || !W1 || !W2 || !Wl)
        return DENET_E_BADARG;
    if (!(C == 64 || C == 256) || K > 128 || shot > 8 || N > 65535) return DENET_E_UNSUPPORTED;
    if (feat_dtype != DENET_F32 && feat_dtype != DENET_BF16) return DENET_E_UNSUPPORTED;
    const int64_t pk = denet_upsample_loss_parts(K, h, w, H, W, 1), pp = denet_upsample_loss_parts(2, h, w, H, W, 0);
    if (pk < 0 || pp < 0) return DENET_E_UNSUPPORTED;
    HeadWs s = carve(ws, N, way, shot, K, C, h, w, pk, pp);
    if (ws_bytes < s.total) return DENET_E_WORKSPACE;
    const int64_t B = N * way * shot, P = h * w, NB = N * way;
    if (P * 4 > 64 * 1024 || B * ((C + 15) / 16) > (int64_t)1 << 30) return DENET_E_UNSUPPORTED;
    const int n_part = (int)denet_classify_bwd_parts(P);
    const int RT = (int)((K + 31) / 32);
    const bool f32 = feat_dtype == DENET_F32;
    hipStream_t st = (hipStream_t)stream;
    float *u1 = s.u, *u2 = u1 + B * C, *u3 = u2 + B * C;

    // B1: dWn slabs; scale[n] is derived from the loss partials inside the kernel
    ClsBwdArgs g{Fq, s.Wn, s.dlog_a, s.dlog_b, nullptr, grad_loss_k, s.part_k, (int)pk, (int)N, (int)K, (int)C,
                 (int)P, n_part, nullptr, s.dWn_part, s.scale};
    {
        const bool on = probe && probe->which == DENET_K_CLASSIFY_BWD && probe->start && probe->stop;
        if (on && hipEventRecord((hipEvent_t)probe->start, st) != hipSuccess) return DENET_E_LAUNCH;
        const float* pp_ptr = have_focal ? s.part_p : nullptr;
        int rc = f32 ? RT_DISPATCH(launch_bwd_dwn, float, RT, g, pp_ptr, (int)NB, (int)pp, losses, n_part, (int)N, st)
                     : RT_DISPATCH(launch_bwd_dwn, unsigned short, RT, g, pp_ptr, (int)NB, (int)pp, losses, n_part,
                                   (int)N, st);
        if (rc) return rc;
        DENET_CHECK_LAUNCH();
        if (on && hipEventRecord((hipEvent_t)probe->stop, st) != hipSuccess) return DENET_E_LAUNCH;
    }
    // B2: chain || dFq
    {
        PeChainArgs ch{s.dWn_part, s.S, s.cnt, s.hidden, s.gate, W1, W2, Wl, gate_mask, labels, n_part, (int)way,
                       (int)shot, (int)K, (float)P, u1, u2, u3, s.coef};
        ClsBwdArgs g2 = g;
        g2.dFq = dFq;
        g2.dWn_part = nullptr;
        g2.scale_out = nullptr;
        const bool on = probe && probe->which == DENET_K_PE_BWD_CHAIN && probe->start && probe->stop;
        if (on && hipEventRecord((hipEvent_t)probe->start, st) != hipSuccess) return DENET_E_LAUNCH;
        int rc = f32 ? RT_DISPATCH(launch_bwd_mid, float, RT, ch, g2, (int)(N * way), n_part, (int)N, st)
                     : RT_DISPATCH(launch_bwd_mid, unsigned short, RT, ch, g2, (int)(N * way), n_part, (int)N, st);
        if (rc) return rc;
        DENET_CHECK_LAUNCH();
        if (on && hipEventRecord((hipEvent_t)probe->stop, st) != hipSuccess) return DENET_E_LAUNCH;
    }
    // B3: parameter gradients || dFs
    {
        PeParamsArgs pa{u1, u2, u3, s.S, s.hidden, s.pool, s.dWn_part, weight_mask, labels, n_part, (int)N, (int)way,
                        (int)B, (int)K, (int)C, (float)P, dW1, db1, dW2, db2, dWl, dbl, dweight};
        const int gx = pe_params_gx((int)K, (int)C);
        const int rpb = C >= 16 ? 16 : (int)C, ncb = (int)((C + rpb - 1) / rpb);
        size_t lds = (size_t)P * 4;
        if (lds < 4096) lds = 4096;
        dim3 grid((unsigned)(4 * gx + B * ncb)), block(256);
        const bool on = probe && probe->which == DENET_K_MASKED_SUM_BWD && probe->start && probe->stop;
        if (on && hipEventRecord((hipEvent_t)probe->start, st) != hipSuccess) return DENET_E_LAUNCH;
        if (f32)
            hipLaunchKernelGGL(head_bwd_tail_kernel<float>, grid, block, lds, st, pa, s.mask, s.coef, (float*)dFs,
                               (int)C, (int)P, rpb, ncb, gx);
        else
            hipLaunchKernelGGL(head_bwd_tail_kernel<unsigned short>, grid, block, lds, st, pa, s.mask, s.coef,
                               (unsigned short*)dFs, (int)C, (int)P, rpb, ncb, gx);
        DENET_CHECK_LAUNCH();
        if (on && hipEventRecord((hipEvent_t)probe->stop, st) != hipSuccess) return DENET_E_LAUNCH;
    }
    return DENET_OK;
}
