// The whole head as two enqueue-only entry points: everything
// DENetTrainer.step does between the feature producer and optimizer.step()
// (trainer/trainers.py:951-975), i.e. K0..K5 forward and K4b..K1b backward, on
// one stream, with a caller-provided workspace and no host synchronisation
// (the reference syncs N*way + 3 times per step: SURVEY.md §1).
#include "denet_common.hpp"

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
    PROBED(DENET_K_MASK_RESIZE, denet_mask_resize(Ys, B, H, W, h, w, s.mask, s.cnt, stream));
    PROBED(DENET_K_MASKED_SUM_FWD, denet_masked_sum_fwd(Fs, feat_dtype, s.mask, B, C, P, s.S, stream));
    PROBED(DENET_K_PROTO_EXTEND_FWD,
           denet_proto_extend_fwd(s.S, s.cnt, W1, b1, W2, b2, Wl, bl, weight, labels, gate_mask, weight_mask, N, way,
                                  shot, K, C, P, s.hidden, s.gate, s.pool, s.Wn, nullptr, stream));
    PROBED(DENET_K_CLASSIFY_FWD,
           denet_classify_fwd(Fq, feat_dtype, s.Wn, labels, N, way, K, C, P, 1, logits_full, logits_binary, stream));
    if (Yq_full) {
        PROBED(DENET_K_WCE_FWD,
               denet_upsample_wce_fwd(logits_full, Yq_full, wgt, N, K, h, w, H, W, s.part_k,
                                      with_grad ? s.dlog_a : nullptr, with_grad ? s.dlog_b : nullptr, stream));
    }
    if (Yq)
        PROBED(DENET_K_FOCAL_FWD,
               denet_upsample_focal_fwd(logits_binary, Yq, NB, h, w, H, W, 0.25f, 2.f, s.part_p, stream));
    if (Yq_full || Yq)
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
                                   float* db2, float* dWl, float* dbl, void* ws, int64_t ws_bytes,
                                   const denet_probe_t* probe, void* stream) {
    if (!ws || !dFq || !dFs) return DENET_E_BADARG;
    if (N <= 0 || way <= 0 || shot <= 0 || K <= 0 || C <= 0 || h <= 0 || w <= 0) return DENET_E_BADARG;
    const int64_t pk = denet_upsample_loss_parts(K, h, w, H, W, 1), pp = denet_upsample_loss_parts(2, h, w, H, W, 0);
    if (pk < 0 || pp < 0) return DENET_E_UNSUPPORTED;
    HeadWs s = carve(ws, N, way, shot, K, C, h, w, pk, pp);
    if (ws_bytes < s.total) return DENET_E_WORKSPACE;
    const int64_t B = N * way * shot, P = h * w;
    const int64_t n_part = denet_classify_bwd_parts(P);
    PROBED(DENET_K_CLASSIFY_BWD,
           denet_classify_bwd(Fq, feat_dtype, s.Wn, s.dlog_a, s.dlog_b, s.scale, grad_loss_k, N, K, C, P, dFq,
                              s.dWn_part, stream));
    if (!dW1 || !db1 || !dW2 || !db2 || !dWl || !dbl || !dweight || !labels || !W1 || !W2 || !Wl)
        return DENET_E_BADARG;
    if (!(C == 64 || C == 256) || K > 128 || shot > 8 || N > 65535) return DENET_E_UNSUPPORTED;
    float *u1 = s.u, *u2 = u1 + B * C, *u3 = u2 + B * C, *dWn = u3 + B * C;
    const float* dWn_src = s.dWn_part;
    if (n_part > 1) {
        PROBED(DENET_K_PE_BWD_REDUCE, denet_pe_bwd_reduce(s.dWn_part, n_part, N, K, C, dWn, stream));
        dWn_src = dWn;
    }
    PROBED(DENET_K_PE_BWD_CHAIN,
           denet_pe_bwd_chain(dWn_src, s.S, s.cnt, s.hidden, s.gate, W1, W2, Wl, labels, gate_mask, N, way, shot, K,
                              C, P, u1, u2, u3, s.coef, stream));
    PROBED(DENET_K_PE_BWD_PARAMS,
           denet_pe_bwd_params(u1, u2, u3, s.S, s.hidden, s.pool, dWn_src, labels, weight_mask, N, way, shot, K, C,
                               P, dW1, db1, dW2, db2, dWl, dbl, dweight, stream));
    PROBED(DENET_K_MASKED_SUM_BWD, denet_masked_sum_bwd(s.mask, s.coef, B, C, P, dFs, feat_dtype, stream));
    return DENET_OK;
}
