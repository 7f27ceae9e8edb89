/*
 * denet_hip.h -- C ABI of libdenet_hip.so: the MI355X (gfx950) kernels behind
 * the DENet few-shot segmentation head.
 *
 * The reference (lizhaoliu-Lec/DENet) is pure Python/PyTorch and has no FFI;
 * every entry point below replaces a chain of stock ATen ops in the reference
 * file:line it cites (paths relative to the reference root).  The host side
 * (denet_amd/, Python) binds these with ctypes; see INTEGRATION.md.
 *
 * Conventions
 *  - All pointers are DEVICE pointers into caller-owned memory unless noted.
 *  - Every call only ENQUEUES work on `stream` (a hipStream_t passed as
 *    void*); nothing here allocates, frees or synchronises.
 *  - Return value: DENET_OK (0) or a negative DENET_E_* code; nothing throws.
 *    denet_strerror() turns a code into text.
 *  - Tensors are contiguous, row-major, in the layouts the reference uses:
 *      features  NCHW        [B, C, h, w]   fp32 (DENET_F32) or bf16 (DENET_BF16)
 *      masks     [B, H, W]   int64
 *      logits    [N, K, h, w] fp32
 *  - B = N*way*shot support images, P = h*w feature positions.
 *  - C (feature depth) must be 64 or 256 (the reference fixes 256: model/head/denet.py:27,47;
 *    64 exists for small test fixtures); K <= 128; shot <= 8.
 */
#ifndef DENET_HIP_H
#define DENET_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define DENET_OK            0
#define DENET_E_BADARG     -1   /* null pointer / non-positive dimension            */
#define DENET_E_UNSUPPORTED -2  /* shape outside what the kernels were built for    */
#define DENET_E_LAUNCH     -3   /* hipLaunchKernel reported an error                */
#define DENET_E_WORKSPACE  -4   /* workspace too small                              */
#define DENET_E_STATIC_LIMIT -5 /* beyond the static scratch of a no-allocation entry point */

#define DENET_F32  0
#define DENET_BF16 1

/* Library version / build arch string ("gfx950"). Host pointers. */
const char* denet_version(void);
const char* denet_strerror(int code);

/* ---- K1: masked pooling -------------------------------------------------
 * Replaces GAM.mask + adaptive_avg_pool2d (model/common.py:234-243) and the
 * pooling branch of MaskedAveragePooling.forward (model/common.py:188-194).
 * One pass over Fs yields everything both need:
 *   mask[b,p] = float(Ys[b, sy(p), sx(p)])     nearest: s = min(floor(d*fp32(in/out)), in-1)
 *   cnt[b]    = sum_p mask[b,p]
 *   S[b,c]    = sum_p mask[b,p] * Fs[b,c,p]
 */
int denet_mask_resize(const int64_t* Ys, int64_t B, int64_t H, int64_t W, int64_t h, int64_t w,
                      float* mask /*[B,h*w]*/, float* cnt /*[B]*/, void* stream);
int denet_masked_sum_fwd(const void* Fs, int feat_dtype, const float* mask, int64_t B, int64_t C,
                         int64_t P, float* S /*[B,C]*/, void* stream);
/* K0 + K1 in one launch: the mask is gathered from Ys [B,H,W] inside the pooling kernel, and
 * mask [B,h*w] / cnt [B] are written out for the backward pass. */
int denet_masked_pool_fwd(const void* Fs, int feat_dtype, const int64_t* Ys, int64_t B, int64_t C,
                          int64_t H, int64_t W, int64_t h, int64_t w, float* mask, float* cnt, float* S,
                          void* stream);
/* Backward of the above (autograd of model/common.py:192,239,243,245):
 *   dFs[b,c,p] = mask[b,p] * coef[b,c]     (rank-1; Fs itself is not re-read) */
int denet_masked_sum_bwd(const float* mask, const float* coef /*[B,C]*/, int64_t B, int64_t C,
                         int64_t P, void* dFs, int feat_dtype, void* stream);
/* The same operation (same reference lines, bit-identical results) for large streams: one block per contiguous
 * chunk of the flat dFs tensor, in address order, non-temporal stores -- the write pattern of a fill kernel.  The
 * fused step uses it for bf16 gradients above 80 MB.  dFs 16-byte aligned, C a multiple of 64, C*P < 2^22. */
int denet_masked_sum_bwd_chunked(const float* mask, const float* coef /*[B,C]*/, int64_t B, int64_t C,
                                 int64_t P, void* dFs, int feat_dtype, void* stream);

/* ---- K2: gate, projection, dynamic extension ----------------------------
 * Replaces GAM.gate (model/common.py:225-232,244), MaskedAveragePooling's
 * division + 1x1 conv (model/common.py:194-196), WeightEstimator._extend
 * (model/common.py:317-339) and the per-episode Python loop (:368).
 *   att = S/P; hidden = relu(W1 att + b1); z = W2 hidden + b2
 *   g = sigmoid(z * gate_mask); pool = g*S/(cnt+1e-3); proto = Wl pool + bl
 *   Wn[n] = weight * weight_mask[n]; Wn[n, labels[n,j]] = mean_s proto[n,j,s]
 *           (j ascending: a later way wins a duplicated label)
 * labels are read ON DEVICE (no .item() sync, cf. model/common.py:328).
 * gate_mask [B,C] / weight_mask [N,K,C]: already-scaled dropout multipliers
 * (0 or 1/(1-p)) or NULL for eval mode.
 * hidden, gate, pool [B,C] are saved for the backward pass.
 * proto [N*way,C] (may be NULL): the shot-mean prototype of every (episode, way)
 * group, for the support_free bookkeeping (model/common.py:332-338).
 */
int denet_proto_extend_fwd(const float* S, const float* cnt,
                           const float* W1, const float* b1, const float* W2, const float* b2,
                           const float* Wl, const float* bl, const float* weight,
                           const int64_t* labels /*[N,way]*/,
                           const float* gate_mask, const float* weight_mask,
                           int64_t N, int64_t way, int64_t shot, int64_t K, int64_t C, int64_t P,
                           float* hidden, float* gate, float* pool, float* Wn /*[N,K,C]*/,
                           float* proto /*[N*way,C] or NULL*/, void* stream);

/* Backward of K2 (autograd of the same lines).  Inputs: dWn [N,K,C] given as
 * `n_part` partial slabs [N,n_part,K,C] that are summed here (K3b writes one
 * slab per position tile).  Outputs: gradients of the seven parameters and
 * coef [B,C] for denet_masked_sum_bwd.  ws: workspace of
 * denet_proto_extend_bwd_workspace(N,way,shot,K,C) bytes. */
int64_t denet_proto_extend_bwd_workspace(int64_t N, int64_t way, int64_t shot, int64_t K, int64_t C);
int denet_proto_extend_bwd(const float* dWn_part, int64_t n_part,
                           const float* S, const float* cnt,
                           const float* hidden, const float* gate, const float* pool,
                           const float* W1, const float* W2, const float* Wl,
                           const int64_t* labels, const float* gate_mask, const float* weight_mask,
                           int64_t N, int64_t way, int64_t shot, int64_t K, int64_t C, int64_t P,
                           float* dW1, float* db1, float* dW2, float* db2, float* dWl, float* dbl,
                           float* dweight /*[K,C]*/, float* coef /*[B,C]*/,
                           void* ws, int64_t ws_bytes, void* stream);

/* The same backward with NO workspace argument and no allocation of any kind (no hipMalloc / hipFree, no
 * allocating library routine, no torch): the scratch lives in a static device array of
 * denet_proto_extend_bwd_static_limit() bytes (1 MiB; every GPU the library runs on holds it until the process
 * ends), so problems with denet_proto_extend_bwd_workspace(...) above that limit return DENET_E_STATIC_LIMIT.
 * No counterpart in the reference: it exists for callers that cannot allocate -- code under HIP graph capture,
 * tight serving loops, harnesses that forbid hipMalloc.  Same arguments, outputs and arithmetic as
 * denet_proto_extend_bwd (it shares its kernels).  All calls on one GPU share that GPU's array: calls that
 * overlap on one GPU (different streams or threads) race; serialise them.  The first call on a device looks the
 * array's address up (hipGetSymbolAddress), so make one call outside any stream capture first. */
int64_t denet_proto_extend_bwd_static_limit(void);
int denet_proto_extend_bwd_static(const float* dWn_part, int64_t n_part,
                                  const float* S, const float* cnt,
                                  const float* hidden, const float* gate, const float* pool,
                                  const float* W1, const float* W2, const float* Wl,
                                  const int64_t* labels, const float* gate_mask, const float* weight_mask,
                                  int64_t N, int64_t way, int64_t shot, int64_t K, int64_t C, int64_t P,
                                  float* dW1, float* db1, float* dW2, float* db2, float* dWl, float* dbl,
                                  float* dweight /*[K,C]*/, float* coef /*[B,C]*/, void* stream);

/* ---- K3: per-pixel classifier (MFMA) + binary-logit extraction ----------
 * Replaces WeightEstimator.infer, training branch (model/common.py:390-394):
 *   logits[n,k,p] = sum_c Fq[n,c,p] * Wn[n,k,c]          (plain dot product)
 * and get_binary_logits(base=True) (utils/operations.py:313-338) per way:
 *   bin[n*way+j, 1, p] = logits[n, labels[n,j], p]
 *   bin[n*way+j, 0, p] = max_{k != labels[n,j]} logits[n,k,p]
 * logits_binary may be NULL.  base=0 selects the base=False variant (row 0 as
 * background, utils/operations.py:334-336).
 */
int denet_classify_fwd(const void* Fq, int feat_dtype, const float* Wn, const int64_t* labels,
                       int64_t N, int64_t way, int64_t K, int64_t C, int64_t P, int base,
                       float* logits /*[N,K,P]*/, float* logits_binary /*[N*way,2,P]*/,
                       void* stream);
/* Shared-matrix variant for mode='deploy' (model/common.py:395-399): Wn is
 * [K,C] and used for every n. */
int denet_classify_shared_fwd(const void* Fq, int feat_dtype, const float* W,
                              int64_t N, int64_t K, int64_t C, int64_t P,
                              float* logits, void* stream);
/* get_binary_logits on an existing logits tensor (utils/operations.py:313-338), same
 * semantics as the epilogue of denet_classify_fwd.  labels [N,way] -> bin [N*way,2,P]. */
int denet_binary_logits(const float* logits, const int64_t* labels, int64_t N, int64_t way, int64_t K,
                        int64_t P, int base, float* logits_binary, void* stream);
/* out[i] = mean_n x[n,i]: the batch mean of extended weights of mode='deploy'
 * (model/common.py:369-372). */
int denet_batch_mean(const float* x, int64_t N, int64_t per, float* out, void* stream);

/* Backward of K3 (autograd of model/common.py:392-393):
 *   dL = scale[n] * (dlog_a + dlog_b)     (the two slabs K4 writes; dlog_b may be NULL,
 *                                          scale may be NULL = 1)
 *   dFq[n,c,p]  = sum_k Wn[n,k,c] * dL[n,k,p]
 *   dWn_part[n,t,k,c] = sum_{p in tile t} dL[n,k,p] * Fq[n,c,p]
 * n_part = denet_classify_bwd_parts(P) tiles. */
int64_t denet_classify_bwd_parts(int64_t P);
int denet_classify_bwd(const void* Fq, int feat_dtype, const float* Wn,
                       const float* dlog_a, const float* dlog_b, const float* scale,
                       const float* gmul /* device scalar multiplied into dL, or NULL */,
                       int64_t N, int64_t K, int64_t C, int64_t P,
                       void* dFq /* may be NULL */, float* dWn_part /* may be NULL */, void* stream);

/* dWn[n,:] = sum_t dWn_part[n,t,:]: the tensor form of the classifier's weight gradient, for
 * callers that run WeightEstimator.extend and .infer as separate autograd nodes. per_n = K*C. */
int denet_sum_parts(const float* part, int64_t N, int64_t n_part, int64_t per_n, float* out, void* stream);

/* ---- K4/K5: fused bilinear up-sampling + losses --------------------------
 * Replaces F.interpolate(bilinear, align_corners=True) of both logit tensors
 * (trainer/trainers.py:954-956), the per-episode weighted cross entropy
 * (trainer/utils.py:270-277 -> loss/losses.py:19-44) and the binary focal loss
 * (loss/losses.py:47-79) without materialising the [N,K,H,W] tensor.
 *   up = bilinear(logits), src = dst*(in-1)/(out-1)
 *   num[n] = sum_q v*wgt[n,y]*(-log softmax(up)[y]);  den[n] = sum_q v*wgt[n,y]
 *   v = (y != 255)
 * With dlog_a/dlog_b non-NULL it also writes the UNSCALED gradient of
 * sum_n num[n] w.r.t. the low-res logits as two slabs (rows touched as the
 * upper / the lower bilinear tap), to be combined as scale[n]*(a+b) with
 * scale[n] = 1/(N*den[n])  (loss_k is a mean over episodes of per-episode
 * weighted means).
 * Focal: Yq [NB,H,W], NB = N*way.
 * part: [N, parts, 2] floats of per-block partial (num, den) sums, with
 * parts = denet_upsample_loss_parts(K, h, w, H, W, with_grad) (K = 2, with_grad = 0 for focal).
 * denet_loss_finalize reduces them: out[0]=loss_k, out[1]=loss_p.
 *   pooled = 0: loss_k = mean_n num[n]/den[n], scale[n] = 1/(N*den[n])   (trainer/utils.py:270-277)
 *   pooled = 1: loss_k = sum num / sum den,    scale[n] = 1/sum den      (loss/losses.py:39-44 on a batch)
 * denet_combine_slabs: out = gmul * scale[n] * (a + b): the dense gradient
 * w.r.t. the logits for callers that need it as a tensor.
 */
int64_t denet_upsample_loss_parts(int64_t K, int64_t h, int64_t w, int64_t H, int64_t W, int with_grad);
int denet_upsample_wce_fwd(const float* logits, const int64_t* Yq_full, const float* wgt /*[N,K] or NULL*/,
                           int64_t N, int64_t K, int64_t h, int64_t w, int64_t H, int64_t W,
                           float* part /*[N,parts,2]*/, float* dlog_a, float* dlog_b, void* stream);
int denet_upsample_focal_fwd(const float* logits_binary, const int64_t* Yq,
                             int64_t NB, int64_t h, int64_t w, int64_t H, int64_t W,
                             float alpha, float gamma, float* part /*[NB,parts,2]*/, void* stream);
int denet_loss_finalize(const float* part_k, const float* part_p, int64_t N, int64_t NB,
                        int64_t parts_k, int64_t parts_p, int pooled, float* out /*[2]*/, float* per_episode /*[N] or NULL*/,
                        float* scale /*[N] or NULL*/, void* stream);
int denet_combine_slabs(const float* a, const float* b /*or NULL*/, const float* scale /*[N] or NULL*/,
                        const float* gmul /*device scalar or NULL*/, int64_t N, int64_t per /*K*h*w*/,
                        float* out, void* stream);
/* Plain up-sampling for callers that need the tensor (DENetTrainer.step returns
 * it, trainer/trainers.py:1003-1005). */
int denet_upsample_bilinear(const float* x, int64_t NC, int64_t h, int64_t w, int64_t H, int64_t W,
                            float* out /*[NC,H,W]*/, void* stream);

/* ---- M1/M2: metrics -------------------------------------------------------
 * FullIOU.update (utils/metrics.py:59-67): mat[t,argmax_k pred] += 1 for
 * 0 <= t < K.  mat is int64 [K,K], accumulated in place (exact integer
 * counts; first index wins ties like torch.argmax).
 *   _confmat       : API-level, preds already up-sampled [N,K,H,W]
 *   _confmat_fused : from low-res logits [N,K,h,w], up-sampling on the fly
 * BinaryIOU.update (utils/metrics.py:24-31): per sample
 *   counts[i] = {sum(true), #(true+pred==2), #(true+pred in {1,2})} int64
 * (the host applies the skip rule and the per-class dictionaries). */
int denet_confmat(const float* preds, const int64_t* label_true, int64_t N, int64_t K, int64_t HW,
                  int64_t* mat, void* stream);
int denet_confmat_fused(const float* logits, const int64_t* label_true, int64_t N, int64_t K,
                        int64_t h, int64_t w, int64_t H, int64_t W, int64_t* mat, void* stream);
int denet_binary_iou_counts(const float* preds2 /*[N,2,HW]*/, const int64_t* label_true,
                            int64_t N, int64_t HW, int64_t* counts /*[N,3]*/, void* stream);
int denet_binary_iou_counts_fused(const float* logits2 /*[N,2,h,w]*/, const int64_t* label_true,
                                  int64_t N, int64_t h, int64_t w, int64_t H, int64_t W,
                                  int64_t* counts /*[N,3]*/, void* stream);

/* ---- T1: the whole head step as two enqueue-only calls -----------------------
 * Everything DENetTrainer.step does between the feature producer and
 * optimizer.step() (trainer/trainers.py:951-975; wrappers trainer/utils.py:
 * 251-303): K0..K5 forward (+ the loss gradient slabs when with_grad), then
 * K3b, K2b, K1b, as three launches whose independent halves run side by side (B1: dWn slabs ||
 * dFq; B2: middle of the sliced K2b chain || dweight; B3: dW1, dW2, dWl, biases || coef + dFs).
 * The gate / projection mat-vecs of K2 and K2b are cut into 16-row slices that run on different
 * CUs; the exchanges between them are the launch boundaries (csrc/dev_chain_sliced.hpp).
 * with_grad = DENET_GRAD_DEFER_LOSSES
 * moves the reduction of the loss partials into B1 (losses[] is then valid after the backward
 * call, and the forward needs Yq_full).  ws: denet_head_step_workspace(...) bytes, 256-B aligned,
 * shared by the two calls of one step.  Yq / Yq_full may be NULL (test mode:
 * no loss, trainer/trainers.py:958).  losses = {loss_k, loss_p} on device.
 * grad_loss_k: device scalar d(objective)/d(loss_k), NULL = 1.  Only loss_k is
 * back-propagated (trainer/trainers.py:972-974).
 *
 * probe (HOST pointer, may be NULL): brackets ONE kernel of the step with the
 * caller's HIP events, recorded on `stream`, so a harness can time that kernel
 * inside its own timed region (bench.py's roofline line). */
#define DENET_GRAD_DEFER_LOSSES  2   /* with_grad value: losses[] is written by denet_head_step_bwd */
#define DENET_K_MASK_RESIZE      1   /* (stand-alone K0 only; the fused step gathers the mask inside K1) */
#define DENET_K_MASKED_SUM_FWD   2   /* K0+K1: mask gather + masked channel sums                 */
#define DENET_K_PROTO_EXTEND_FWD 3
#define DENET_K_CLASSIFY_FWD     4
#define DENET_K_WCE_FWD          5   /* K4 (+K5 in the same launch when both label maps are given) */
#define DENET_K_FOCAL_FWD        6   /* K5 alone (no Yq_full)                                      */
#define DENET_K_LOSS_FINALIZE    7
#define DENET_K_CLASSIFY_BWD     8   /* B1: dWn slabs || dFq (+ deferred loss reduction)           */
#define DENET_K_PE_BWD_CHAIN     9   /* B2: sliced K2b chain, middle || dweight                    */
#define DENET_K_MASKED_SUM_BWD   10  /* B3: dW1, dW2, dWl, biases || coef + dFs = mask*coef        */
/* denet_head_step_bwd flags.  The dFs blocks are the last consumer of the step and nothing reads
 * their output, so a data-parallel caller may run the backward in two calls and start the
 * all-reduce of the parameter gradients between them (it then overlaps with the 70 MB dFs stream):
 * first NO_DFS (every gradient except dFs), then ONLY_DFS with the same arguments. */
#define DENET_BWD_HAVE_FOCAL     1   /* the forward was given Yq                                   */
#define DENET_BWD_NO_DFS         2   /* everything except the dFs blocks                           */
#define DENET_BWD_ONLY_DFS       4   /* only the dFs blocks (after a NO_DFS call of the same step)  */
typedef struct denet_probe {
    int32_t which;      /* DENET_K_* */
    int32_t reserved;
    void* start;        /* hipEvent_t */
    void* stop;         /* hipEvent_t */
} denet_probe_t;

int64_t denet_head_step_workspace(int64_t N, int64_t way, int64_t shot, int64_t K, int64_t C,
                                  int64_t h, int64_t w, int64_t H, int64_t W);

/* Which kernel variant each launch of the step runs for a shape.  The launchers pick variants from the
 * call's sizes alone (cache policy of the once-touched streams, classifier layout, loss-kernel path); this
 * query runs the SAME decision functions and reports the outcome, so a test can assert that the shape it
 * checks against the oracle exercises the variant a benchmark times.  No counterpart in the reference (one
 * ATen op chain for every size).  `out` is a HOST pointer; nothing is enqueued. */
#define DENET_CLS_TILE8   0   /* eight waves per 64-position tile (K <= 32, < 384 tiles in the batch) */
#define DENET_CLS_UNITS   1   /* K > 32: (half tile, row tile) units dealt to eight waves             */
#define DENET_CLS_WAVE4   2   /* a wave per tile, four tiles per block                                 */
#define DENET_CLS_WAVE4_BF16 3 /* bf16 features, 32 < K <= 96: the same block shape on the bf16 matrix pipe,
                                  class weights split into three bf16 planes (exact)                    */
typedef struct denet_step_plan {
    int32_t pool_launches;    /* 1: mask gathered inside the pooling kernel; 2: flat mask resize first       */
    int32_t pool_long_rows;   /* rolling-loop pooling kernel (rows of more than 512 16-byte vectors)         */
    int32_t pool_nt_loads;    /* Fs read with non-temporal loads (stream > 160 MB)                           */
    int32_t classify_layout;  /* DENET_CLS_*                                                                 */
    int32_t classify_groups;  /* DENET_CLS_UNITS: blocks per episode, else 0                                 */
    int32_t loss_chain_cell;  /* 8 / 9: chain kernel with that many label columns per cell; 0: exact kernel  */
    int32_t loss_chunk;       /* chain kernel: classes per table chunk                                       */
    int32_t loss_xsplit;      /* column ranges per low-res row                                               */
    int32_t bwd_persistent;   /* B1 runs the persistent-block kernel                                         */
    int32_t bwd_groups;       /* dWn slabs per episode written by B1                                         */
    int32_t bwd_dfq_nt;       /* dFq leaves B1 with non-temporal stores                                      */
    int32_t dfs_nt;           /* dFs leaves B3 with non-temporal stores                                      */
    int32_t dfs_flat_writer;  /* fp32 + dfs_nt: B3 writes the block's 16 rows as one span instead of a row per wave */
    int32_t dfs_chunked;      /* bf16 + dfs_nt: B3's blocks only form coefficients, a second launch writes the flat
                                 tensor one contiguous chunk per block, in address order                              */
    int32_t bwd_bf16_pipe;    /* 32 < K <= 96: B1 on the bf16 matrix pipe (dL, and fp32 Fq / Wn, as exact bf16 planes)   */
    int32_t bwd_xcd_pairs;    /* B1's grid pairs the two roles of a tile group on one XCD (one-round grids only)        */
} denet_step_plan_t;
int denet_head_step_plan(int64_t N, int64_t way, int64_t shot, int64_t K, int64_t C, int64_t h, int64_t w,
                         int64_t H, int64_t W, int feat_dtype, denet_step_plan_t* out);
int denet_head_step_fwd(const void* Fq, const void* Fs, int feat_dtype, const int64_t* Ys,
                        const int64_t* labels, const int64_t* Yq, const int64_t* Yq_full,
                        const float* wgt, const float* weight, const float* W1, const float* b1,
                        const float* W2, const float* b2, const float* Wl, const float* bl,
                        const float* gate_mask, const float* weight_mask,
                        int64_t N, int64_t way, int64_t shot, int64_t K, int64_t C,
                        int64_t h, int64_t w, int64_t H, int64_t W, int with_grad,
                        float* logits_full, float* logits_binary, float* losses,
                        void* ws, int64_t ws_bytes, const denet_probe_t* probe, void* stream);
int denet_head_step_bwd(const void* Fq, int feat_dtype, const int64_t* labels,
                        const float* W1, const float* W2, const float* Wl,
                        const float* gate_mask, const float* weight_mask, const float* grad_loss_k,
                        int64_t N, int64_t way, int64_t shot, int64_t K, int64_t C, int64_t h, int64_t w,
                        int64_t H, int64_t W, void* dFq, void* dFs, float* dweight, float* dW1, float* db1, float* dW2,
                        float* db2, float* dWl, float* dbl,
                        float* losses /* [2]; NULL unless the forward ran with DENET_GRAD_DEFER_LOSSES */,
                        int flags /* DENET_BWD_* */, void* ws, int64_t ws_bytes,
                        const denet_probe_t* probe, void* stream);

/* ---- T1 in train mode: dropout generated inside the kernels --------------------
 * WeightEstimator in train() mode (model/common.py:231 `nn.Dropout` of the gate, :326
 * `self.dropout(self.weight)` per episode; the trainer calls model.train() at
 * trainer/trainers.py:938).  The reference draws the masks from torch's generator; here the
 * multipliers (0 with probability p, else 1/(1-p)) are a pure function of (seed, stream id,
 * element index): Philox4x32-10, key = seed, counter = (index / 4, stream id, 0, 0), drop iff
 * the 32-bit output < round(p * 2^32).  The forward applies them in F2 (gate) and while F3
 * assembles Wn; the backward regenerates them from the same `drop`, so nothing but the seed is
 * kept between the calls.  The caller advances `seed` every step.  A non-NULL mask pointer wins
 * over the generator (injected masks: parity tests); `drop` NULL and both pointers NULL = eval
 * mode (= denet_head_step_fwd / _bwd, which are these calls with drop = NULL).
 * p in [0, 1); p = 0 leaves that site without dropout.
 * denet_dropout_mask writes the multipliers of one stream as a tensor (tests; callers of the
 * stand-alone denet_proto_extend_fwd / _bwd who want the same masks as pointers). */
#define DENET_DROPOUT_GATE    1   /* element (b, c) of the gate pre-activation, [N*way*shot][C] */
#define DENET_DROPOUT_WEIGHT  2   /* element (n, k, c) of the episode's class weights, [N][K][C] */
typedef struct denet_dropout {
    uint64_t seed;
    float p_gate;       /* model/common.py:231, 0.5 in the reference */
    float p_weight;     /* model/common.py:274 drop_rate, 0.5 in the reference */
} denet_dropout_t;
int denet_dropout_mask(uint64_t seed, int stream_id, float p, int64_t n, float* out, void* stream);
int denet_head_step_train_fwd(const void* Fq, const void* Fs, int feat_dtype, const int64_t* Ys,
                        const int64_t* labels, const int64_t* Yq, const int64_t* Yq_full,
                        const float* wgt, const float* weight, const float* W1, const float* b1,
                        const float* W2, const float* b2, const float* Wl, const float* bl,
                        const float* gate_mask, const float* weight_mask,
                        const denet_dropout_t* drop /* HOST pointer, read during the call */,
                        int64_t N, int64_t way, int64_t shot, int64_t K, int64_t C,
                        int64_t h, int64_t w, int64_t H, int64_t W, int with_grad,
                        float* logits_full, float* logits_binary, float* losses,
                        void* ws, int64_t ws_bytes, const denet_probe_t* probe, void* stream);
int denet_head_step_train_bwd(const void* Fq, int feat_dtype, const int64_t* labels,
                        const float* W1, const float* W2, const float* Wl,
                        const float* gate_mask, const float* weight_mask,
                        const denet_dropout_t* drop /* the forward's */, const float* grad_loss_k,
                        int64_t N, int64_t way, int64_t shot, int64_t K, int64_t C, int64_t h, int64_t w,
                        int64_t H, int64_t W, void* dFq, void* dFs, float* dweight, float* dW1, float* db1, float* dW2,
                        float* db2, float* dWl, float* dbl, float* losses, int flags,
                        void* ws, int64_t ws_bytes, const denet_probe_t* probe, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* DENET_HIP_H */
