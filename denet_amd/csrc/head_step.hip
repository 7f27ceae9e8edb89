// The whole head as two enqueue-only entry points: everything
// DENetTrainer.step does between the feature producer and optimizer.step()
// (trainer/trainers.py:951-975), i.e. K0..K5 forward and K4b..K1b backward, on
// one stream, with a caller-provided workspace and no host synchronisation
// (the reference syncs N*way + 3 times per step: SURVEY.md §1).
#include <type_traits>

#include "denet_common.hpp"
#include "dev_chain_sliced.hpp"
#include "dev_classify.hpp"
#include "dev_masked_sum.hpp"
#include "dev_proto_extend.hpp"
#include "dev_upsample_loss.hpp"

struct HeadWs {
    float *mask, *cnt, *S, *hidden, *gate, *pool, *Wn, *part_k, *part_p, *scale, *dlog_a, *dlog_b, *dWn_part,
        *pre1_part, *proto_part, *dh_part, *dpool, *u2, *u3, *coef;
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
    TAKE(pre1_part, B * (C / 16) * C);       // sliced chain: split-K partials (dev_chain_sliced.hpp)
    TAKE(proto_part, NB * (C / 16) * C);
    TAKE(dh_part, B * (C / 16) * C);
    TAKE(dpool, NB * C);
    TAKE(u2, B * C);
    TAKE(u3, B * C);
    TAKE(coef, B * C);                       // large batches: coefficients of the chunked dFs writer
#undef TAKE
    s.total = off * 4;
    return s;
}

// ---------------------------------------------------------------------------
// Fused launches of the backward half.  At batch 8 every kernel of the step is latency-bound
// (5-30 us for a few MB), so the step time is the LENGTH of the dependency chain; independent
// pieces therefore share a launch, block roles picked by blockIdx:
//   B1  dWn slabs (K3b, first half)  ||  dFq (K3b, second half)   [+ the loss reduction, in block (0,0)]
//   B2  sliced chain, middle: dproto -> dpool -> dz -> split-K partials of W2^T dz  ||  dweight
//   B3  dW1, dW2, dWl and the bias gradients  ||  coef + dFs = mask * coef (K1b)
// ---------------------------------------------------------------------------
#define B1_NW 8           // waves per block of the first backward launch (one 32-channel tile per wave)
template <typename T, int RT>
__global__ __launch_bounds__(64 * B1_NW) void head_bwd_dwn_kernel(ClsBwdArgs g, ClsBwdArgs g2, const float* __restrict__ part_p,
                                                          int NB, int parts_p, float* __restrict__ losses) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    const int n_part = g.n_part;
    if ((int)blockIdx.x < n_part) {
        classify_bwd_body<T, RT, B1_NW>(g, blockIdx.y, blockIdx.x, sm);
        if (losses != nullptr && blockIdx.x == 0 && blockIdx.y == 0) {
            __syncthreads();
            loss_finalize_body(g.part_k, part_p, g.N, NB, g.parts_k, parts_p, 0, losses, nullptr, nullptr, sm);
        }
    } else {
        classify_bwd_body<T, RT, B1_NW>(g2, blockIdx.y, blockIdx.x - n_part, sm);
    }
}

// B1's persistent launches: the two roles of a tile group read the same dL tiles at the same time.  Workgroups go
// to the 8 XCDs round-robin by linear id, so the (1-D) grid is laid out in runs of 8 dWn blocks followed by the 8
// dFq blocks of the SAME (episode, group) pairs: both roles of a group share an XCD and the second reader of a dL
// tile finds it in that XCD's L2.  The pairs are numbered through all episodes, so only the last run is padded
// (a per-episode layout left two XCDs idle at C3's six groups: B1 37.7 -> 60.5 us).
// Used while all blocks are resident at once (one round: both roles of a pair then run side by side; C3 B1 37.4 ->
// 35.6 us, C5 41.4 -> 40.4, C2 14.8 -> 14.7); with several rounds of blocks it measured slower (C2 x 128: step
// 1.000 vs 0.993 ms).
#ifndef DENET_B1_XCD_PAIRS
#define DENET_B1_XCD_PAIRS 1
#endif
static inline bool b1_paired(int ngrp, int N) { return DENET_B1_XCD_PAIRS && 2 * N * ngrp <= 320; }
static inline dim3 b1_grid(int ngrp, int N) {
    if (b1_paired(ngrp, N)) return dim3((unsigned)(((N * ngrp + 7) / 8) * 16));
    return dim3((unsigned)(2 * ngrp), (unsigned)N);
}
__device__ __forceinline__ bool b1_role(int paired, int ngrp, int N, int& n, int& grp, bool& dwn) {
    const int x = blockIdx.x;
    if (paired) {
        const int L = (x >> 4) * 8 + (x & 7);
        dwn = ((x >> 3) & 1) == 0;
        n = L / ngrp;
        grp = L - n * ngrp;
        return L < N * ngrp;
    }
    dwn = x < ngrp;
    grp = dwn ? x : x - ngrp;
    n = blockIdx.y;
    return true;
}

// K > 32: persistent blocks (dev_classify.hpp, classify_bwd_wide_body); g.n_part = groups per episode
template <typename T, int RT, bool NTQ>
__global__ __launch_bounds__(64 * CLW_BWD_NW) void head_bwd_dwn_wide_kernel(ClsBwdArgs g, ClsBwdArgs g2,
                                                                            const float* __restrict__ part_p, int NB,
                                                                            int parts_p, float* __restrict__ losses,
                                                                            int paired) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    int n, grp;
    bool dwn;
    if (!b1_role(paired, g.n_part, g.N, n, grp, dwn)) return;
    if (dwn) {
        classify_bwd_wide_body<T, RT, true>(g, n, grp, sm);
        if (losses != nullptr && grp == 0 && n == 0) {
            __syncthreads();
            loss_finalize_body(g.part_k, part_p, g.N, NB, g.parts_k, parts_p, 0, losses, nullptr, nullptr, sm);
        }
    } else {
        classify_bwd_wide_body<T, RT, false, NTQ>(g2, n, grp, sm);
    }
}

// bf16 features, 32 < K <= 96: the same blocks on the bf16 matrix pipe (dev_classify.hpp, classify_bwd_wide_bf16_body)
template <typename T, int RT, bool NTQ>
__global__ __launch_bounds__(64 * CLW_BWD_NW) void head_bwd_dwn_wide_bf16_kernel(ClsBwdArgs g, ClsBwdArgs g2,
                                                                                 const float* __restrict__ part_p, int NB,
                                                                                 int parts_p, float* __restrict__ losses,
                                                                                 int paired) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    int n, grp;
    bool dwn;
    if (!b1_role(paired, g.n_part, g.N, n, grp, dwn)) return;
    if (dwn) {
        classify_bwd_wide_bf16_body<T, RT, true>(g, n, grp, sm);
        if (losses != nullptr && grp == 0 && n == 0) {
            __syncthreads();
            loss_finalize_body(g.part_k, part_p, g.N, NB, g.parts_k, parts_p, 0, losses, nullptr, nullptr, sm);
        }
    } else {
        classify_bwd_wide_bf16_body<T, RT, false, NTQ>(g2, n, grp, sm);
    }
}

template <int CJ, int NS>
__global__ __launch_bounds__(SC_THREADS) void head_bwd_mid_kernel(ChainMidArgs a, PeParamsArgs pa, int nchain) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    constexpr int NSL = 64 * CJ / SC_SL;
    if ((int)blockIdx.x < nchain)
        chain_mid_bwd_body<CJ, NS>(a, blockIdx.x / NSL, blockIdx.x % NSL, sm);
    else
        proto_extend_bwd_params_body(pa, 3, blockIdx.x - nchain, sm);     // dweight: needs only B1's slabs
}

template <typename T, int NSLT, bool NT>
__global__ __launch_bounds__(256) void head_bwd_tail_kernel(PeParamsArgs pa, CoefArgs ca, const float* __restrict__ mask,
                                                           T* __restrict__ dFs, int C, int P, int rpb, int ncb, int gx,
                                                           int pfirst) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    // pfirst: the parameter-gradient blocks come FIRST in the grid.  They loop over all B support images and,
    // dispatched behind the dFs blocks of a long stream, they start when it is nearly written and run on past its
    // end.  Step time first / last: 640 images (C2 x 128) 979 / 991 us, 200 (C3) 319.6 / 320.8, 160 (C2 x 32)
    // 247.0 / 243.5, 80 (C2 x 16) 147.8 / 143.3, 40 (C2) 88.1 / 86.1 -- first pays only where the loops are long, so
    // the launcher asks for it from 320 images on.
    const int bx = blockIdx.x, npar = 3 * gx, nms = gridDim.x - npar;
    const bool par = pfirst ? bx < npar : bx >= nms;
    if (!par) {
        const int d = pfirst ? bx - npar : bx;
        masked_sum_bwd_body<T, NSLT, NT>(mask, nullptr, dFs, C, P, rpb, d / ncb, d % ncb, sm, ca);
    } else {
        const int r = pfirst ? bx : bx - nms;
        proto_extend_bwd_params_body(pa, r / gx, r % gx, sm);
    }
}

#ifndef DENET_TAIL_PARAMS_FIRST_B
#define DENET_TAIL_PARAMS_FIRST_B 320   // B3: parameter-gradient blocks first from this many support images on (see the kernel)
#endif
#ifndef DENET_DFS_NT_MB
#define DENET_DFS_NT_MB 80          // dFs streams above this size leave with non-temporal stores (see B3 below)
#endif
#ifndef DENET_DFQ_NT_MB
#define DENET_DFQ_NT_MB 80          // dFq streams UP TO this size leave B1 with non-temporal stores (dev_classify.hpp, NTQ)
#endif
// the backward launchers' choices, from the call's sizes alone (also reported by denet_head_step_plan)
static inline bool bwd_persistent(int64_t P) { return P >= 4; }       // quad loads need P >= 4
#ifndef DENET_BWD_BF16_PIPE
#define DENET_BWD_BF16_PIPE 1
#endif
// B1 on the bf16 matrix pipe (dev_classify.hpp, classify_bwd_wide_bf16_body) for 32 < K <= 96, both feature types
// (its 8-position Fq fragments need rows of >= 8).  K <= 32 stays on the fp32 instruction: there the operand
// splits cost more than the cheaper MFMAs save (fp32 features, C2's shape: step 87.1 vs 86.3 us at 8 episodes,
// 246 vs 242 at 32, 1025 vs 991 at 128).
static inline bool bwd_bf16_pipe(int64_t K, int64_t P, size_t sF) {
    return DENET_BWD_BF16_PIPE && K > 32 && K <= 96 && P >= 8;
}
static inline bool bwd_dfq_nt(int64_t N, int64_t C, int64_t P, size_t sF) {
    return N * C * P * (int64_t)sF <= ((int64_t)DENET_DFQ_NT_MB << 20);
}
// dFs streams go around the caches (dev_masked_sum.hpp, NT) unless the whole step's working set stays
// resident in the 256 MB last-level cache.  Measured (C2 shape, fp32, step time cached vs NT): 69 MB
// 86.9 vs 90.7 us, 103 MB 127.8 vs 125.3, 138 MB 161.5 vs 144.2, 207 MB 221.6 vs 204.9, 275 MB 287.1 vs
// 260.1; bf16 at C5's shape: 67 MB 158.7 vs 156.4, 134 MB 250.1 vs 238.0, 269 MB 427.1 vs 418.6.
static inline bool dfs_nt(int64_t B, int64_t C, int64_t P, size_t sF) {
    return B * C * P * (int64_t)sF > ((int64_t)DENET_DFS_NT_MB << 20);
}
#ifndef DENET_DFS_CHUNKED
#define DENET_DFS_CHUNKED 1
#endif
static inline bool dfs_chunked(bool nt, bool f32) { return DENET_DFS_CHUNKED && nt && !f32; }
template <typename T, int RT>
static int launch_bwd_dwn(const ClsBwdArgs& g, const ClsBwdArgs& g2, const float* part_p, int NB, int parts_p,
                          float* losses, int n_part, int N, bool wide, hipStream_t st) {
    if constexpr (RT == 2 || RT == 3) {
        if (wide && bwd_bf16_pipe(g.K, g.P, sizeof(T))) {
            // (the finalize body of block (0, 0) reuses the LDS: both allocations are far above what it needs)
            const size_t ldb = CLWB_LDS(RT) > 4096 ? CLWB_LDS(RT) : 4096;
            if (bwd_dfq_nt(N, g.C, g.P, sizeof(T))) {
                if (denet_ensure_dynamic_lds(reinterpret_cast<const void*>(head_bwd_dwn_wide_bf16_kernel<T, RT, true>), ldb))
                    return DENET_E_LAUNCH;
                hipLaunchKernelGGL((head_bwd_dwn_wide_bf16_kernel<T, RT, true>), b1_grid(n_part, N),
                                   dim3(64 * CLW_BWD_NW), ldb, st, g, g2, part_p, NB, parts_p, losses, (int)b1_paired(n_part, N));
            } else {
                if (denet_ensure_dynamic_lds(reinterpret_cast<const void*>(head_bwd_dwn_wide_bf16_kernel<T, RT, false>), ldb))
                    return DENET_E_LAUNCH;
                hipLaunchKernelGGL((head_bwd_dwn_wide_bf16_kernel<T, RT, false>), b1_grid(n_part, N),
                                   dim3(64 * CLW_BWD_NW), ldb, st, g, g2, part_p, NB, parts_p, losses, (int)b1_paired(n_part, N));
            }
            return DENET_OK;
        }
    }
    if (wide) {                         // persistent blocks (quad loads need P >= 4; see cls_bwd_parts below)
        const size_t ldw = CLW_BWD_LDS(RT);
        if (bwd_dfq_nt(N, g.C, g.P, sizeof(T))) {
            if (denet_ensure_dynamic_lds(reinterpret_cast<const void*>(head_bwd_dwn_wide_kernel<T, RT, true>), ldw))
                return DENET_E_LAUNCH;
            hipLaunchKernelGGL((head_bwd_dwn_wide_kernel<T, RT, true>), b1_grid(n_part, N),
                               dim3(64 * CLW_BWD_NW), ldw, st, g, g2, part_p, NB, parts_p, losses, (int)b1_paired(n_part, N));
        } else {
            if (denet_ensure_dynamic_lds(reinterpret_cast<const void*>(head_bwd_dwn_wide_kernel<T, RT, false>), ldw))
                return DENET_E_LAUNCH;
            hipLaunchKernelGGL((head_bwd_dwn_wide_kernel<T, RT, false>), b1_grid(n_part, N),
                               dim3(64 * CLW_BWD_NW), ldw, st, g, g2, part_p, NB, parts_p, losses, (int)b1_paired(n_part, N));
        }
        return DENET_OK;
    }
    const size_t lds = CLS_BWD_LDS_NW(g.K, B1_NW);
    if (denet_ensure_dynamic_lds(reinterpret_cast<const void*>(head_bwd_dwn_kernel<T, RT>), lds)) return DENET_E_LAUNCH;
    hipLaunchKernelGGL((head_bwd_dwn_kernel<T, RT>), dim3((unsigned)(2 * n_part), (unsigned)N), dim3(64 * B1_NW), lds,
                       st, g, g2, part_p, NB, parts_p, losses);
    return DENET_OK;
}

static int launch_bwd_mid(const ChainMidArgs& a, const PeParamsArgs& pa, int groups, int C, int shot, hipStream_t st) {
    const int nsl = C / SC_SL, nchain = groups * nsl;
    dim3 grid((unsigned)(nchain + pa.K * (C >> 4))), block(SC_THREADS);
    const size_t lds = SC_CM_LDS > 4096 ? SC_CM_LDS : 4096;
#define CM_GO(CJ, NS) hipLaunchKernelGGL((head_bwd_mid_kernel<CJ, NS>), grid, block, lds, st, a, pa, nchain)
#define CM_NS(CJ)                     \
    if (shot == 1) CM_GO(CJ, 1);      \
    else if (shot <= 3) CM_GO(CJ, 3); \
    else if (shot <= 5) CM_GO(CJ, 5); \
    else CM_GO(CJ, 8)
    if (C == 256) { CM_NS(4); }
    else if (C == 64) { CM_NS(1); }
    else return DENET_E_UNSUPPORTED;
#undef CM_NS
#undef CM_GO
    return DENET_OK;
}

static int launch_gate_proj(const GateProjArgs& a, int groups, int C, int shot, hipStream_t st) {
    const int nsl = C / SC_SL;
    dim3 grid((unsigned)(groups * nsl)), block(SC_THREADS);
    const size_t lds = SC_GP_LDS(C);
#define GP_GO(CJ, NS) hipLaunchKernelGGL((gate_proj_kernel<CJ, NS>), grid, block, lds, st, a)
#define GP_NS(CJ)                     \
    if (shot == 1) GP_GO(CJ, 1);      \
    else if (shot <= 3) GP_GO(CJ, 3); \
    else if (shot <= 5) GP_GO(CJ, 5); \
    else GP_GO(CJ, 8)
    if (C == 256) { GP_NS(4); }
    else if (C == 64) { GP_NS(1); }
    else return DENET_E_UNSUPPORTED;
#undef GP_NS
#undef GP_GO
    DENET_CHECK_LAUNCH();
    return DENET_OK;
}

// internal launchers of the sliced chain's forward neighbours (masked_sum.hip, classify.hip)
int denet_masked_pool_fwd_chain(const void* Fs, int feat_dtype, const int64_t* Ys, int64_t B, int64_t C, int64_t H,
                                int64_t W, int64_t h, int64_t w, float* mask, float* cnt, float* S, const float* W1,
                                float* pre1_part, void* stream);
int denet_classify_build_fwd(const void* Fq, int feat_dtype, const float* weight, const float* weight_mask,
                             const float* proto_part, const float* bl, const int64_t* labels, int64_t N, int64_t way,
                             int64_t K, int64_t C, int64_t P, float* Wn_out, float* logits, float* logits_binary,
                             const DropGen* wdrop, void* stream);

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

extern "C" int denet_head_step_plan(int64_t N, int64_t way, int64_t shot, int64_t K, int64_t C, int64_t h, int64_t w,
                                    int64_t H, int64_t W, int feat_dtype, denet_step_plan_t* out) {
    if (!out) return DENET_E_BADARG;
    if (N <= 0 || way <= 0 || shot <= 0 || K <= 0 || C <= 0 || h <= 0 || w <= 0 || H <= 0 || W <= 0)
        return DENET_E_BADARG;
    if (!(C == 64 || C == 256) || K > 128 || shot > 8) return DENET_E_UNSUPPORTED;
    if (feat_dtype != DENET_F32 && feat_dtype != DENET_BF16) return DENET_E_UNSUPPORTED;
    const int64_t B = N * way * shot, P = h * w;
    const size_t sF = feat_dtype == DENET_F32 ? 4 : 2;
    const int RT = (int)((K + 31) / 32);
    denet_step_plan_t p{};
    const PoolPlan pl = denet_pool_plan(feat_dtype, B, C, P, true);
    p.pool_launches = pl.two_launch ? 2 : 1;
    p.pool_long_rows = pl.long_rows;
    p.pool_nt_loads = pl.nt_loads;
    int64_t groups = 0;
    p.classify_layout = denet_classify_fwd_layout(feat_dtype, N, K, P, &groups);
    p.classify_groups = p.classify_layout == DENET_CLS_UNITS ? (int32_t)groups : 0;
    LossPlan lp;
    if (!denet_loss_plan(K, h, w, H, W, true, &lp)) return DENET_E_UNSUPPORTED;
    p.loss_chain_cell = lp.chain_cell;
    p.loss_chunk = lp.chunk;
    p.loss_xsplit = lp.xsplit;
    p.bwd_persistent = bwd_persistent(P);
    p.bwd_groups = p.bwd_persistent ? cls_bwd_wide_groups(N, P, RT) : (int)denet_classify_bwd_parts(P);
    p.bwd_dfq_nt = p.bwd_persistent && bwd_dfq_nt(N, C, P, sF);
    p.bwd_bf16_pipe = p.bwd_persistent && bwd_bf16_pipe(K, P, sF);
    p.bwd_xcd_pairs = p.bwd_persistent && b1_paired(p.bwd_groups, (int)N);
    p.dfs_nt = dfs_nt(B, C, P, sF);
    // non-temporal streams: fp32 = the flat span writer inside B3, bf16 = coefficient blocks + dfs_chunk_kernel
    p.dfs_chunked = dfs_chunked(p.dfs_nt, sF == 4);
    p.dfs_flat_writer = p.dfs_nt && sF == 4;
    *out = p;
    return DENET_OK;
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

// Dropout generators of a train-mode call: off for a null `drop`; p outside [0, 1) is rejected.
static int dropgens(const denet_dropout_t* drop, DropGen& gd, DropGen& wd) {
    gd = wd = DropGen{0u, 0u, 0u, 1.f, 0};
    if (!drop) return DENET_OK;
    if (!(drop->p_gate >= 0.f && drop->p_gate < 1.f) || !(drop->p_weight >= 0.f && drop->p_weight < 1.f))
        return DENET_E_BADARG;
    gd = make_dropgen(drop->seed, drop->p_gate);
    wd = make_dropgen(drop->seed, drop->p_weight);
    return DENET_OK;
}

extern "C" int denet_head_step_train_fwd(const void* Fq, const void* Fs, int feat_dtype, const int64_t* Ys,
                                   const int64_t* labels, const int64_t* Yq, const int64_t* Yq_full,
                                   const float* wgt, const float* weight, const float* W1, const float* b1,
                                   const float* W2, const float* b2, const float* Wl, const float* bl,
                                   const float* gate_mask, const float* weight_mask,
                                   const denet_dropout_t* drop, int64_t N, int64_t way,
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
    if (!(C == 64 || C == 256) || K > 128 || shot > 8) return DENET_E_UNSUPPORTED;
    if (!Fq || !Fs || !Ys || !labels || !weight || !W1 || !b1 || !W2 || !b2 || !Wl || !bl) return DENET_E_BADARG;
    DropGen gd, wd;
    TRY(dropgens(drop, gd, wd));
    PROBED(DENET_K_MASKED_SUM_FWD,
           denet_masked_pool_fwd_chain(Fs, feat_dtype, Ys, B, C, H, W, h, w, s.mask, s.cnt, s.S, W1, s.pre1_part,
                                       stream));
    {
        GateProjArgs ga{s.pre1_part, s.S, s.cnt, b1, W2, b2, Wl, gate_mask, (int)shot, s.hidden, s.gate, s.pool,
                        s.proto_part, gd};
        PROBED(DENET_K_PROTO_EXTEND_FWD, launch_gate_proj(ga, (int)NB, (int)C, (int)shot, (hipStream_t)stream));
    }
    PROBED(DENET_K_CLASSIFY_FWD,
           denet_classify_build_fwd(Fq, feat_dtype, weight, weight_mask, s.proto_part, bl, labels, N, way, K, C, P,
                                    s.Wn, logits_full, logits_binary, &wd, stream));
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

extern "C" int denet_head_step_fwd(const void* Fq, const void* Fs, int feat_dtype, const int64_t* Ys,
                                   const int64_t* labels, const int64_t* Yq, const int64_t* Yq_full,
                                   const float* wgt, const float* weight, const float* W1, const float* b1,
                                   const float* W2, const float* b2, const float* Wl, const float* bl,
                                   const float* gate_mask, const float* weight_mask, int64_t N, int64_t way,
                                   int64_t shot, int64_t K, int64_t C, int64_t h, int64_t w, int64_t H, int64_t W,
                                   int with_grad, float* logits_full, float* logits_binary,
                                   float* losses /*[2]: loss_k, loss_p*/, void* ws, int64_t ws_bytes,
                                   const denet_probe_t* probe, void* stream) {
    return denet_head_step_train_fwd(Fq, Fs, feat_dtype, Ys, labels, Yq, Yq_full, wgt, weight, W1, b1, W2, b2, Wl, bl,
                                     gate_mask, weight_mask, nullptr, N, way, shot, K, C, h, w, H, W, with_grad,
                                     logits_full, logits_binary, losses, ws, ws_bytes, probe, stream);
}

extern "C" int denet_head_step_train_bwd(const void* Fq, int feat_dtype, const int64_t* labels, const float* W1,
                                   const float* W2, const float* Wl, const float* gate_mask,
                                   const float* weight_mask, const denet_dropout_t* drop,
                                   const float* grad_loss_k /*device scalar or NULL*/,
                                   int64_t N, int64_t way, int64_t shot, int64_t K, int64_t C, int64_t h, int64_t w,
                                   int64_t H, int64_t W, void* dFq, void* dFs, float* dweight, float* dW1, float* db1, float* dW2,
                                   float* db2, float* dWl, float* dbl, float* losses /*NULL unless deferred*/,
                                   int flags, void* ws, int64_t ws_bytes, const denet_probe_t* probe,
                                   void* stream) {
    if (!ws || !dFq || !dFs || !Fq) return DENET_E_BADARG;
    if (N <= 0 || way <= 0 || shot <= 0 || K <= 0 || C <= 0 || h <= 0 || w <= 0) return DENET_E_BADARG;
    const bool have_focal = (flags & DENET_BWD_HAVE_FOCAL) != 0;
    const bool no_dfs = (flags & DENET_BWD_NO_DFS) != 0, only_dfs = (flags & DENET_BWD_ONLY_DFS) != 0;
    if (no_dfs && only_dfs) return DENET_E_BADARG;
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
    const int RT = (int)((K + 31) / 32);
    // dWn slabs per episode: one per position tile, or (K > 32, persistent blocks) one per tile group
    // persistent blocks at every K (quad loads need P >= 4).  K <= 32 against one tile per block: C2 step
    // 88.2 vs 89.7 us at 8 episodes (14 slabs per episode instead of 27 for B2 to sum), 1.000 vs 1.024 ms at 128
    // (B2 18 vs 57 us)
    const bool wide = bwd_persistent(P);
    const int n_part = wide ? cls_bwd_wide_groups(N, P, RT) : (int)denet_classify_bwd_parts(P);
    const bool f32 = feat_dtype == DENET_F32;
    hipStream_t st = (hipStream_t)stream;
    DropGen gd, wd;
    TRY(dropgens(drop, gd, wd));

    ClsBwdArgs g{Fq, s.Wn, s.dlog_a, s.dlog_b, nullptr, grad_loss_k, s.part_k, (int)pk, (int)N, (int)K, (int)C,
                 (int)P, n_part, nullptr, s.dWn_part, s.scale};
    // B1: dWn slabs || dFq; scale[n] is derived from the loss partials inside the kernel
    if (!only_dfs) {
        ClsBwdArgs g2 = g;
        g2.dFq = dFq;
        g2.dWn_part = nullptr;
        g2.scale_out = nullptr;
        const bool on = probe && probe->which == DENET_K_CLASSIFY_BWD && probe->start && probe->stop;
        if (on && hipEventRecord((hipEvent_t)probe->start, st) != hipSuccess) return DENET_E_LAUNCH;
        const float* pp_ptr = have_focal ? s.part_p : nullptr;
        int rc = f32 ? RT_DISPATCH(launch_bwd_dwn, float, RT, g, g2, pp_ptr, (int)NB, (int)pp, losses, n_part, (int)N,
                                   wide, st)
                     : RT_DISPATCH(launch_bwd_dwn, unsigned short, RT, g, g2, pp_ptr, (int)NB, (int)pp, losses,
                                   n_part, (int)N, wide, st);
        if (rc) return rc;
        DENET_CHECK_LAUNCH();
        if (on && hipEventRecord((hipEvent_t)probe->stop, st) != hipSuccess) return DENET_E_LAUNCH;
    }
    PeParamsArgs pa{nullptr, s.u2, s.u3, s.S, s.hidden, s.pool, s.dWn_part, weight_mask, labels, n_part, (int)N, (int)way,
                    (int)B, (int)K, (int)C, (float)P, dW1, db1, dW2, db2, dWl, dbl, dweight, s.dh_part, wd};
    // B2: sliced chain, middle || dweight
    if (!only_dfs) {
        ChainMidArgs cm{s.dWn_part, s.S, s.cnt, s.gate, gate_mask, Wl, W2, labels, n_part, (int)way, (int)shot,
                        (int)K, s.u2, s.u3, s.dpool, s.dh_part, gd};
        const bool on = probe && probe->which == DENET_K_PE_BWD_CHAIN && probe->start && probe->stop;
        if (on && hipEventRecord((hipEvent_t)probe->start, st) != hipSuccess) return DENET_E_LAUNCH;
        int rc = launch_bwd_mid(cm, pa, (int)NB, (int)C, (int)shot, st);
        if (rc) return rc;
        DENET_CHECK_LAUNCH();
        if (on && hipEventRecord((hipEvent_t)probe->stop, st) != hipSuccess) return DENET_E_LAUNCH;
    }
    // B3: parameter gradients || coef + dFs
    {
        const bool nt = dfs_nt(B, C, P, f32 ? 4 : 2);
        // bf16 streams that go around the caches are written by a kernel of their own, one block per contiguous
        // chunk in address order (dev_masked_sum.hpp: dfs_chunk_kernel); the dFs blocks of this launch then only
        // compute their slice's coefficients.  C5 (269 MB): B3 86.1 -> 60.6 us, step 396 -> 371 us.  fp32 keeps the
        // flat span writer inside this launch: chunked it measured 86.4 vs 82.7 us at C3 (344 MB) and 332 vs 284 us
        // at 1.1 GB (one short-lived thread per 16 bytes: the mask load's latency is not hidden).
        const bool chunked = dfs_chunked(nt, f32) && ((uintptr_t)dFs & 15) == 0 && C * P < (1 << 22) && B <= 65535;
        CoefArgs ca{s.dh_part, s.hidden, W1, s.dpool, s.gate, s.cnt, (int)shot, (int)(C / 16), (float)P, SC_EPS,
                    chunked ? s.coef : nullptr};
        // one launch, or its two halves apart (DENET_BWD_NO_DFS / DENET_BWD_ONLY_DFS): the kernel takes
        // its first gridDim.x - 3*gx blocks as dFs blocks
        const int gx = only_dfs ? 0 : (int)C / WG_ROWS;      // blocks of each of the roles dW1, dW2, dWl
        const int rpb = 16, ncb = (int)(C / 16);
        const int64_t ndfs = no_dfs ? 0 : B * ncb;
        dim3 grid((unsigned)(3 * gx + ndfs)), block(256);
        const bool on = probe && probe->which == DENET_K_MASKED_SUM_BWD && probe->start && probe->stop;
        if (on && hipEventRecord((hipEvent_t)probe->start, st) != hipSuccess) return DENET_E_LAUNCH;
        // LDS: the mask image of the dFs blocks (dev_masked_sum.hpp) + their 64 + 16 floats
        size_t lds = (size_t)((f32 ? (nt ? mask_image_floats<4, MaskImage<float, true>::PLANAR>((int)P)
                                         : mask_image_floats<4, MaskImage<float, false>::PLANAR>((int)P))
                                   : mask_image_floats<8, true>((int)P)) + 64 + 16) * 4;
        if (lds < 4096) lds = 4096;
#ifdef DENET_TAIL_LDS_MIN_NT
        if (nt && lds < (size_t)DENET_TAIL_LDS_MIN_NT) lds = (size_t)DENET_TAIL_LDS_MIN_NT;   // diagnostic: fewer resident blocks
#endif
#define TAIL_GO(TT, NSLT)                                                                                           \
    if (nt) {                                                                                                       \
        if (denet_ensure_dynamic_lds(reinterpret_cast<const void*>(head_bwd_tail_kernel<TT, NSLT, true>), lds))     \
            return DENET_E_LAUNCH;                                                                                  \
        hipLaunchKernelGGL((head_bwd_tail_kernel<TT, NSLT, true>), grid, block, lds, st, pa, ca, s.mask, (TT*)dFs,  \
                           (int)C, (int)P, rpb, ncb, gx, (int)(nt && B >= DENET_TAIL_PARAMS_FIRST_B));                                                         \
    } else {                                                                                                        \
        if (denet_ensure_dynamic_lds(reinterpret_cast<const void*>(head_bwd_tail_kernel<TT, NSLT, false>), lds))    \
            return DENET_E_LAUNCH;                                                                                  \
        hipLaunchKernelGGL((head_bwd_tail_kernel<TT, NSLT, false>), grid, block, lds, st, pa, ca, s.mask, (TT*)dFs, \
                           (int)C, (int)P, rpb, ncb, gx, (int)(nt && B >= DENET_TAIL_PARAMS_FIRST_B));                                                         \
    }
        if (f32) {
            if (C == 256) { TAIL_GO(float, 16); } else { TAIL_GO(float, 4); }
        } else {
            if (C == 256) { TAIL_GO(unsigned short, 16); } else { TAIL_GO(unsigned short, 4); }
        }
#undef TAIL_GO
        DENET_CHECK_LAUNCH();
        if (chunked && !no_dfs) {
#ifndef DENET_DFS_CHUNK_U_F32
#define DENET_DFS_CHUNK_U_F32 2
#endif
#ifndef DENET_DFS_CHUNK_U_BF16
#define DENET_DFS_CHUNK_U_BF16 2
#endif
            // (the parameter-gradient blocks were also tried inside this launch, where a 48 us stream would hide their
            // 10.5 us: the combined kernel's register count cuts the chunk blocks' occupancy, C5 B3 60.5 -> 68 us)
            const int64_t span = C * P;
            if (f32) {
                constexpr int U = DENET_DFS_CHUNK_U_F32;
                const int64_t per = 256 * U * 4;
                hipLaunchKernelGGL((dfs_chunk_kernel<float, U>), dim3((unsigned)((span + per - 1) / per), (unsigned)B),
                                   dim3(256), 0, st, s.mask, s.coef, (float*)dFs, (int)C, (int)P);
            } else {
                constexpr int U = DENET_DFS_CHUNK_U_BF16;
                const int64_t per = 256 * U * 8;
                hipLaunchKernelGGL((dfs_chunk_kernel<unsigned short, U>),
                                   dim3((unsigned)((span + per - 1) / per), (unsigned)B), dim3(256), 0, st, s.mask, s.coef,
                                   (unsigned short*)dFs, (int)C, (int)P);
            }
            DENET_CHECK_LAUNCH();
        }
        if (on && hipEventRecord((hipEvent_t)probe->stop, st) != hipSuccess) return DENET_E_LAUNCH;
    }
    return DENET_OK;
}

extern "C" int denet_head_step_bwd(const void* Fq, int feat_dtype, const int64_t* labels, const float* W1,
                                   const float* W2, const float* Wl, const float* gate_mask,
                                   const float* weight_mask, const float* grad_loss_k /*device scalar or NULL*/,
                                   int64_t N, int64_t way, int64_t shot, int64_t K, int64_t C, int64_t h, int64_t w,
                                   int64_t H, int64_t W, void* dFq, void* dFs, float* dweight, float* dW1, float* db1, float* dW2,
                                   float* db2, float* dWl, float* dbl, float* losses /*NULL unless deferred*/,
                                   int flags, void* ws, int64_t ws_bytes, const denet_probe_t* probe,
                                   void* stream) {
    return denet_head_step_train_bwd(Fq, feat_dtype, labels, W1, W2, Wl, gate_mask, weight_mask, nullptr, grad_loss_k,
                                     N, way, shot, K, C, h, w, H, W, dFq, dFs, dweight, dW1, db1, dW2, db2, dWl, dbl,
                                     losses, flags, ws, ws_bytes, probe, stream);
}

// The multipliers the train-mode step generates for (seed, stream_id, p), as a tensor: out[i], i < n.
// stream_id DENET_DROPOUT_GATE: element (b, c) of the gate pre-activation [B][C]; DENET_DROPOUT_WEIGHT:
// element (n, k, c) of the per-episode class weights [N][K][C].
__global__ void __launch_bounds__(256) dropout_mask_kernel(DropGen g, unsigned stream_id, int64_t n, float* out) {
    const int64_t i4 = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i4 * 4 >= n) return;
    f32x4 m = {1.f, 1.f, 1.f, 1.f};
    if (g.on) m = drop_mult4(g, stream_id, (unsigned)i4);
    for (int t = 0; t < 4; ++t)
        if (i4 * 4 + t < n) out[i4 * 4 + t] = m[t];
}

extern "C" int denet_dropout_mask(uint64_t seed, int stream_id, float p, int64_t n, float* out, void* stream) {
    if (!out || n <= 0 || !(p >= 0.f && p < 1.f)) return DENET_E_BADARG;
    if (stream_id != DENET_DROPOUT_GATE && stream_id != DENET_DROPOUT_WEIGHT) return DENET_E_BADARG;
    if (n > ((int64_t)1 << 34)) return DENET_E_UNSUPPORTED;          // 32-bit counter of 4-element groups
    const int64_t groups = (n + 3) / 4;
    hipLaunchKernelGGL(dropout_mask_kernel, dim3((unsigned)((groups + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                       make_dropgen(seed, p), (unsigned)stream_id, n, out);
    DENET_CHECK_LAUNCH();
    return DENET_OK;
}
