// K2 / K2b: channel gate, prototype projection, dynamic class-weight
// extension, and their backward.
//
// Work per support image is three C x C mat-vecs, so this stage is bound by
// streaming the three weight matrices from L2 into a CU (768 KB at C = 256),
// not by HBM.  One 512-thread workgroup owns one (episode, way) group, so the
// `shot` images of a group share every weight row it reads, the shot-mean is
// local, and the class label is read on the device (the reference syncs the
// host once per way: model/common.py:328).
//
// Forward mat-vec layout: a wave covers 4 weight rows per step, 16 lanes per
// row, each lane 16-byte loads along the row (coalesced 256-B segments); the
// 16-lane partial sums are combined inside the DPP row.
// Backward (transposed) mat-vec layout: one thread per column, the row loop
// split over 512/C thread groups; loads are coalesced across columns.
#include "denet_common.hpp"
#include "dev_proto_extend.hpp"
#include <mutex>

#define PE_THREADS 512


// ---------------------------------------------------------------------------
// Forward.  1024 threads = 16 waves; a wave covers 4 weight rows per step, 16 lanes per row
// (16-byte loads along the row, coalesced 256-B segments), so one step of the block covers 64 rows
// and a C x C layer is C/64 steps.  The three layers form ONE sequence of 3*C/64 steps whose weight
// loads run through a 4-slot register ring, three steps ahead of the FMAs and across the layer
// boundaries (the weights do not depend on the activations), so the block pays the memory latency
// once instead of once per step.  Activations come from LDS; bias, activation and the global
// stores happen in a block-wide pass at each layer boundary.  The step loop is fully unrolled
// (trip count is a template constant), which keeps every ring index static.
// ---------------------------------------------------------------------------
#define PF_THREADS 1024
#define PF_ROWS (PF_THREADS / 16)      // weight rows covered by one step of the block
#define PF_DEPTH 3

// z[s][o] = sum_i W[o][i] * xs[s][i]  for the 4 rows of this wave in step `it`, s < NS (no bias)
template <int CJ, int NS>
__device__ __forceinline__ void layer_step(const f32x4 (&wv)[CJ], const float* xs, float* z, int it) {
    constexpr int C = CJ * 64;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int r = lane >> 4, q = lane & 15;
    float acc[NS];
#pragma unroll
    for (int s = 0; s < NS; ++s) {
        float a = 0.f;
#pragma unroll
        for (int j = 0; j < CJ; ++j) {
            const f32x4 xv = *reinterpret_cast<const f32x4*>(xs + s * C + q * 4 + 64 * j);
            a += wv[j][0] * xv[0] + wv[j][1] * xv[1] + wv[j][2] * xv[2] + wv[j][3] * xv[3];
        }
        acc[s] = a;
    }
#pragma unroll
    for (int s = 0; s < NS; ++s) acc[s] = row16_sum(acc[s]);
    if (q == 0) {
        const int o = it * PF_ROWS + wave * 4 + r;
#pragma unroll
        for (int s = 0; s < NS; ++s) z[s * C + o] = acc[s];
    }
}

// NS >= shot: images of the group handled together (shot <= 8).
template <int CJ, int NS>
__global__ __launch_bounds__(PF_THREADS) void proto_extend_fwd_kernel(
    const float* __restrict__ S, const float* __restrict__ cnt, const float* __restrict__ W1,
    const float* __restrict__ b1, const float* __restrict__ W2, const float* __restrict__ b2,
    const float* __restrict__ Wl, const float* __restrict__ bl, const float* __restrict__ weight,
    const int64_t* __restrict__ labels, const float* __restrict__ gate_mask,
    const float* __restrict__ weight_mask, int way, int shot, int K, float Pf,
    float* __restrict__ hidden, float* __restrict__ gate, float* __restrict__ pool, float* __restrict__ Wn,
    float* __restrict__ proto) {
    constexpr int C = CJ * 64;
    constexpr int STEPS = C / PF_ROWS, T = 3 * STEPS;
    extern __shared__ __attribute__((aligned(16))) float sm[];
    float* xa = sm;                    // layer input (att, hidden, mean pool)  [NS][C]
    float* zb = xa + NS * C;           // layer output before bias             [NS][C]
    float* sraw = zb + NS * C;         // S                                    [NS][C]
    float* gmk = sraw + NS * C;        // gate mask                            [NS][C]
    float* cden = gmk + NS * C;        // cnt + eps [NS]
    const int grp = blockIdx.x;
    const int n = grp / way, j = grp - n * way;
    const int64_t b0 = (int64_t)grp * shot;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wrow = wave * 4 + (lane >> 4), wcol = (lane & 15) * 4;
    const int ns = shot;

    f32x4 ring[PF_DEPTH + 1][CJ];
    auto issue = [&](int t, f32x4 (&dst)[CJ]) {
        const float* W = (t < STEPS) ? W1 : (t < 2 * STEPS ? W2 : Wl);
        const int it = t % STEPS;
        const float* p = W + (int64_t)(it * PF_ROWS + wrow) * C + wcol;
#pragma unroll
        for (int jj = 0; jj < CJ; ++jj) dst[jj] = *reinterpret_cast<const f32x4*>(p + 64 * jj);
    };
    STAMP(0);
#pragma unroll
    for (int t = 0; t < PF_DEPTH; ++t)
        if (t < T) issue(t, ring[t]);

    // Wn[n] = dropout(weight) for the rows no way of this episode overwrites (model/common.py:317,326)
    const int64_t* lab = labels + (int64_t)n * way;
    float* Wn_n = Wn + (int64_t)n * K * C;
    if (j == 0) {
        const float* wm = weight_mask ? weight_mask + (int64_t)n * K * C : nullptr;
        for (int idx = tid * 4; idx < K * C; idx += PF_THREADS * 4) {
            const int k = idx / C;
            bool novel = false;
            for (int jj = 0; jj < way; ++jj) novel = novel || (lab[jj] == k);
            if (novel) continue;
            f32x4 v = *reinterpret_cast<const f32x4*>(weight + idx);
            if (wm) {
                f32x4 m = *reinterpret_cast<const f32x4*>(wm + idx);
                v[0] *= m[0]; v[1] *= m[1]; v[2] *= m[2]; v[3] *= m[3];
            }
            *reinterpret_cast<f32x4*>(Wn_n + idx) = v;
        }
    }
    // att = mean over all P positions of mask*F  (model/common.py:243); unused slots are zero
    for (int i = tid; i < NS * C; i += PF_THREADS) {
        const bool ok = i < ns * C;
        const float sv = ok ? S[b0 * C + i] : 0.f;
        sraw[i] = sv;
        xa[i] = sv / Pf;
        gmk[i] = (ok && gate_mask) ? gate_mask[b0 * C + i] : 1.f;
    }
    if (tid < NS) cden[tid] = (tid < ns ? cnt[b0 + tid] : 1.f) + PE_EPS;
    __syncthreads();
    STAMP(1);

#pragma unroll
    for (int t = 0; t < T; ++t) {
        if (t + PF_DEPTH < T) issue(t + PF_DEPTH, ring[(t + PF_DEPTH) & PF_DEPTH]);
        const int layer = t / STEPS, it = t % STEPS;
        if (layer < 2) layer_step<CJ, NS>(ring[t & PF_DEPTH], xa, zb, it);
        else layer_step<CJ, 1>(ring[t & PF_DEPTH], xa, zb, it);
        if (it == STEPS - 1) {
            __syncthreads();
            if (layer == 0) {
                // hidden = relu(W1 att + b1)  (model/common.py:226-228)
                for (int i = tid; i < NS * C; i += PF_THREADS) {
                    const float v = fmaxf(zb[i] + b1[i % C], 0.f);
                    xa[i] = v;
                    if (i < ns * C) hidden[b0 * C + i] = v;
                }
                STAMP(2);
            } else if (layer == 1) {
                // g = sigmoid(dropout(W2 hidden + b2)); pool = g*S/(cnt+eps)  (model/common.py:229-232,192-194)
                for (int i = tid; i < ns * C; i += PF_THREADS) {
                    const int o = i % C, s = i / C;
                    const float v = (zb[i] + b2[o]) * gmk[i];
                    const float g = 1.f / (1.f + expf(-v));
                    gate[b0 * C + i] = g;
                    const float pl = g * sraw[i] / cden[s];
                    pool[b0 * C + i] = pl;
                    zb[i] = pl;
                }
                __syncthreads();
                // mean_s (Wl pool_s + bl) = Wl (mean_s pool_s) + bl: one mat-vec per group (model/common.py:196,319)
                for (int c = tid; c < C; c += PF_THREADS) {
                    float tsum = 0.f;
                    for (int s = 0; s < ns; ++s) tsum += zb[s * C + c];
                    xa[c] = tsum / (float)shot;
                }
                STAMP(3);
            }
            __syncthreads();
        }
    }
    STAMP(4);
    // Wn[n, label_j] = prototype  (model/common.py:327-334); a later way overwrites an earlier one
    const int64_t mine = lab[j];
    bool win = mine >= 0 && mine < K;
    for (int jj = j + 1; jj < way; ++jj) win = win && (lab[jj] != mine);
    if (win)
        for (int c = tid; c < C; c += PF_THREADS) Wn_n[mine * C + c] = zb[c] + bl[c];
    if (proto)
        for (int c = tid; c < C; c += PF_THREADS) proto[(int64_t)grp * C + c] = zb[c] + bl[c];
    STAMP(5);
}

// ---------------------------------------------------------------------------
// C ABI
// ---------------------------------------------------------------------------
static inline bool pos(int64_t v) { return v > 0 && v < ((int64_t)1 << 30); }
static inline bool c_ok(int64_t C) { return C == 64 || C == 256; }

extern "C" int denet_proto_extend_fwd(const float* S, const float* cnt, const float* W1, const float* b1,
                                      const float* W2, const float* b2, const float* Wl, const float* bl,
                                      const float* weight, const int64_t* labels, const float* gate_mask,
                                      const float* weight_mask, int64_t N, int64_t way, int64_t shot,
                                      int64_t K, int64_t C, int64_t P, float* hidden, float* gate,
                                      float* pool, float* Wn, float* proto, void* stream) {
    if (!S || !cnt || !W1 || !b1 || !W2 || !b2 || !Wl || !bl || !weight || !labels || !hidden || !gate ||
        !pool || !Wn)
        return DENET_E_BADARG;
    if (!pos(N) || !pos(way) || !pos(shot) || !pos(K) || !pos(C) || !pos(P)) return DENET_E_BADARG;
    if (!c_ok(C) || K > 128 || shot > 8) return DENET_E_UNSUPPORTED;
    dim3 grid((unsigned)(N * way)), block(PF_THREADS);
    hipStream_t st = (hipStream_t)stream;
#define PE_LAUNCH2(CJ, NS)                                                                                  \
    hipLaunchKernelGGL((proto_extend_fwd_kernel<CJ, NS>), grid, block, (size_t)(4 * NS * C + NS + 4) * 4, st, S, \
                       cnt, W1, b1, W2, b2, Wl, bl, weight, labels, gate_mask, weight_mask, (int)way, (int)shot, \
                       (int)K, (float)P, hidden, gate, pool, Wn, proto)
#define PE_LAUNCH(CJ)                                  \
    if (shot == 1) PE_LAUNCH2(CJ, 1);                  \
    else if (shot <= 3) PE_LAUNCH2(CJ, 3);             \
    else if (shot <= 5) PE_LAUNCH2(CJ, 5);             \
    else PE_LAUNCH2(CJ, 8)
    switch (C) {
        case 64: PE_LAUNCH(1); break;
        case 128: PE_LAUNCH(2); break;
        case 256: PE_LAUNCH(4); break;
        default: return DENET_E_UNSUPPORTED;
    }
#undef PE_LAUNCH2
#undef PE_LAUNCH
    DENET_CHECK_LAUNCH();
    return DENET_OK;
}

extern "C" int64_t denet_proto_extend_bwd_workspace(int64_t N, int64_t way, int64_t shot, int64_t K, int64_t C) {
    if (N <= 0 || way <= 0 || shot <= 0 || K <= 0 || C <= 0) return DENET_E_BADARG;
    return (3 * N * way * shot * C + N * K * C) * 4;
}

// The three launches of K2b, separately callable so that the fused step can time each one.
int denet_pe_bwd_reduce(const float* dWn_part, int64_t n_part, int64_t N, int64_t K, int64_t C, float* dWn,
                        void* stream) {
    const int64_t per_n = K * C;
    hipLaunchKernelGGL(reduce_parts_kernel, dim3((unsigned)((per_n / 4 + 255) / 256), (unsigned)N), dim3(256), 0,
                       (hipStream_t)stream, dWn_part, (int)n_part, per_n, dWn);
    DENET_CHECK_LAUNCH();
    return DENET_OK;
}

int denet_pe_bwd_chain(const float* dWn, int64_t n_part, const float* S, const float* cnt, const float* hidden, const float* gate,
                       const float* W1, const float* W2, const float* Wl, const int64_t* labels,
                       const float* gate_mask, int64_t N, int64_t way, int64_t shot, int64_t K, int64_t C,
                       int64_t P, float* u1, float* u2, float* u3, float* coef, void* stream) {
    hipStream_t st = (hipStream_t)stream;
    const size_t lds = PE_CHAIN_LDS(C);
    PeChainArgs a{dWn, S, cnt, hidden, gate, W1, W2, Wl, gate_mask, labels, (int)n_part, (int)way, (int)shot,
                  (int)K, (float)P, u1, u2, u3, coef};
#define CH_LAUNCH(CC, CHK)                                                                                   \
    if (denet_ensure_dynamic_lds(reinterpret_cast<const void*>(proto_extend_bwd_chain_kernel<CC, CHK>), lds)) \
        return DENET_E_LAUNCH;                                                                               \
    hipLaunchKernelGGL((proto_extend_bwd_chain_kernel<CC, CHK>), dim3((unsigned)(N * way)), dim3(CH_THREADS),  \
                       lds, st, a)
    switch (C) {
        case 64: CH_LAUNCH(64, 8); break;
        case 256: CH_LAUNCH(256, 16); break;
        default: return DENET_E_UNSUPPORTED;
    }
#undef CH_LAUNCH
    DENET_CHECK_LAUNCH();
    return DENET_OK;
}

int denet_pe_bwd_params(const float* u1, const float* u2, const float* u3, const float* S, const float* hidden,
                        const float* pool, const float* dWn, int64_t n_part, const int64_t* labels,
                        const float* weight_mask, int64_t N, int64_t way, int64_t shot, int64_t K, int64_t C, int64_t P, float* dW1,
                        float* db1, float* dW2, float* db2, float* dWl, float* dbl, float* dweight, void* stream) {
    const int64_t B = N * way * shot;
    const unsigned gx = (unsigned)pe_params_gx((int)K, (int)C);
    PeParamsArgs a{u1, u2, u3, S, hidden, pool, dWn, weight_mask, labels, (int)n_part, (int)N, (int)way, (int)B,
                   (int)K, (int)C, (float)P, dW1, db1, dW2, db2, dWl, dbl, dweight};
    hipLaunchKernelGGL(proto_extend_bwd_params_kernel, dim3(gx, 4), dim3(256), 0, (hipStream_t)stream, a);
    DENET_CHECK_LAUNCH();
    return DENET_OK;
}

extern "C" int denet_proto_extend_bwd(const float* dWn_part, int64_t n_part, const float* S, const float* cnt,
                                      const float* hidden, const float* gate, const float* pool,
                                      const float* W1, const float* W2, const float* Wl, const int64_t* labels,
                                      const float* gate_mask, const float* weight_mask, int64_t N, int64_t way,
                                      int64_t shot, int64_t K, int64_t C, int64_t P, float* dW1, float* db1,
                                      float* dW2, float* db2, float* dWl, float* dbl, float* dweight,
                                      float* coef, void* ws, int64_t ws_bytes, void* stream) {
    if (!dWn_part || !S || !cnt || !hidden || !gate || !pool || !W1 || !W2 || !Wl || !labels || !dW1 || !db1 ||
        !dW2 || !db2 || !dWl || !dbl || !dweight || !coef || !ws)
        return DENET_E_BADARG;
    if (!pos(N) || !pos(way) || !pos(shot) || !pos(K) || !pos(C) || !pos(P) || !pos(n_part))
        return DENET_E_BADARG;
    if (!c_ok(C) || K > 128 || shot > 8) return DENET_E_UNSUPPORTED;
    const int64_t B = N * way * shot;
    if (ws_bytes < denet_proto_extend_bwd_workspace(N, way, shot, K, C)) return DENET_E_WORKSPACE;
    if (N > 65535) return DENET_E_UNSUPPORTED;
    float* u1 = (float*)ws;
    float* u2 = u1 + B * C;
    float* u3 = u2 + B * C;
    float* dWn = u3 + B * C;                      // [N,K,C] tile-reduced
    int rc;
    if (n_part > 1) {
        if ((rc = denet_pe_bwd_reduce(dWn_part, n_part, N, K, C, dWn, stream))) return rc;
        dWn_part = dWn;
    }
    if ((rc = denet_pe_bwd_chain(dWn_part, 1, S, cnt, hidden, gate, W1, W2, Wl, labels, gate_mask, N, way, shot, K, C,
                                 P, u1, u2, u3, coef, stream)))
        return rc;
    return denet_pe_bwd_params(u1, u2, u3, S, hidden, pool, dWn_part, 1, labels, weight_mask, N, way, shot, K, C, P,
                               dW1, db1, dW2, db2, dWl, dbl, dweight, stream);
}

// ---------------------------------------------------------------------------
// The same backward for callers that cannot allocate (HIP graph capture, tight serving loops, harnesses that
// forbid hipMalloc): the scratch that denet_proto_extend_bwd takes as `ws` lives in a file-scope device array.
// The array's NAME is only a host-side stand-in: its device address is looked up once per device with
// hipGetSymbolAddress and that pointer -- never the name -- is what the kernels receive.
// ---------------------------------------------------------------------------
#define PE_STATIC_SCRATCH_BYTES (1 << 20)
static __device__ __align__(256) float scratch[PE_STATIC_SCRATCH_BYTES / 4];

#define PE_MAX_DEVICES 64
static float* g_scratch_ptr[PE_MAX_DEVICES];
static std::mutex g_scratch_mu;

// device address of `scratch` on the CURRENT device (each GPU holds its own copy at its own address)
static int pe_static_scratch(float** out) {
    int dev = -1;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= PE_MAX_DEVICES) return DENET_E_LAUNCH;
    std::lock_guard<std::mutex> lock(g_scratch_mu);
    if (!g_scratch_ptr[dev]) {
        void* p = nullptr;
        if (hipGetSymbolAddress(&p, HIP_SYMBOL(scratch)) != hipSuccess || !p) return DENET_E_LAUNCH;
        g_scratch_ptr[dev] = static_cast<float*>(p);
    }
    *out = g_scratch_ptr[dev];
    return DENET_OK;
}

extern "C" int64_t denet_proto_extend_bwd_static_limit(void) { return PE_STATIC_SCRATCH_BYTES; }

extern "C" int denet_proto_extend_bwd_static(const float* dWn_part, int64_t n_part, const float* S, const float* cnt,
                                             const float* hidden, const float* gate, const float* pool,
                                             const float* W1, const float* W2, const float* Wl,
                                             const int64_t* labels, const float* gate_mask,
                                             const float* weight_mask, int64_t N, int64_t way, int64_t shot,
                                             int64_t K, int64_t C, int64_t P, float* dW1, float* db1, float* dW2,
                                             float* db2, float* dWl, float* dbl, float* dweight, float* coef,
                                             void* stream) {
    const int64_t need = denet_proto_extend_bwd_workspace(N, way, shot, K, C);
    if (need < 0) return (int)need;
    if (need > PE_STATIC_SCRATCH_BYTES) return DENET_E_STATIC_LIMIT;
    float* ptr = nullptr;
    const int rc = pe_static_scratch(&ptr);
    if (rc) return rc;
    // every intermediate in the scratch (u1, u2, u3, the tile-reduced dWn) is fully written by the kernels
    // before it is read, so nothing of the previous call's contents has to be cleared
    return denet_proto_extend_bwd(dWn_part, n_part, S, cnt, hidden, gate, pool, W1, W2, Wl, labels, gate_mask,
                                  weight_mask, N, way, shot, K, C, P, dW1, db1, dW2, db2, dWl, dbl, dweight, coef,
                                  ptr, PE_STATIC_SCRATCH_BYTES, stream);
}

// dWn[n] = sum over position tiles of the slabs K3b wrote, as a tensor (the gradient of
// WeightEstimator.infer w.r.t. its `weight` argument when extend and infer are called apart).
extern "C" int denet_sum_parts(const float* part, int64_t N, int64_t n_part, int64_t per_n, float* out,
                               void* stream) {
    if (!part || !out) return DENET_E_BADARG;
    if (!pos(N) || !pos(n_part) || !pos(per_n)) return DENET_E_BADARG;
    if (N > 65535 || (per_n & 3)) return DENET_E_UNSUPPORTED;
    hipLaunchKernelGGL(reduce_parts_kernel, dim3((unsigned)((per_n / 4 + 255) / 256), (unsigned)N), dim3(256), 0,
                       (hipStream_t)stream, part, (int)n_part, per_n, out);
    DENET_CHECK_LAUNCH();
    return DENET_OK;
}
