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

#define PE_THREADS 512
#define PE_EPS 1e-3f      // model/common.py:170

// y[s][o] = b[o] + sum_i W[o][i] * xs[s][i]   for s < shot, o < C ; result handed to epi(o, s, y)
template <int CJ, typename Epi>
__device__ __forceinline__ void dense_rows(const float* __restrict__ W, const float* __restrict__ bias,
                                           const float* xs, int shot, Epi epi) {
    constexpr int C = CJ * 64;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int r = lane >> 4, q = lane & 15;
    f32x4 wv[CJ], wn[CJ];
    int o = wave * 4 + r;
#pragma unroll
    for (int j = 0; j < CJ; ++j) wv[j] = *reinterpret_cast<const f32x4*>(W + (int64_t)o * C + q * 4 + 64 * j);
    for (; o < C; o += 32) {
        const int on = o + 32;
        if (on < C) {
#pragma unroll
            for (int j = 0; j < CJ; ++j)
                wn[j] = *reinterpret_cast<const f32x4*>(W + (int64_t)on * C + q * 4 + 64 * j);
        }
        const float bo = bias[o];
        for (int s = 0; s < shot; ++s) {
            float acc = 0.f;
#pragma unroll
            for (int j = 0; j < CJ; ++j) {
                f32x4 xv = *reinterpret_cast<const f32x4*>(xs + s * C + q * 4 + 64 * j);
                acc += wv[j][0] * xv[0] + wv[j][1] * xv[1] + wv[j][2] * xv[2] + wv[j][3] * xv[3];
            }
            acc = row16_sum(acc);
            if (q == 0) epi(o, s, acc + bo);
        }
#pragma unroll
        for (int j = 0; j < CJ; ++j) wv[j] = wn[j];
    }
}

template <int CJ>
__global__ __launch_bounds__(PE_THREADS) void proto_extend_fwd_kernel(
    const float* __restrict__ S, const float* __restrict__ cnt, const float* __restrict__ W1,
    const float* __restrict__ b1, const float* __restrict__ W2, const float* __restrict__ b2,
    const float* __restrict__ Wl, const float* __restrict__ bl, const float* __restrict__ weight,
    const int64_t* __restrict__ labels, const float* __restrict__ gate_mask,
    const float* __restrict__ weight_mask, int way, int shot, int K, float Pf,
    float* __restrict__ hidden, float* __restrict__ gate, float* __restrict__ pool, float* __restrict__ Wn) {
    constexpr int C = CJ * 64;
    extern __shared__ __attribute__((aligned(16))) float sm[];
    float* xa = sm;                    // att     [shot][C]
    float* xb = xa + shot * C;         // hidden  [shot][C]
    float* xc = xb + shot * C;         // pool    [shot][C]
    float* pbar = xc + shot * C;       // [C]
    const int grp = blockIdx.x;
    const int n = grp / way, j = grp - n * way;
    const int64_t b0 = (int64_t)grp * shot;
    const int tid = threadIdx.x;

    // att = mean over all P positions of mask*F  (model/common.py:243)
    for (int i = tid; i < shot * C; i += PE_THREADS) xa[i] = S[b0 * C + i] / Pf;
    __syncthreads();
    // hidden = relu(W1 att + b1)  (model/common.py:226-228)
    dense_rows<CJ>(W1, b1, xa, shot, [&](int o, int s, float v) {
        v = fmaxf(v, 0.f);
        xb[s * C + o] = v;
        hidden[(b0 + s) * C + o] = v;
    });
    __syncthreads();
    // g = sigmoid(dropout(W2 hidden + b2)); pool = g*S/(cnt+eps)  (model/common.py:229-232,192-194)
    dense_rows<CJ>(W2, b2, xb, shot, [&](int o, int s, float v) {
        const int64_t idx = (b0 + s) * C + o;
        if (gate_mask) v *= gate_mask[idx];
        const float g = 1.f / (1.f + expf(-v));
        gate[idx] = g;
        const float pl = g * S[idx] / (cnt[b0 + s] + PE_EPS);
        pool[idx] = pl;
        xc[s * C + o] = pl;
    });
    __syncthreads();
    // proto = Wl pool + bl, then mean over shot  (model/common.py:196,319)
    {
        float run = 0.f;
        const float inv = (float)shot;
        dense_rows<CJ>(Wl, bl, xc, shot, [&](int o, int s, float v) {
            run = (s == 0) ? v : run + v;
            if (s == shot - 1) pbar[o] = run / inv;
        });
    }
    __syncthreads();

    // Wn[n] = dropout(weight); Wn[n, label_j] = pbar  (model/common.py:317-334)
    const int64_t* lab = labels + (int64_t)n * way;
    const int64_t mine = lab[j];
    bool win = mine >= 0 && mine < K;
    for (int jj = j + 1; jj < way; ++jj) win = win && (lab[jj] != mine);     // a later way overwrites
    float* Wn_n = Wn + (int64_t)n * K * C;
    if (win)
        for (int c = tid; c < C; c += PE_THREADS) Wn_n[mine * C + c] = pbar[c];
    if (j == 0) {
        const float* wm = weight_mask ? weight_mask + (int64_t)n * K * C : nullptr;
        for (int idx = tid * 4; idx < K * C; idx += PE_THREADS * 4) {
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
}

// ---------------------------------------------------------------------------
// Backward, stage A: per-group chain.  Transposed mat-vecs for up to 8 images
// at a time (vin laid out [C][8] in LDS, zero padded).
//   y[s][i] = sum_o W[o][i] * vin[o][s]
// Each thread owns column i = tid % C for the row range of its split; partials
// are combined through LDS.  Result handed to epi(i, s, y) by split 0.
// ---------------------------------------------------------------------------
template <typename Epi>
__device__ __forceinline__ void dense_cols8(const float* __restrict__ W, const float* vin /*[C][8]*/,
                                            float* red /*[nsplit][8][C]*/, int C, int ns, Epi epi) {
    const int tid = threadIdx.x;
    const int nsplit = PE_THREADS / C;            // C in {64,128,256,512}
    const int i = tid % C, sp = tid / C;
    const int rows = C / nsplit;
    const int o0 = sp * rows;
    float acc[8];
#pragma unroll
    for (int s = 0; s < 8; ++s) acc[s] = 0.f;
#pragma unroll 4
    for (int o = o0; o < o0 + rows; ++o) {
        const float wv = W[(int64_t)o * C + i];
        const f32x4 v0 = *reinterpret_cast<const f32x4*>(vin + o * 8);
        const f32x4 v1 = *reinterpret_cast<const f32x4*>(vin + o * 8 + 4);
        acc[0] += wv * v0[0]; acc[1] += wv * v0[1]; acc[2] += wv * v0[2]; acc[3] += wv * v0[3];
        acc[4] += wv * v1[0]; acc[5] += wv * v1[1]; acc[6] += wv * v1[2]; acc[7] += wv * v1[3];
    }
#pragma unroll
    for (int s = 0; s < 8; ++s) red[(sp * 8 + s) * C + i] = acc[s];
    __syncthreads();
    if (sp == 0) {
#pragma unroll
        for (int s = 0; s < 8; ++s) {
            if (s < ns) {
                float t = 0.f;
                for (int q = 0; q < nsplit; ++q) t += red[(q * 8 + s) * C + i];
                epi(i, s, t);
            }
        }
    }
    __syncthreads();
}

__global__ __launch_bounds__(PE_THREADS) void proto_extend_bwd_chain_kernel(
    const float* __restrict__ dWn_part, int n_part, const float* __restrict__ S,
    const float* __restrict__ cnt, const float* __restrict__ hidden, const float* __restrict__ gate,
    const float* __restrict__ W1, const float* __restrict__ W2, const float* __restrict__ Wl,
    const int64_t* __restrict__ labels, const float* __restrict__ gate_mask, int way, int shot, int K,
    int C, float Pf, float* __restrict__ u1 /*dpre1 [B,C]*/, float* __restrict__ u2 /*dz [B,C]*/,
    float* __restrict__ u3 /*dproto [B,C]*/, float* __restrict__ coef /*[B,C]*/) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    float* vin = sm;                 // [C][8]
    float* red = vin + C * 8;        // [nsplit*8][C]  (nsplit*C = 512)
    float* dpool = red + PE_THREADS * 8;   // [C]
    const int grp = blockIdx.x;
    const int n = grp / way, j = grp - n * way;
    const int64_t b0 = (int64_t)grp * shot;
    const int tid = threadIdx.x;

    // dproto[s] = dWn[n, label_j] / shot for every shot (mean backward), if this way owns the row
    const int64_t* lab = labels + (int64_t)n * way;
    const int64_t mine = lab[j];
    bool win = mine >= 0 && mine < K;
    for (int jj = j + 1; jj < way; ++jj) win = win && (lab[jj] != mine);
    for (int c = tid; c < C; c += PE_THREADS) {
        float t = 0.f;
        if (win) {
            const float* src = dWn_part + (((int64_t)n * n_part) * K + mine) * C + c;
            for (int q = 0; q < n_part; ++q) t += src[(int64_t)q * K * C];
        }
        t /= (float)shot;
#pragma unroll
        for (int s = 0; s < 8; ++s) vin[c * 8 + s] = (s == 0) ? t : 0.f;
        for (int s = 0; s < shot; ++s) u3[(b0 + s) * C + c] = t;
    }
    __syncthreads();
    // dpool[i] = sum_o Wl[o][i] * dproto[o]   (identical for every shot of the group)
    dense_cols8(Wl, vin, red, C, 1, [&](int i, int, float v) { dpool[i] = v; });

    for (int s0 = 0; s0 < shot; s0 += 8) {
        const int ns = min(8, shot - s0);
        // dz = dpool * S/(cnt+eps) * g(1-g) * gate_mask      (pool = g*S/(cnt+eps))
        for (int idx = tid; idx < C * 8; idx += PE_THREADS) {
            const int c = idx >> 3, s = idx & 7;
            float dz = 0.f;
            if (s < ns) {
                const int64_t gi = (b0 + s0 + s) * C + c;
                const float g = gate[gi];
                const float den = cnt[b0 + s0 + s] + PE_EPS;
                dz = dpool[c] * S[gi] / den * g * (1.f - g);
                if (gate_mask) dz *= gate_mask[gi];
                u2[gi] = dz;
            }
            vin[idx] = dz;
        }
        __syncthreads();
        // dhidden = W2^T dz ; dpre1 = dhidden * (hidden > 0)
        dense_cols8(W2, vin, red, C, ns, [&](int i, int s, float v) {
            const int64_t gi = (b0 + s0 + s) * C + i;
            const float d = hidden[gi] > 0.f ? v : 0.f;
            u1[gi] = d;
            // stash for the next mat-vec (vin is free: all reads finished before the reduce barrier)
            vin[i * 8 + s] = d;
        });
        // zero the padding columns of vin (s >= ns) -- they still hold dz of this chunk
        for (int idx = tid; idx < C * 8; idx += PE_THREADS)
            if ((idx & 7) >= ns) vin[idx] = 0.f;
        __syncthreads();
        // datt = W1^T dpre1 ; coef = dpool*g/(cnt+eps) + datt/P
        dense_cols8(W1, vin, red, C, ns, [&](int i, int s, float v) {
            const int64_t gi = (b0 + s0 + s) * C + i;
            const float den = cnt[b0 + s0 + s] + PE_EPS;
            coef[gi] = dpool[i] * gate[gi] / den + v / Pf;
        });
    }
}

// ---------------------------------------------------------------------------
// Backward, stage B: parameter gradients (sums over all B support images).
//   blockIdx.y = 0: dW1 = dpre1^T att   1: dW2 = dz^T hidden   2: dWl = dproto^T pool
//   blockIdx.y = 3: dweight[k] = sum_n [k not novel in n] weight_mask[n,k] * dWn[n,k]
// ---------------------------------------------------------------------------
#define WG_ROWS 4
__global__ __launch_bounds__(256) void proto_extend_bwd_params_kernel(
    const float* __restrict__ u1, const float* __restrict__ u2, const float* __restrict__ u3,
    const float* __restrict__ S, const float* __restrict__ hidden, const float* __restrict__ pool,
    const float* __restrict__ dWn_part, int n_part, const int64_t* __restrict__ labels,
    const float* __restrict__ weight_mask, int N, int way, int B, int K, int C, float Pf,
    float* __restrict__ dW1, float* __restrict__ db1, float* __restrict__ dW2, float* __restrict__ db2,
    float* __restrict__ dWl, float* __restrict__ dbl, float* __restrict__ dweight) {
    const int which = blockIdx.y;
    const int tid = threadIdx.x;
    if (which == 3) {
        const int k = blockIdx.x;
        if (k >= K) return;
        for (int c = tid; c < C; c += 256) {
            float t = 0.f;
            for (int n = 0; n < N; ++n) {
                bool novel = false;
                for (int jj = 0; jj < way; ++jj) novel = novel || (labels[(int64_t)n * way + jj] == k);
                if (novel) continue;
                const float* src = dWn_part + (((int64_t)n * n_part) * K + k) * C + c;
                float a = 0.f;
                for (int q = 0; q < n_part; ++q) a += src[(int64_t)q * K * C];
                if (weight_mask) a *= weight_mask[((int64_t)n * K + k) * C + c];
                t += a;
            }
            dweight[(int64_t)k * C + c] = t;
        }
        return;
    }
    const int o0 = blockIdx.x * WG_ROWS;
    if (o0 >= C) return;
    const float* U = which == 0 ? u1 : (which == 1 ? u2 : u3);
    const float* V = which == 0 ? S : (which == 1 ? hidden : pool);
    const float vs = which == 0 ? 1.f / Pf : 1.f;
    float* dW = which == 0 ? dW1 : (which == 1 ? dW2 : dWl);
    float* db = which == 0 ? db1 : (which == 1 ? db2 : dbl);
    for (int i = tid; i < C; i += 256) {
        float acc[WG_ROWS];
#pragma unroll
        for (int r = 0; r < WG_ROWS; ++r) acc[r] = 0.f;
#pragma unroll 4
        for (int b = 0; b < B; ++b) {
            const float v = V[(int64_t)b * C + i];
#pragma unroll
            for (int r = 0; r < WG_ROWS; ++r) acc[r] += U[(int64_t)b * C + o0 + r] * v;
        }
#pragma unroll
        for (int r = 0; r < WG_ROWS; ++r) dW[(int64_t)(o0 + r) * C + i] = acc[r] * vs;
    }
    if (tid < WG_ROWS) {
        float t = 0.f;
        for (int b = 0; b < B; ++b) t += U[(int64_t)b * C + o0 + tid];
        db[o0 + tid] = t;
    }
}

// ---------------------------------------------------------------------------
// C ABI
// ---------------------------------------------------------------------------
static inline bool pos(int64_t v) { return v > 0 && v < ((int64_t)1 << 30); }
static inline bool c_ok(int64_t C) { return C == 64 || C == 128 || C == 256 || C == 512; }

extern "C" int denet_proto_extend_fwd(const float* S, const float* cnt, const float* W1, const float* b1,
                                      const float* W2, const float* b2, const float* Wl, const float* bl,
                                      const float* weight, const int64_t* labels, const float* gate_mask,
                                      const float* weight_mask, int64_t N, int64_t way, int64_t shot,
                                      int64_t K, int64_t C, int64_t P, float* hidden, float* gate,
                                      float* pool, float* Wn, void* stream) {
    if (!S || !cnt || !W1 || !b1 || !W2 || !b2 || !Wl || !bl || !weight || !labels || !hidden || !gate ||
        !pool || !Wn)
        return DENET_E_BADARG;
    if (!pos(N) || !pos(way) || !pos(shot) || !pos(K) || !pos(C) || !pos(P)) return DENET_E_BADARG;
    if (!c_ok(C) || K > 128) return DENET_E_UNSUPPORTED;
    const size_t lds = (size_t)(3 * shot + 1) * C * 4;
    if (lds > 64 * 1024) return DENET_E_UNSUPPORTED;
    dim3 grid((unsigned)(N * way)), block(PE_THREADS);
    hipStream_t st = (hipStream_t)stream;
#define PE_LAUNCH(CJ)                                                                                      \
    hipLaunchKernelGGL(proto_extend_fwd_kernel<CJ>, grid, block, lds, st, S, cnt, W1, b1, W2, b2, Wl, bl,   \
                       weight, labels, gate_mask, weight_mask, (int)way, (int)shot, (int)K, (float)P,       \
                       hidden, gate, pool, Wn)
    switch (C) {
        case 64: PE_LAUNCH(1); break;
        case 128: PE_LAUNCH(2); break;
        case 256: PE_LAUNCH(4); break;
        default: PE_LAUNCH(8); break;
    }
#undef PE_LAUNCH
    DENET_CHECK_LAUNCH();
    return DENET_OK;
}

extern "C" int64_t denet_proto_extend_bwd_workspace(int64_t B, int64_t C) { return 3 * B * C * 4; }

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
    if (!c_ok(C) || K > 128) return DENET_E_UNSUPPORTED;
    const int64_t B = N * way * shot;
    if (ws_bytes < denet_proto_extend_bwd_workspace(B, C)) return DENET_E_WORKSPACE;
    float* u1 = (float*)ws;
    float* u2 = u1 + B * C;
    float* u3 = u2 + B * C;
    hipStream_t st = (hipStream_t)stream;
    const size_t lds = (size_t)(C * 8 + PE_THREADS * 8 + C) * 4;
    hipLaunchKernelGGL(proto_extend_bwd_chain_kernel, dim3((unsigned)(N * way)), dim3(PE_THREADS), lds, st,
                       dWn_part, (int)n_part, S, cnt, hidden, gate, W1, W2, Wl, labels, gate_mask, (int)way,
                       (int)shot, (int)K, (int)C, (float)P, u1, u2, u3, coef);
    DENET_CHECK_LAUNCH();
    const unsigned gx = (unsigned)((C / WG_ROWS) > K ? (C / WG_ROWS) : K);
    hipLaunchKernelGGL(proto_extend_bwd_params_kernel, dim3(gx, 4), dim3(256), 0, st, u1, u2, u3, S, hidden,
                       pool, dWn_part, (int)n_part, labels, weight_mask, (int)N, (int)way, (int)B, (int)K,
                       (int)C, (float)P, dW1, db1, dW2, db2, dWl, dbl, dweight);
    DENET_CHECK_LAUNCH();
    return DENET_OK;
}
