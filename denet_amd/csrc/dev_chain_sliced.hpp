// Sliced chain (fused step): the gate / projection stage K2 and its backward K2b as kernels that use
// the WHOLE chip instead of one CU per (episode, way) group.
//
// The stage is three dependent C x C mat-vecs per support image.  One workgroup per group has to pull
// all three weight matrices (768 KB at C = 256) through a single CU, which takes 14 us forward and
// 24 us backward at batch 8 whatever the FMA rate.  Here every mat-vec is cut into C/16 slices of 16
// rows (or columns) that run on different CUs, and the all-to-all between two mat-vecs is a launch
// boundary that the producer / consumer kernels on either side absorb:
//
//   forward   F1  pooling block (image, 16 channels):  pre1_part = W1[:, slice] att[slice]      (split-K)
//             F2  block (group, 16 rows q):  hidden = relu(b1 + sum pre1_part)   (every block, redundantly)
//                                            gate[q], pool[q] = rows q of W2                  (row slice)
//                                            proto_part = Wl[:, q] mean_s pool[q]               (split-K)
//             F3  classifier block:  Wn[label] = bl + sum proto_part  while staging Wn in LDS
//   backward  B1  dWn slabs (unchanged)
//             B2  block (group, 16 columns i):  dproto = sum of the label row's slabs / shot
//                                            dpool[i] = Wl[:, i]^T dproto                       (column slice)
//                                            dz[i] = dpool[i] * dgate[i]
//                                            dh_part = W2[i, :]^T dz[i]                          (split-K)
//             B3  dFs block (image, 16 channels):  dpre1 = relu'(hidden) * sum dh_part
//                                            datt[c] = W1[:, c]^T dpre1 ; coef ; dFs = mask * coef
// Every block reads 16-48 KB of weights instead of 768 KB, every cross-block sum has a fixed order
// (deterministic), and nothing spins on another block: the exchanges are ordinary kernel boundaries.
#pragma once
#include "denet_common.hpp"

#define SC_THREADS 256
#define SC_SL 16                 // slice width
#define SC_EPS 1e-3f             // model/common.py:170

// ---------------------------------------------------------------------------
// F2: gate layer 2 (row slice) + projection (split-K), model/common.py:225-232,192-196,319
// ---------------------------------------------------------------------------
struct GateProjArgs {
    const float *pre1_part /*[B][NSL][C]*/, *S, *cnt, *b1, *W2, *b2, *Wl, *gate_mask;
    int shot;
    float *hidden, *gate, *pool /*[B][C]*/, *proto_part /*[G][NSL][C]*/;
    DropGen gdrop;        // used where gate_mask is null (train-mode entry points); zero = off
};
#define SC_GP_LDS(C) ((size_t)(8 * (C) + 16) * 4)

// Block (grp, sl); 256 threads; C = 64 * CJ; NS >= shot images handled together.
template <int CJ, int NS>
__device__ __forceinline__ void gate_proj_body(const GateProjArgs& a, int grp, int sl, float* sm) {
    constexpr int C = 64 * CJ, NSL = C / SC_SL;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int ns = a.shot;
    const int64_t b0 = (int64_t)grp * ns;
    float* xa = sm;                  // hidden [NS][C]
    float* pm = xa + NS * C;         // mean_s pool of the slice [16]

    // weights first: they depend on nothing
    const int wrow = wave * 4 + (lane >> 4), q = SC_SL * sl + wrow, wcol = (lane & 15) * 4;
    f32x4 w2v[CJ];
#pragma unroll
    for (int jj = 0; jj < CJ; ++jj)
        w2v[jj] = *reinterpret_cast<const f32x4*>(a.W2 + (int64_t)q * C + wcol + 64 * jj);
    const int o = min(tid, C - 1);
    const bool live = tid < C;
    f32x4 wlv[4];
#pragma unroll
    for (int m = 0; m < 4; ++m) wlv[m] = *reinterpret_cast<const f32x4*>(a.Wl + (int64_t)o * C + SC_SL * sl + 4 * m);
    // per-row scalars of the gate epilogue (same address for the 16 lanes of a row: one broadcast load)
    const float b2q = a.b2[q];
    float sS[NS], sG[NS], sC[NS];
#pragma unroll
    for (int s = 0; s < NS; ++s) {
        const int64_t b = b0 + min(s, ns - 1);
        sS[s] = a.S[b * C + q];
        sG[s] = a.gate_mask ? a.gate_mask[b * C + q]
                            : (a.gdrop.on ? drop_mult1(a.gdrop, DENET_DROP_GATE, (unsigned)(b * C + q)) : 1.f);
        sC[s] = a.cnt[b] + SC_EPS;
    }
    // hidden = relu(b1 + W1 att): sum of the pooling kernel's split-K partials, every image of the group
    {
        const float bias = a.b1[o];
        float hs[NS];
#pragma unroll
        for (int s = 0; s < NS; ++s) {
            const float* pp = a.pre1_part + ((b0 + min(s, ns - 1)) * NSL) * C + o;
            float t = 0.f;
#pragma unroll
            for (int cb = 0; cb < NSL; ++cb) t += pp[(int64_t)cb * C];
            hs[s] = fmaxf(t + bias, 0.f);
        }
        if (live) {
#pragma unroll
            for (int s = 0; s < NS; ++s) {
                xa[s * C + o] = s < ns ? hs[s] : 0.f;
                if (sl == 0 && s < ns) a.hidden[(b0 + s) * C + o] = hs[s];
            }
        }
    }
    __syncthreads();

    // rows q of layer 2: 16 lanes per row, 16-byte pieces along the row
    float z[NS];
#pragma unroll
    for (int s = 0; s < NS; ++s) {
        float t = 0.f;
#pragma unroll
        for (int jj = 0; jj < CJ; ++jj) {
            const f32x4 xv = *reinterpret_cast<const f32x4*>(xa + s * C + wcol + 64 * jj);
            t += w2v[jj][0] * xv[0] + w2v[jj][1] * xv[1] + w2v[jj][2] * xv[2] + w2v[jj][3] * xv[3];
        }
        z[s] = row16_sum(t);
    }
    if ((lane & 15) == 0) {
        // g = sigmoid(dropout(W2 hidden + b2)); pool = g * S / (cnt + eps)
        float tsum = 0.f;
#pragma unroll
        for (int s = 0; s < NS; ++s) {
            if (s < ns) {
                const float v = (z[s] + b2q) * sG[s];
                const float g = 1.f / (1.f + expf(-v));
                const float pl = g * sS[s] / sC[s];
                a.gate[(b0 + s) * C + q] = g;
                a.pool[(b0 + s) * C + q] = pl;
                tsum += pl;
            }
        }
        pm[wrow] = tsum / (float)ns;            // mean over the shots (model/common.py:319)
    }
    __syncthreads();
    // projection, split-K over this block's 16 inputs:  proto_part[r] = sum_t Wl[r][16 sl + t] * pm[t]
    if (live) {
        float t = 0.f;
#pragma unroll
        for (int e = 0; e < 16; ++e) t = fmaf(wlv[e >> 2][e & 3], pm[e], t);
        a.proto_part[((int64_t)grp * NSL + sl) * C + o] = t;
    }
}

template <int CJ, int NS>
__global__ __launch_bounds__(SC_THREADS) void gate_proj_kernel(GateProjArgs a) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    constexpr int NSL = 64 * CJ / SC_SL;
    gate_proj_body<CJ, NS>(a, blockIdx.x / NSL, blockIdx.x % NSL, sm);
}

// ---------------------------------------------------------------------------
// B2: dproto -> dpool (column slice of Wl^T) -> dz -> split-K partial of W2^T dz
// ---------------------------------------------------------------------------
struct ChainMidArgs {
    const float *dWn_part /*[N][n_part][K][C]*/, *S, *cnt, *gate, *gate_mask, *Wl, *W2;
    const int64_t* labels;
    int n_part, way, shot, K;
    float *u2 /*dz [B][C]*/, *u3 /*dproto [B][C]*/, *dpool /*[G][C]*/, *dh_part /*[B][NSL][C]*/;
    DropGen gdrop;        // as in GateProjArgs: regenerates the forward's gate multipliers
};
#define SC_CM_LDS ((size_t)(64 + 16 * 8) * 4)

template <int CJ, int NS>
__device__ __forceinline__ void chain_mid_bwd_body(const ChainMidArgs& a, int grp, int sl, float* sm) {
    constexpr int C = 64 * CJ, NSL = C / SC_SL;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int ns = a.shot, way = a.way, K = a.K;
    const int n = grp / way, j = grp - n * way;
    const int64_t b0 = (int64_t)grp * ns;
    float* red = sm;                 // [4][16]
    float* dzs = red + 64;           // [8][16]
    const int o = min(tid, C - 1);
    const bool live = tid < C;
    const int i0 = SC_SL * sl;

    // weights first
    f32x4 wlv[4];
#pragma unroll
    for (int m = 0; m < 4; ++m) wlv[m] = *reinterpret_cast<const f32x4*>(a.Wl + (int64_t)o * C + i0 + 4 * m);
    float w2c[16];
#pragma unroll
    for (int t = 0; t < 16; ++t) w2c[t] = a.W2[(int64_t)(i0 + t) * C + o];
    // factor of dz for (image s, column i0 + t), thread idx = s * 16 + t:
    //   pool = g * S/(cnt+eps), g = sigmoid(v * gate_mask)  ->  dv = dpool * S/(cnt+eps) * g(1-g) * gate_mask
    const int fs = min(tid >> 4, ns - 1), ft = tid & 15;
    float fz;
    {
        const int64_t gi = (b0 + fs) * C + i0 + ft;
        const float g = a.gate[gi];
        const float den = a.cnt[b0 + fs] + SC_EPS;
        fz = a.S[gi] / den * g * (1.f - g);
        if (a.gate_mask) fz *= a.gate_mask[gi];
        else if (a.gdrop.on) fz *= drop_mult1(a.gdrop, DENET_DROP_GATE, (unsigned)gi);
    }
    // dproto = dWn[n, label_j] / shot for every shot (mean backward), if this way owns the row
    const int64_t* lab = a.labels + (int64_t)n * way;
    const int64_t mine = lab[j];
    bool win = mine >= 0 && mine < K;
    for (int jj = j + 1; jj < way; ++jj) win = win && (lab[jj] != mine);
    float dp = 0.f;
    {
        const int64_t row = win ? mine : 0;
        const float* src = a.dWn_part + (((int64_t)n * a.n_part) * K + row) * C + o;
        // 16 slabs at a time, clamped unconditional loads + select: a remainder loop would pay one
        // memory round trip per slab
        for (int q0 = 0; q0 < a.n_part; q0 += 16) {
            float v[16];
#pragma unroll
            for (int u = 0; u < 16; ++u) v[u] = src[(int64_t)min(q0 + u, a.n_part - 1) * K * C];
#pragma unroll
            for (int u = 0; u < 16; ++u) dp += (q0 + u < a.n_part) ? v[u] : 0.f;
        }
        dp = (win && live) ? dp / (float)ns : 0.f;
    }
    if (sl == 0 && live)
        for (int s = 0; s < ns; ++s) a.u3[(b0 + s) * C + o] = dp;

    // dpool[i0 + t] = sum_o Wl[o][i0 + t] * dproto[o]
#pragma unroll
    for (int t = 0; t < 16; ++t) {
        const float v = wave_sum(wlv[t >> 2][t & 3] * dp);
        if (lane == t) red[wave * 16 + t] = v;
    }
    __syncthreads();
    if (tid < 16 * NS) {
        const float dpl = ((red[ft] + red[16 + ft]) + red[32 + ft]) + red[48 + ft];
        const float dz = dpl * fz;
        const bool ok = (tid >> 4) < ns;
        dzs[tid] = ok ? dz : 0.f;
        if (ok) a.u2[(b0 + fs) * C + i0 + ft] = dz;
        if (tid < 16) a.dpool[(int64_t)grp * C + i0 + ft] = dpl;
    }
    __syncthreads();
    // dh_part[s][o] = sum_t W2[i0 + t][o] * dz[s][t]
    if (live) {
#pragma unroll
        for (int s = 0; s < NS; ++s) {
            if (s < ns) {
                float t = 0.f;
#pragma unroll
                for (int e = 0; e < 16; ++e) t = fmaf(w2c[e], dzs[s * 16 + e], t);
                a.dh_part[((b0 + s) * NSL + sl) * C + o] = t;
            }
        }
    }
}
