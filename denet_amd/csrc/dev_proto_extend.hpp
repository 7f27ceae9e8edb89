// Device bodies of K2b (backward of gate / projection / extension), shared by the stand-alone
// kernels (proto_extend.hip) and the fused step kernels (head_step.hip).
#pragma once
#include "denet_common.hpp"

#define PE_EPS 1e-3f      // model/common.py:170

// ---------------------------------------------------------------------------
// Backward, stage A: per-group chain of three transposed mat-vecs for up to 8 images at a time
// (vin laid out [C][8] in LDS, zero padded):   y[s][i] = sum_o W[o][i] * vin[o][s]
// 512 threads: thread (i = tid % C, sp = tid / C) owns column i for R = C*C/512 consecutive rows,
// taken in chunks of CH rows.  As in the forward kernel the three mat-vecs are one sequence of
// chunks whose loads run through a 4-slot register ring three chunks ahead of the FMAs.
// ---------------------------------------------------------------------------
#define CH_THREADS 512

// NV = number of live columns of vin (images of the group): only those are read and multiplied.
// The first mat-vec of the chain has ONE (dproto is the same for every shot), the others `shot`.
template <int CH, int NV>
__device__ __forceinline__ void cols_chunk(const float (&wr)[CH], const float* vin /*[C][8]*/, int o0, float (&acc)[8]) {
#pragma unroll
    for (int t = 0; t < CH; ++t) {
        const float* vp = vin + (o0 + t) * 8;
        if constexpr (NV == 1) {
            acc[0] += wr[t] * vp[0];
        } else if constexpr (NV <= 4) {
            const f32x4 v0 = *reinterpret_cast<const f32x4*>(vp);
#pragma unroll
            for (int s = 0; s < NV; ++s) acc[s] += wr[t] * v0[s];
        } else {
            const f32x4 v0 = *reinterpret_cast<const f32x4*>(vp);
            acc[0] += wr[t] * v0[0]; acc[1] += wr[t] * v0[1]; acc[2] += wr[t] * v0[2]; acc[3] += wr[t] * v0[3];
            if constexpr (NV == 5) {
                acc[4] += wr[t] * vp[4];
            } else {
                const f32x4 v1 = *reinterpret_cast<const f32x4*>(vp + 4);
#pragma unroll
                for (int s = 4; s < NV; ++s) acc[s] += wr[t] * v1[s - 4];
            }
        }
    }
}

// sum of the splits' partial results for (column i, image s)
__device__ __forceinline__ float cols_total(const float* red, int C, int i, int s) {
    const int nsplit = CH_THREADS / C;
    float t = 0.f;
    for (int q = 0; q < nsplit; ++q) t += red[(q * 8 + s) * C + i];
    return t;
}

struct PeChainArgs {
    const float *dWn_part, *S, *cnt, *hidden, *gate, *W1, *W2, *Wl, *gate_mask;
    const int64_t* labels;
    int n_part, way, shot, K;
    float Pf;
    float *u1 /*dpre1 [B,C]*/, *u2 /*dz [B,C]*/, *u3 /*dproto [B,C]*/, *coef /*[B,C]*/;
};
#define PE_CHAIN_LDS(C) ((size_t)((C) * 8 + CH_THREADS * 8 + (C) + 3 * (C) * 8) * 4)

// Body for group `grp` = (episode, way); 512 threads; sm: PE_CHAIN_LDS(C) bytes.
template <int C, int CH, int NS>
__device__ __forceinline__ void proto_extend_bwd_chain_body_ns(const PeChainArgs& a, int grp, float* sm) {
    const float* __restrict__ dWn_part = a.dWn_part;
    const float* __restrict__ S = a.S;
    const float* __restrict__ cnt = a.cnt;
    const float* __restrict__ hidden = a.hidden;
    const float* __restrict__ gate = a.gate;
    const float* __restrict__ W1 = a.W1;
    const float* __restrict__ W2 = a.W2;
    const float* __restrict__ Wl = a.Wl;
    const float* __restrict__ gate_mask = a.gate_mask;
    const int64_t* __restrict__ labels = a.labels;
    const int n_part = a.n_part, way = a.way, shot = a.shot, K = a.K;
    const float Pf = a.Pf;
    float* __restrict__ u1 = a.u1;
    float* __restrict__ u2 = a.u2;
    float* __restrict__ u3 = a.u3;
    float* __restrict__ coef = a.coef;
    constexpr int R = C * C / CH_THREADS;        // rows per thread
    constexpr int NCH = R / CH, T = 3 * NCH;     // chunks per mat-vec, chunks in all
    float* vin = sm;                 // [C][8]
    float* red = vin + C * 8;        // [nsplit*8][C]  (nsplit*C = 512)
    float* dpool = red + CH_THREADS * 8;   // [C]
    float* pdz = dpool + C;          // [C][8]  S/(cnt+eps) * g(1-g) * gate_mask   (factor of dz)
    float* phid = pdz + C * 8;       // [C][8]  hidden > 0
    float* pco = phid + C * 8;       // [C][8]  g/(cnt+eps)                          (factor of coef)
    const int n = grp / way, j = grp - n * way;
    const int64_t b0 = (int64_t)grp * shot;
    const int tid = threadIdx.x;
    const int ci = tid % C, sp = tid / C;
    const int ns = shot;

    float ring[4][CH];
    auto issue = [&](int t, float (&dst)[CH]) {
        const float* W = (t < NCH) ? Wl : (t < 2 * NCH ? W2 : W1);
        const int ch = t % NCH;
        const float* p = W + (int64_t)(sp * R + ch * CH) * C + ci;
#pragma unroll
        for (int q = 0; q < CH; ++q) dst[q] = p[(int64_t)q * C];
    };
    STAMP(0);
#pragma unroll
    for (int t = 0; t < 3; ++t)
        if (t < T) issue(t, ring[t]);
    // Everything the layer boundaries need from global memory is independent of the chain: fetch it
    // now, under the weight stream, so that a boundary costs LDS reads instead of a memory round trip.
    for (int idx = tid; idx < C * 8; idx += CH_THREADS) {
        const int c = idx >> 3, s = idx & 7;
        float fz = 0.f, fh = 0.f, fc = 0.f;
        if (s < ns) {
            const int64_t gi = (b0 + s) * C + c;
            const float g = gate[gi];
            const float den = cnt[b0 + s] + PE_EPS;
            fz = S[gi] / den * g * (1.f - g);
            if (gate_mask) fz *= gate_mask[gi];
            fh = hidden[gi] > 0.f ? 1.f : 0.f;
            fc = g / den;
        }
        pdz[idx] = fz;
        phid[idx] = fh;
        pco[idx] = fc;
    }

    // dproto[s] = dWn[n, label_j] / shot for every shot (mean backward), if this way owns the row
    const int64_t* lab = labels + (int64_t)n * way;
    const int64_t mine = lab[j];
    bool win = mine >= 0 && mine < K;
    for (int jj = j + 1; jj < way; ++jj) win = win && (lab[jj] != mine);
    {
        // sum of the row's slabs: the 512/C thread groups take every (512/C)-th slab, all loads in flight
        constexpr int NSP = CH_THREADS / C;
        float t = 0.f;
        if (win) {
            const float* src = dWn_part + (((int64_t)n * n_part) * K + mine) * C + ci;
#pragma unroll 16
            for (int q = sp; q < n_part; q += NSP) t += src[(int64_t)q * K * C];
        }
        red[sp * C + ci] = t;
        __syncthreads();
        if (tid < C) {
            float tt = 0.f;
#pragma unroll
            for (int q = 0; q < NSP; ++q) tt += red[q * C + tid];
            tt /= (float)shot;
#pragma unroll
            for (int s = 0; s < 8; ++s) vin[tid * 8 + s] = (s == 0) ? tt : 0.f;
            for (int s = 0; s < shot; ++s) u3[(b0 + s) * C + tid] = tt;
        }
    }
    __syncthreads();

    STAMP(1);
    float acc[8];
#pragma unroll
    for (int s = 0; s < 8; ++s) acc[s] = 0.f;
#pragma unroll
    for (int t = 0; t < T; ++t) {
        if (t + 3 < T) issue(t + 3, ring[(t + 3) & 3]);
        const int mv = t / NCH, ch = t % NCH;
        if (mv == 0) cols_chunk<CH, 1>(ring[t & 3], vin, sp * R + ch * CH, acc);
        else cols_chunk<CH, NS>(ring[t & 3], vin, sp * R + ch * CH, acc);
        if (ch == NCH - 1) {
            if (mv == 0) { STAMP(2); } else if (mv == 1) { STAMP(4); } else { STAMP(6); }
#pragma unroll
            for (int s = 0; s < 8; ++s) { red[(sp * 8 + s) * C + ci] = acc[s]; acc[s] = 0.f; }
            __syncthreads();
            if (mv == 0) {
                // dpool[i] = sum_o Wl[o][i] * dproto[o]   (identical for every shot of the group)
                for (int i = tid; i < C; i += CH_THREADS) dpool[i] = cols_total(red, C, i, 0);
                __syncthreads();
                // dz = dpool * S/(cnt+eps) * g(1-g) * gate_mask      (pool = g*S/(cnt+eps))
                for (int idx = tid; idx < C * 8; idx += CH_THREADS) {
                    const int c = idx >> 3, s = idx & 7;
                    const float dz = dpool[c] * pdz[idx];
                    if (s < ns) u2[(b0 + s) * C + c] = dz;
                    vin[idx] = dz;
                }
            } else if (mv == 1) {
                // dhidden = W2^T dz ; dpre1 = dhidden * (hidden > 0)
                for (int idx = tid; idx < C * 8; idx += CH_THREADS) {
                    const int c = idx >> 3, s = idx & 7;
                    float d = 0.f;
                    if (s < ns) {
                        d = phid[idx] > 0.f ? cols_total(red, C, c, s) : 0.f;
                        u1[(b0 + s) * C + c] = d;
                    }
                    vin[idx] = d;
                }
            } else {
                // datt = W1^T dpre1 ; coef = dpool*g/(cnt+eps) + datt/P
                for (int idx = tid; idx < C * ns; idx += CH_THREADS) {
                    const int s = idx / C, c = idx - s * C;
                    coef[(b0 + s) * C + c] = dpool[c] * pco[c * 8 + s] + cols_total(red, C, c, s) / Pf;
                }
            }
            __syncthreads();
            if (mv == 0) { STAMP(3); } else if (mv == 1) { STAMP(5); } else { STAMP(7); }
        }
    }
}

// dispatch on the number of images per group (shot <= 8)
template <int C, int CH>
__device__ __forceinline__ void proto_extend_bwd_chain_body(const PeChainArgs& a, int grp, float* sm) {
    if (a.shot == 1) proto_extend_bwd_chain_body_ns<C, CH, 1>(a, grp, sm);
    else if (a.shot <= 3) proto_extend_bwd_chain_body_ns<C, CH, 3>(a, grp, sm);
    else if (a.shot <= 5) proto_extend_bwd_chain_body_ns<C, CH, 5>(a, grp, sm);
    else proto_extend_bwd_chain_body_ns<C, CH, 8>(a, grp, sm);
}

template <int C, int CH>
__global__ __launch_bounds__(CH_THREADS) void proto_extend_bwd_chain_kernel(PeChainArgs a) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    proto_extend_bwd_chain_body<C, CH>(a, blockIdx.x, sm);
}

// dWn[n,k,c] = sum over position tiles of the partial slabs K3b wrote (fixed order: deterministic)
static __global__ __launch_bounds__(256) void reduce_parts_kernel(const float* __restrict__ part, int n_part, int64_t per_n,
                                                          float* __restrict__ out) {
    const int n = blockIdx.y;
    const int64_t i4 = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 4;
    if (i4 >= per_n) return;
    const float* src = part + (int64_t)n * n_part * per_n + i4;
    f32x4 a = {0.f, 0.f, 0.f, 0.f};
#pragma unroll 9
    for (int q = 0; q < n_part; ++q) {
        const f32x4 v = *reinterpret_cast<const f32x4*>(src + (int64_t)q * per_n);
        a[0] += v[0]; a[1] += v[1]; a[2] += v[2]; a[3] += v[3];
    }
    *reinterpret_cast<f32x4*>(out + (int64_t)n * per_n + i4) = a;
}

// ---------------------------------------------------------------------------
// Backward, stage B: parameter gradients (sums over all B support images).
//   blockIdx.y = 0: dW1 = dpre1^T att   1: dW2 = dz^T hidden   2: dWl = dproto^T pool
//   blockIdx.y = 3: dweight[k] = sum_n [k not novel in n] weight_mask[n,k] * dWn[n,k]
// ---------------------------------------------------------------------------
#define WG_ROWS 1
struct PeParamsArgs {
    // u1 = dpre1 [B,C]; in the sliced chain (dh_part != nullptr) it is not materialised: role 0 forms
    // column o0 of it from the split-K partials dh_part [B][C/16][C] and the relu mask of `hidden`
    const float *u1, *u2, *u3, *S, *hidden, *pool, *dWn_part, *weight_mask;
    const int64_t* labels;
    int n_part, N, way, B, K, C;
    float Pf;
    float *dW1, *db1, *dW2, *db2, *dWl, *dbl, *dweight;
    const float* dh_part = nullptr;
    DropGen wdrop = {0u, 0u, 0u, 1.f, 0};   // used where weight_mask is null (train-mode entry points)
};
// blocks per role: roles 0..2 need C/WG_ROWS blocks, role 3 needs K * C/16
__host__ __device__ __forceinline__ int pe_params_gx(int K, int C) {
    const int a = C / WG_ROWS, b = K * (C >> 4);
    return a > b ? a : b;
}

// Body for (which, bx); 256 threads; smem: 4 KB (256 float4).
__device__ __forceinline__ void proto_extend_bwd_params_body(const PeParamsArgs& a, int which, int bx, float* smem) {
    const float* __restrict__ u1 = a.u1;
    const float* __restrict__ u2 = a.u2;
    const float* __restrict__ u3 = a.u3;
    const float* __restrict__ S = a.S;
    const float* __restrict__ hidden = a.hidden;
    const float* __restrict__ pool = a.pool;
    const float* __restrict__ dWn_part = a.dWn_part;
    const float* __restrict__ weight_mask = a.weight_mask;
    const int64_t* __restrict__ labels = a.labels;
    const int n_part = a.n_part, N = a.N, way = a.way, B = a.B, K = a.K, C = a.C;
    const float Pf = a.Pf;
    float* __restrict__ dW1 = a.dW1;
    float* __restrict__ db1 = a.db1;
    float* __restrict__ dW2 = a.dW2;
    float* __restrict__ db2 = a.db2;
    float* __restrict__ dWl = a.dWl;
    float* __restrict__ dbl = a.dbl;
    float* __restrict__ dweight = a.dweight;
    const int tid = threadIdx.x;
    if (which == 3) {
        // dweight[k, 16-column chunk] = sum over (n, tile) of the partial slabs, novel rows and dropped
        // entries excluded.  Block = (k, chunk): K * C/16 blocks keep every CU streaming at large batches;
        // the 256 threads are 4 float4 columns x 64 groups that split the (n, tile) pairs, so every
        // thread has its loads in flight together; the groups are then summed through LDS in a fixed order.
        f32x4* redw = reinterpret_cast<f32x4*>(smem);
        const int nchunk = C >> 4;
        const int k = bx / nchunk, chunk = bx - k * nchunk;
        if (k >= K) return;
        const int c4 = tid & 3, grp = tid >> 2;
        const int col = chunk * 16 + c4 * 4;
        const int pairs = N * n_part;
        f32x4 t = {0.f, 0.f, 0.f, 0.f};
#pragma unroll 8
        for (int p = grp; p < pairs; p += 64) {
            const int n = p / n_part, q = p - n * n_part;
            // all loads unconditional and independent of each other (labels, slab, mask): the novel
            // rows are dropped by a select, not by a branch around the loads
            f32x4 v = *reinterpret_cast<const f32x4*>(dWn_part + (((int64_t)n * n_part + q) * K + k) * C + col);
            bool novel = false;
            for (int jj = 0; jj < way; ++jj) novel = novel || (labels[(int64_t)n * way + jj] == k);
            if (weight_mask) {
                const f32x4 m = *reinterpret_cast<const f32x4*>(weight_mask + ((int64_t)n * K + k) * C + col);
                v[0] *= m[0]; v[1] *= m[1]; v[2] *= m[2]; v[3] *= m[3];
            } else if (a.wdrop.on) {
                const f32x4 m = drop_mult4(a.wdrop, DENET_DROP_WEIGHT, (unsigned)((((int64_t)n * K + k) * C + col) >> 2));
                v[0] *= m[0]; v[1] *= m[1]; v[2] *= m[2]; v[3] *= m[3];
            }
            const float keep = novel ? 0.f : 1.f;
            t[0] = fmaf(keep, v[0], t[0]); t[1] = fmaf(keep, v[1], t[1]);
            t[2] = fmaf(keep, v[2], t[2]); t[3] = fmaf(keep, v[3], t[3]);
        }
        redw[tid] = t;
        __syncthreads();
        if (tid < 4) {
            f32x4 r = redw[tid];
            for (int g2 = 1; g2 < 64; ++g2) {
                const f32x4 v = redw[g2 * 4 + tid];
                r[0] += v[0]; r[1] += v[1]; r[2] += v[2]; r[3] += v[3];
            }
            *reinterpret_cast<f32x4*>(dweight + (int64_t)k * C + chunk * 16 + tid * 4) = r;
        }
        return;
    }
    const int o0 = bx * WG_ROWS;
    if (o0 >= C) return;
    const float* U = which == 0 ? u1 : (which == 1 ? u2 : u3);
    const float* V = which == 0 ? S : (which == 1 ? hidden : pool);
    const float vs = which == 0 ? 1.f / Pf : 1.f;
    float* dW = which == 0 ? dW1 : (which == 1 ? dW2 : dWl);
    float* db = which == 0 ? db1 : (which == 1 ? db2 : dbl);
    if (which == 0 && a.dh_part != nullptr) {
        // sliced chain: dpre1[b][o0] for 1024 images at a time in LDS, then the same row product
        const float* __restrict__ dh_part = a.dh_part;
        const int nsl = C >> 4;
        float acc = 0.f, sdb = 0.f;                       // C <= 256: one output column per thread
        for (int bb = 0; bb < B; bb += 1024) {
            const int nb = min(1024, B - bb);
            __syncthreads();
            for (int b = tid; b < nb; b += 256) {
                const float* dp = dh_part + ((int64_t)(bb + b) * nsl) * C + o0;
                float t = 0.f;
                for (int q0 = 0; q0 < nsl; q0 += 16) {       // clamped unconditional loads + select
                    float v[16];
#pragma unroll
                    for (int u = 0; u < 16; ++u) v[u] = dp[(int64_t)min(q0 + u, nsl - 1) * C];
#pragma unroll
                    for (int u = 0; u < 16; ++u) t += (q0 + u < nsl) ? v[u] : 0.f;
                }
                smem[b] = hidden[(int64_t)(bb + b) * C + o0] > 0.f ? t : 0.f;
            }
            __syncthreads();
            if (tid < C) {
#pragma unroll 16
                for (int b = 0; b < nb; ++b) {
                    const float u = smem[b];
                    acc = fmaf(u, V[(int64_t)(bb + b) * C + tid], acc);
                    sdb += u;
                }
            }
        }
        if (tid < C) dW[(int64_t)o0 * C + tid] = acc * vs;
        if (tid == 0) db[o0] = sdb;
        return;
    }
    // One output row per block (WG_ROWS == 1).  U[b, o0] is the same address for every lane, so the
    // bias gradient db[o0] = sum_b U[b, o0] falls out of the same loads; 16 images in flight at a time.
    for (int i = tid; i < C; i += 256) {
        float acc = 0.f, sdb = 0.f;
#pragma unroll 16
        for (int b = 0; b < B; ++b) {
            const float u = U[(int64_t)b * C + o0];
            acc = fmaf(u, V[(int64_t)b * C + i], acc);
            sdb += u;
        }
        dW[(int64_t)o0 * C + i] = acc * vs;
        if (i == 0) db[o0] = sdb;
    }
}

static __global__ __launch_bounds__(256) void proto_extend_bwd_params_kernel(PeParamsArgs a) {
    __shared__ __attribute__((aligned(16))) float smem[1024];
    proto_extend_bwd_params_body(a, blockIdx.y, blockIdx.x, smem);
}

