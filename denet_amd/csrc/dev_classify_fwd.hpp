// Device code of K3 forward: operand loads, Wn staging / assembly, and the body of the eight-/four-waves-per-tile
// kernel (classify.hip).  A header of its own so that other launches can host classifier blocks as a block role;
// the round-3 experiments that did (base rows inside the pooling or the gate / projection launch, F2 + F3 merged
// behind arrival counters) all measured slower than the launch boundary: profiles/r3_split_classifier_stamps.txt.
#pragma once
#include "denet_common.hpp"
#include "dev_classify.hpp"

// Two adjacent positions of one channel row.  Every load is UNCONDITIONAL (out-of-range lanes read a
// clamped, valid address and are zeroed by a select afterwards): a load inside a branch makes the
// compiler drain vmcnt at the join, which turns a burst of 32 independent loads into 32 round trips.
template <typename T, bool FAST>
__device__ __forceinline__ void load2(const T* row, int pa, int P, float& a, float& b);
template <>
__device__ __forceinline__ void load2<float, true>(const float* row, int pa, int P, float& a, float& b) {
    const f32x2_u v = *reinterpret_cast<const f32x2_u*>(row + pa);
    a = v[0]; b = v[1];
}
template <>
__device__ __forceinline__ void load2<float, false>(const float* row, int pa, int P, float& a, float& b) {
    const float x = row[min(pa, P - 1)], y = row[min(pa + 1, P - 1)];
    a = pa < P ? x : 0.f;
    b = pa + 1 < P ? y : 0.f;
}
template <>
__device__ __forceinline__ void load2<unsigned short, true>(const unsigned short* row, int pa, int P, float& a,
                                                            float& b) {
    const u16x2_u v = *reinterpret_cast<const u16x2_u*>(row + pa);
    a = bf16_to_f32(v[0]); b = bf16_to_f32(v[1]);
}
template <>
__device__ __forceinline__ void load2<unsigned short, false>(const unsigned short* row, int pa, int P, float& a,
                                                             float& b) {
    const unsigned short x = row[min(pa, P - 1)], y = row[min(pa + 1, P - 1)];
    a = pa < P ? bf16_to_f32(x) : 0.f;
    b = pa + 1 < P ? bf16_to_f32(y) : 0.f;
}

// Branch-free form for every tile, partial ones included: ONE 8-byte (4-byte for bf16) load at a clamped
// position (raw2_load), the pair shifted / zeroed by selects at the consumer (raw2_get).  (A `fast ? wide :
// narrow` branch around the loads of a software-pipelined loop makes the compiler drain vmcnt(0) at the join:
// the prefetch then runs serially.)
template <typename T> struct Raw2;
template <> struct Raw2<float> { typedef f32x2_u type; };
template <> struct Raw2<unsigned short> { typedef unsigned type; };
__device__ __forceinline__ f32x2_u raw2_load(const float* row, int pc) { return *reinterpret_cast<const f32x2_u*>(row + pc); }
__device__ __forceinline__ unsigned raw2_load(const unsigned short* row, int pc) {
    const u16x2_u v = *reinterpret_cast<const u16x2_u*>(row + pc);
    return (unsigned)v[0] | ((unsigned)v[1] << 16);
}
// sh = pa - pc: 0 = both positions valid, 1 = the second loaded element is position pa (the last position of an
// odd P), >= 2 = the lane's columns lie past P.  Output columns depend on their own B column only and columns
// past P are never stored, so only the shifted case needs a select.
__device__ __forceinline__ void raw2_get(const f32x2_u v, int sh, float& a, float& b) {
    a = sh == 1 ? v[1] : v[0];
    b = v[1];
}
__device__ __forceinline__ void raw2_get(const unsigned v, int sh, float& a, float& b) {
    const float v0 = __uint_as_float(v << 16), v1 = __uint_as_float(v & 0xffff0000u);
    a = sh == 1 ? v1 : v0;
    b = v1;
}

// BUILD (fused step, sliced chain): Wn[n] does not exist yet.  The block assembles it in LDS while it
// stages the tile -- base rows = weight * weight_mask[n] (dropout, model/common.py:326), the rows the
// episode's ways overwrite = bl + sum of the projection's split-K partials (model/common.py:327-334,
// a later way wins a duplicate label) -- and the block of tile 0 writes it out for the backward pass.
struct WnBuild {
    const float *weight /*[K][C]*/, *weight_mask /*[N][K][C] or null*/, *proto_part /*[N*way][C/16][C]*/, *bl;
    float* Wn_out /*[N][K][C]*/;
    DropGen wdrop;        // used where weight_mask is null (train-mode entry points); zero = off
};

// Stage Wn[n] -> LDS [K][C+1] (16-byte loads, all of a thread's loads issued before the first store); BUILD:
// assemble it on the way (see WnBuild).  NT threads, NSL = C / 16, SU = 16-byte loads per thread and pass.
template <bool BUILD, int NT, int NSL, int SU>
__device__ __forceinline__ void stage_wn(float* sm, const float* __restrict__ Wn, int64_t wn_stride,
                                         const int64_t* __restrict__ labels, int way, int K, int C, int n,
                                         const WnBuild& wb) {
    constexpr int NW = NT / 64;
    const int tid = threadIdx.x;
    const int ldw = C + 1;
    // the rows this episode's ways overwrite: thread group jw = tid / 64 sums the partials of way jw
    // (16-byte column pieces, all loads in flight together, fixed order)
    f32x4 pv = {0.f, 0.f, 0.f, 0.f};
    int prow = -1;
    if constexpr (BUILD) {
        const int jw = min(tid >> 6, way - 1), c4 = min(tid & 63, (C >> 2) - 1);
        const int64_t* lab = labels + (int64_t)n * way;
        const int64_t mine = lab[jw];
        bool win = (tid >> 6) < way && (tid & 63) < (C >> 2) && mine >= 0 && mine < K;
        for (int jj = jw + 1; jj < way; ++jj) win = win && (lab[jj] != mine);
        const float* pp = wb.proto_part + ((int64_t)(n * way + jw) * NSL) * C + c4 * 4;
        f32x4 t[NSL];
#pragma unroll
        for (int q = 0; q < NSL; ++q) t[q] = *reinterpret_cast<const f32x4*>(pp + (int64_t)q * C);
        f32x4 acc4 = t[0];
#pragma unroll
        for (int q = 1; q < NSL; ++q) { acc4[0] += t[q][0]; acc4[1] += t[q][1]; acc4[2] += t[q][2]; acc4[3] += t[q][3]; }
        const f32x4 b4 = *reinterpret_cast<const f32x4*>(wb.bl + c4 * 4);
        pv = f32x4{acc4[0] + b4[0], acc4[1] + b4[1], acc4[2] + b4[2], acc4[3] + b4[3]};
        prow = win ? (int)mine * ldw + c4 * 4 : -1;
    }
    // stage Wn[n] -> LDS [K][C+1]: 16-byte loads, all of a thread's loads issued before the first store
    {
        const float* src = BUILD ? wb.weight : Wn + (int64_t)n * wn_stride;
        const float* msk = (BUILD && wb.weight_mask) ? wb.weight_mask + (int64_t)n * K * C : nullptr;
        const int64_t* lab = labels + (int64_t)n * way;
        const int nv = (K * C) >> 2;
        // loads per thread and pass: the whole matrix in one burst up to K = 96 (a pass is one L2 round trip)
        const int cshift = 31 - __builtin_clz((unsigned)C);
#pragma unroll 1
        for (int base = 0; base < nv; base += NT * SU) {
            f32x4 v[SU];
#pragma unroll
            for (int u = 0; u < SU; ++u) {
                const int i4 = base + u * NT + tid;
#ifdef DENET_DIAG_NO_WN
                v[u] = f32x4{1.f, 2.f, 3.f, 4.f};
#else
                v[u] = *reinterpret_cast<const f32x4*>(src + (int64_t)min(i4, nv - 1) * 4);
#endif
            }
            if (BUILD && msk != nullptr) {
                f32x4 m[SU];
#pragma unroll
                for (int u = 0; u < SU; ++u) {
                    const int i4 = base + u * NT + tid;
                    m[u] = *reinterpret_cast<const f32x4*>(msk + (int64_t)min(i4, nv - 1) * 4);
                }
#pragma unroll
                for (int u = 0; u < SU; ++u) { v[u][0] *= m[u][0]; v[u][1] *= m[u][1]; v[u][2] *= m[u][2]; v[u][3] *= m[u][3]; }
            } else if (BUILD && wb.wdrop.on) {
                // generated under the weight loads' latency; element (n, k, c) = flat index 4 * (n * nv + i4)
#pragma unroll
                for (int u = 0; u < SU; ++u) {
                    const int i4 = min(base + u * NT + tid, nv - 1);
                    const f32x4 m = drop_mult4(wb.wdrop, DENET_DROP_WEIGHT, (unsigned)(n * nv + i4));
                    v[u][0] *= m[0]; v[u][1] *= m[1]; v[u][2] *= m[2]; v[u][3] *= m[3];
                }
            }
#pragma unroll
            for (int u = 0; u < SU; ++u) {
                const int i4 = base + u * NT + tid;
                if (i4 < nv) {
                    const int idx = i4 * 4;
                    const int k = idx >> cshift, c = idx & (C - 1);      // C is 64 or 256
                    bool keep = true;                 // BUILD: the rows a way overwrites are written below
                    if constexpr (BUILD)
                        for (int jj = 0; jj < way; ++jj) keep = keep && (lab[jj] != k);
                    if (keep) {
                        float* d = sm + k * ldw + c;
                        d[0] = v[u][0]; d[1] = v[u][1]; d[2] = v[u][2]; d[3] = v[u][3];
                    }
                }
            }
        }
    }
    if constexpr (BUILD) {
        if (prow >= 0) {
            float* d = sm + prow;
            d[0] = pv[0]; d[1] = pv[1]; d[2] = pv[2]; d[3] = pv[3];
        }
        // more than NW ways: the remaining ones, NW at a time (winners own distinct rows)
        for (int jw0 = NW; jw0 < way; jw0 += NW) {
            const int jw = jw0 + (tid >> 6), c4 = tid & 63;
            if (jw < way && c4 < (C >> 2)) {
                const int64_t* lab = labels + (int64_t)n * way;
                const int64_t mine = lab[jw];
                bool win = mine >= 0 && mine < K;
                for (int jj = jw + 1; jj < way; ++jj) win = win && (lab[jj] != mine);
                if (win) {
                    const float* pp = wb.proto_part + ((int64_t)(n * way + jw) * NSL) * C + c4 * 4;
                    f32x4 acc4 = *reinterpret_cast<const f32x4*>(pp);
                    for (int q = 1; q < NSL; ++q) {
                        const f32x4 tq = *reinterpret_cast<const f32x4*>(pp + (int64_t)q * C);
                        acc4[0] += tq[0]; acc4[1] += tq[1]; acc4[2] += tq[2]; acc4[3] += tq[3];
                    }
                    const f32x4 b4 = *reinterpret_cast<const f32x4*>(wb.bl + c4 * 4);
                    float* d = sm + (int)mine * ldw + c4 * 4;
                    d[0] = acc4[0] + b4[0]; d[1] = acc4[1] + b4[1]; d[2] = acc4[2] + b4[2]; d[3] = acc4[3] + b4[3];
                }
            }
        }
    }
}

// NW waves per block split the channels: NLD = C / (2 NW) k-steps per wave.  Block = (episode n, 64-position tile
// `tile` of `ntiles`).
template <typename T, int RT, int NLD, bool BUILD, int NW>
__device__ __forceinline__ void classify_fwd_tile_body(const T* __restrict__ Fq, const float* __restrict__ Wn,
                                                       int64_t wn_stride, const int64_t* __restrict__ labels,
                                                       int way, int K, int C, int P, int base,
                                                       float* __restrict__ logits, float* __restrict__ bin,
                                                       const WnBuild& wb, int n, int tile, int ntiles, float* sm) {
    const int p0 = tile * CL_TP;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int j = lane & 31, hh = lane >> 5;
    const int ldw = C + 1;

    constexpr int NT = 64 * NW;
    const int cper = C / NW;                       // channels per wave
    const int cbeg = wave * cper;
    const int pa = p0 + 2 * j;
    const bool fast = (p0 + CL_TP <= P);           // block-uniform: only the last tile of an episode is partial
    const T* fq = Fq + (int64_t)n * C * P;
    // The streamed operand first: up to 64 channels (32 k-steps, 16 KB per wave) of Fq go in flight
    // before anything else, so the Wn staging below hides under their latency.
    STAMP(0);
    constexpr int chunk = 2 * NLD;                 // channels per pass of this wave (== cper: C is 64 or 256)
    float b0[NLD], b1[NLD];
#ifdef DENET_DIAG_NO_FQ
#pragma unroll
    for (int u = 0; u < NLD; ++u) { b0[u] = (float)u; b1[u] = 1.f; }
#else
    if (fast) {
#pragma unroll
        for (int u = 0; u < NLD; ++u) load2<T, true>(fq + (int64_t)(cbeg + 2 * u + hh) * P, pa, P, b0[u], b1[u]);
    } else {
#pragma unroll
        for (int u = 0; u < NLD; ++u) load2<T, false>(fq + (int64_t)(cbeg + 2 * u + hh) * P, pa, P, b0[u], b1[u]);
    }
#endif

    constexpr int SUc = RT == 1 ? 8 : RT == 2 ? 16 : RT == 3 ? 24 : 16;
    stage_wn<BUILD, NT, NLD * NW / 8, SUc>(sm, Wn, wn_stride, labels, way, K, C, n, wb);
    __syncthreads();
    STAMP(1);
    if constexpr (BUILD) {
        // Wn[n] for the backward pass: tile t writes rows t, t + tiles, ... (one 1 KB row = one store per thread)
        float* dst = wb.Wn_out + (int64_t)n * K * C;
        for (int k = tile; k < K; k += ntiles)
            for (int c = tid; c < C; c += NT) dst[k * C + c] = sm[k * ldw + c];
    }

    f32x16 acc[RT][2];
#pragma unroll
    for (int rt = 0; rt < RT; ++rt)
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[rt][t][e] = 0.f;
    int arow[RT];
    bool aok[RT];
#pragma unroll
    for (int rt = 0; rt < RT; ++rt) {
        const int row = 32 * rt + j;
        aok[rt] = row < K;
        arow[rt] = (aok[rt] ? row : K - 1) * ldw;
    }

    for (int c = cbeg; c < cbeg + cper; c += chunk) {
#pragma unroll
        for (int u = 0; u < NLD; ++u) {
            const int cc = c + 2 * u + hh;
#pragma unroll
            for (int rt = 0; rt < RT; ++rt) {
                float a = sm[arow[rt] + cc];
                a = aok[rt] ? a : 0.f;
                acc[rt][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b0[u], acc[rt][0], 0, 0, 0);
                acc[rt][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b1[u], acc[rt][1], 0, 0, 0);
            }
        }
        if (c + chunk < cbeg + cper) {             // only if a wave owns more than one chunk
#pragma unroll
            for (int u = 0; u < NLD; ++u)
                load2<T, false>(fq + (int64_t)(c + chunk + 2 * u + hh) * P, pa, P, b0[u], b1[u]);
        }
    }
    STAMP(2);
    __syncthreads();            // all waves done with the Wn tile; LDS is reused for the partials

    // partial tiles: part[wave][k][CL_LD]
    float* part = sm + (int64_t)wave * K * CL_LD;
#pragma unroll
    for (int rt = 0; rt < RT; ++rt)
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const int row = 32 * rt + (e & 3) + 8 * (e >> 2) + 4 * hh;
            if (row < K) {
                f32x2 v;
                v[0] = acc[rt][0][e];
                v[1] = acc[rt][1][e];
                *reinterpret_cast<f32x2*>(part + row * CL_LD + 2 * j) = v;
            }
        }
    __syncthreads();
    {
        const int col = tid & 63, rg = tid >> 6;
        const int p = p0 + col;
        const int ws = K * CL_LD;
        for (int k = rg; k < K; k += NW) {
            const int o = k * CL_LD + col;
            float v = ((sm[o] + sm[ws + o]) + sm[2 * ws + o]) + sm[3 * ws + o];
            if constexpr (NW == 8) v += ((sm[4 * ws + o] + sm[5 * ws + o]) + sm[6 * ws + o]) + sm[7 * ws + o];
            sm[o] = v;
            if (p < P) logits[((int64_t)n * K + k) * P + p] = v;
        }
    }
    STAMP(3);
    if (bin == nullptr) return;
    __syncthreads();
    // binary logits: ch1 = row label, ch0 = max over the other rows (utils/operations.py:327-337)
    // "max over the other rows" for every way at once: the column's largest entry, its row, and the
    // largest of the remaining entries (max over k != lab is the second one if lab is that row, else the
    // first).  The K rows are scanned by NW thread groups in parallel and merged in group order -- a serial
    // scan of 81 dependent LDS reads per (way, column) took 4.1 of a block's 21.5 us at K = 81.
    const bool par = base && K >= 8;
    float* t2 = sm + K * CL_LD;                      // [3][NW][64], over the dead partial tiles of waves >= 1
    if (par) {
        const int col = tid & 63, grp = tid >> 6;
        float m1 = -INFINITY, m2 = -INFINITY;
        int i1 = -1;
        for (int k = grp; k < K; k += NW) {
            const float v = sm[k * CL_LD + col];
            if (v > m1) { m2 = m1; m1 = v; i1 = k; }
            else if (v > m2) m2 = v;
        }
        t2[grp * 64 + col] = m1;
        t2[(NW + grp) * 64 + col] = m2;
        t2[(2 * NW + grp) * 64 + col] = __int_as_float(i1);
        __syncthreads();
    }
    for (int item = tid; item < way * CL_TP; item += NT) {
        const int jw = item >> 6, col = item & 63;
        const int p = p0 + col;
        if (p >= P) continue;
        const int64_t lab = labels[(int64_t)n * way + jw];
        float fg = 0.f, bg;
        if (lab >= 0 && lab < K) fg = sm[(int)lab * CL_LD + col];
        if (par) {
            float m1 = -INFINITY, m2 = -INFINITY;
            int i1 = -1;
#pragma unroll
            for (int gq = 0; gq < NW; ++gq) {
                const float a1 = t2[gq * 64 + col], a2 = t2[(NW + gq) * 64 + col];
                const int ai = __float_as_int(t2[(2 * NW + gq) * 64 + col]);
                if (a1 > m1) { m2 = fmaxf(m1, a2); m1 = a1; i1 = ai; }
                else m2 = fmaxf(m2, a1);
            }
            bg = ((int64_t)i1 == lab) ? m2 : m1;
        } else if (base) {
            bg = -INFINITY;
            for (int k = 0; k < K; ++k)
                if (k != lab) bg = fmaxf(bg, sm[k * CL_LD + col]);
        } else {
            bg = sm[col];
        }
        float* dst = bin + ((int64_t)(n * way + jw) * 2) * P + p;
        dst[0] = bg;
        dst[P] = fg;
    }
    STAMP(4);
}

template <typename T, int RT, int NLD, bool BUILD, int NW>
__global__ __launch_bounds__(64 * NW) void classify_fwd_kernel(const T* __restrict__ Fq, const float* __restrict__ Wn,
                                                          int64_t wn_stride, const int64_t* __restrict__ labels,
                                                          int way, int K, int C, int P, int base,
                                                          float* __restrict__ logits, float* __restrict__ bin,
                                                          WnBuild wb) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    classify_fwd_tile_body<T, RT, NLD, BUILD, NW>(Fq, Wn, wn_stride, labels, way, K, C, P, base, logits, bin, wb,
                                                  blockIdx.y, blockIdx.x, gridDim.x, sm);
}
