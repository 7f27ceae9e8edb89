// Device body of K3b (classifier backward), shared by the stand-alone kernel (classify.hip) and the
// fused step kernels (head_step.hip).
#pragma once
#include "denet_common.hpp"

#define CL_TP 64          // positions per block
#define CL_LD 66          // LDS row stride of a 64-wide tile (even: 8-byte reads)

typedef float f32x2_u __attribute__((ext_vector_type(2), aligned(4)));
typedef float f32x4_u __attribute__((ext_vector_type(4), aligned(4)));
typedef unsigned short u16x2_u __attribute__((ext_vector_type(2), aligned(2)));

// ---------------------------------------------------------------------------
// K3b: backward.  One block per (episode, 64-position tile), NW waves:
//   dL[k][p]   = scale[n] * (dlog_a + dlog_b)                     -> LDS
//   dFq[c][p]  = sum_k Wn[k][c] * dL[k][p]                        (A = Wn^T from L2, B = dL from LDS)
//   dWn_part[tile][k][c] = sum_{p in tile} dL[k][p] * Fq[c][p]    (A = dL from LDS, B = Fq^T: each
//        lane holds 32 consecutive positions of one channel row in registers; the MFMA k index of
//        step s is position 32*(lane>>5) + s for both operands).  The MFMA wants lane <-> channel,
//        memory wants lane <-> position: the [32 channels x 64 positions] tile is therefore loaded
//        with lanes along the positions (a wave load = four 256-byte row pieces; lane <-> channel
//        loads touch 64 cache lines per instruction and re-fetch every line eight times through the
//        16 KB L1) and turned through a wave-private LDS tile.
// The partial dWn slabs are summed by K2b (no float atomics: deterministic).
// scale[n] comes from `scale`, or, when part_k != nullptr, is derived here from the loss kernel's
// per-block partial sums exactly as loss_finalize does (and written to `scale_out` by tile 0), which
// takes the finalize launch off the critical path of the fused step.
// ---------------------------------------------------------------------------
struct ClsBwdArgs {
    const void* Fq;
    const float *Wn, *dlog_a, *dlog_b, *scale, *gmul, *part_k;
    int parts_k, N, K, C, P, n_part;
    void* dFq;
    float *dWn_part, *scale_out;
};
#define CL_TS 68          // LDS row stride of a wave's transposition tile (16-byte rows, conflict-free b128)
#define CLS_BWD_DL(K) ((((K) + 1) & ~1) * CL_LD)                       // floats: dL [KP][CL_LD]
#define CLS_BWD_LDS_NW(K, NW) ((size_t)(CLS_BWD_DL(K) + (NW) * 32 * CL_TS) * 4)
#define CLS_BWD_LDS(K) CLS_BWD_LDS_NW(K, 4)

// 1/(N * sum of the episode's den partials): every lane of the calling wave gets the value
__device__ __forceinline__ float episode_den(const float* __restrict__ part_k, int parts_k, int n) {
    float de = 0.f;
    const int lane = threadIdx.x & 63;
    for (int y = lane; y < parts_k; y += 64) de += part_k[((int64_t)n * parts_k + y) * 2 + 1];
    return wave_sum(de);
}

template <typename T, int RT, int NW>
__device__ __forceinline__ void classify_bwd_body(const ClsBwdArgs& g, int n, int tile, float* sm) {
    const T* __restrict__ Fq = reinterpret_cast<const T*>(g.Fq);
    const float* __restrict__ dlog_a = g.dlog_a;
    const float* __restrict__ dlog_b = g.dlog_b;
    T* __restrict__ dFq = reinterpret_cast<T*>(g.dFq);
    float* __restrict__ dWn_part = g.dWn_part;
    const int K = g.K, C = g.C, P = g.P, n_part = g.n_part;
    constexpr int NT = NW * 64;
    const int p0 = tile * CL_TP;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int j = lane & 31, hh = lane >> 5;
    const int KP = (K + 1) & ~1;
    const float* wn = g.Wn + (int64_t)n * K * C;
    const bool fast = (p0 + CL_TP <= P);
    const int nct = C >> 5;                      // 32-channel tiles: 2 (C=64) or 8 (C=256)
    constexpr int TPW = (8 + NW - 1) / NW;       // channel tiles per wave
    constexpr int NA = RT * 16;
    // With K <= 32 every global operand of this block fits in registers, so ALL of them are
    // requested before the first wait: the block then pays one memory latency, not five.
    constexpr bool PRE = (RT == 1);
    float av[TPW][PRE ? NA : 1];
    f32x4 fraw[PRE ? TPW : 1][8];

    auto load_av = [&](int ct, float* dst) {
#pragma unroll
        for (int u = 0; u < NA; ++u) {
            const int kk = 2 * u + hh;
            const float v = wn[(int64_t)min(kk, K - 1) * C + ct * 32 + j];     // unconditional load, then select
            dst[u] = kk < K ? v : 0.f;
        }
    };
    // raw tile of this wave: lane = (row r4 = lane>>4 of a group of 4 channels, 4 positions 4*(lane&15)..)
    const int r4 = lane >> 4, c16 = lane & 15;
    auto load_raw = [&](int ct, f32x4 (&raw)[8]) {
        const int pb = p0 + 4 * c16;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const T* rowp = Fq + ((int64_t)n * C + ct * 32 + 4 * i + r4) * P;
            if (fast) {
                if constexpr (sizeof(T) == 4) {
                    const f32x4_u v = *reinterpret_cast<const f32x4_u*>((const float*)rowp + pb);
                    raw[i] = f32x4{v[0], v[1], v[2], v[3]};
                } else {
                    const u16x2_u v0 = *reinterpret_cast<const u16x2_u*>((const unsigned short*)rowp + pb);
                    const u16x2_u v1 = *reinterpret_cast<const u16x2_u*>((const unsigned short*)rowp + pb + 2);
                    raw[i] = f32x4{bf16_to_f32(v0[0]), bf16_to_f32(v0[1]), bf16_to_f32(v1[0]), bf16_to_f32(v1[1])};
                }
            } else {
                // partial last tile: clamped unconditional loads, zeroed by a select (no load inside a branch)
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float v = FeatIO<T>::load(rowp + min(pb + e, P - 1));
                    raw[i][e] = (pb + e < P) ? v : 0.f;
                }
            }
        }
    };
    float* tsm = sm + CLS_BWD_DL(K) + wave * 32 * CL_TS;     // this wave's transposition tile [32][CL_TS]
    auto turn = [&](const f32x4 (&raw)[8], float* dst) {
        // same wave writes and reads: LDS operations of a wave complete in order
#pragma unroll
        for (int i = 0; i < 8; ++i) *reinterpret_cast<f32x4*>(tsm + (4 * i + r4) * CL_TS + 4 * c16) = raw[i];
#pragma unroll
        for (int m = 0; m < 8; ++m) {
            const f32x4 v = *reinterpret_cast<const f32x4*>(tsm + j * CL_TS + 32 * hh + 4 * m);
            dst[4 * m] = v[0]; dst[4 * m + 1] = v[1]; dst[4 * m + 2] = v[2]; dst[4 * m + 3] = v[3];
        }
    };
    if constexpr (PRE) {
#pragma unroll
        for (int i = 0; i < TPW; ++i) {
            const int ct = wave + NW * i;
            if (ct < nct) {
                if (dFq != nullptr) load_av(ct, av[i]);
                if (dWn_part != nullptr) load_raw(ct, fraw[i]);
            }
        }
    }
    float sc;
    if (g.part_k != nullptr) {
        sc = 1.f / ((float)g.N * episode_den(g.part_k, g.parts_k, n));
        if (g.scale_out != nullptr && tile == 0 && tid == 0) g.scale_out[n] = sc;
    } else {
        sc = g.scale ? g.scale[n] : 1.f;
    }
    sc *= (g.gmul ? g.gmul[0] : 1.f);

#pragma unroll 1
    for (int base = 0; base < KP * CL_TP; base += NT * 6) {
        float va[6], vb[6];
#pragma unroll
        for (int u = 0; u < 6; ++u) {
            const int idx = base + u * NT + tid;
            const int k = idx >> 6, col = idx & 63;
            const int p = p0 + col;
            const bool ok = idx < KP * CL_TP && k < K && p < P;
            const int64_t gi = ((int64_t)n * K + min(k, K - 1)) * P + min(p, P - 1);
            const float xa = dlog_a[gi];
            const float xb = dlog_b ? dlog_b[gi] : 0.f;
            va[u] = ok ? xa : 0.f;
            vb[u] = ok ? xb : 0.f;
        }
#pragma unroll
        for (int u = 0; u < 6; ++u) {
            const int idx = base + u * NT + tid;
            if (idx < KP * CL_TP) sm[(idx >> 6) * CL_LD + (idx & 63)] = (va[u] + vb[u]) * sc;
        }
    }
    __syncthreads();

    // ---- dFq ----
    if (dFq != nullptr) {
        const int pa = p0 + 2 * j;
#pragma unroll
        for (int i = 0; i < TPW; ++i) {
            const int ct = wave + NW * i;
            if (ct >= nct) continue;
            const int c0 = ct * 32;
            f32x16 a0, a1;
#pragma unroll
            for (int e = 0; e < 16; ++e) { a0[e] = 0.f; a1[e] = 0.f; }
            float avl[NA];
            if constexpr (!PRE) load_av(ct, avl);
#pragma unroll
            for (int u = 0; u < NA; ++u) {
                const int kk = 2 * u + hh;
                if (2 * u < KP) {
                    const float a = PRE ? av[i][PRE ? u : 0] : avl[u];
                    const f32x2 b = *reinterpret_cast<const f32x2*>(sm + kk * CL_LD + 2 * j);
                    a0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b[0], a0, 0, 0, 0);
                    a1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b[1], a1, 0, 0, 0);
                }
            }
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int c = c0 + (e & 3) + 8 * (e >> 2) + 4 * hh;
                T* dst = dFq + ((int64_t)n * C + c) * P + pa;
                if (pa + 1 < P) {
                    // two adjacent positions in one store: a wave store is two contiguous 256-byte row pieces
                    if constexpr (sizeof(T) == 4) {
                        *reinterpret_cast<f32x2_u*>(dst) = f32x2_u{a0[e], a1[e]};
                    } else {
                        *reinterpret_cast<u16x2_u*>(dst) = u16x2_u{f32_to_bf16(a0[e]), f32_to_bf16(a1[e])};
                    }
                } else if (pa < P) {
                    FeatIO<T>::store(dst, a0[e]);
                }
            }
        }
    }

    // ---- dWn partial ----
    if (dWn_part != nullptr) {
#pragma unroll
        for (int i = 0; i < TPW; ++i) {
            const int ct = wave + NW * i;
            if (ct >= nct) continue;
            const int c0 = ct * 32;
            float fvl[32];
            if constexpr (PRE) {
                turn(fraw[i], fvl);
            } else {
                f32x4 raw[8];
                load_raw(ct, raw);
                turn(raw, fvl);
            }
#pragma unroll
            for (int rt = 0; rt < RT; ++rt) {
                const int row = 32 * rt + j;
                const bool ok = row < K;
                const float* arow = sm + (ok ? row : 0) * CL_LD + 32 * hh;
                f32x16 acc;
#pragma unroll
                for (int e = 0; e < 16; ++e) acc[e] = 0.f;
#pragma unroll
                for (int s = 0; s < 32; ++s) {
                    float a = arow[s];
                    a = ok ? a : 0.f;
                    const float bq = fvl[s];
                    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, bq, acc, 0, 0, 0);
                }
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    const int kr = 32 * rt + (e & 3) + 8 * (e >> 2) + 4 * hh;
                    if (kr < K)
                        dWn_part[(((int64_t)n * n_part + tile) * K + kr) * C + c0 + j] = acc[e];
                }
            }
        }
    }
}

template <typename T, int RT>
__global__ __launch_bounds__(256) void classify_bwd_kernel(ClsBwdArgs g) {
    extern __shared__ __attribute__((aligned(16))) float sm[];   // dL [KP][CL_LD]
    classify_bwd_body<T, RT, 4>(g, blockIdx.y, blockIdx.x, sm);
}

