// Device body of K3b (classifier backward), shared by the stand-alone kernel (classify.hip) and the
// fused step kernels (head_step.hip).
#pragma once
#include <type_traits>

#include "denet_common.hpp"

#define CL_TP 64          // positions per block
#define CL_LD 66          // LDS row stride of a 64-wide tile (even: 8-byte reads)

typedef float f32x2_u __attribute__((ext_vector_type(2), aligned(4)));
typedef float f32x4_u __attribute__((ext_vector_type(4), aligned(4)));
typedef unsigned short u16x2_u __attribute__((ext_vector_type(2), aligned(2)));

// bf16 matrix pipe (v_mfma_f32_32x32x16_bf16): operand fragments are 8 bf16 = 4 dwords per lane
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4_u2 __attribute__((ext_vector_type(4), aligned(2)));     // 8 bf16 at any element address

// w = hi + mid + lo, each a bf16 (returned in the high half of a dword); exact for finite w
__device__ __forceinline__ void split_bf16x3(float w, unsigned& h, unsigned& m, unsigned& l) {
    h = __float_as_uint(w) & 0xffff0000u;
    const float r1 = w - __uint_as_float(h);               // <= 16 significant bits
    m = __float_as_uint(r1) & 0xffff0000u;
    const float r2 = r1 - __uint_as_float(m);              // <= 8 significant bits
    l = __float_as_uint(r2) & 0xffff0000u;
}

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

// 1 / (N * den) for the episode's gradient; 0 when every label pixel of the episode is ignored (den == 0): the
// reference's loss is 0/0 = NaN there while its gradient is exactly zero (nll_loss's backward skips ignored targets,
// trainer/utils.py:272-277 + ATen), and inf * 0 would make ours NaN
__device__ __forceinline__ float episode_scale(int N, float den) { return den > 0.f ? 1.f / ((float)N * den) : 0.f; }

// 1/(N * sum of the episode's den partials): every lane of the calling wave gets the value
__device__ __forceinline__ float episode_den(const float* __restrict__ part_k, int parts_k, int n) {
    float de = 0.f;
    const int lane = threadIdx.x & 63;
    for (int y = lane; y < parts_k; y += 64) de += part_k[((int64_t)n * parts_k + y) * 2 + 1];
    return wave_sum(de);
}

// the calling lane's share of the episode's den partials (same order of additions as episode_den)
__device__ __forceinline__ float part_k_lane(const float* __restrict__ part_k, int parts_k, int n) {
    float de = 0.f;
    const int lane = threadIdx.x & 63;
    for (int y = lane; y < parts_k; y += 64) de += part_k[((int64_t)n * parts_k + y) * 2 + 1];
    return de;
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
        sc = episode_scale(g.N, episode_den(g.part_k, g.parts_k, n));
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

// ---------------------------------------------------------------------------
// K3b for K > 32 in the fused step: PERSISTENT blocks.  A block = (episode n, group of consecutive
// position tiles, role) with 8 waves, one 32-channel tile per wave (C = 256):
//   role DWN: the wave's dWn accumulators (RT x 16 registers) stay resident over the group's tiles and ONE
//             slab per group is written, so n_part = groups per episode (6 at C3, 15 at C5) instead of
//             tiles per episode (27 / 103): the slab traffic of this kernel and of its consumers (B2's
//             dproto and dweight) shrinks by that factor;
//   role DFQ: the wave's Wn^T fragments (RT x 16 registers) are loaded once per block, not once per tile.
// dL = scale * (dlog_a + dlog_b) of tile t+1 is fetched into registers and the Fq tile of t+1 requested
// while the MFMAs of tile t run (all loads unconditional: the last iteration re-requests its own tile), then
// stored to the other of two LDS buffers: one barrier per tile.  Same arithmetic order per output element
// as classify_bwd_body within a tile; across tiles the slab sum order changes (tiles of a group are
// accumulated in the MFMA accumulators, groups in B2), which the element-wise tolerance of the tests covers.
// ---------------------------------------------------------------------------
#define CLW_BWD_NW 8
#define CLW_LD 68         // dL row stride: 16-byte rows, conflict-free 16-byte reads down a column of rows
#define CLW_DL(RT) ((RT) * 32 * CLW_LD)                                  // floats: dL [RT*32][CLW_LD], rows >= K zero
#define CLW_BWD_LDS(RT) ((size_t)(2 * CLW_DL(RT) + CLW_BWD_NW * 32 * CL_TS) * 4)

// diagnostic builds (scripts/diag/bwd_wide_stamps.hip): phase stamps of thread 0 of block (n 0, group 0) of one role
#ifdef DENET_STAMPS
#ifndef DENET_STAMP_ROLE_DWN
#define DENET_STAMP_ROLE_DWN 1
#endif
#define WSTAMP(i) do { if (n == 0 && grp == 0 && threadIdx.x == 0 && DWN == (bool)(DENET_STAMP_ROLE_DWN)) g_stamps[i] = __builtin_amdgcn_s_memrealtime(); } while (0)
#else
#define WSTAMP(i) do { } while (0)
#endif

// NTQ: dFq leaves with non-temporal stores (it is not read again in the step).  Measured, step time without / with:
// C2 87.25 / 86.6 us, C3 339.4 / 338.9, C5 396.5 / 395.5, C2 at 128 episodes (220 MB of dFq) 0.993 / 1.023 ms --
// the launcher uses it for streams up to 80 MB.
template <typename T, int RT, bool DWN, bool NTQ = false>
__device__ __forceinline__ void classify_bwd_wide_body(const ClsBwdArgs& g, int n, int grp, float* sm) {
#ifdef DENET_DIAG_BW_ONLYDWN
    if (!DWN) return;
#endif
#ifdef DENET_DIAG_BW_ONLYDFQ
    if (DWN) return;
#endif
    const T* __restrict__ Fq = reinterpret_cast<const T*>(g.Fq);
    const float* __restrict__ dlog_a = g.dlog_a;
    const float* __restrict__ dlog_b = g.dlog_b;
    T* __restrict__ dFq = reinterpret_cast<T*>(g.dFq);
    float* __restrict__ dWn_part = g.dWn_part;
    const int K = g.K, C = g.C, P = g.P, ngrp = g.n_part;
    constexpr int NT = CLW_BWD_NW * 64, NA = RT * 16;
    const int ntile = (P + CL_TP - 1) / CL_TP;
    const int t0 = (int)((int64_t)grp * ntile / ngrp), t1 = (int)((int64_t)(grp + 1) * ntile / ngrp);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int j = lane & 31, hh = lane >> 5;
    const float* wn = g.Wn + (int64_t)n * K * C;
    const int ct = wave;
    const bool have = ct < (C >> 5);               // C = 64: waves 2.. only help with the staging
    const int cc = min(ct, (C >> 5) - 1);
    const int r4 = lane >> 4, c16 = lane & 15;
    float* dl0 = sm;
    float* dl1 = sm + CLW_DL(RT);
    float* tsm = sm + 2 * CLW_DL(RT) + wave * 32 * CL_TS;
    // rows K .. RT*32-1 of both buffers stay zero: the MFMA row tiles read them without a guard
    for (int idx = K * CLW_LD + tid; idx < RT * 32 * CLW_LD; idx += NT) { dl0[idx] = 0.f; dl1[idx] = 0.f; }

    const bool have_b = dlog_b != nullptr;
    const float* __restrict__ dlb = have_b ? dlog_b : dlog_a;      // no branch around a load
    // dL staging: thread <-> RT quads (row k = idx >> 4, four positions 4 * (idx & 15)), idx = u * NT + tid
    // Every quad load starts at min(p, P - 4): the partial last tile of an episode then needs no scalar path
    // and no branch -- the lanes whose quad was moved back pick their elements (and zeros for the positions
    // past P) when the tile goes to LDS: quad_fix(loaded, sh), sh = p - min(p, P - 4).
    auto quad_fix = [&](const f32x4& l, int sh) {
        f32x4 r;
        r[0] = sh == 0 ? l[0] : sh == 1 ? l[1] : sh == 2 ? l[2] : sh == 3 ? l[3] : 0.f;
        r[1] = sh == 0 ? l[1] : sh == 1 ? l[2] : sh == 2 ? l[3] : 0.f;
        r[2] = sh == 0 ? l[2] : sh == 1 ? l[3] : 0.f;
        r[3] = sh == 0 ? l[3] : 0.f;
        return r;
    };
    // dL staging: thread <-> RT quads (row k = idx >> 4, four positions 4 * (idx & 15)), idx = u * NT + tid
    auto load_dl_u = [&](int tile, int u, f32x4& xa, f32x4& xb) {
        const int idx = u * NT + tid;
        const int k = min(idx >> 4, K - 1), pq = min(tile * CL_TP + 4 * (idx & 15), P - 4);
        const f32x4_u a = *reinterpret_cast<const f32x4_u*>(dlog_a + ((int64_t)n * K + k) * P + pq);
        const f32x4_u b = *reinterpret_cast<const f32x4_u*>(dlb + ((int64_t)n * K + k) * P + pq);
        xa = f32x4{a[0], a[1], a[2], a[3]};
        xb = f32x4{b[0], b[1], b[2], b[3]};
    };
    auto store_dl = [&](float* buf, int tile, const f32x4 (&va)[RT], const f32x4 (&vb)[RT], float sc) {
        const bool part = (tile + 1) * CL_TP > P;          // uniform
#pragma unroll
        for (int u = 0; u < RT; ++u) {
            const int idx = u * NT + tid;
            if ((idx >> 4) < K) {
                const int pq = tile * CL_TP + 4 * (idx & 15);
                f32x4 v;
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = (va[u][e] + (have_b ? vb[u][e] : 0.f)) * sc;
                if (part) v = quad_fix(v, pq - min(pq, P - 4));
                *reinterpret_cast<f32x4*>(buf + (idx >> 4) * CLW_LD + 4 * (idx & 15)) = v;
            }
        }
    };
    // Fq tile of this wave's 32 channels: load i = channels 4i .. 4i+3 (one per 16 lanes), 4 positions per lane
    auto load_raw_i = [&](int tile, int i, f32x4& r) {
        const int pb = min(tile * CL_TP + 4 * c16, P - 4);
        const T* rowp = Fq + ((int64_t)n * C + cc * 32 + 4 * i + r4) * P;
        if constexpr (sizeof(T) == 4) {
            const f32x4_u v = *reinterpret_cast<const f32x4_u*>((const float*)rowp + pb);
            r = f32x4{v[0], v[1], v[2], v[3]};
        } else {
            const u16x2_u v0 = *reinterpret_cast<const u16x2_u*>((const unsigned short*)rowp + pb);
            const u16x2_u v1 = *reinterpret_cast<const u16x2_u*>((const unsigned short*)rowp + pb + 2);
            r = f32x4{bf16_to_f32(v0[0]), bf16_to_f32(v0[1]), bf16_to_f32(v1[0]), bf16_to_f32(v1[1])};
        }
    };
    auto turn = [&](int tile, const f32x4 (&raw)[8], float (&dst)[32]) {
        const bool part = (tile + 1) * CL_TP > P;          // uniform
        const int pb = tile * CL_TP + 4 * c16;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            f32x4 v = raw[i];
            if (part) v = quad_fix(v, pb - min(pb, P - 4));
            *reinterpret_cast<f32x4*>(tsm + (4 * i + r4) * CL_TS + 4 * c16) = v;
        }
#pragma unroll
        for (int m = 0; m < 8; ++m) {
            const f32x4 v = *reinterpret_cast<const f32x4*>(tsm + j * CL_TS + 32 * hh + 4 * m);
            dst[4 * m] = v[0]; dst[4 * m + 1] = v[1]; dst[4 * m + 2] = v[2]; dst[4 * m + 3] = v[3];
        }
    };
    // one result row pair of a finished dFq tile: channel ct*32 + (e&3) + 8*(e>>2) + 4*hh, positions pa, pa+1
    auto store_dfq_e = [&](int tile, int e, float x0, float x1) {
        const int pa = tile * CL_TP + 2 * j;
        const int c = ct * 32 + (e & 3) + 8 * (e >> 2) + 4 * hh;
        T* dst = dFq + ((int64_t)n * C + c) * P + pa;
#ifdef DENET_DIAG_BW_NOSTORE
        if (x0 != 12345.f) return;
#endif
        if (pa + 1 < P) {
            if constexpr (sizeof(T) == 4) {
                const f32x2_u v = {x0, x1};
                if constexpr (NTQ) __builtin_nontemporal_store(v, reinterpret_cast<f32x2_u*>(dst));
                else *reinterpret_cast<f32x2_u*>(dst) = v;
            } else {
                const u16x2_u v = {f32_to_bf16(x0), f32_to_bf16(x1)};
                if constexpr (NTQ) __builtin_nontemporal_store(v, reinterpret_cast<u16x2_u*>(dst));
                else *reinterpret_cast<u16x2_u*>(dst) = v;
            }
        } else if (pa < P) {
            FeatIO<T>::store(dst, x0);
        }
    };

    // everything this block needs first is requested before the first wait
    f32x4 va[RT], vb[RT];
    f32x4 raw[8];
    float av[DWN ? 1 : NA];
    WSTAMP(0);
    BSTAMP(0);
    // the episode's loss denominator: its partials are requested first and reduced after the big loads are out
    float de = 0.f;
    if (g.part_k != nullptr) de = part_k_lane(g.part_k, g.parts_k, n);
#pragma unroll
    for (int u = 0; u < RT; ++u) load_dl_u(t0, u, va[u], vb[u]);
    if constexpr (DWN) {
#pragma unroll
        for (int i = 0; i < 8; ++i) load_raw_i(t0, i, raw[i]);
    } else {
#pragma unroll
        for (int u = 0; u < NA; ++u) {
            const int kk = 2 * u + hh;
            const float v = wn[(int64_t)min(kk, K - 1) * C + cc * 32 + j];
            av[u] = kk < K ? v : 0.f;
        }
    }
    float sc;
    if (g.part_k != nullptr) {
        sc = episode_scale(g.N, wave_sum(de));
        if (g.scale_out != nullptr && grp == 0 && tid == 0) g.scale_out[n] = sc;
    } else {
        sc = g.scale ? g.scale[n] : 1.f;
    }
    sc *= (g.gmul ? g.gmul[0] : 1.f);
    store_dl(dl0, t0, va, vb, sc);
    __syncthreads();
    WSTAMP(1);

    f32x16 acc[DWN ? RT : 1];
    if constexpr (DWN) {
#pragma unroll
        for (int rt = 0; rt < RT; ++rt)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[rt][e] = 0.f;
    }
    // dFq: the finished tile waits in s0 / s1 and is stored between the MFMAs of the next tile
    f32x16 s0, s1;
    int ts = -1;                                    // tile held in s0 / s1 (-1: none)
#pragma unroll
    for (int e = 0; e < 16; ++e) { s0[e] = 0.f; s1[e] = 0.f; }

    // The loads of tile t+1 and the stores of tile t-1 are ISSUED BETWEEN the MFMA groups of tile t, a few per
    // group: a burst of them ahead of the MFMAs stalls the wave in the memory pipeline's issue queue (every CU
    // asks for ~100 KB at once; measured 4 us per tile, scripts/diag/bwd_wide_stamps.hip) and the MFMAs of an
    // in-order wave wait behind it.  The last iteration re-requests its own tile (no branch around loads).
#pragma unroll 1
    for (int t = t0; t < t1; ++t) {
        const float* cur = ((t - t0) & 1) ? dl1 : dl0;
        float* nxt = ((t - t0) & 1) ? dl0 : dl1;
        const int tn = min(t + 1, t1 - 1);
        float fvl[DWN ? 32 : 1];
        if constexpr (DWN) turn(t, raw, fvl);
        if (t - t0 < 4) WSTAMP(2 + 4 * (t - t0));
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (DWN) {
            constexpr int NG = RT * 2;             // groups of 16 MFMAs
            // A = dL rows from LDS, 16 positions (4 x 16 bytes) per group, requested one group ahead
            f32x4 A[2][4];
            auto lda = [&](int gi, f32x4 (&d)[4]) {
                const float* r = cur + (32 * (gi >> 1) + j) * CLW_LD + 32 * hh + 16 * (gi & 1);
#pragma unroll
                for (int q = 0; q < 4; ++q) d[q] = *reinterpret_cast<const f32x4*>(r + 4 * q);
            };
            lda(0, A[0]);
#pragma unroll
            for (int gi = 0; gi < NG; ++gi) {
                if (gi + 1 < NG) lda(gi + 1, A[(gi + 1) & 1]);
#ifndef DENET_DIAG_BW_NODL
                if (gi < RT) load_dl_u(tn, gi, va[gi], vb[gi]);
#endif
#pragma unroll
                for (int i = 0; i < 8; ++i)
                    if ((i * NG) / 8 == gi) load_raw_i(tn, i, raw[i]);
                __builtin_amdgcn_sched_barrier(0);
#ifdef DENET_DIAG_BW_NOMFMA
                if (have && g.P < 0) {
#else
                if (have) {
#endif
#pragma unroll
                    for (int q = 0; q < 4; ++q)
#pragma unroll
                        for (int e = 0; e < 4; ++e)
                            acc[gi >> 1] = __builtin_amdgcn_mfma_f32_32x32x2f32(
                                A[gi & 1][q][e], fvl[16 * (gi & 1) + 4 * q + e], acc[gi >> 1], 0, 0, 0);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        } else {
            constexpr int NG = RT * 4;             // groups of 8 MFMAs (8 dL rows)
            f32x16 a0, a1;
#pragma unroll
            for (int e = 0; e < 16; ++e) { a0[e] = 0.f; a1[e] = 0.f; }
            // B = dL rows from LDS, 8 rows (4 x 8 bytes per lane) per group, requested one group ahead;
            // groups past K are skipped (uniform)
            f32x2 Bf[2][4];
            auto ldb = [&](int gi, f32x2 (&d)[4]) {
#pragma unroll
                for (int q = 0; q < 4; ++q)
                    d[q] = *reinterpret_cast<const f32x2*>(cur + (8 * gi + 2 * q + hh) * CLW_LD + 2 * j);
            };
            ldb(0, Bf[0]);
#pragma unroll
            for (int gi = 0; gi < NG; ++gi) {
                if (gi + 1 < NG) ldb(gi + 1, Bf[(gi + 1) & 1]);
#ifndef DENET_DIAG_BW_NODL
                if ((gi & 1) == 0 && (gi >> 1) < RT) load_dl_u(tn, gi >> 1, va[gi >> 1], vb[gi >> 1]);
#endif
                if (have && ts >= 0) {
#pragma unroll
                    for (int e = 0; e < 16; ++e)
                        if ((e * NG) / 16 == gi) store_dfq_e(ts, e, s0[e], s1[e]);
                }
                __builtin_amdgcn_sched_barrier(0);
#ifdef DENET_DIAG_BW_NOMFMA
                if (have && 8 * gi < K && g.P < 0) {
#else
                if (have && 8 * gi < K) {
#endif
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        a0 = __builtin_amdgcn_mfma_f32_32x32x2f32(av[4 * gi + q], Bf[gi & 1][q][0], a0, 0, 0, 0);
                        a1 = __builtin_amdgcn_mfma_f32_32x32x2f32(av[4 * gi + q], Bf[gi & 1][q][1], a1, 0, 0, 0);
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
            }
            s0 = a0;
            s1 = a1;
            ts = t;
        }
        if (t - t0 < 4) WSTAMP(4 + 4 * (t - t0));
        store_dl(nxt, tn, va, vb, sc);
        __syncthreads();
        if (t - t0 < 4) WSTAMP(5 + 4 * (t - t0));
    }
    WSTAMP(30);

    if constexpr (DWN) {
        if (have) {
#pragma unroll
            for (int rt = 0; rt < RT; ++rt)
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    const int kr = 32 * rt + (e & 3) + 8 * (e >> 2) + 4 * hh;
                    if (kr < K) dWn_part[(((int64_t)n * ngrp + grp) * K + kr) * C + ct * 32 + j] = acc[rt][e];
                }
        }
    } else {
        if (have && ts >= 0) {
#pragma unroll
            for (int e = 0; e < 16; ++e) store_dfq_e(ts, e, s0[e], s1[e]);
        }
    }
    WSTAMP(31);
    BSTAMP(1);
}

// ---------------------------------------------------------------------------
// The persistent backward for bf16 features with 32 < K <= 96, on the bf16 matrix pipe (round 3).  Same blocks,
// roles, tile loop, double-buffered dL and load / store interleaving as classify_bwd_wide_body; what changes is
// the arithmetic and what it lets go of.  dL = scale * (dlog_a + dlog_b) is split into bf16 planes when it goes
// to LDS (dL = h + m + l exactly, split_bf16x3), Fq is bf16 already:
//   role dWn:  dWn[k][c] += sum_p dL[k][p] Fq[c][p]: the contraction runs over POSITIONS, and eight consecutive
//              positions of a channel row are 16 contiguous bytes of Fq -- the B fragment is ONE global load per
//              lane and k-step, no transposition tile in LDS (the fp32 kernel turns every Fq tile through LDS:
//              lane <-> channel for the MFMA, lane <-> position for memory).  A = dL planes [3][32 RT][72] bf16
//              (144-byte rows: sixteen lanes' 16-byte reads cover the 64 banks).  3 planes x RT x 4 k-steps =
//              36 MFMAs of 32 cycles per tile and wave instead of 96 of 64; every product is exact.
//   role dFq:  dFq[c][p] = sum_k Wn[k][c] dL[k][p]: the contraction runs over CLASSES, so dL goes to LDS
//              transposed, [2][64 positions][32 RT + 8] bf16 (row of position p = 32 (p & 1) + (p >> 1): the two
//              column tiles of the MFMA are the even and the odd positions, which keeps dFq's 4-byte pair stores),
//              and Wn^T sits in registers as two planes, split once per block.  dFq leaves as bf16 (8 significand
//              bits), so the terms of order <= 1 suffice: Wh dh + Wh dm + Wm dh, error < 2^-15 of a product;
//              3 x 2 RT k-steps x 2 tiles = 36 MFMAs per tile and wave instead of 96.
// Measured with the MFMAs taken out of the fp32 kernel (-DDENET_DIAG_BW_NOMFMA): B1 65.8 -> 38 us at C5, i.e. the
// matrix work was 27 us of the launch.
// ---------------------------------------------------------------------------
#define CLWB_LDP 72                                  // bf16 per row of the [class][position] planes (dWn role)
#define CLWB_LDK(RT) (32 * (RT) + 8)                 // bf16 per row of the [position][class] planes (dFq role)
#define CLWB_BUF_DWN(RT) (3 * 32 * (RT) * CLWB_LDP)  // bf16 elements per buffer
#define CLWB_BUF_DFQ(RT) (3 * 64 * CLWB_LDK(RT))   // (two planes for bf16 features, three for fp32)
#define CLWB_BUF(RT) (CLWB_BUF_DWN(RT) > CLWB_BUF_DFQ(RT) ? CLWB_BUF_DWN(RT) : CLWB_BUF_DFQ(RT))
#define CLWB_LDS(RT) ((size_t)2 * CLWB_BUF(RT) * 2)

// elements i + sh of a fragment of 8 bf16, zeros from the end (sh >= 8: all zeros): the fragment was loaded
// `sh` positions before the ones it stands for, because those run past the end of the row
__device__ __forceinline__ u32x4 bf16x8_shift_down(const u32x4 v, int sh) {
    unsigned long long lo = (unsigned long long)v[0] | ((unsigned long long)v[1] << 32);
    unsigned long long hi = (unsigned long long)v[2] | ((unsigned long long)v[3] << 32);
    const int s = 16 * sh;
    if (s >= 128) { lo = 0ull; hi = 0ull; }
    else if (s >= 64) { lo = hi >> (s - 64); hi = 0ull; }
    else if (s > 0) { lo = (lo >> s) | (hi << (64 - s)); hi >>= s; }
    return u32x4{(unsigned)lo, (unsigned)(lo >> 32), (unsigned)hi, (unsigned)(hi >> 32)};
}

template <typename T, int RT, bool DWN, bool NTQ = false>
__device__ __forceinline__ void classify_bwd_wide_bf16_body(const ClsBwdArgs& g, int n, int grp, float* smf) {
    constexpr bool F32 = sizeof(T) == 4;           // fp32 features: Fq is split in registers, all six terms of order <= 2
    constexpr int NPQ = F32 ? 3 : 2;               // planes of dL^T and Wn^T in the dFq role
    const T* __restrict__ Fq = reinterpret_cast<const T*>(g.Fq);
    const float* __restrict__ dlog_a = g.dlog_a;
    const float* __restrict__ dlog_b = g.dlog_b;
    T* __restrict__ dFq = reinterpret_cast<T*>(g.dFq);
    float* __restrict__ dWn_part = g.dWn_part;
    const int K = g.K, C = g.C, P = g.P, ngrp = g.n_part;
    constexpr int NT = CLW_BWD_NW * 64, NR = 32 * RT, KS = 2 * RT;      // KS: class k-steps of 16 (dFq role)
    constexpr int LDP = CLWB_LDP, LDK = CLWB_LDK(RT), BUF = CLWB_BUF(RT);
    static_assert(NPQ * 64 * CLWB_LDK(RT) <= CLWB_BUF(RT), "dFq-role planes must fit the buffer");
    const int ntile = (P + CL_TP - 1) / CL_TP;
    const int t0 = (int)((int64_t)grp * ntile / ngrp), t1 = (int)((int64_t)(grp + 1) * ntile / ngrp);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int j = lane & 31, hh = lane >> 5;
    const float* wn = g.Wn + (int64_t)n * K * C;
    const int ct = wave;
    const bool have = ct < (C >> 5);               // C = 64: waves 2.. only help with the staging
    const int cc = min(ct, (C >> 5) - 1);
    unsigned short* sm = reinterpret_cast<unsigned short*>(smf);
    unsigned short* dl0 = sm;
    unsigned short* dl1 = sm + BUF;
    // what store_dl never writes stays zero: class rows (dWn role) / class columns (dFq role) K .. 32 RT - 1
    if constexpr (DWN) {
        for (int idx = tid; idx < 2 * 3 * (NR - K) * LDP; idx += NT) {
            const int per = (NR - K) * LDP, q = idx / per, r = idx - q * per;      // q = buffer * 3 + plane
            sm[(q / 3) * BUF + ((q % 3) * NR + K) * LDP + r] = 0;
        }
    } else {
        for (int idx = tid; idx < 2 * NPQ * 64 * (NR - K); idx += NT) {
            const int per = NR - K, q = idx / per, r = idx - q * per;              // q = buffer * (NPQ * 64) + plane * 64 + row
            sm[(q / (NPQ * 64)) * BUF + (q % (NPQ * 64)) * LDK + K + r] = 0;
        }
    }

    const bool have_b = dlog_b != nullptr;
    const float* __restrict__ dlb = have_b ? dlog_b : dlog_a;      // no branch around a load
    // dL staging as in classify_bwd_wide_body: thread <-> RT quads (row k = idx >> 4, four positions 4 * (idx & 15)),
    // every quad load starts at min(p, P - 4), the lanes whose quad was moved back pick their elements (quad_fix)
    auto quad_fix = [&](const f32x4& l, int sh) {
        f32x4 r;
        r[0] = sh == 0 ? l[0] : sh == 1 ? l[1] : sh == 2 ? l[2] : sh == 3 ? l[3] : 0.f;
        r[1] = sh == 0 ? l[1] : sh == 1 ? l[2] : sh == 2 ? l[3] : 0.f;
        r[2] = sh == 0 ? l[2] : sh == 1 ? l[3] : 0.f;
        r[3] = sh == 0 ? l[3] : 0.f;
        return r;
    };
    auto load_dl_u = [&](int tile, int u, f32x4& xa, f32x4& xb) {
        const int idx = u * NT + tid;
        const int k = min(idx >> 4, K - 1), pq = min(tile * CL_TP + 4 * (idx & 15), P - 4);
        const f32x4_u a = *reinterpret_cast<const f32x4_u*>(dlog_a + ((int64_t)n * K + k) * P + pq);
        const f32x4_u b = *reinterpret_cast<const f32x4_u*>(dlb + ((int64_t)n * K + k) * P + pq);
        xa = f32x4{a[0], a[1], a[2], a[3]};
        xb = f32x4{b[0], b[1], b[2], b[3]};
    };
    auto store_dl = [&](unsigned short* buf, int tile, const f32x4 (&va)[RT], const f32x4 (&vb)[RT], float sc) {
        const bool part = (tile + 1) * CL_TP > P;          // uniform
#pragma unroll
        for (int u = 0; u < RT; ++u) {
            const int idx = u * NT + tid, k = idx >> 4, q4 = 4 * (idx & 15);
            if (k < K) {
                const int pq = tile * CL_TP + q4;
                f32x4 v;
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = (va[u][e] + (have_b ? vb[u][e] : 0.f)) * sc;
                if (part) v = quad_fix(v, pq - min(pq, P - 4));
                unsigned h[4], m[4], l[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) split_bf16x3(v[e], h[e], m[e], l[e]);
                if constexpr (DWN) {
                    unsigned short* d = buf + k * LDP + q4;
                    *reinterpret_cast<u32x2*>(d) = u32x2{(h[0] >> 16) | h[1], (h[2] >> 16) | h[3]};
                    *reinterpret_cast<u32x2*>(d + NR * LDP) = u32x2{(m[0] >> 16) | m[1], (m[2] >> 16) | m[3]};
                    *reinterpret_cast<u32x2*>(d + 2 * NR * LDP) = u32x2{(l[0] >> 16) | l[1], (l[2] >> 16) | l[3]};
                } else {
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const int p = q4 + e, row = 32 * (p & 1) + (p >> 1);
                        buf[row * LDK + k] = (unsigned short)(h[e] >> 16);
                        buf[(64 + row) * LDK + k] = (unsigned short)(m[e] >> 16);
                        if constexpr (NPQ == 3) buf[(128 + row) * LDK + k] = (unsigned short)(l[e] >> 16);
                    }
                }
            }
        }
    };
    // dWn role, B fragment of k-step ks of a tile: 8 positions of channel cc*32 + j from tile*64 + 16 ks + 8 hh on.
    // The load starts at min(position, P - 8) (always inside the row); fq_fix shifts the fragment and zeroes the
    // positions past P (partial last tile of an episode only).  bf16: the 16 bytes ARE the fragment; fp32: 32 bytes,
    // split into three bf16 planes when the tile becomes current.
    typedef typename std::conditional<F32, f32x4, u32x4>::type fqraw_t;
    constexpr int NFR = F32 ? 2 : 1;               // raw vectors per k-step
    const T* fqrow = Fq + ((int64_t)n * C + cc * 32 + j) * P;
    auto load_fq = [&](int tile, int ks, fqraw_t (&r)[NFR]) {
        const int p0 = tile * CL_TP + 16 * ks + 8 * hh;
        const T* src = fqrow + max(min(p0, P - 8), 0);
        if constexpr (F32) {
            const f32x4_u v0 = *reinterpret_cast<const f32x4_u*>(src), v1 = *reinterpret_cast<const f32x4_u*>(src + 4);
            r[0] = f32x4{v0[0], v0[1], v0[2], v0[3]};
            r[1] = f32x4{v1[0], v1[1], v1[2], v1[3]};
        } else {
            const u32x4_u2 v = *reinterpret_cast<const u32x4_u2*>(src);
            r[0] = u32x4{v[0], v[1], v[2], v[3]};
        }
    };
    // -> the k-step's B planes (bf16: one; fp32: hi / mid / lo)
    auto fq_planes = [&](int tile, int ks, const fqraw_t (&r)[NFR], bool part, u32x4 (&b)[F32 ? 3 : 1]) {
        const int p0 = tile * CL_TP + 16 * ks + 8 * hh, sh = p0 - min(p0, P - 8);
        if constexpr (F32) {
            float x[8] = {r[0][0], r[0][1], r[0][2], r[0][3], r[1][0], r[1][1], r[1][2], r[1][3]};
            if (part) {                             // uniform
                float y[8];
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    float v = 0.f;
#pragma unroll
                    for (int q = 0; i + q < 8; ++q) v = sh == q ? x[i + q] : v;
                    y[i] = v;
                }
#pragma unroll
                for (int i = 0; i < 8; ++i) x[i] = y[i];
            }
            unsigned h[8], m[8], l[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) split_bf16x3(x[i], h[i], m[i], l[i]);
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                b[0][q] = (h[2 * q] >> 16) | h[2 * q + 1];
                b[1][q] = (m[2 * q] >> 16) | m[2 * q + 1];
                b[2][q] = (l[2 * q] >> 16) | l[2 * q + 1];
            }
        } else {
            b[0] = part ? bf16x8_shift_down(r[0], sh) : r[0];
        }
    };
    auto store_dfq_e = [&](int tile, int e, float x0, float x1) {
        const int pa = tile * CL_TP + 2 * j;
        const int c = ct * 32 + (e & 3) + 8 * (e >> 2) + 4 * hh;
        T* dst = dFq + ((int64_t)n * C + c) * P + pa;
        if (pa + 1 < P) {
            if constexpr (F32) {
                const f32x2_u v = {x0, x1};
                if constexpr (NTQ) __builtin_nontemporal_store(v, reinterpret_cast<f32x2_u*>(dst));
                else *reinterpret_cast<f32x2_u*>(dst) = v;
            } else {
                const u16x2_u v = {f32_to_bf16(x0), f32_to_bf16(x1)};
                if constexpr (NTQ) __builtin_nontemporal_store(v, reinterpret_cast<u16x2_u*>(dst));
                else *reinterpret_cast<u16x2_u*>(dst) = v;
            }
        } else if (pa < P) {
            FeatIO<T>::store(dst, x0);
        }
    };

    // everything this block needs first is requested before the first wait
    f32x4 va[RT], vb[RT];
    fqraw_t fqn[DWN ? 4 : 1][NFR];
    u32x4 wa[DWN ? 1 : KS][NPQ];
    WSTAMP(0);
    BSTAMP(0);
    float de = 0.f;
    if (g.part_k != nullptr) de = part_k_lane(g.part_k, g.parts_k, n);
#pragma unroll
    for (int u = 0; u < RT; ++u) load_dl_u(t0, u, va[u], vb[u]);
    if constexpr (DWN) {
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) load_fq(t0, ks, fqn[ks]);
    } else {
        // Wn^T as two bf16 planes: lane (channel cc*32 + j, half hh) holds classes 16 ks + 8 hh + i
        float w[KS][8];
#pragma unroll
        for (int ks = 0; ks < KS; ++ks)
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const int kk = 16 * ks + 8 * hh + i;
                const float v = wn[(int64_t)min(kk, K - 1) * C + cc * 32 + j];
                w[ks][i] = kk < K ? v : 0.f;
            }
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            unsigned h[8], m[8], l[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) split_bf16x3(w[ks][i], h[i], m[i], l[i]);
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                wa[ks][0][q] = (h[2 * q] >> 16) | h[2 * q + 1];
                wa[ks][1][q] = (m[2 * q] >> 16) | m[2 * q + 1];
                if constexpr (NPQ == 3) wa[ks][2][q] = (l[2 * q] >> 16) | l[2 * q + 1];
            }
        }
    }
    float sc;
    if (g.part_k != nullptr) {
        sc = episode_scale(g.N, wave_sum(de));
        if (g.scale_out != nullptr && grp == 0 && tid == 0) g.scale_out[n] = sc;
    } else {
        sc = g.scale ? g.scale[n] : 1.f;
    }
    sc *= (g.gmul ? g.gmul[0] : 1.f);
    store_dl(dl0, t0, va, vb, sc);
    __syncthreads();
    WSTAMP(1);

    f32x16 acc[DWN ? RT : 1];
    if constexpr (DWN) {
#pragma unroll
        for (int rt = 0; rt < RT; ++rt)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[rt][e] = 0.f;
    }
    // dFq: the finished tile waits in s0 / s1 and is stored between the MFMAs of the next tile
    f32x16 s0, s1;
    int ts = -1;                                    // tile held in s0 / s1 (-1: none)
#pragma unroll
    for (int e = 0; e < 16; ++e) { s0[e] = 0.f; s1[e] = 0.f; }

#pragma unroll 1
    for (int t = t0; t < t1; ++t) {
        const unsigned short* cur = ((t - t0) & 1) ? dl1 : dl0;
        unsigned short* nxt = ((t - t0) & 1) ? dl0 : dl1;
        const int tn = min(t + 1, t1 - 1);
        if (t - t0 < 4) WSTAMP(2 + 4 * (t - t0));
        if constexpr (DWN) {
            constexpr int NB = F32 ? 3 : 1;
            u32x4 fqc[4][NB];
            const bool part = (t + 1) * CL_TP > P;         // uniform
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) fq_planes(t, ks, fqn[ks], part, fqc[ks]);
            __builtin_amdgcn_sched_barrier(0);
            const unsigned short* arow = cur + j * LDP + 8 * hh;
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                u32x4 A[3][RT];
#pragma unroll
                for (int s = 0; s < 3; ++s)
#pragma unroll
                    for (int rt = 0; rt < RT; ++rt)
                        A[s][rt] = *reinterpret_cast<const u32x4*>(arow + (s * NR + 32 * rt) * LDP + 16 * ks);
                // the loads of tile t+1, a few per k-step (see classify_bwd_wide_body)
                if (ks < RT) load_dl_u(tn, ks, va[ks], vb[ks]);
                load_fq(tn, ks, fqn[ks]);
                __builtin_amdgcn_sched_barrier(0);
                if (have) {
#define CLWB_TERM(SA, SB)                                                                                       \
    _Pragma("unroll") for (int rt = 0; rt < RT; ++rt) acc[rt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(        \
        __builtin_bit_cast(bf16x8, A[SA][rt]), __builtin_bit_cast(bf16x8, fqc[ks][SB]), acc[rt], 0, 0, 0);
                    if constexpr (F32) {           // dL plane x Fq plane, smallest terms first
                        CLWB_TERM(1, 1) CLWB_TERM(2, 0) CLWB_TERM(0, 2) CLWB_TERM(1, 0) CLWB_TERM(0, 1) CLWB_TERM(0, 0)
                    } else {
                        CLWB_TERM(2, 0) CLWB_TERM(1, 0) CLWB_TERM(0, 0)
                    }
#undef CLWB_TERM
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        } else {
            f32x16 a0, a1;
#pragma unroll
            for (int e = 0; e < 16; ++e) { a0[e] = 0.f; a1[e] = 0.f; }
            const unsigned short* brow = cur + j * LDK + 8 * hh;
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                // B = dL^T planes: rows of the even (tile 0) and odd (tile 1) positions, classes 16 ks + 8 hh ..
                u32x4 Bf[NPQ][2];                  // [plane][tile]
#pragma unroll
                for (int s = 0; s < NPQ; ++s)
#pragma unroll
                    for (int tt = 0; tt < 2; ++tt)
                        Bf[s][tt] = *reinterpret_cast<const u32x4*>(brow + (s * 64 + 32 * tt) * LDK + 16 * ks);
                if (ks < RT) load_dl_u(tn, ks, va[ks], vb[ks]);
                if (have && ts >= 0) {
#pragma unroll
                    for (int e = 0; e < 16; ++e)
                        if ((e * KS) / 16 == ks) store_dfq_e(ts, e, s0[e], s1[e]);
                }
                __builtin_amdgcn_sched_barrier(0);
                if (have && 16 * ks < K) {
#define CLWB_TERM(SA, SB)                                                                                       \
    a0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, wa[ks][SA]),                        \
                                                 __builtin_bit_cast(bf16x8, Bf[SB][0]), a0, 0, 0, 0);           \
    a1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, wa[ks][SA]),                        \
                                                 __builtin_bit_cast(bf16x8, Bf[SB][1]), a1, 0, 0, 0);
                    if constexpr (F32) {           // Wn plane x dL plane, smallest terms first
                        CLWB_TERM(1, 1) CLWB_TERM(2, 0) CLWB_TERM(0, 2) CLWB_TERM(1, 0) CLWB_TERM(0, 1) CLWB_TERM(0, 0)
                    } else {
                        CLWB_TERM(1, 0) CLWB_TERM(0, 1) CLWB_TERM(0, 0)
                    }
#undef CLWB_TERM
                }
                __builtin_amdgcn_sched_barrier(0);
            }
            s0 = a0;
            s1 = a1;
            ts = t;
        }
        if (t - t0 < 4) WSTAMP(4 + 4 * (t - t0));
        store_dl(nxt, tn, va, vb, sc);
        __syncthreads();
        if (t - t0 < 4) WSTAMP(5 + 4 * (t - t0));
    }
    WSTAMP(30);

    if constexpr (DWN) {
        if (have) {
#pragma unroll
            for (int rt = 0; rt < RT; ++rt)
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    const int kr = 32 * rt + (e & 3) + 8 * (e >> 2) + 4 * hh;
                    if (kr < K) dWn_part[(((int64_t)n * ngrp + grp) * K + kr) * C + ct * 32 + j] = acc[rt][e];
                }
        }
    } else {
        if (have && ts >= 0) {
#pragma unroll
            for (int e = 0; e < 16; ++e) store_dfq_e(ts, e, s0[e], s1[e]);
        }
    }
    WSTAMP(31);
    BSTAMP(1);
}

// groups of position tiles per episode for the persistent kernel: one round of blocks over the chip
// (2 roles x N x groups <= ~256 CUs) once there is more work than that, otherwise one tile per block.
// K <= 32 is cut into 512 blocks (two rounds of shorter blocks) once each of them still gets three tiles or
// more (measured, C2 step: 288.0 vs 294.8 us at 32 episodes, 0.987 vs 0.996 ms at 128; at 16 episodes
// 167.7 vs 163.4 us and at 8 episodes 92.1 vs 88.2 us the other way round).
__host__ __forceinline__ int cls_bwd_wide_groups(int64_t N, int64_t P, int RT) {
    const int64_t ntile = (P + CL_TP - 1) / CL_TP;
    const int64_t tgt = (RT == 1 && 2 * N * ntile >= 1536) ? 512 : 256;
    const int64_t tpb = (2 * N * ntile + tgt - 1) / tgt;
    return (int)((ntile + tpb - 1) / tpb);
}

template <typename T, int RT>
__global__ __launch_bounds__(256) void classify_bwd_kernel(ClsBwdArgs g) {
    extern __shared__ __attribute__((aligned(16))) float sm[];   // dL [KP][CL_LD]
    classify_bwd_body<T, RT, 4>(g, blockIdx.y, blockIdx.x, sm);
}

