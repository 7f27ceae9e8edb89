// K3 / K3e / K3b: per-pixel classifier on the matrix cores, binary-logit
// extraction, and the classifier's backward.
//
//   logits[n,k,p] = sum_c Wn[n,k,c] * Fq[n,c,p]        (model/common.py:390-394)
//
// Per episode this is a [K x C] x [C x P] GEMM with K = 21/81, C = 256,
// P = 1681: 9.7 FLOP/B at K = 21, i.e. HBM-bound, so the layout is chosen for
// the Fq stream, not for MFMA utilisation:
//   * Fq is NCHW, so for a fixed channel the positions are contiguous.  The
//     exact-fp32 MFMA v_mfma_f32_32x32x2_f32 wants B[k = lane>>5][j = lane&31];
//     each lane loads TWO adjacent positions (8 B) of channel c + (lane>>5) and
//     feeds them to two column tiles (column j of tile t is position 2j + t),
//     so a wave-load is two fully used 256-B row segments and nothing is staged
//     through LDS for the streamed operand.
//   * The class-weight tile Wn[n] (21.5 KB at K = 21) is staged once per block
//     in LDS, row stride C + 1 so the A-operand read (32 rows, one column) is
//     bank-conflict free.
//   * The 4 waves of a block split the channels (the reduction dimension) and
//     are summed through LDS in a fixed order (deterministic), which gives
//     N * ceil(P/64) * 4 waves of parallelism at batch 8.
// fp32 in, fp32 accumulate: bit-for-bit an fmaf chain per output (guide §3).
#include "denet_common.hpp"

#define CL_TP 64          // positions per block
#define CL_LD 66          // LDS row stride of a 64-wide tile (even: 8-byte reads)

typedef float f32x2_u __attribute__((ext_vector_type(2), aligned(4)));
typedef float f32x4_u __attribute__((ext_vector_type(4), aligned(4)));
typedef unsigned short u16x2_u __attribute__((ext_vector_type(2), aligned(2)));

template <typename T>
__device__ __forceinline__ void load2(const T* p, bool ok0, bool ok1, bool fast, float& a, float& b);
template <>
__device__ __forceinline__ void load2<float>(const float* p, bool ok0, bool ok1, bool fast, float& a, float& b) {
    if (fast) {
        f32x2_u v = *reinterpret_cast<const f32x2_u*>(p);
        a = v[0]; b = v[1];
    } else {
        a = ok0 ? p[0] : 0.f;
        b = ok1 ? p[1] : 0.f;
    }
}
template <>
__device__ __forceinline__ void load2<unsigned short>(const unsigned short* p, bool ok0, bool ok1, bool fast,
                                                     float& a, float& b) {
    if (fast) {
        u16x2_u v = *reinterpret_cast<const u16x2_u*>(p);
        a = bf16_to_f32(v[0]); b = bf16_to_f32(v[1]);
    } else {
        a = ok0 ? bf16_to_f32(p[0]) : 0.f;
        b = ok1 ? bf16_to_f32(p[1]) : 0.f;
    }
}

template <typename T, int RT, int NLD>
__global__ __launch_bounds__(256) void classify_fwd_kernel(const T* __restrict__ Fq, const float* __restrict__ Wn,
                                                          int64_t wn_stride, const int64_t* __restrict__ labels,
                                                          int way, int K, int C, int P, int base,
                                                          float* __restrict__ logits, float* __restrict__ bin) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    const int n = blockIdx.y, p0 = blockIdx.x * CL_TP;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int j = lane & 31, hh = lane >> 5;
    const int ldw = C + 1;

    const int cper = C >> 2;                       // channels per wave: 16, 32, 64 or 128
    const int cbeg = wave * cper;
    const int pa = p0 + 2 * j;
    const bool ok0 = pa < P, ok1 = pa + 1 < P;
    const bool fast = (p0 + CL_TP <= P);
    const T* fq = Fq + (int64_t)n * C * P + pa;
    // The streamed operand first: up to 64 channels (32 k-steps, 16 KB per wave) of Fq go in flight
    // before anything else, so the Wn staging below hides under their latency.
    STAMP(0);
    constexpr int chunk = 2 * NLD;                 // channels per pass of this wave (== cper: C is 64 or 256)
    float b0[NLD], b1[NLD];
#pragma unroll
    for (int u = 0; u < NLD; ++u) load2<T>(fq + (int64_t)(cbeg + 2 * u + hh) * P, ok0, ok1, fast, b0[u], b1[u]);

    // stage Wn[n] -> LDS [K][C+1]: 16-byte loads, all of a thread's loads issued before the first store
    {
        const float* src = Wn + (int64_t)n * wn_stride;
        const int nv = (K * C) >> 2;
#pragma unroll 1
        for (int base = 0; base < nv; base += 256 * 8) {
            f32x4 v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int i4 = base + u * 256 + tid;
                if (i4 < nv) v[u] = *reinterpret_cast<const f32x4*>(src + (int64_t)i4 * 4);
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int i4 = base + u * 256 + tid;
                if (i4 < nv) {
                    const int idx = i4 * 4;
                    const int k = idx / C, c = idx - k * C;
                    float* d = sm + k * ldw + c;
                    d[0] = v[u][0]; d[1] = v[u][1]; d[2] = v[u][2]; d[3] = v[u][3];
                }
            }
        }
    }
    __syncthreads();
    STAMP(1);

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
                load2<T>(fq + (int64_t)(c + chunk + 2 * u + hh) * P, ok0, ok1, fast, b0[u], b1[u]);
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
        for (int k = rg; k < K; k += 4) {
            const int o = k * CL_LD + col;
            const float v = ((sm[o] + sm[ws + o]) + sm[2 * ws + o]) + sm[3 * ws + o];
            sm[o] = v;
            if (p < P) logits[((int64_t)n * K + k) * P + p] = v;
        }
    }
    STAMP(3);
    if (bin == nullptr) return;
    __syncthreads();
    // binary logits: ch1 = row label, ch0 = max over the other rows (utils/operations.py:327-337)
    for (int item = tid; item < way * CL_TP; item += 256) {
        const int jw = item >> 6, col = item & 63;
        const int p = p0 + col;
        if (p >= P) continue;
        const int64_t lab = labels[(int64_t)n * way + jw];
        float fg = 0.f, bg;
        if (lab >= 0 && lab < K) fg = sm[(int)lab * CL_LD + col];
        if (base) {
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

// ---------------------------------------------------------------------------
// K3b: backward.  One block per (episode, 64-position tile):
//   dL[k][p]   = scale[n] * (dlog_a + dlog_b)                     -> LDS
//   dFq[c][p]  = sum_k Wn[k][c] * dL[k][p]                        (A = Wn^T from L2, B = dL from LDS)
//   dWn_part[tile][k][c] = sum_{p in tile} dL[k][p] * Fq[c][p]    (A = dL from LDS, B = Fq^T: each
//        lane holds 32 consecutive positions of one channel row in registers; the MFMA k index of
//        step s is position 32*(lane>>5) + s for both operands)
// The partial dWn slabs are summed by K2b (no float atomics: deterministic).
// ---------------------------------------------------------------------------
template <typename T, int RT>
__global__ __launch_bounds__(256) void classify_bwd_kernel(const T* __restrict__ Fq, const float* __restrict__ Wn,
                                                          const float* __restrict__ dlog_a,
                                                          const float* __restrict__ dlog_b,
                                                          const float* __restrict__ scale,
                                                          const float* __restrict__ gmul, int K, int C, int P,
                                                          int n_part, T* __restrict__ dFq,
                                                          float* __restrict__ dWn_part) {
    extern __shared__ __attribute__((aligned(16))) float sm[];   // dL [KP][CL_LD]
    const int n = blockIdx.y, tile = blockIdx.x, p0 = tile * CL_TP;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int j = lane & 31, hh = lane >> 5;
    const int KP = (K + 1) & ~1;
    const float sc = (scale ? scale[n] : 1.f) * (gmul ? gmul[0] : 1.f);
    const float* wn = Wn + (int64_t)n * K * C;
    const bool fast = (p0 + CL_TP <= P);
    const int nct = C >> 5;                      // 32-channel tiles: 2 (C=64) or 8 (C=256); <= 2 per wave
    constexpr int NA = RT * 16;
    // With K <= 32 every global operand of this block fits in registers, so ALL of them are
    // requested before the first wait: the block then pays one memory latency, not five.
    constexpr bool PRE = (RT == 1);
    float av[2][PRE ? NA : 1];
    float fv[2][PRE ? 32 : 1];

    auto load_av = [&](int ct, float* dst) {
#pragma unroll
        for (int u = 0; u < NA; ++u) {
            const int kk = 2 * u + hh;
            dst[u] = kk < K ? wn[(int64_t)kk * C + ct * 32 + j] : 0.f;
        }
    };
    auto load_fv = [&](int ct, float* dst) {
        const int pb = p0 + 32 * hh;
        const T* src = Fq + ((int64_t)n * C + ct * 32 + j) * P + pb;
        if (fast) {
            if constexpr (sizeof(T) == 4) {
#pragma unroll
                for (int m = 0; m < 8; ++m) {
                    f32x4_u v = *reinterpret_cast<const f32x4_u*>((const float*)src + 4 * m);
                    dst[4 * m] = v[0]; dst[4 * m + 1] = v[1]; dst[4 * m + 2] = v[2]; dst[4 * m + 3] = v[3];
                }
            } else {
#pragma unroll
                for (int s = 0; s < 32; ++s) dst[s] = FeatIO<T>::load(src + s);
            }
        } else {
#pragma unroll
            for (int s = 0; s < 32; ++s) dst[s] = (pb + s < P) ? FeatIO<T>::load(src + s) : 0.f;
        }
    };
    if constexpr (PRE) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int ct = wave + 4 * i;
            if (ct < nct) {
                if (dFq != nullptr) load_av(ct, av[i]);
                if (dWn_part != nullptr) load_fv(ct, fv[i]);
            }
        }
    }

#pragma unroll 1
    for (int base = 0; base < KP * CL_TP; base += 256 * 6) {
        float va[6], vb[6];
#pragma unroll
        for (int u = 0; u < 6; ++u) {
            const int idx = base + u * 256 + tid;
            const int k = idx >> 6, col = idx & 63;
            const int p = p0 + col;
            va[u] = 0.f; vb[u] = 0.f;
            if (idx < KP * CL_TP && k < K && p < P) {
                const int64_t gi = ((int64_t)n * K + k) * P + p;
                va[u] = dlog_a[gi];
                if (dlog_b) vb[u] = dlog_b[gi];
            }
        }
#pragma unroll
        for (int u = 0; u < 6; ++u) {
            const int idx = base + u * 256 + tid;
            if (idx < KP * CL_TP) sm[(idx >> 6) * CL_LD + (idx & 63)] = (va[u] + vb[u]) * sc;
        }
    }
    __syncthreads();

    // ---- dFq ----
    if (dFq != nullptr) {
        const int pa = p0 + 2 * j;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int ct = wave + 4 * i;
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
                if (pa < P) FeatIO<T>::store(dst, a0[e]);
                if (pa + 1 < P) FeatIO<T>::store(dst + 1, a1[e]);
            }
        }
    }

    // ---- dWn partial ----
    if (dWn_part != nullptr) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int ct = wave + 4 * i;
            if (ct >= nct) continue;
            const int c0 = ct * 32;
            float fvl[32];
            if constexpr (!PRE) load_fv(ct, fvl);
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
                    const float bq = PRE ? fv[i][PRE ? s : 0] : fvl[s];
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
// K3e stand-alone: get_binary_logits on an existing logits tensor (utils/operations.py:313-338)
// and the batch mean of extended weights for mode='deploy' (model/common.py:369-372).
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void binary_logits_kernel(const float* __restrict__ logits,
                                                           const int64_t* __restrict__ labels, int way, int K,
                                                           int P, int base, float* __restrict__ bin) {
    const int nj = blockIdx.y, n = nj / way;
    const int p = blockIdx.x * 256 + threadIdx.x;
    if (p >= P) return;
    const int64_t lab = labels[nj];
    const float* src = logits + (int64_t)n * K * P + p;
    float fg = 0.f, bg;
    if (lab >= 0 && lab < K) fg = src[(int64_t)lab * P];
    if (base) {
        bg = -INFINITY;
        for (int k = 0; k < K; ++k)
            if (k != lab) bg = fmaxf(bg, src[(int64_t)k * P]);
    } else {
        bg = src[0];
    }
    float* dst = bin + ((int64_t)nj * 2) * P + p;
    dst[0] = bg;
    dst[P] = fg;
}

__global__ __launch_bounds__(256) void batch_mean_kernel(const float* __restrict__ x, int N, int64_t per,
                                                        float* __restrict__ out) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= per) return;
    float t = 0.f;
    for (int n = 0; n < N; ++n) t += x[(int64_t)n * per + i];
    out[i] = t / (float)N;
}

// ---------------------------------------------------------------------------
// C ABI
// ---------------------------------------------------------------------------
static inline bool pos(int64_t v) { return v > 0 && v < ((int64_t)1 << 30); }
static inline bool c_ok(int64_t C) { return C == 64 || C == 256; }

template <typename KernelT>
static int set_lds(KernelT kern, size_t lds) {
    return denet_ensure_dynamic_lds(reinterpret_cast<const void*>(kern), lds);
}

template <typename T>
static int launch_classify_fwd(const T* Fq, const float* Wn, int64_t wn_stride, const int64_t* labels, int64_t N,
                               int64_t way, int64_t K, int64_t C, int64_t P, int base, float* logits, float* bin,
                               hipStream_t st) {
    const int RT = (int)((K + 31) / 32);
    const size_t a = (size_t)K * (C + 1), b = (size_t)4 * K * CL_LD;
    const size_t lds = (a > b ? a : b) * 4;
    if (lds > 160 * 1024) return DENET_E_UNSUPPORTED;
    dim3 grid((unsigned)((P + CL_TP - 1) / CL_TP), (unsigned)N), block(256);
#define CL_FWD2(R, L)                                                                                       \
    {                                                                                                       \
        int rc = set_lds(classify_fwd_kernel<T, R, L>, lds);                                                \
        if (rc) return rc;                                                                                  \
        hipLaunchKernelGGL((classify_fwd_kernel<T, R, L>), grid, block, lds, st, Fq, Wn, wn_stride, labels, \
                           (int)way, (int)K, (int)C, (int)P, base, logits, bin);                            \
    }
#define CL_FWD(R)                       \
    if (C == 64) CL_FWD2(R, 8) else CL_FWD2(R, 32)
    switch (RT) {
        case 1: CL_FWD(1); break;
        case 2: CL_FWD(2); break;
        case 3: CL_FWD(3); break;
        case 4: CL_FWD(4); break;
        default: return DENET_E_UNSUPPORTED;
    }
#undef CL_FWD2
#undef CL_FWD
    DENET_CHECK_LAUNCH();
    return DENET_OK;
}

extern "C" int denet_classify_fwd(const void* Fq, int feat_dtype, const float* Wn, const int64_t* labels, int64_t N,
                                  int64_t way, int64_t K, int64_t C, int64_t P, int base, float* logits,
                                  float* logits_binary, void* stream) {
    if (!Fq || !Wn || !logits) return DENET_E_BADARG;
    if (logits_binary && !labels) return DENET_E_BADARG;
    if (!pos(N) || !pos(K) || !pos(C) || !pos(P) || way < 0) return DENET_E_BADARG;
    if (!c_ok(C) || K > 128 || N > 65535) return DENET_E_UNSUPPORTED;
    hipStream_t st = (hipStream_t)stream;
    if (feat_dtype == DENET_F32)
        return launch_classify_fwd<float>((const float*)Fq, Wn, K * C, labels, N, way, K, C, P, base, logits,
                                          logits_binary, st);
    if (feat_dtype == DENET_BF16)
        return launch_classify_fwd<unsigned short>((const unsigned short*)Fq, Wn, K * C, labels, N, way, K, C, P,
                                                   base, logits, logits_binary, st);
    return DENET_E_UNSUPPORTED;
}

extern "C" int denet_classify_shared_fwd(const void* Fq, int feat_dtype, const float* W, int64_t N, int64_t K,
                                         int64_t C, int64_t P, float* logits, void* stream) {
    if (!Fq || !W || !logits) return DENET_E_BADARG;
    if (!pos(N) || !pos(K) || !pos(C) || !pos(P)) return DENET_E_BADARG;
    if (!c_ok(C) || K > 128 || N > 65535) return DENET_E_UNSUPPORTED;
    hipStream_t st = (hipStream_t)stream;
    if (feat_dtype == DENET_F32)
        return launch_classify_fwd<float>((const float*)Fq, W, 0, nullptr, N, 0, K, C, P, 1, logits, nullptr, st);
    if (feat_dtype == DENET_BF16)
        return launch_classify_fwd<unsigned short>((const unsigned short*)Fq, W, 0, nullptr, N, 0, K, C, P, 1,
                                                   logits, nullptr, st);
    return DENET_E_UNSUPPORTED;
}

extern "C" int64_t denet_classify_bwd_parts(int64_t P) { return (P + CL_TP - 1) / CL_TP; }

template <typename T>
static int launch_classify_bwd(const T* Fq, const float* Wn, const float* da, const float* db, const float* scale,
                               const float* gmul, int64_t N, int64_t K, int64_t C, int64_t P, T* dFq, float* dWn_part, hipStream_t st) {
    const int RT = (int)((K + 31) / 32);
    const int n_part = (int)denet_classify_bwd_parts(P);
    const size_t lds = (size_t)((K + 1) & ~1) * CL_LD * 4;
    dim3 grid((unsigned)n_part, (unsigned)N), block(256);
#define CL_BWD(R)                                                                                          \
    hipLaunchKernelGGL((classify_bwd_kernel<T, R>), grid, block, lds, st, Fq, Wn, da, db, scale, gmul, (int)K, \
                       (int)C, (int)P, n_part, dFq, dWn_part)
    switch (RT) {
        case 1: CL_BWD(1); break;
        case 2: CL_BWD(2); break;
        case 3: CL_BWD(3); break;
        case 4: CL_BWD(4); break;
        default: return DENET_E_UNSUPPORTED;
    }
#undef CL_BWD
    DENET_CHECK_LAUNCH();
    return DENET_OK;
}

extern "C" int denet_classify_bwd(const void* Fq, int feat_dtype, const float* Wn, const float* dlog_a,
                                  const float* dlog_b, const float* scale, const float* gmul, int64_t N,
                                  int64_t K, int64_t C, int64_t P, void* dFq, float* dWn_part, void* stream) {
    if (!Fq || !Wn || !dlog_a) return DENET_E_BADARG;
    if (!pos(N) || !pos(K) || !pos(C) || !pos(P)) return DENET_E_BADARG;
    if (!c_ok(C) || K > 128 || N > 65535) return DENET_E_UNSUPPORTED;
    hipStream_t st = (hipStream_t)stream;
    if (feat_dtype == DENET_F32)
        return launch_classify_bwd<float>((const float*)Fq, Wn, dlog_a, dlog_b, scale, gmul, N, K, C, P, (float*)dFq,
                                          dWn_part, st);
    if (feat_dtype == DENET_BF16)
        return launch_classify_bwd<unsigned short>((const unsigned short*)Fq, Wn, dlog_a, dlog_b, scale, gmul, N, K, C, P,
                                                   (unsigned short*)dFq, dWn_part, st);
    return DENET_E_UNSUPPORTED;
}

extern "C" int denet_binary_logits(const float* logits, const int64_t* labels, int64_t N, int64_t way, int64_t K,
                                   int64_t P, int base, float* logits_binary, void* stream) {
    if (!logits || !labels || !logits_binary) return DENET_E_BADARG;
    if (!pos(N) || !pos(way) || !pos(K) || !pos(P)) return DENET_E_BADARG;
    if (N * way > 65535) return DENET_E_UNSUPPORTED;
    hipLaunchKernelGGL(binary_logits_kernel, dim3((unsigned)((P + 255) / 256), (unsigned)(N * way)), dim3(256), 0,
                       (hipStream_t)stream, logits, labels, (int)way, (int)K, (int)P, base, logits_binary);
    DENET_CHECK_LAUNCH();
    return DENET_OK;
}

extern "C" int denet_batch_mean(const float* x, int64_t N, int64_t per, float* out, void* stream) {
    if (!x || !out) return DENET_E_BADARG;
    if (!pos(N) || !pos(per)) return DENET_E_BADARG;
    hipLaunchKernelGGL(batch_mean_kernel, dim3((unsigned)((per + 255) / 256)), dim3(256), 0, (hipStream_t)stream, x,
                       (int)N, per, out);
    DENET_CHECK_LAUNCH();
    return DENET_OK;
}
