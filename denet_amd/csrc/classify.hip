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
#include <type_traits>

#include "denet_common.hpp"
#include "dev_classify.hpp"

#include "dev_classify_fwd.hpp"

// Epilogue of the wave-per-tile kernels: logits from the accumulators (C/D layout of the 32x32 MFMAs: column =
// lane & 31, row = (e & 3) + 8 (e >> 2) + 4 (lane >> 5)), then the binary logits.  A wave holds NTL column tiles;
// column j of tile t is position pa + t (NTL = 2: pa = p0 + 2 j, the two tiles interleave; NTL = 1: pa = p0 + j).
template <int RT, int NTL>
__device__ __forceinline__ void wave_tile_epilogue(f32x16 (&acc)[RT][NTL], const int64_t* __restrict__ labels, int way,
                                                   int K, int P, int base, float* __restrict__ logits,
                                                   float* __restrict__ bin, int n, int pa, int hh) {
    // ---- logits: straight from the accumulators ----
    float* lg = logits + (int64_t)n * K * P;
#pragma unroll
    for (int rt = 0; rt < RT; ++rt)
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const int row = 32 * rt + (e & 3) + 8 * (e >> 2) + 4 * hh;
            if (row < K) {
                float* dst = lg + (int64_t)row * P + pa;
                if constexpr (NTL == 2) {
                    if (pa + 1 < P) *reinterpret_cast<f32x2_u*>(dst) = f32x2_u{acc[rt][0][e], acc[rt][1][e]};
                    else if (pa < P) dst[0] = acc[rt][0][e];
                } else {
                    if (pa < P) dst[0] = acc[rt][0][e];
                }
            }
        }
    if (bin == nullptr) return;

    // ---- binary logits (utils/operations.py:327-337) ----
    // per lane and position: the largest and the second largest entry over this lane's rows (ties count: two
    // equal maxima give m2 == m1), then merged across the two lane halves.  max over k != lab is then
    // m2 if the label row holds the maximum (fg == m1), else m1 -- no row index needed.
    float m1[NTL], m2[NTL];
#pragma unroll
    for (int t = 0; t < NTL; ++t) { m1[t] = -INFINITY; m2[t] = -INFINITY; }
#pragma unroll
    for (int rt = 0; rt < RT; ++rt)
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const int row = 32 * rt + (e & 3) + 8 * (e >> 2) + 4 * hh;
#pragma unroll
            for (int t = 0; t < NTL; ++t) {
                const float v = row < K ? acc[rt][t][e] : -INFINITY;
                m2[t] = __builtin_amdgcn_fmed3f(m1[t], m2[t], v);       // second largest of {m1 >= m2, v}
                m1[t] = fmaxf(m1[t], v);
            }
        }
#pragma unroll
    for (int t = 0; t < NTL; ++t) {
        const float o1 = __shfl_xor(m1[t], 32, 64), o2 = __shfl_xor(m2[t], 32, 64);
        const float hi = fmaxf(m1[t], o1), lo = fminf(m1[t], o1);
        m2[t] = fmaxf(lo, fmaxf(m2[t], o2));
        m1[t] = hi;
    }
    for (int jw = 0; jw < way; ++jw) {
        const int64_t lab64 = labels[(int64_t)n * way + jw];
        const bool labok = lab64 >= 0 && lab64 < K;
        const int lab = labok ? (int)lab64 : 0;
        // the accumulator that holds row `sel`: (rt, e) are wave-uniform, the lane half is (row >> 2) & 1
        auto pick = [&](int sel, float (&out)[NTL]) {
            const int r = sel & 31, idx = (sel >> 5) * 16 + (r & 3) + 4 * (r >> 3), half = (r >> 2) & 1;
            float v[NTL];
#pragma unroll
            for (int t = 0; t < NTL; ++t) v[t] = 0.f;
#pragma unroll
            for (int rt = 0; rt < RT; ++rt)
#pragma unroll
                for (int e = 0; e < 16; ++e)
                    if (idx == rt * 16 + e) {                                               // uniform branch
#pragma unroll
                        for (int t = 0; t < NTL; ++t) v[t] = acc[rt][t][e];
                    }
#pragma unroll
            for (int t = 0; t < NTL; ++t) {
                const float o = __shfl_xor(v[t], 32, 64);
                out[t] = hh == half ? v[t] : o;
            }
        };
        float fg[NTL], outb[NTL];
        pick(lab, fg);
        if (base) {
#pragma unroll
            for (int t = 0; t < NTL; ++t) outb[t] = (labok && fg[t] == m1[t]) ? m2[t] : m1[t];
        } else {
            pick(0, outb);
        }
        if (hh == 0) {
            float* dst = bin + ((int64_t)(n * way + jw) * 2) * P + pa;
            if constexpr (NTL == 2) {
                const float f0 = labok ? fg[0] : 0.f, f1 = labok ? fg[1] : 0.f;
                if (pa + 1 < P) {
                    *reinterpret_cast<f32x2_u*>(dst) = f32x2_u{outb[0], outb[1]};
                    *reinterpret_cast<f32x2_u*>(dst + P) = f32x2_u{f0, f1};
                } else if (pa < P) {
                    dst[0] = outb[0];
                    dst[P] = f0;
                }
            } else if (pa < P) {
                dst[0] = outb[0];
                dst[P] = labok ? fg[0] : 0.f;
            }
        }
    }
}

// ---------------------------------------------------------------------------
// K > 32 (COCO: K = 81): Wn[n] is 83 KB of LDS, one block per CU, and the layout above re-stages it for every
// 64 positions (C3: 6 us of a block's 19 in staging, 3.6 in the cross-wave reduction, the MFMA pipe busy a
// third of the time).  Here a block keeps Wn[n] resident for FOUR position tiles: wave w owns tile 4*g + w and
// ALL channels, so nothing is reduced across waves and no partial tile goes through LDS; Fq is streamed in
// chunks of 32 channels, the next chunk's loads in flight under the current chunk's MFMAs (96 of them, ~6k
// cycles); logits are stored straight from the accumulators (a row of a wave store = 256 contiguous bytes) and
// the binary logits come from a per-lane top-2 scan of the accumulators merged across the two lane halves.
// ---------------------------------------------------------------------------
#define CLW_NLD 16         // k-steps (pairs of channels) per chunk
template <typename T, int RT, bool BUILD, int CJ>
__global__ __launch_bounds__(256) void classify_fwd_wide4_kernel(const T* __restrict__ Fq, const float* __restrict__ Wn,
                                                               int64_t wn_stride, const int64_t* __restrict__ labels,
                                                               int way, int K, int P, int base,
                                                               float* __restrict__ logits, float* __restrict__ bin,
                                                               WnBuild wb) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    constexpr int C = 64 * CJ;
    const int n = blockIdx.y;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int j = lane & 31, hh = lane >> 5;
    const int ldw = C + 1;
    const int ntile = (P + CL_TP - 1) / CL_TP;
    const int tile = blockIdx.x * 4 + wave;
    const bool active = tile < ntile;
    const int p0 = (active ? tile : ntile - 1) * CL_TP;
    const int pa = p0 + 2 * j;
    const T* fq = Fq + (int64_t)n * C * P;

    typedef typename Raw2<T>::type raw_t;
    const int pc = max(min(pa, P - 2), 0), sh = pa - pc;
    raw_t bA[CLW_NLD], bB[CLW_NLD];
    // raw loads only; __builtin_amdgcn_sched_barrier keeps the compiler from sinking them to their uses
    auto load_chunk = [&](int c, raw_t (&x)[CLW_NLD]) {
#pragma unroll
#ifdef DENET_DIAG_NO_FQ
        for (int u = 0; u < CLW_NLD; ++u) x[u] = raw_t{};
#else
        for (int u = 0; u < CLW_NLD; ++u) x[u] = raw2_load(fq + (int64_t)(c + 2 * u + hh) * P, pc);
#endif
        __builtin_amdgcn_sched_barrier(0);
    };
    STAMP(0);
    load_chunk(0, bA);                             // in flight under the Wn staging

    constexpr int SUc = RT == 1 ? 8 : RT == 2 ? 16 : RT == 3 ? 24 : 16;
    stage_wn<BUILD, 256, C / 16, SUc>(sm, Wn, wn_stride, labels, way, K, C, n, wb);
    // rows K .. 32*RT-1 of the LDS image are zeros: the MFMA A operand needs no row mask (a select between the
    // LDS read and the MFMA also re-writes a register the previous MFMA may still be reading)
    for (int i = K * ldw + tid; i < 32 * RT * ldw; i += 256) sm[i] = 0.f;
    __syncthreads();
    if constexpr (BUILD) {
        float* dst = wb.Wn_out + (int64_t)n * K * C;
        for (int k = blockIdx.x; k < K; k += gridDim.x)
            for (int c = tid; c < C; c += 256) dst[k * C + c] = sm[k * ldw + c];
    }
    if (!active) return;                           // (no barrier below this line)
    STAMP(1);

    f32x16 acc[RT][2];
#pragma unroll
    for (int rt = 0; rt < RT; ++rt)
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[rt][t][e] = 0.f;
    int arow[RT];
#pragma unroll
    for (int rt = 0; rt < RT; ++rt) arow[rt] = (32 * rt + j) * ldw;
    auto mfma_chunk = [&](int c, const raw_t (&x)[CLW_NLD]) {
        // the chunk's A operands (RT x 16 LDS words per lane) in one burst, then MFMAs back to back: with one
        // wave per SIMD an LDS read issued right before its MFMA exposes its latency every other k-step
        float av[CLW_NLD][RT];
#pragma unroll
        for (int u = 0; u < CLW_NLD; ++u)
#pragma unroll
            for (int rt = 0; rt < RT; ++rt) av[u][rt] = sm[arow[rt] + c + 2 * u + hh];
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int u = 0; u < CLW_NLD; ++u) {
            float x0, x1;
            raw2_get(x[u], sh, x0, x1);
#pragma unroll
            for (int rt = 0; rt < RT; ++rt) {
                acc[rt][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[u][rt], x0, acc[rt][0], 0, 0, 0);
                acc[rt][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[u][rt], x1, acc[rt][1], 0, 0, 0);
            }
        }
        __builtin_amdgcn_sched_barrier(0);
    };
    constexpr int CH = 2 * CLW_NLD;                // channels per chunk
#pragma unroll 1
    for (int c = 0; c < C; c += 2 * CH) {
        load_chunk(c + CH, bB);                    // C is a multiple of 64: chunk pairs are always whole
        mfma_chunk(c, bA);
        load_chunk(c + 2 * CH < C ? c + 2 * CH : 0, bA);     // (the last prefetch re-reads chunk 0: unused)
        mfma_chunk(c + CH, bB);
    }

    STAMP(2);
    wave_tile_epilogue<RT, 2>(acc, labels, way, K, P, base, logits, bin, n, pa, hh);
    STAMP(4);
}

// ---------------------------------------------------------------------------
// bf16 features, K <= 96: the same block shape (a wave per 64-position tile, four tiles per block, Wn[n] resident
// in LDS) on the bf16 matrix pipe, v_mfma_f32_32x32x16_bf16 = 16x the rate of the fp32 instruction.  The streamed
// operand is bf16 already; the fp32 class weights are split into three bf16 planes, hi + mid + lo = the fp32
// value EXACTLY (8 + 8 + 8 significand bits, truncating splits), so every product bf16 x bf16 is exact in fp32
// and the result differs from the fp32-MFMA kernel only in the order of the fp32 additions:
//   logits = sum_c (hi + mid + lo)[k][c] * Fq[c][p]          3 MFMAs of 16 channels instead of 8 of 2.
// Matrix work per SIMD at K = 81, C = 256: 288 MFMAs x 32 cycles = 9.2 k cycles against 768 x 64 = 49 k, which
// leaves the kernel bound by its loads -- so ALL of the wave's Fq (C/2 dwords per lane: 8 channels x 2 adjacent
// positions per 16-channel k-step) is requested in two halves right after the Wn planes are written (Wn first: vmcnt
// retires in order); the MFMA loop consumes the k-steps as their loads retire.
//   A operand (Wn planes): lane (r = lane & 31, h = lane >> 5) holds row r, channels c0 + 8h .. 8h+7 = one
//     ds_read_b128 of plane[s][32 rt + r]; plane rows are C + 8 bf16 (528 B: 16 lanes x 16 B cover the 64 banks).
//   B operand (Fq): the same lane holds channels c0 + 8h + i of column r; the dword loaded for channel i carries
//     positions pa (low half) and pa + 1 (high half) = column r of tile 0 and of tile 1, so two v_perm_b32 per
//     channel pair build both tiles' fragments.
// LDS: 3 x 32 RT x (C + 8) x 2 B = 152 KB at RT = 3 (the planes are written straight from the global loads,
// stage_wn_planes: no fp32 image in between).
// ---------------------------------------------------------------------------
// (bf16x8, u32x4, split_bf16x3: dev_classify.hpp)

// Wn[n] from global memory straight into the three bf16 planes (no fp32 image, no second pass through LDS): the
// loads, row selection and dropout of stage_wn, then split_bf16x3 on the loaded registers and 8-byte plane
// stores.  Rows K .. 32 RT - 1 are written as zeros.  BUILD: the block also writes its share of the assembled
// fp32 Wn[n] (rows k with k % gridDim.x == blockIdx.x) for the backward pass, from the same registers.
// NT threads; the caller puts a barrier behind it.
template <int RT, bool BUILD, int CJ, int NT>
__device__ __forceinline__ void stage_wn_planes(float* sm, const float* __restrict__ Wn, int64_t wn_stride,
                                                const int64_t* __restrict__ labels, int way, int K, int n,
                                                const WnBuild& wb) {
    constexpr int C = 64 * CJ, NR = 32 * RT, LDD = (C + 8) / 2, NSL = C / 16, NW = NT / 64;
    constexpr int NV = NR * C / 4, SU = (NV + NT - 1) / NT;        // 16-byte pieces, per thread
    constexpr int cshift = CJ == 1 ? 6 : 8;
    const int tid = threadIdx.x;
    unsigned* pl = reinterpret_cast<unsigned*>(sm);
    typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
    auto put = [&](int k, int c, const f32x4& v) {                 // c: a multiple of 4
        unsigned h[4], m[4], l[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) split_bf16x3(v[e], h[e], m[e], l[e]);
        unsigned* d = pl + k * LDD + (c >> 1);
        *reinterpret_cast<u32x2*>(d) = u32x2{(h[0] >> 16) | h[1], (h[2] >> 16) | h[3]};
        *reinterpret_cast<u32x2*>(d + NR * LDD) = u32x2{(m[0] >> 16) | m[1], (m[2] >> 16) | m[3]};
        *reinterpret_cast<u32x2*>(d + 2 * NR * LDD) = u32x2{(l[0] >> 16) | l[1], (l[2] >> 16) | l[3]};
    };
    auto mine_row = [&](int k) { return (k % (int)gridDim.x) == (int)blockIdx.x; };
    float* wout = BUILD ? wb.Wn_out + (int64_t)n * K * C : nullptr;
    // the rows this episode's ways overwrite: thread group jw = tid / 64 sums the partials of way jw
    f32x4 pv = {0.f, 0.f, 0.f, 0.f};
    int pk = -1, pc = 0;
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
        pk = win ? (int)mine : -1;
        pc = c4 * 4;
    }
    {
        const float* src = BUILD ? wb.weight : Wn + (int64_t)n * wn_stride;
        const float* msk = (BUILD && wb.weight_mask) ? wb.weight_mask + (int64_t)n * K * C : nullptr;
        const int64_t* lab = labels + (int64_t)n * way;
        const int nv = (K * C) >> 2;
        f32x4 v[SU];
#pragma unroll
        for (int u = 0; u < SU; ++u) v[u] = *reinterpret_cast<const f32x4*>(src + (int64_t)min(u * NT + tid, nv - 1) * 4);
        if (BUILD && msk != nullptr) {
            f32x4 mm[SU];
#pragma unroll
            for (int u = 0; u < SU; ++u) mm[u] = *reinterpret_cast<const f32x4*>(msk + (int64_t)min(u * NT + tid, nv - 1) * 4);
#pragma unroll
            for (int u = 0; u < SU; ++u) { v[u][0] *= mm[u][0]; v[u][1] *= mm[u][1]; v[u][2] *= mm[u][2]; v[u][3] *= mm[u][3]; }
        } else if (BUILD && wb.wdrop.on) {
#pragma unroll
            for (int u = 0; u < SU; ++u) {
                const int i4 = min(u * NT + tid, nv - 1);
                const f32x4 m = drop_mult4(wb.wdrop, DENET_DROP_WEIGHT, (unsigned)(n * nv + i4));
                v[u][0] *= m[0]; v[u][1] *= m[1]; v[u][2] *= m[2]; v[u][3] *= m[3];
            }
        }
#pragma unroll
        for (int u = 0; u < SU; ++u) {
            const int i4 = u * NT + tid;
            if (i4 < NV) {
                const int idx = i4 * 4;
                const int k = idx >> cshift, c = idx & (C - 1);
                if (k >= K) {
                    put(k, c, f32x4{0.f, 0.f, 0.f, 0.f});
                } else {
                    bool keep = true;                 // BUILD: the rows a way overwrites are written below
                    if constexpr (BUILD)
                        for (int jj = 0; jj < way; ++jj) keep = keep && (lab[jj] != k);
                    if (keep) {
                        put(k, c, v[u]);
                        if (BUILD && mine_row(k)) *reinterpret_cast<f32x4*>(wout + k * C + c) = v[u];
                    }
                }
            }
        }
    }
    if constexpr (BUILD) {
        if (pk >= 0) {
            put(pk, pc, pv);
            if (mine_row(pk)) *reinterpret_cast<f32x4*>(wout + pk * C + pc) = pv;
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
                    const f32x4 r = {acc4[0] + b4[0], acc4[1] + b4[1], acc4[2] + b4[2], acc4[3] + b4[3]};
                    put((int)mine, c4 * 4, r);
                    if (mine_row((int)mine)) *reinterpret_cast<f32x4*>(wout + (int)mine * C + c4 * 4) = r;
                }
            }
        }
    }
}

#define CLB_LDS(RT, C) ((size_t)3 * 32 * (RT) * ((C) + 8) * 2)
template <int RT, bool BUILD, int CJ>
__global__ __launch_bounds__(256) void classify_fwd_wide4_bf16_kernel(const unsigned short* __restrict__ Fq,
                                                                    const float* __restrict__ Wn, int64_t wn_stride,
                                                                    const int64_t* __restrict__ labels, int way, int K,
                                                                    int P, int base, float* __restrict__ logits,
                                                                    float* __restrict__ bin, WnBuild wb) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    constexpr int C = 64 * CJ, NR = 32 * RT;
    constexpr int KS = C / 16;                     // k-steps of 16 channels
    constexpr int LDD = (C + 8) / 2;               // dwords per plane row
    const int n = blockIdx.y;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int j = lane & 31, hh = lane >> 5;
    const int ntile = (P + CL_TP - 1) / CL_TP;
    const int tile = blockIdx.x * 4 + wave;
    const bool active = tile < ntile;
    const int p0 = (active ? tile : ntile - 1) * CL_TP;
    const int pa = p0 + 2 * j;
    const int pc = max(min(pa, P - 2), 0), sh = pa - pc;
    const unsigned short* fq = Fq + (int64_t)n * C * P;

    STAMP(0);
    // Wn first (L2 hits, back in a round trip), THEN the Fq requests: vmcnt retires in order, so a wait on Wn
    // behind 128 HBM loads would be a wait on all of them.  The wave's B operand: channel 16 q + 8 hh + i,
    // positions pc, pc + 1 (unconditional, clamped), requested in two halves -- a wave keeps ~64 loads in flight,
    // so asking for all C/2 at once stalls the issue for a round trip; the second half goes out before the MFMAs
    // of the first.
    unsigned x[KS][8];
    auto request = [&](int q0, int q1) {
#pragma unroll
        for (int q = q0; q < q1; ++q)
#pragma unroll
            for (int i = 0; i < 8; ++i) {
#ifdef DENET_DIAG_NO_FQ
                x[q][i] = 0u;
#else
                x[q][i] = raw2_load(fq + (unsigned)(16 * q + 8 * hh + i) * (unsigned)P, pc);      // C * P < 2^31
#endif
            }
        __builtin_amdgcn_sched_barrier(0);
    };
    constexpr int KH = KS > 4 ? KS / 2 : KS;
    stage_wn_planes<RT, BUILD, CJ, 256>(sm, Wn, wn_stride, labels, way, K, n, wb);
    STAMP(1);
    __builtin_amdgcn_sched_barrier(0);
    request(0, KH);
    STAMP(2);
    __syncthreads();
    STAMP(3);
    request(KH, KS);
    if (!active) return;                           // (no barrier below this line)
    STAMP(4);

    f32x16 acc[RT][2];
#pragma unroll
    for (int rt = 0; rt < RT; ++rt)
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[rt][t][e] = 0.f;
    const unsigned* pla = reinterpret_cast<const unsigned*>(sm) + j * LDD + 4 * hh;
    // A fragments one k-step ahead of their MFMAs (one wave per SIMD: an LDS read issued right before its MFMA
    // exposes its latency every time)
    u32x4 af[2][3 * RT];
    auto load_a = [&](int q, u32x4 (&a)[3 * RT]) {
#pragma unroll
        for (int s = 0; s < 3; ++s)
#pragma unroll
            for (int rt = 0; rt < RT; ++rt)
                a[s * RT + rt] = *reinterpret_cast<const u32x4*>(pla + (s * NR + 32 * rt) * LDD + 8 * q);
    };
    load_a(0, af[0]);
#pragma unroll
    for (int q = 0; q < KS; ++q) {
        if (q + 1 < KS) load_a(q + 1, af[(q + 1) & 1]);
        u32x4 b0, b1;
#pragma unroll
        for (int m = 0; m < 4; ++m) {
            const unsigned lo = __builtin_amdgcn_perm(x[q][2 * m + 1], x[q][2 * m], 0x05040100u);   // positions pc
            const unsigned hi = __builtin_amdgcn_perm(x[q][2 * m + 1], x[q][2 * m], 0x07060302u);   // positions pc + 1
            b0[m] = sh == 1 ? hi : lo;             // sh == 1: the last position of an odd P sits in the high half
            b1[m] = hi;
        }
        const bf16x8 B0 = __builtin_bit_cast(bf16x8, b0), B1 = __builtin_bit_cast(bf16x8, b1);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int s = 0; s < 3; ++s)
#pragma unroll
            for (int rt = 0; rt < RT; ++rt) {
                const bf16x8 A = __builtin_bit_cast(bf16x8, af[q & 1][s * RT + rt]);
                acc[rt][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A, B0, acc[rt][0], 0, 0, 0);
                acc[rt][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A, B1, acc[rt][1], 0, 0, 0);
            }
        __builtin_amdgcn_sched_barrier(0);
    }
    STAMP(5);
    wave_tile_epilogue<RT, 2>(acc, labels, way, K, P, base, logits, bin, n, pa, hh);
    STAMP(6);
}

// K > 32: the MFMA work of an episode is spread over as many CUs as there are (one block per CU: Wn[n] takes
// 32 RT rows of LDS) and, inside a block, evenly over the four SIMDs.  A block owns a contiguous range of
// 32-POSITION half tiles of one episode, the ranges cut so that N x groups ~ 256 blocks (C3: 12 blocks of 4-5 half
// tiles per episode; C5: 32 of 6-7).  The unit of work is (half tile, 32-row tile of Wn) = 128 MFMAs; the waves
// take the block's units round-robin, so a SIMD carries ceil(units / 4) x 128 MFMAs whatever the range length (a
// wave per half tile put 768 on the SIMD that got two of them).  A = Wn rows from LDS, B = 32 positions of Fq,
// one dword per lane and k-step (lane = position j, k half hh), the first chunk of the next unit requested under
// the last MFMAs of the current one; logits straight from the accumulators; the binary logits need the
// maximum over ALL rows of a position, so every unit leaves its top two (and the rows the ways name) in LDS and
// the block merges them after a barrier.
#ifndef CLW_NW
#define CLW_NW 8
#endif
#define CLW_MAXH 8            // half tiles per block, at most
#define CLW_XM(RT) (CLW_MAXH * (RT) * 64)                   // floats: [half tile][rt][m1, m2][32]
template <typename T, int RT, bool BUILD, int CJ>
__global__ __launch_bounds__(64 * CLW_NW) void classify_fwd_wide_kernel(const T* __restrict__ Fq, const float* __restrict__ Wn,
                                                                      int64_t wn_stride, const int64_t* __restrict__ labels,
                                                                      int way, int K, int P, int base,
                                                                      float* __restrict__ logits, float* __restrict__ bin,
                                                                      WnBuild wb) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    constexpr int C = 64 * CJ, NT = 64 * CLW_NW;
    const int n = blockIdx.y;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int j = lane & 31, hh = lane >> 5;
    const int ldw = C + 1;
    const int nht = (P + 31) >> 5;                 // half tiles of the episode
    const int h0 = (int)((int64_t)blockIdx.x * nht / gridDim.x), h1 = (int)((int64_t)(blockIdx.x + 1) * nht / gridDim.x);
    const int nh = h1 - h0, U = nh * RT;           // the host guarantees nh <= CLW_MAXH
    float* xm = sm + 32 * RT * ldw;                // [nh][RT][2][32]
    float* xf = xm + CLW_XM(RT);                   // [nh][way + 1][32]: the rows the ways name, and row 0
    const T* fqn = Fq + (int64_t)n * C * P;

    // The wave's work is a flat sequence of chunks (16 k-steps = 32 channels of one unit), G = units x C/32 of
    // them.  Four chunk buffers form a ring: the B operand of chunk g+3 is requested before the MFMAs of chunk g,
    // so a load has 48 MFMAs (~1.3 us, twice that with the SIMD's other wave) to arrive -- with one chunk of
    // distance the waves sat in s_waitcnt (38 ns per MFMA against 27.5).  Columns are independent: a clamped
    // column (past P) is computed and dropped; chunk indices past G re-request the last chunk (no branch
    // around a load).
    constexpr int CH = 2 * CLW_NLD, CPU = C / CH;  // channels per chunk, chunks per unit (8, or 2 at C = 64)
    const int nu = wave < U ? (U - wave + CLW_NW - 1) / CLW_NW : 0, G = nu * CPU;
    float b0[CLW_NLD], b1[CLW_NLD], b2[CLW_NLD], b3[CLW_NLD];
    // 32-bit element offsets from the episode's base (C * P < 2^31): one v_add per load instead of a 64-bit
    // multiply-add (quarter rate) and a 64-bit shift-add, which cost more issue time than the MFMAs they fed
    const unsigned P2 = 2u * (unsigned)P;
    auto issue = [&](int g, float (&x)[CLW_NLD]) {
        const int gg = max(min(g, G - 1), 0);
        const int u = min(wave + (gg / CPU) * CLW_NW, U - 1), c = (gg % CPU) * CH;
        unsigned off = (unsigned)(c + hh) * (unsigned)P + (unsigned)min((h0 + u / RT) * 32 + j, P - 1);
#pragma unroll
        for (int q = 0; q < CLW_NLD; ++q) {
#ifdef DENET_DIAG_NO_FQ
            x[q] = 0.f;
#else
            x[q] = FeatIO<T>::load(fqn + off);
#endif
            off += P2;
        }
        __builtin_amdgcn_sched_barrier(0);
    };
    STAMP(0);
    issue(0, b0);                                  // in flight under the Wn staging
    issue(1, b1);
    issue(2, b2);

    // (two passes of 2 RT quads per thread: one pass of 4 RT spills beside the three chunks in flight)
    stage_wn<BUILD, NT, C / 16, (32 * RT * C / 8 + NT - 1) / NT>(sm, Wn, wn_stride, labels, way, K, C, n, wb);
    // rows K .. 32*RT-1 of the LDS image are zeros: the MFMA A operand needs no row mask
    for (int i = K * ldw + tid; i < 32 * RT * ldw; i += NT) sm[i] = 0.f;
    __syncthreads();
    if constexpr (BUILD) {
        float* dst = wb.Wn_out + (int64_t)n * K * C;
        for (int k = blockIdx.x; k < K; k += gridDim.x)
            for (int c = tid; c < C; c += NT) dst[k * C + c] = sm[k * ldw + c];
    }
    STAMP(1);

    float* lg = logits + (int64_t)n * K * P;
    // Four accumulators, one per k-step mod 4 (independent chains).  With the Fq loads stubbed out the loop runs
    // at 72 cycles per MFMA and SIMD (66 is the pipe's rate), with them at 96: what bounds the kernel is the
    // matrix work per SIMD, ceil(units / 4) x 128 MFMAs, plus the time its loads take to issue
    // (scripts/diag/classify_stamps.hip, -DDENET_DIAG_NO_FQ); one, two or four chains measured the same.
    f32x16 accs[4];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int e = 0; e < 16; ++e) accs[a][e] = 0.f;
    auto mfma_chunk = [&](int g, const float (&x)[CLW_NLD]) {
        if (g >= G) return;                        // uniform (C = 64 only: G need not be a multiple of 4)
        const int u = wave + (g / CPU) * CLW_NW, c = (g % CPU) * CH;
        const int arow = (32 * (u % RT) + j) * ldw + c + hh;
        float av[CLW_NLD];
#pragma unroll
        for (int q = 0; q < CLW_NLD; ++q) av[q] = sm[arow + 2 * q];
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int q = 0; q < CLW_NLD; ++q)
            accs[q & 3] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[q], x[q], accs[q & 3], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
    };
    // the unit whose last chunk is g is complete: logits from the accumulators, its share of the binary logits
    auto finish_unit = [&](int g) {
        if (g >= G || (g % CPU) != CPU - 1) return;          // uniform
        const int u = wave + (g / CPU) * CLW_NW;
        const int hl = u / RT, rt = u - hl * RT;
        const int p = (h0 + hl) * 32 + j;
        f32x16 acc;
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            acc[e] = (accs[0][e] + accs[1][e]) + (accs[2][e] + accs[3][e]);
            accs[0][e] = 0.f; accs[1][e] = 0.f; accs[2][e] = 0.f; accs[3][e] = 0.f;
        }
        // ---- logits: 32 lanes = 128 contiguous bytes of a row ----
        if (p < P) {
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int row = 32 * rt + (e & 3) + 8 * (e >> 2) + 4 * hh;
                if (row < K) lg[(int64_t)row * P + p] = acc[e];
            }
        }
        if (bin != nullptr) {
            // this unit's largest and second largest entry per position (ties count: two equal maxima give
            // m2 == m1), merged across the two lane halves
            float m1 = -INFINITY, m2 = -INFINITY;
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int row = 32 * rt + (e & 3) + 8 * (e >> 2) + 4 * hh;
                const float v = row < K ? acc[e] : -INFINITY;
                m2 = __builtin_amdgcn_fmed3f(m1, m2, v);       // second largest of {m1 >= m2, v}
                m1 = fmaxf(m1, v);
            }
            const float o1 = __shfl_xor(m1, 32, 64), o2 = __shfl_xor(m2, 32, 64);
            const float hi = fmaxf(m1, o1), lo = fminf(m1, o1);
            m2 = fmaxf(lo, fmaxf(m2, o2));
            m1 = hi;
            if (hh == 0) {
                xm[((hl * RT + rt) * 2 + 0) * 32 + j] = m1;
                xm[((hl * RT + rt) * 2 + 1) * 32 + j] = m2;
            }
            // the rows the ways name (and row 0 for base == 0), if this unit holds them
            for (int s = 0; s <= way; ++s) {
                int64_t sel = 0;
                if (s < way) sel = labels[(int64_t)n * way + s];
                else if (base) continue;
                if (sel < 32 * rt || sel >= 32 * rt + 32 || sel >= K) continue;           // uniform
                const int r = (int)sel & 31, idx = (r & 3) + 4 * (r >> 3), half = (r >> 2) & 1;
                float v0 = 0.f;
#pragma unroll
                for (int e = 0; e < 16; ++e)
                    if (idx == e) v0 = acc[e];                                            // uniform branch
                const float w0 = __shfl_xor(v0, 32, 64);
                if (hh == 0) xf[(hl * (way + 1) + s) * 32 + j] = half == 0 ? v0 : w0;
            }
        }
    };
#pragma unroll 1
    for (int g = 0; g < G; g += 4) {
        issue(g + 3, b3);
        mfma_chunk(g, b0);
        issue(g + 4, b0);
        mfma_chunk(g + 1, b1);
        if constexpr (CPU == 2) finish_unit(g + 1);
        issue(g + 5, b1);
        mfma_chunk(g + 2, b2);
        issue(g + 6, b2);
        mfma_chunk(g + 3, b3);
        finish_unit(g + 3);
    }
    STAMP(2);
    if (bin == nullptr) return;
    __syncthreads();
    STAMP(3);
    // ---- binary logits (utils/operations.py:327-337): ch1 = the label's row, ch0 = max over the other rows =
    // the second largest entry if the label row holds the maximum (fg == m1), else the largest ----
    for (int i = tid; i < nh * 32; i += NT) {
        const int hl = i >> 5, jj = i & 31;
        const int p = (h0 + hl) * 32 + jj;
        if (p >= P) continue;
        float m1 = xm[(hl * RT * 2 + 0) * 32 + jj], m2 = xm[(hl * RT * 2 + 1) * 32 + jj];
#pragma unroll
        for (int rt = 1; rt < RT; ++rt) {
            const float b1 = xm[((hl * RT + rt) * 2 + 0) * 32 + jj], b2 = xm[((hl * RT + rt) * 2 + 1) * 32 + jj];
            const float hi = fmaxf(m1, b1), lo = fminf(m1, b1);
            m2 = fmaxf(lo, fmaxf(m2, b2));
            m1 = hi;
        }
        for (int jw = 0; jw < way; ++jw) {
            const int64_t lab64 = labels[(int64_t)n * way + jw];
            const bool labok = lab64 >= 0 && lab64 < K;
            const float fg = labok ? xf[(hl * (way + 1) + jw) * 32 + jj] : 0.f;
            float outb;
            if (base) outb = (labok && fg == m1) ? m2 : m1;
            else outb = xf[(hl * (way + 1) + way) * 32 + jj];
            float* dst = bin + ((int64_t)(n * way + jw) * 2) * P + p;
            dst[0] = outb;
            dst[P] = fg;
        }
    }
    STAMP(4);
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

// Forward layout for a shape (also reported by denet_head_step_plan): DENET_CLS_TILE8 = eight waves per
// 64-position tile (K <= 32, few tiles), DENET_CLS_UNITS = layout (a) below, DENET_CLS_WAVE4 = layout (b),
// DENET_CLS_WAVE4_BF16 = layout (b) on the bf16 matrix pipe (bf16 features, 32 < K <= 96).
// `groups` (layout (a): blocks per episode) is returned through the pointer when it is not null.
int denet_classify_fwd_layout(int feat_dtype, int64_t N, int64_t K, int64_t P, int64_t* groups_out) {
    const int RT = (int)((K + 31) / 32);
    if (groups_out) *groups_out = 0;
    // bf16 features: the split-weight bf16 MFMA kernel wherever Wn[n] stays resident (its three planes fit in
    // LDS up to K = 96); its matrix work is a fifth of the fp32 instruction's, so the unit layout has nothing to win
    if (feat_dtype == DENET_BF16 && (RT == 2 || RT == 3)) return DENET_CLS_WAVE4_BF16;
    // K <= 32 with many tiles: layout (b) (a wave per tile, Wn staged once per four tiles) instead of the
    // latency-oriented eight-waves-per-tile kernel
    // (C2: 12.9 vs 18.2 us at 8 episodes = 216 tiles, 23.9 vs 19.9 at 16, 144 vs 72 at 128)
    const bool wide1 = RT == 1 && N * ((P + CL_TP - 1) / CL_TP) >= 384;
    if (RT < 2 && !wide1) return DENET_CLS_TILE8;
    // K > 32: Wn[n] resident in LDS.  Two layouts: (a) units of (32-position half tile, 32-row tile) dealt to
    // 8 waves, N x groups ~ 256 blocks -- a SIMD carries ceil(units / 4) x 128 MFMAs; (b) a wave per 64-position
    // tile and four tiles per block -- 768 MFMAs per SIMD at RT = 3, but every Fq element loaded once, 8 bytes
    // per lane (layout (a) reads it RT times, 4 bytes per lane).  (a) where it takes the matrix work per SIMD
    // to two thirds or less (C3: 512 against 768, F3 46 -> 39 us), (b) otherwise (C5: 768 either way, and
    // (a) measured 51 us against 48).
    const int64_t nht = (P + 31) / 32;
    int64_t groups = 256 / N;
    if (groups > nht) groups = nht;
    if (groups < (nht + CLW_MAXH - 1) / CLW_MAXH) groups = (nht + CLW_MAXH - 1) / CLW_MAXH;
    if (groups_out) *groups_out = groups;
    const int64_t units = ((nht + groups - 1) / groups) * RT;
    const bool by_units = !wide1 && ((units + 3) / 4) * 128 * 3 <= (int64_t)RT * 256 * 2;
    return by_units ? DENET_CLS_UNITS : DENET_CLS_WAVE4;
}

template <typename T, bool BUILD = false>
static int launch_classify_fwd(const T* Fq, const float* Wn, int64_t wn_stride, const int64_t* labels, int64_t N,
                               int64_t way, int64_t K, int64_t C, int64_t P, int base, float* logits, float* bin,
                               hipStream_t st, WnBuild wb = WnBuild{nullptr, nullptr, nullptr, nullptr, nullptr}) {
    const int RT = (int)((K + 31) / 32);
    int64_t groups = 0;
    const int layout =
        denet_classify_fwd_layout(std::is_same<T, unsigned short>::value ? DENET_BF16 : DENET_F32, N, K, P, &groups);
    if (layout == DENET_CLS_WAVE4_BF16) {
        if constexpr (std::is_same<T, unsigned short>::value) {
            const size_t ldsb = CLB_LDS(RT, C);
            if (ldsb > 160 * 1024) return DENET_E_UNSUPPORTED;
            const int ntile = (int)((P + CL_TP - 1) / CL_TP);
            dim3 gridb((unsigned)((ntile + 3) / 4), (unsigned)N), blockb(256);
#define CL_WIDE4B(R, J)                                                                                       \
    {                                                                                                         \
        int rc = set_lds(classify_fwd_wide4_bf16_kernel<R, BUILD, J>, ldsb);                                  \
        if (rc) return rc;                                                                                    \
        hipLaunchKernelGGL((classify_fwd_wide4_bf16_kernel<R, BUILD, J>), gridb, blockb, ldsb, st, Fq, Wn, wn_stride, \
                           labels, (int)way, (int)K, (int)P, base, logits, bin, wb);                          \
    }
            if (RT == 2) { if (C == 64) CL_WIDE4B(2, 1) else CL_WIDE4B(2, 4) }
            else { if (C == 64) CL_WIDE4B(3, 1) else CL_WIDE4B(3, 4) }
#undef CL_WIDE4B
            DENET_CHECK_LAUNCH();
            return DENET_OK;
        } else {
            return DENET_E_UNSUPPORTED;
        }
    }
    if (layout != DENET_CLS_TILE8) {
        const bool by_units = layout == DENET_CLS_UNITS;
        if (!by_units) {
            const size_t lds4 = (size_t)32 * RT * (C + 1) * 4;          // rows K .. 32*RT-1 are zero padding
            if (lds4 > 160 * 1024 || RT > 4) return DENET_E_UNSUPPORTED;
            const int ntile = (int)((P + CL_TP - 1) / CL_TP);
            dim3 grid4((unsigned)((ntile + 3) / 4), (unsigned)N), block4(256);
#define CL_WIDE4(R, J)                                                                                        \
    {                                                                                                         \
        int rc = set_lds(classify_fwd_wide4_kernel<T, R, BUILD, J>, lds4);                                    \
        if (rc) return rc;                                                                                    \
        hipLaunchKernelGGL((classify_fwd_wide4_kernel<T, R, BUILD, J>), grid4, block4, lds4, st, Fq, Wn, wn_stride, \
                           labels, (int)way, (int)K, (int)P, base, logits, bin, wb);                          \
    }
            switch (RT) {
                case 1: if (C == 64) CL_WIDE4(1, 1) else CL_WIDE4(1, 4) break;
                case 2: if (C == 64) CL_WIDE4(2, 1) else CL_WIDE4(2, 4) break;
                case 3: if (C == 64) CL_WIDE4(3, 1) else CL_WIDE4(3, 4) break;
                default: if (C == 64) CL_WIDE4(4, 1) else CL_WIDE4(4, 4) break;
            }
#undef CL_WIDE4
            DENET_CHECK_LAUNCH();
            return DENET_OK;
        }
        // LDS: Wn[n] (rows K .. 32*RT-1 zero padding) + the per-unit top-two and label rows for the binary logits
        const size_t ldsw = ((size_t)32 * RT * (C + 1) + CLW_XM(RT) + (size_t)CLW_MAXH * (way + 1) * 32) * 4;
        if (ldsw > 160 * 1024 || RT > 4) return DENET_E_UNSUPPORTED;
        dim3 gridw((unsigned)groups, (unsigned)N), blockw(64 * CLW_NW);
#define CL_WIDE(R, J)                                                                                         \
    {                                                                                                         \
        int rc = set_lds(classify_fwd_wide_kernel<T, R, BUILD, J>, ldsw);                                     \
        if (rc) return rc;                                                                                    \
        hipLaunchKernelGGL((classify_fwd_wide_kernel<T, R, BUILD, J>), gridw, blockw, ldsw, st, Fq, Wn, wn_stride, \
                           labels, (int)way, (int)K, (int)P, base, logits, bin, wb);                          \
    }
        switch (RT) {
            case 2: if (C == 64) CL_WIDE(2, 1) else CL_WIDE(2, 4) break;
            case 3: if (C == 64) CL_WIDE(3, 1) else CL_WIDE(3, 4) break;
            default: if (C == 64) CL_WIDE(4, 1) else CL_WIDE(4, 4) break;
        }
#undef CL_WIDE
        DENET_CHECK_LAUNCH();
        return DENET_OK;
    }
    // K <= 32: eight waves per block (32 MFMA k-steps per wave instead of 64; there is one block per CU
    // at batch 8, so the extra waves are free); larger K keeps four (the partial tiles must fit in LDS)
    const int NW = RT == 1 ? 8 : 4;
    const size_t a = (size_t)K * (C + 1), b = (size_t)NW * K * CL_LD;
    const size_t lds = (a > b ? a : b) * 4;
    if (lds > 160 * 1024) return DENET_E_UNSUPPORTED;
    dim3 grid((unsigned)((P + CL_TP - 1) / CL_TP), (unsigned)N), block((unsigned)(64 * NW));
#define CL_FWD3(R, L, W)                                                                                    \
    {                                                                                                       \
        int rc = set_lds(classify_fwd_kernel<T, R, L, BUILD, W>, lds);                                      \
        if (rc) return rc;                                                                                  \
        hipLaunchKernelGGL((classify_fwd_kernel<T, R, L, BUILD, W>), grid, block, lds, st, Fq, Wn, wn_stride, \
                           labels, (int)way, (int)K, (int)C, (int)P, base, logits, bin, wb);                \
    }
    if (C == 64) CL_FWD3(1, 4, 8) else CL_FWD3(1, 16, 8)
#undef CL_FWD3
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

// Fused step: classifier that assembles Wn[n] itself (see WnBuild) and writes it to Wn_out.
int denet_classify_build_fwd(const void* Fq, int feat_dtype, const float* weight, const float* weight_mask,
                             const float* proto_part, const float* bl, const int64_t* labels, int64_t N, int64_t way,
                             int64_t K, int64_t C, int64_t P, float* Wn_out, float* logits, float* logits_binary,
                             const DropGen* wdrop, void* stream) {
    if (!Fq || !weight || !proto_part || !bl || !labels || !Wn_out || !logits || !logits_binary) return DENET_E_BADARG;
    if (!pos(N) || !pos(K) || !pos(C) || !pos(P) || !pos(way)) return DENET_E_BADARG;
    if (!c_ok(C) || K > 128 || N > 65535) return DENET_E_UNSUPPORTED;
    hipStream_t st = (hipStream_t)stream;
    WnBuild wb{weight, weight_mask, proto_part, bl, Wn_out, wdrop ? *wdrop : DropGen{0u, 0u, 0u, 1.f, 0}};
    if (feat_dtype == DENET_F32)
        return launch_classify_fwd<float, true>((const float*)Fq, nullptr, 0, labels, N, way, K, C, P, 1, logits,
                                                logits_binary, st, wb);
    if (feat_dtype == DENET_BF16)
        return launch_classify_fwd<unsigned short, true>((const unsigned short*)Fq, nullptr, 0, labels, N, way, K, C,
                                                         P, 1, logits, logits_binary, st, wb);
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
    const size_t lds = CLS_BWD_LDS(K);
    dim3 grid((unsigned)n_part, (unsigned)N), block(256);
    ClsBwdArgs g{Fq, Wn, da, db, scale, gmul, nullptr, 0, (int)N, (int)K, (int)C, (int)P, n_part, dFq, dWn_part,
                 nullptr};
#define CL_BWD(R)                                                              \
    {                                                                          \
        int rc = set_lds(classify_bwd_kernel<T, R>, lds);                      \
        if (rc) return rc;                                                     \
        hipLaunchKernelGGL((classify_bwd_kernel<T, R>), grid, block, lds, st, g); \
    }
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
