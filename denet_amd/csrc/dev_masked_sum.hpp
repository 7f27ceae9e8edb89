// Device bodies of K0/K1/K1b, shared by the stand-alone kernels (masked_sum.hip) and the fused
// step kernels (head_step.hip).
#pragma once
#include "denet_common.hpp"

// ---------------------------------------------------------------------------
// K1: S[b,c] = sum_p mask[b,p] * Fs[b,c,p]
// ---------------------------------------------------------------------------
template <typename T> struct Vec16;
template <> struct Vec16<float> {
    static constexpr int N = 4, LOG = 2;
    typedef f32x4 type;
    static __device__ __forceinline__ float elem(const type& v, int e) { return v[e]; }
};
template <> struct Vec16<unsigned short> {
    static constexpr int N = 8, LOG = 3;
    typedef u16x8 type;
    static __device__ __forceinline__ float elem(const type& v, int e) { return bf16_to_f32(v[e]); }
};

// LDS image of a mask row for the two streaming kernels: V planes (V = elements per 16-byte vector), plane q
// holds mask[q], mask[q + V], ...  Lane i of a wave reads element e of ITS vector, mask[head + V i + e], at
// plane[(head + e) % V][(head + e) / V + i]: consecutive lanes, consecutive banks, one address register per
// plane and the vector index as an immediate offset.  (With the row stored as [P] floats the lanes sit 4 V
// bytes apart: 8-way bank conflicts at V = 4, 16-way at V = 8 -- the bf16 pooling stream ran LDS-bound at
// 3.2 TB/s against 6.0 for fp32 over the same bytes, scripts/diag/pool_stamps.)  The image is followed by a
// tail of MASK_TAIL floats: vectors past the end of a row are read unclamped and discarded by a select.
#define MASK_TAIL 1024
// bf16 always uses the planar image.  fp32 (8-way conflicts, four reads per vector) uses it where it measured
// faster -- the short-row pooling kernel (C2 F1 18.8 -> 17.9 us, C3 85.1 -> 83.9) and the flat dFs writer (C3 B3
// 84.9 -> 83.2 us) -- and keeps the row as [P] floats in the long-row pooling kernel (the planar addressing costs
// it 47 registers: 6.0 -> 5.2 TB/s at 81 x 81) and in the row-per-wave dFs writer (C2 B3 16.6 vs 16.95 us).
// FAST = !LONG for the pooling kernel, NT for the dFs writer.
#ifndef DENET_POOL_PLANAR_F32
#define DENET_POOL_PLANAR_F32 0
#endif
template <typename T, bool FAST> struct MaskImage;
template <bool FAST> struct MaskImage<float, FAST> { static constexpr bool PLANAR = FAST || DENET_POOL_PLANAR_F32; };
template <bool FAST> struct MaskImage<unsigned short, FAST> { static constexpr bool PLANAR = true; };
template <int V> __host__ __device__ __forceinline__ int mask_plane_stride(int P) { return ((P + V - 1) / V + 2) | 1; }
template <int V, bool PLANAR> __host__ __device__ __forceinline__ int mask_image_floats(int P) {
    return (PLANAR ? V * mask_plane_stride<V>(P) : ((P + 3) & ~3)) + MASK_TAIL;
}

// Body of K1 for block (cblk, b): `sm` [P] floats of LDS.  With Ys != nullptr the mask is gathered
// here from the int64 label map (K0 fused: nearest resize, model/common.py:189-190) and the block
// with cblk == 0 also writes mask[b,:] and cnt[b] for the backward pass; otherwise it is read from `mask` (and
// with cnt != nullptr the block with cblk == 0 still writes cnt[b]: mask_resize_flat_kernel leaves it to us).
// LONG: rows of more than 512 vectors (P > 2048 fp32 / 4096 bf16 elements) -- a separate instantiation, so that
// the register sets of the two streaming loops do not add up in one kernel.
// NTL: the feature stream is read with non-temporal loads.  Fs is read once per step; once the step's working set
// is larger than the 256 MB last-level cache, letting the stream through the cache evicts what IS re-read (Fq by
// the classifier forward and backward, the logits and their gradient slabs) and the stream itself runs slower:
// C3 step 354.8 -> 337.3 us (F1 84 -> 74), C5 420.5 -> 397.9 us (F1 74 -> 62); at C2 (8 episodes, everything
// stays resident) 87.7 -> 91.0 us the other way, so the launcher picks by size like it does for the dFs stores.
template <typename T, bool LONG = false, bool NTL = false>
__device__ __forceinline__ void masked_sum_fwd_body(const T* __restrict__ Fs, const int64_t* __restrict__ Ys,
                                                    float* __restrict__ mask, float* __restrict__ cnt,
                                                    float* __restrict__ S, int C, int P, int rows_per_block, int H,
                                                    int W, int h, int w, int b, int cblk, float* sm,
                                                    const float* __restrict__ W1 = nullptr,
                                                    float* __restrict__ pre1_part = nullptr) {
    const int c0 = cblk * rows_per_block;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float* mrow = mask + (int64_t)b * P;
    // Sliced chain, first mat-vec (gate layer 1, model/common.py:226) as a split-K partial: this block
    // owns att[c0..c0+16) of image b, so it can form  pre1_part[b][cblk][o] = sum_r W1[o][c0+r] * att[c0+r]
    // for every output row o.  rows_per_block is 16 here.
    constexpr int V = Vec16<T>::N, LV = Vec16<T>::LOG;
    constexpr bool PLANAR = MaskImage<T, !LONG>::PLANAR;
    const int PS = mask_plane_stride<V>(P), MP = mask_image_floats<V, PLANAR>(P);
    auto midx = [&](int p) { return PLANAR ? (p & (V - 1)) * PS + (p >> LV) : p; };
    float* sl = sm + MP + 4;                               // [16] this block's att = S / (h*w)
    const int wq = threadIdx.x & 3, wog = threadIdx.x >> 2;
    // W1 slice: four lanes share a row (16 B each, 64 B per row), four loads cover rows og, og+64,
    // og+128, og+192.  It is requested behind the LAST round of feature loads (with the first round it
    // delays the stream: 21.0 vs 20.3 us at C2 batch 8) and waits in LDS, each thread parking and
    // re-reading its own 64 bytes.
    float* wsl = sm + MP + 4 + 16 + (int)threadIdx.x * 16;
    auto fetch_w1 = [&]() {
        if (W1 == nullptr) return;
#pragma unroll
        for (int m = 0; m < 4; ++m)
            *reinterpret_cast<f32x4*>(wsl + 4 * m) =
                *reinterpret_cast<const f32x4*>(W1 + (int64_t)min(wog + 64 * m, C - 1) * C + c0 + 4 * wq);
    };
    typedef typename Vec16<T>::type vec_t;
    auto fs_load = [](const vec_t* p) {
        if constexpr (NTL) return __builtin_nontemporal_load(p);
        else return *p;
    };
    constexpr int NV = 8;                                 // vector loads per lane per row on the fast path
    constexpr int NR = 2;                                 // rows of a wave in flight together (more costs occupancy)
    constexpr bool short_rows = !LONG;                    // the launcher picks: P / V <= NV * 64

    // ---- fast path, part 1: EVERY feature load of the wave's first NR rows is issued before anything
    // else -- they do not depend on the mask -- all unconditional (clamped index, value selected later:
    // a load inside a branch makes the compiler drain vmcnt at the join), so the block pays ONE memory
    // round trip for the mask gather and the feature rows together.
    vec_t v[NR][NV];
    float hv[NR], tv[NR];
    int heads[NR], nvecs[NR];
    auto issue = [&](int r0) {
#pragma unroll
        for (int q = 0; q < NR; ++q) {
            const int r = r0 + 4 * q, c = c0 + r;
            const bool live = (r < rows_per_block) && (c < C);
            const T* row = Fs + ((int64_t)b * C + (live ? c : c0)) * P;
            const int mis = (int)(((uintptr_t)row / sizeof(T)) % V);
            int head = mis ? V - mis : 0;
            if (head > P) head = P;
            const int nvec = (P - head) / V;
            heads[q] = head; nvecs[q] = nvec;
            const vec_t* rv = reinterpret_cast<const vec_t*>(row + head);
            hv[q] = FeatIO<T>::load(row + min(lane, P - 1));
            tv[q] = FeatIO<T>::load(row + min(head + nvec * V + lane, P - 1));
#pragma unroll
            for (int u = 0; u < NV; ++u) v[q][u] = fs_load(rv + min(lane + 64 * u, nvec > 0 ? nvec - 1 : 0));
        }
    };
    // mask[head + V i + e] for vector i = lane + (immediate): one LDS address per plane (row-major: one in all)
    auto plane_bases = [&](int head, const float* (&pb)[V]) {
#pragma unroll
        for (int e = 0; e < V; ++e) pb[e] = PLANAR ? sm + midx(head + e) + lane : sm + head + lane * V + e;
    };
    auto dotp = [&](const vec_t& x, const float* const (&pb)[V], int off) {
        float a = 0.f;
#pragma unroll
        for (int e = 0; e < V; ++e) a += Vec16<T>::elem(x, e) * pb[e][PLANAR ? off : off * V];
        return a;
    };
    auto consume = [&](int r0) {
        float accs[NR];
#pragma unroll
        for (int q = 0; q < NR; ++q) {
            const int head = heads[q], nvec = nvecs[q], t0 = head + nvec * V;
            float acc = (lane < head) ? hv[q] * sm[midx(min(lane, P - 1))] : 0.f;
            const float* pb[V];
            plane_bases(head, pb);
#pragma unroll
            for (int u = 0; u < NV; ++u) {
                const float d = dotp(v[q][u], pb, 64 * u);          // (past the row's end: tail words, discarded)
                acc += (lane + 64 * u < nvec) ? d : 0.f;
            }
            acc += (t0 + lane < P) ? tv[q] * sm[midx(min(t0 + lane, P - 1))] : 0.f;
            accs[q] = wave_sum(acc);
        }
#pragma unroll
        for (int q = 0; q < NR; ++q) {
            const int r = r0 + 4 * q;
            if (r < rows_per_block && c0 + r < C && lane == 0) {
                S[(int64_t)b * C + c0 + r] = accs[q];
                if (W1 != nullptr) sl[r] = accs[q] / (float)P;        // att = S / (h*w)
            }
        }
    };
    auto chain_tail = [&]() {
        if (W1 == nullptr) return;
        __syncthreads();
        const f32x4 at = *reinterpret_cast<const f32x4*>(sl + 4 * wq);
        float mine = 0.f;
#pragma unroll
        for (int m = 0; m < 4; ++m) {
            const f32x4 wv = *reinterpret_cast<const f32x4*>(wsl + 4 * m);
            float a = wv[0] * at[0];
            a = fmaf(wv[1], at[1], a);
            a = fmaf(wv[2], at[2], a);
            a = fmaf(wv[3], at[3], a);
            a += dpp_f32<0xB1>(a);                          // the quad's four column groups
            a += dpp_f32<0x4E>(a);
            mine = (wq == m) ? a : mine;                    // lane q keeps row og + 64 q
        }
        const int o = wog + 64 * wq;
        if (o < C) {
            const int ncb = C / rows_per_block;
            pre1_part[((int64_t)b * ncb + cblk) * C + o] = mine;
        }
    };
    if constexpr (short_rows) issue(wave);
    // ---- the mask: gathered here from the int64 label map (K0 fused: nearest resize,
    // model/common.py:189-190; the block with cblk == 0 also writes mask[b,:] and cnt[b] for the backward
    // pass), or read from `mask`.
    if (Ys != nullptr) {
        const float sy = nearest_scale(H, h), sx = nearest_scale(W, w);
        const int64_t* src = Ys + (int64_t)b * H * W;
        float acc = 0.f;
#pragma unroll 8
        for (int p = threadIdx.x; p < P; p += 256) {
            const int y = p / w, x = p - y * w;
            const float m = (float)src[(int64_t)nearest_src(y, sy, H) * W + nearest_src(x, sx, W)];
            sm[midx(p)] = m;
            acc += m;
            if (cblk == 0) mrow[p] = m;
        }
        if (cblk == 0) {
            // every partial sum is a small integer: exact in any order
            acc = wave_sum(acc);
            float* red = sm + MP;
            if (lane == 0) red[wave] = acc;
            __syncthreads();
            if (threadIdx.x == 0) cnt[b] = (red[0] + red[1]) + (red[2] + red[3]);
        }
    } else {
        float acc = 0.f;
#pragma unroll 8
        for (int p = threadIdx.x; p < P; p += 256) {
            const float m = mrow[p];
            sm[midx(p)] = m;
            acc += m;
        }
        if (cnt != nullptr && cblk == 0) {                 // (mask resized by a launch that leaves the count to us)
            acc = wave_sum(acc);
            float* red = sm + MP;
            if (lane == 0) red[wave] = acc;
            __syncthreads();
            if (threadIdx.x == 0) cnt[b] = (red[0] + red[1]) + (red[2] + red[3]);
        }
    }
    __syncthreads();

    if constexpr (short_rows) {
        consume(wave);
        bool fetched = false;
        for (int r = wave + 4 * NR; r < rows_per_block; r += 4 * NR) {
            issue(r);
            if (r + 4 * NR >= rows_per_block) { fetch_w1(); fetched = true; }
            consume(r);
        }
        if (!fetched) fetch_w1();
        chain_tail();
        return;
    } else {
    // ---- long rows: stream each row with a rolling loop
    fetch_w1();
    for (int r = wave; r < rows_per_block; r += 4) {
        const int c = c0 + r;
        if (c >= C) break;
        const T* row = Fs + ((int64_t)b * C + c) * P;
        const int mis = (int)(((uintptr_t)row / sizeof(T)) % V);
        int head = mis ? V - mis : 0;
        if (head > P) head = P;
        const float hv1 = FeatIO<T>::load(row + min(lane, P - 1));
        float acc = (lane < head) ? hv1 * sm[midx(min(lane, P - 1))] : 0.f;
        const int nvec = (P - head) / V;
        const vec_t* rv = reinterpret_cast<const vec_t*>(row + head);
        const int t0 = head + nvec * V;
        const float tv1 = FeatIO<T>::load(row + min(t0 + lane, P - 1));
        // eight clamped, unconditional vector loads per lane and trip (8 KB per wave), the NEXT trip's loads
        // requested before the current trip is consumed (two register sets): a wave that waits for every trip
        // before it asks for the next one streams at one trip per memory round trip
        vec_t va[NV], vb[NV];
        auto fetch = [&](int i0, vec_t (&vv)[NV]) {
#pragma unroll
            for (int u = 0; u < NV; ++u) vv[u] = fs_load(rv + max(min(i0 + 64 * u, nvec - 1), 0));
        };
        const float* pb[V];
        plane_bases(head, pb);
        // (vector i up to nvec + 959 is read and discarded by the select: one float per plane in a planar image,
        // i.e. inside its MASK_TAIL; V floats per vector in the row-major fp32 image, where those reads fall into
        // the parked W1 slices or the padding launch_pool allocates behind the image)
        auto eat = [&](int i0, int off, const vec_t (&vv)[NV]) {
#pragma unroll
            for (int u = 0; u < NV; ++u) {
                const float d = dotp(vv[u], pb, off + 64 * u);
                acc += (i0 + off + 64 * u < nvec) ? d : 0.f;
            }
        };
        fetch(lane, va);
        for (int i0 = lane; i0 < nvec; i0 += 2 * 64 * NV) {
            fetch(i0 + 64 * NV, vb);
            eat(i0, 0, va);
            fetch(i0 + 2 * 64 * NV, va);
            eat(i0, 64 * NV, vb);
#pragma unroll
            for (int e = 0; e < V; ++e) pb[e] += (PLANAR ? 1 : V) * 2 * 64 * NV;
        }
        acc += (t0 + lane < P) ? tv1 * sm[midx(min(t0 + lane, P - 1))] : 0.f;
        acc = wave_sum(acc);
        if (lane == 0) {
            S[(int64_t)b * C + c] = acc;
            if (W1 != nullptr) sl[r] = acc / (float)P;
        }
    }
    chain_tail();
    }
}

// (short rows, fp32: 134 VGPRs, 3 waves per SIMD; forcing 4 with amdgpu_waves_per_eu spills 7 registers and the
// stream runs 10 % slower at every batch size: more resident blocks do not buy bandwidth)
template <typename T, bool LONG, bool NTL>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(3))) void masked_sum_fwd_kernel(const T* __restrict__ Fs,
                                                            const int64_t* __restrict__ Ys,
                                                            float* __restrict__ mask, float* __restrict__ cnt,
                                                            float* __restrict__ S, int C, int P,
                                                            int rows_per_block, int H, int W, int h, int w,
                                                            const float* __restrict__ W1,
                                                            float* __restrict__ pre1_part) {
    extern __shared__ __attribute__((aligned(16))) float sm[];   // [P] (+ 4 + 16)
    // 1-D grid of ncb * B blocks; the ncb channel blocks of an image are placed on one XCD (shared gather)
    const int ncb = (C + rows_per_block - 1) / rows_per_block;
    const int v = xcd_swizzle(blockIdx.x, gridDim.x);
    BSTAMP(0);
    masked_sum_fwd_body<T, LONG, NTL>(Fs, Ys, mask, cnt, S, C, P, rows_per_block, H, W, h, w, v / ncb, v % ncb, sm, W1,
                                      pre1_part);
    BSTAMP(1);
}

// ---------------------------------------------------------------------------
// K1b: dFs[b,c,p] = mask[b,p] * coef[b,c]
// ---------------------------------------------------------------------------
// q = e / d, rem = e - q*d for 0 <= e < 2^22, 0 < d <= 2^14, inv = 1/d (float estimate + one-step correction)
__device__ __forceinline__ int wc_divmod_p(int e, int d, float inv, int& rem) {
    int q = (int)(((float)e + 0.5f) * inv);
    int r = e - q * d;
    if (r < 0) { --q; r += d; }
    if (r >= d) { ++q; r -= d; }
    rem = r;
    return q;
}
// Sliced chain, last step (fused step only): the block forms coef[b][c0..c0+16) itself from the
// split-K partials of the previous launch,
//   dpre1[b][j] = (hidden[b][j] > 0) * sum_sl dh_part[b][sl][j]             (relu backward)
//   datt[b][c]  = sum_j W1[j][c] * dpre1[b][j]                                (W1^T mat-vec, column slice)
//   coef[b][c]  = dpool[grp][c] * gate[b][c] / (cnt[b] + eps) + datt[b][c] / P
// thread j holds the 64-byte slice W1[j][c0..c0+16); the 16 sums over j are wave reductions
// combined through LDS in a fixed order.  Needs C <= 256 and rows_per_block == 16.
struct CoefArgs {
    const float *dh_part, *hidden, *W1, *dpool, *gate, *cnt;
    int shot, nsl;
    float Pf, eps;
    float* coef_out;      // not null: the block only writes its 16 coefficients to coef_out [B][C] (the stream is
                          // written by dfs_chunk_kernel afterwards)
};

// NT: non-temporal stores.  A cached dFs stream leaves the last-level cache full of dirty lines whose write-back
// the NEXT streaming kernels pay for (the pooling kernel of the next step: 117 us after a cached 344 MB dFs
// stream, 89 us otherwise; scripts/diag/pool_after_dfs.py); sent around the cache the stream costs this kernel
// ~10-25 us more and the next one more than that less.  Only when the step's whole working set fits the 256 MB
// cache (C2 at 8 episodes: 69 MB of dFs) are cached stores faster for both; the launcher picks
// (head_step.hip: 80 MB).
// bf16 keeps the row-per-wave writer with non-temporal stores (the flat writer's per-element plane arithmetic
// makes it VALU-bound there: C5 B3 97 us flat against 86 us by rows)
#ifndef DENET_NT_BF16_ROWS
#define DENET_NT_BF16_ROWS 1
#endif
template <typename T, int NSLT, bool NT = false>      // NSLT = C/16 for the sliced chain, 0: coef is given
__device__ __forceinline__ void masked_sum_bwd_body(const float* __restrict__ mask, const float* __restrict__ coef,
                                                    T* __restrict__ dFs, int C, int P, int rows_per_block, int b,
                                                    int cblk, float* sm, const CoefArgs& ca) {
    const int c0 = cblk * rows_per_block;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const float* mrow = mask + (int64_t)b * P;
    // the mask row in LDS: same image as the pooling kernel's (planar for bf16, MaskImage)
    constexpr int V = Vec16<T>::N, LV = Vec16<T>::LOG;
    constexpr bool PLANAR = MaskImage<T, NT>::PLANAR;
    const int PS = mask_plane_stride<V>(P), MP = mask_image_floats<V, PLANAR>(P);
    auto midx = [&](int p) { return PLANAR ? (p & (V - 1)) * PS + (p >> LV) : p; };
    float* red = sm + MP;                    // [4][16]
    float* coefs = red + 64;                 // [16]
    constexpr bool SLICED = NSLT > 0;
    // every global load of the block is requested before the first wait: the coef operands and the
    // mask row (8 clamped loads per thread cover P <= 2048; longer rows finish in a loop below)
    f32x4 w1v[4];
    float dhv[SLICED ? NSLT : 1];
    float hid = 0.f, cg = 0.f, cd = 0.f, cc = 1.f;
    if constexpr (SLICED) {
        const int j = min((int)threadIdx.x, C - 1);
        const float* wp = ca.W1 + (int64_t)j * C + c0;
#pragma unroll
        for (int m = 0; m < 4; ++m) w1v[m] = *reinterpret_cast<const f32x4*>(wp + 4 * m);
        hid = ca.hidden[(int64_t)b * C + j];
        const float* dp = ca.dh_part + (int64_t)b * NSLT * C + j;
#pragma unroll
        for (int q = 0; q < NSLT; ++q) dhv[q] = dp[(int64_t)q * C];
        // factors of the first term, for the 16 finishing threads
        const int t16 = threadIdx.x & 15;
        cg = ca.gate[(int64_t)b * C + c0 + t16];
        cd = ca.dpool[(int64_t)(b / ca.shot) * C + c0 + t16];
        cc = ca.cnt[b] + ca.eps;
    }
    if constexpr (SLICED) {
        if (ca.coef_out != nullptr) {                 // block-uniform: coefficients only
            float dh = 0.f;
#pragma unroll
            for (int q = 0; q < NSLT; ++q) dh += dhv[q];
            const float dpre = ((int)threadIdx.x < C && hid > 0.f) ? dh : 0.f;
#pragma unroll
            for (int t = 0; t < 16; ++t) {
                const float v = wave_sum(w1v[t >> 2][t & 3] * dpre);
                if (lane == t) red[wave * 16 + t] = v;
            }
            __syncthreads();
            if (threadIdx.x < 16 && c0 + (int)threadIdx.x < C) {
                const int t = threadIdx.x;
                const float datt = ((red[t] + red[16 + t]) + red[32 + t]) + red[48 + t];
                ca.coef_out[(int64_t)b * C + c0 + t] = cd * (cg / cc) + datt / ca.Pf;
            }
            return;
        }
    }
    float mv[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) mv[u] = mrow[min((int)threadIdx.x + 256 * u, P - 1)];
    if constexpr (SLICED) {
        float dh = 0.f;
#pragma unroll
        for (int q = 0; q < NSLT; ++q) dh += dhv[q];
        const float dpre = ((int)threadIdx.x < C && hid > 0.f) ? dh : 0.f;
#pragma unroll
        for (int t = 0; t < 16; ++t) {
            const float v = wave_sum(w1v[t >> 2][t & 3] * dpre);
            if (lane == t) red[wave * 16 + t] = v;
        }
    }
#pragma unroll
    for (int u = 0; u < 8; ++u)
        if ((int)threadIdx.x + 256 * u < P) sm[midx(threadIdx.x + 256 * u)] = mv[u];
    for (int p = threadIdx.x + 2048; p < P; p += 256) sm[midx(p)] = mrow[p];
    if constexpr (SLICED) {
        __syncthreads();
        if (threadIdx.x < 16) {
            const int t = threadIdx.x;
            const float datt = ((red[t] + red[16 + t]) + red[32 + t]) + red[48 + t];
            coefs[t] = cd * (cg / cc) + datt / ca.Pf;
        }
    } else {
        if ((int)threadIdx.x < rows_per_block) coefs[threadIdx.x] = coef[(int64_t)b * C + min(c0 + (int)threadIdx.x, C - 1)];
    }
    __syncthreads();

    typedef typename Vec16<T>::type vec_t;
    constexpr bool ROWS = !NT || (V == 8 && DENET_NT_BF16_ROWS);
    if constexpr (ROWS) {
        // streams the last-level cache absorbs (C2, C5): one wave per channel row, four rows of a block in flight
        // (measured against the flat writer below: 17.1 vs 17.7 us at C2, 69.4 vs 71.4 us at C5)
        for (int r = wave; r < rows_per_block; r += 4) {
            const int c = c0 + r;
            if (c >= C) break;
            const float k = coefs[r];
            T* row = dFs + ((int64_t)b * C + c) * P;
            const int mis = (int)(((uintptr_t)row / sizeof(T)) % V);
            int head = mis ? V - mis : 0;
            if (head > P) head = P;
            if (lane < head) FeatIO<T>::store(row + lane, k * sm[midx(lane)]);
            const int nvec = (P - head) / V;
            vec_t* rv = reinterpret_cast<vec_t*>(row + head);
            // mask[head + V i + e]: one LDS address per plane, the trip as an immediate offset (row-major: one in all)
            const float* pb[V];
#pragma unroll
            for (int e = 0; e < V; ++e) pb[e] = PLANAR ? sm + midx(head + e) : sm + head + e;
#pragma unroll 4
            for (int i = lane; i < nvec; i += 64) {
                vec_t v;
#pragma unroll
                for (int e = 0; e < V; ++e) {
                    const float x = k * pb[e][PLANAR ? i : i * V];
                    if constexpr (V == 4) v[e] = x;
                    else v[e] = f32_to_bf16(x);
                }
                if constexpr (NT) __builtin_nontemporal_store(v, rv + i);
                else rv[i] = v;
            }
            const int t0 = head + nvec * V;
            if (t0 + lane < P) FeatIO<T>::store(row + t0 + lane, k * sm[midx(t0 + lane)]);
        }
    } else {
        // The block's 16 channel rows are ONE contiguous span of rows * P elements (NCHW): all 256 threads write it
        // front to back, 4 KB per trip, instead of one wave per row (four 1 KB streams 6.7 KB apart per block and
        // ~7000 of them in flight on the chip, every one opening a DRAM page of its own).  Element e of the span is
        // coefs[e / P] * mask[e % P]; a 16-byte vector may straddle two rows.
        const int rows = min(rows_per_block, C - c0);
        const int total = rows * P;
        T* out = dFs + ((int64_t)b * C + c0) * P;
        const float invP = 1.f / (float)P;
        auto value = [&](int e) {
            int p;
            const int r = wc_divmod_p(e, P, invP, p);
            return coefs[r] * sm[midx(p)];
        };
        const int mis = (int)(((uintptr_t)out / sizeof(T)) % V);
        int head = mis ? V - mis : 0;
        if (head > total) head = total;
        if ((int)threadIdx.x < head) FeatIO<T>::store(out + threadIdx.x, value(threadIdx.x));
        const int nvec = (total - head) / V;
        vec_t* ov = reinterpret_cast<vec_t*>(out + head);
        // (row, position) of the thread's vector advance by 256 * V elements per trip: one divmod up front, then
        // an add and a compare per trip; a vector that straddles two rows (one in ~P / V) takes the select path
        int p0;
        int r0 = wc_divmod_p(head + (int)threadIdx.x * V, P, invP, p0);
        const int stepq = (256 * V) / P, stepr = 256 * V - stepq * P;
#pragma unroll 2
        for (int i = threadIdx.x; i < nvec; i += 256) {
            const float k0 = coefs[min(r0, rows - 1)];
            vec_t v;
            if (p0 + V <= P) {
#pragma unroll
                for (int e = 0; e < V; ++e) {
                    const float x = k0 * sm[midx(p0 + e)];
                    if constexpr (V == 4) v[e] = x;
                    else v[e] = f32_to_bf16(x);
                }
            } else {
                const float k1 = coefs[min(r0 + 1, rows - 1)];
#pragma unroll
                for (int e = 0; e < V; ++e) {
                    const bool wrap = p0 + e >= P;
                    const float x = (wrap ? k1 : k0) * sm[midx(wrap ? p0 + e - P : p0 + e)];
                    if constexpr (V == 4) v[e] = x;
                    else v[e] = f32_to_bf16(x);
                }
            }
            __builtin_nontemporal_store(v, ov + i);
            p0 += stepr;
            r0 += stepq;
            if (p0 >= P) { p0 -= P; ++r0; }
        }
        const int t0 = head + nvec * V;
        if (t0 + (int)threadIdx.x < total) FeatIO<T>::store(out + t0 + threadIdx.x, value(t0 + threadIdx.x));
    }
}

// The dFs stream of large batches: dFs[r][p] = coef[r] * mask[r / C][p] over the FLAT tensor, one block per
// contiguous chunk of 256 * U 16-byte vectors, blocks in address order.  The chip then writes ONE compact moving
// window (resident blocks x chunk) like a fill kernel instead of one 107 KB span per resident 16-row block:
// scripts/diag/write_ceiling.hip measures 6.6-6.7 TB/s for this pattern against 5.0 for per-block spans (and 6.3-6.6
// for hipMemset) on 344 MB.  No LDS: the mask (B * P floats, L2 / last-level-cache resident) is read with one
// unaligned 16-byte load per 16 bytes stored (rows are only element-aligned: P is odd), the row's coefficient with
// a 4-byte load; a vector that straddles two rows (one in ~P / V) is formed element by element.  Needs the
// coefficients as a tensor (CoefArgs::coef_out: the tail kernel's dFs blocks in coefficient-only mode).
typedef float f32x4_mu __attribute__((ext_vector_type(4), aligned(4)));
// grid (chunks per image, B): block (x, b) writes elements [x * CHUNK, (x + 1) * CHUNK) of image b's C * P span
// (C * P < 2^22 elements, so every index fits the float-reciprocal divmod; an image's span starts 16-byte aligned:
// C is a multiple of 64).  blockIdx.x runs fastest: dispatch order = address order.
template <typename T, int U>
__global__ __launch_bounds__(256) void dfs_chunk_kernel(const float* __restrict__ mask, const float* __restrict__ coef,
                                                       T* __restrict__ dFs, int C, int P) {
    constexpr int V = Vec16<T>::N;
    typedef typename Vec16<T>::type vec_t;
    constexpr int CHUNK = 256 * U * V;               // elements per block
    const int b = blockIdx.y;
    const int span = C * P;
    const float invP = 1.f / (float)P;
    const float* mr = mask + (int64_t)b * P;
    const float* cf = coef + (int64_t)b * C;
    T* out = dFs + (int64_t)b * span;
    f32x4_mu m[U][V / 4];
    float k0[U];
    int pp[U], rr[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
        // clamped, unconditional loads (vectors past the span's end are not stored)
        const int e0 = min((int)blockIdx.x * CHUNK + (u * 256 + (int)threadIdx.x) * V, span - 1);
        int p;
        const int r = wc_divmod_p(e0, P, invP, p);
        pp[u] = p;
        rr[u] = r;
        k0[u] = cf[r];
#pragma unroll
        for (int q = 0; q < V / 4; ++q) m[u][q] = *reinterpret_cast<const f32x4_mu*>(mr + max(min(p + 4 * q, P - 4), 0));
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
        const int e0 = (int)blockIdx.x * CHUNK + (u * 256 + (int)threadIdx.x) * V;
        if (e0 >= span) continue;
        const int p = pp[u];
        if (p + V <= P) {                            // (then e0 + V <= span as well)
            vec_t v;
#pragma unroll
            for (int e = 0; e < V; ++e) {
                const float x = k0[u] * m[u][e >> 2][e & 3];
                if constexpr (V == 4) v[e] = x;
                else v[e] = f32_to_bf16(x);
            }
            __builtin_nontemporal_store(v, reinterpret_cast<vec_t*>(out + e0));
        } else {
            // the vector crosses a row boundary (one in ~P / V) or the span's end: element by element
            for (int e = 0; e < V && e0 + e < span; ++e) {
                const bool wrap = p + e >= P;
                const float x = cf[rr[u] + (wrap ? 1 : 0)] * mr[wrap ? p + e - P : p + e];
                FeatIO<T>::store(out + e0 + e, x);
            }
        }
    }
}

template <typename T>
__global__ __launch_bounds__(256) void masked_sum_bwd_kernel(const float* __restrict__ mask,
                                                            const float* __restrict__ coef,
                                                            T* __restrict__ dFs, int C, int P,
                                                            int rows_per_block) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    masked_sum_bwd_body<T, 0>(mask, coef, dFs, C, P, rows_per_block, blockIdx.y, blockIdx.x, sm, CoefArgs{});
}

