// K0/K1/K1b: nearest mask resize, masked channel sums, and their backward.
//
// HBM-bound streaming kernels.  Fs is NCHW, so a (sample, channel) row is P
// contiguous elements; one wave owns a row at a time, reads it with 16-byte
// loads (after peeling to a 16-byte boundary: P = 1681 is odd, so rows are
// only element-aligned), multiplies by the sample's mask staged once per block
// in LDS, and wave-reduces.  Algorithmic bytes: B*C*P*sizeof(T) (+ B*P*4 mask).
#include "denet_common.hpp"

// ---------------------------------------------------------------------------
// K0: mask[b,p] = float(Ys[b, sy, sx]), cnt[b] = sum_p mask[b,p]
// reference: model/common.py:189-190, 237-238 (F.interpolate nearest)
// ---------------------------------------------------------------------------
// One block of 1024 threads per image; every thread requests its GB elements (clamped, unconditional) before it
// uses the first: as a loop of 256 threads over the row the kernel paid one memory round trip per 256 elements
// (seven in a row at 41 x 41, five unrolled trips at 81 x 81: 14 us for 80 images at C5).
#define MR_T 1024
template <int GB>
__global__ __launch_bounds__(MR_T) void mask_resize_kernel(const int64_t* __restrict__ Ys, int H, int W, int h, int w,
                                                           float* __restrict__ mask, float* __restrict__ cnt) {
    const int b = blockIdx.x;
    const int P = h * w;
    const float sy = nearest_scale(H, h), sx = nearest_scale(W, w);
    const int64_t* src = Ys + (int64_t)b * H * W;
    float acc = 0.f;
    for (int p0 = 0; p0 < P; p0 += MR_T * GB) {
        int64_t lv[GB];
#pragma unroll
        for (int u = 0; u < GB; ++u) {
            const int p = min(p0 + (int)threadIdx.x + MR_T * u, P - 1);
            const int y = p / w, x = p - y * w;
            lv[u] = src[(int64_t)nearest_src(y, sy, H) * W + nearest_src(x, sx, W)];
        }
#pragma unroll
        for (int u = 0; u < GB; ++u) {
            const int p = p0 + (int)threadIdx.x + MR_T * u;
            if (p < P) {
                const float m = (float)lv[u];
                mask[(int64_t)b * P + p] = m;
                acc += m;
            }
        }
    }
    // every partial sum is a small integer: exact in any order
    __shared__ float red[MR_T / 64];
    acc = wave_sum(acc);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) {
        float t = 0.f;
#pragma unroll
        for (int i = 0; i < MR_T / 64; ++i) t += red[i];
        cnt[b] = t;
    }
}

// The same resize without the count, flat over all images (thread <-> four elements of [B * P]): for the fused
// step, where one block per image leaves most of the chip idle (80 images at C5) and the pooling kernel's first
// channel block of an image sums the row it stages anyway.
#define MRF_E 4
__global__ __launch_bounds__(256) void mask_resize_flat_kernel(const int64_t* __restrict__ Ys, int H, int W, int h, int w,
                                                               int total, float* __restrict__ mask) {
    const int P = h * w;
    const float sy = nearest_scale(H, h), sx = nearest_scale(W, w);
    const int e0 = blockIdx.x * (256 * MRF_E) + threadIdx.x;
    int64_t lv[MRF_E];
#pragma unroll
    for (int u = 0; u < MRF_E; ++u) {
        const int e = min(e0 + 256 * u, total - 1);
        const int b = e / P, p = e - b * P;
        const int y = p / w, x = p - y * w;
        lv[u] = Ys[((int64_t)b * H + nearest_src(y, sy, H)) * W + nearest_src(x, sx, W)];
    }
#pragma unroll
    for (int u = 0; u < MRF_E; ++u) {
        const int e = e0 + 256 * u;
        if (e < total) mask[e] = (float)lv[u];
    }
}

#include "dev_masked_sum.hpp"

// ---------------------------------------------------------------------------
// C ABI
// ---------------------------------------------------------------------------
static inline bool fits_int(int64_t v) { return v > 0 && v < (int64_t)1 << 30; }

extern "C" int denet_mask_resize(const int64_t* Ys, int64_t B, int64_t H, int64_t W, int64_t h, int64_t w,
                                 float* mask, float* cnt, void* stream) {
    if (!Ys || !mask || !cnt) return DENET_E_BADARG;
    if (!fits_int(B) || !fits_int(H) || !fits_int(W) || !fits_int(h) || !fits_int(w) || !fits_int(h * w))
        return DENET_E_BADARG;
    if (h * w <= 2 * MR_T)
        hipLaunchKernelGGL(mask_resize_kernel<2>, dim3((unsigned)B), dim3(MR_T), 0, (hipStream_t)stream, Ys, (int)H,
                           (int)W, (int)h, (int)w, mask, cnt);
    else
        hipLaunchKernelGGL(mask_resize_kernel<8>, dim3((unsigned)B), dim3(MR_T), 0, (hipStream_t)stream, Ys, (int)H,
                           (int)W, (int)h, (int)w, mask, cnt);
    DENET_CHECK_LAUNCH();
    return DENET_OK;
}

#ifndef DENET_RPB
#define DENET_RPB 16
#endif
static inline int pick_rows_per_block(int64_t C) { return C >= DENET_RPB ? DENET_RPB : (int)C; }

// The launcher's choices for the pooling stage, from the call's sizes alone (also reported by
// denet_head_step_plan, so that tests can tell which variant a shape runs).
PoolPlan denet_pool_plan(int feat_dtype, int64_t B, int64_t C, int64_t P, bool fused) {
    const bool f32 = feat_dtype == DENET_F32;
    const int rpb = pick_rows_per_block(C);
    const int64_t nblk = ((C + rpb - 1) / rpb) * B;
    PoolPlan pl;
    // in the fused step the mask is resized in a launch of its own from 64 images or 64 x 64 features on
    // (denet_masked_pool_fwd_chain below)
    pl.two_launch = fused && (P >= 4096 || B >= 64);
    // short rows (up to 512 16-byte vectors: whole rows in flight) or long rows (rolling loop)
    pl.long_rows = P / (f32 ? 4 : 8) > 512;
    // feature streams of more than 160 MB are read around the caches (dev_masked_sum.hpp, NTL).  Step time
    // without / with, C2's shape: 138 MB 144.2 / 147.1 us, 172 MB 175.7 / 172.9, 207 MB 204.6 / 192.6, 241 MB
    // 243.6 / 228.9; C5's shape (bf16): 134 MB 237.1 / 237.0, 202 MB 332.9 / 316.1, 269 MB 420.5 / 397.9
    pl.nt_loads = nblk * rpb * P * (f32 ? 4 : 2) > ((int64_t)160 << 20);
    return pl;
}

// the pooling kernel: short rows (up to 512 16-byte vectors: whole rows in flight) or long rows (rolling loop)
static int launch_pool(const void* Fs, int feat_dtype, const int64_t* Ys, float* mask, float* cnt, float* S, int C, int P,
                       int rpb, int H, int W, int h, int w, const float* W1, float* pre1_part, dim3 grid, hipStream_t st) {
    const bool f32 = feat_dtype == DENET_F32;
    const PoolPlan pl = denet_pool_plan(feat_dtype, (int64_t)grid.x / ((C + rpb - 1) / rpb), C, P, false);
    const bool longr = pl.long_rows;
    // LDS: the mask image (dev_masked_sum.hpp) + 4 + 16 floats (+ the parked W1 slices).  The long-row loop reads
    // the image up to 959 vectors past the row's end (discarded by a select): a planar image holds one float per
    // vector and plane, so its MASK_TAIL covers that; the row-major fp32 image holds four, and those reads land
    // in the parked W1 slices -- or, without them (stand-alone call), in 3072 floats of padding
    const bool rowmajor = f32 && longr && !MaskImage<float, false>::PLANAR;
    size_t lds = (size_t)((f32 ? (longr ? mask_image_floats<4, MaskImage<float, false>::PLANAR>(P)
                                          : mask_image_floats<4, MaskImage<float, true>::PLANAR>(P))
                                   : mask_image_floats<8, true>(P)) + 4 + 16) * 4 +
                       (W1 ? 256 * 64 : (rowmajor ? 3072 * 4 : 0));
#define POOL_GO2(TT, LG, NL)                                                                                      \
    do {                                                                                                          \
        if (denet_ensure_dynamic_lds(reinterpret_cast<const void*>(masked_sum_fwd_kernel<TT, LG, NL>), lds))      \
            return DENET_E_LAUNCH;                                                                                \
        hipLaunchKernelGGL((masked_sum_fwd_kernel<TT, LG, NL>), grid, dim3(256), lds, st, (const TT*)Fs, Ys, mask, cnt, S, \
                           C, P, rpb, H, W, h, w, W1, pre1_part);                                                 \
    } while (0)
#define POOL_GO(TT, LG)                                                                                           \
    do {                                                                                                          \
        if (ntl) POOL_GO2(TT, LG, true); else POOL_GO2(TT, LG, false);                                            \
    } while (0)
    const bool ntl = pl.nt_loads;
#ifdef DENET_POOL_LDS_MIN_NTL
    if (ntl && lds < (size_t)DENET_POOL_LDS_MIN_NTL) lds = (size_t)DENET_POOL_LDS_MIN_NTL;        // diagnostic: fewer resident blocks
#endif
    if (f32) {
        if (longr) POOL_GO(float, true); else POOL_GO(float, false);
    } else {
        if (longr) POOL_GO(unsigned short, true); else POOL_GO(unsigned short, false);
    }
#undef POOL_GO2
#undef POOL_GO
    return DENET_OK;
}

extern "C" int denet_masked_sum_fwd(const void* Fs, int feat_dtype, const float* mask, int64_t B, int64_t C,
                                    int64_t P, float* S, void* stream) {
    if (!Fs || !mask || !S) return DENET_E_BADARG;
    if (!fits_int(B) || !fits_int(C) || !fits_int(P)) return DENET_E_BADARG;
    if (P * 4 > 64 * 1024 || B > 65535) return DENET_E_UNSUPPORTED;
    const int rpb = pick_rows_per_block(C);
    dim3 grid((unsigned)(((C + rpb - 1) / rpb) * B));
    if (feat_dtype != DENET_F32 && feat_dtype != DENET_BF16) return DENET_E_UNSUPPORTED;
    if (int rc = launch_pool(Fs, feat_dtype, nullptr, (float*)mask, nullptr, S, (int)C, (int)P, rpb, 0, 0, 0, 0, nullptr,
                             nullptr, grid, (hipStream_t)stream))
        return rc;
    DENET_CHECK_LAUNCH();
    return DENET_OK;
}

extern "C" int denet_masked_sum_bwd(const float* mask, const float* coef, int64_t B, int64_t C, int64_t P,
                                    void* dFs, int feat_dtype, void* stream) {
    if (!dFs || !mask || !coef) return DENET_E_BADARG;
    if (!fits_int(B) || !fits_int(C) || !fits_int(P)) return DENET_E_BADARG;
    if (P * 4 > 64 * 1024 || B > 65535) return DENET_E_UNSUPPORTED;
    const int rpb = pick_rows_per_block(C);
    dim3 grid((unsigned)((C + rpb - 1) / rpb), (unsigned)B);
    // mask image (dev_masked_sum.hpp) + 64 floats + the block's 16 coefficients
    if (feat_dtype == DENET_F32) {
        const size_t lds = (size_t)(mask_image_floats<4, MaskImage<float, false>::PLANAR>((int)P) + 64 + 16) * 4;
        if (denet_ensure_dynamic_lds(reinterpret_cast<const void*>(masked_sum_bwd_kernel<float>), lds)) return DENET_E_LAUNCH;
        hipLaunchKernelGGL(masked_sum_bwd_kernel<float>, grid, dim3(256), lds, (hipStream_t)stream, mask, coef,
                           (float*)dFs, (int)C, (int)P, rpb);
    } else if (feat_dtype == DENET_BF16) {
        const size_t lds = (size_t)(mask_image_floats<8, true>((int)P) + 64 + 16) * 4;
        if (denet_ensure_dynamic_lds(reinterpret_cast<const void*>(masked_sum_bwd_kernel<unsigned short>), lds))
            return DENET_E_LAUNCH;
        hipLaunchKernelGGL(masked_sum_bwd_kernel<unsigned short>, grid, dim3(256), lds, (hipStream_t)stream, mask,
                           coef, (unsigned short*)dFs, (int)C, (int)P, rpb);
    } else {
        return DENET_E_UNSUPPORTED;
    }
    DENET_CHECK_LAUNCH();
    return DENET_OK;
}

// K1b for large streams: the same dFs = mask x coef written by dfs_chunk_kernel (one block per contiguous chunk of
// the flat tensor, in address order, non-temporal stores) -- what the fused step runs for bf16 streams above 80 MB.
// Same arguments and results as denet_masked_sum_bwd; dFs must be 16-byte aligned, C a multiple of 64.
extern "C" int denet_masked_sum_bwd_chunked(const float* mask, const float* coef, int64_t B, int64_t C, int64_t P,
                                            void* dFs, int feat_dtype, void* stream) {
    if (!dFs || !mask || !coef) return DENET_E_BADARG;
    if (!fits_int(B) || !fits_int(C) || !fits_int(P)) return DENET_E_BADARG;
    if (P < 4 || (C & 63) || C * P >= (1 << 22) || B > 65535 || ((uintptr_t)dFs & 15)) return DENET_E_UNSUPPORTED;
    const int64_t span = C * P;
    if (feat_dtype == DENET_F32) {
        const int64_t per = 256 * 2 * 4;
        hipLaunchKernelGGL((dfs_chunk_kernel<float, 2>), dim3((unsigned)((span + per - 1) / per), (unsigned)B), dim3(256),
                           0, (hipStream_t)stream, mask, coef, (float*)dFs, (int)C, (int)P);
    } else if (feat_dtype == DENET_BF16) {
        const int64_t per = 256 * 2 * 8;
        hipLaunchKernelGGL((dfs_chunk_kernel<unsigned short, 2>), dim3((unsigned)((span + per - 1) / per), (unsigned)B),
                           dim3(256), 0, (hipStream_t)stream, mask, coef, (unsigned short*)dFs, (int)C, (int)P);
    } else {
        return DENET_E_UNSUPPORTED;
    }
    DENET_CHECK_LAUNCH();
    return DENET_OK;
}

// K0 + K1 in one launch: the mask is gathered from Ys inside the pooling kernel (and written out,
// with cnt, by the first channel block of every image for the backward pass).  With W1 != nullptr
// (sliced chain of the fused step; C must be a multiple of 16 and <= 256) every block also writes
// its split-K partial of the first gate layer: pre1_part [B][C/16][C].
int denet_masked_pool_fwd_chain(const void* Fs, int feat_dtype, const int64_t* Ys, int64_t B, int64_t C, int64_t H,
                                int64_t W, int64_t h, int64_t w, float* mask, float* cnt, float* S, const float* W1,
                                float* pre1_part, void* stream) {
    if (!Fs || !Ys || !mask || !cnt || !S) return DENET_E_BADARG;
    if ((W1 == nullptr) != (pre1_part == nullptr)) return DENET_E_BADARG;
    if (!fits_int(B) || !fits_int(C) || !fits_int(H) || !fits_int(W) || !fits_int(h) || !fits_int(w) ||
        !fits_int(h * w) || !fits_int(H * W))
        return DENET_E_BADARG;
    const int64_t P = h * w;
    if (P * 4 > 64 * 1024 || B > 65535) return DENET_E_UNSUPPORTED;
    if (W1 && ((C & 15) || C > 256)) return DENET_E_UNSUPPORTED;
    const int rpb = pick_rows_per_block(C);
    dim3 grid((unsigned)(((C + rpb - 1) / rpb) * B));
    // The in-kernel gather touches one 64-byte sector of the label map per mask element, once per channel
    // block: C/16 times 64 P bytes per image against C P sF bytes of features.  At 41 x 41 that is cheaper
    // than a launch of its own; from P = 4096 on (81 x 81: 420 KB of sectors per block against 210 KB of bf16
    // features) the mask is resized once per image first and the blocks read it back as P floats.  The same
    // from 64 images on: the gather is 1681 scattered 8-byte loads per block, sixteen times per image, and
    // costs the stream 15 us of 86 at 200 images (C3) and 40 of 258 at 640 (scripts/diag/pool_stamps: modes
    // 1 / 0 / 2); the extra launch costs ~4 us, which only small batches cannot afford (C2 step: 88.7 vs 87.2 us
    // at 40 images, 160.5 vs 161.7 us at 80).
    if (denet_pool_plan(feat_dtype, B, C, P, true).two_launch) {
        if (!fits_int(B * P)) return DENET_E_UNSUPPORTED;
        const int total = (int)(B * P);
        hipLaunchKernelGGL(mask_resize_flat_kernel, dim3((unsigned)((total + 256 * MRF_E - 1) / (256 * MRF_E))), dim3(256),
                           0, (hipStream_t)stream, Ys, (int)H, (int)W, (int)h, (int)w, total, mask);
        DENET_CHECK_LAUNCH();
        Ys = nullptr;                                 // the pooling blocks read `mask`; block 0 of an image writes cnt
    }
    if (feat_dtype != DENET_F32 && feat_dtype != DENET_BF16) return DENET_E_UNSUPPORTED;
    if (int rc = launch_pool(Fs, feat_dtype, Ys, mask, cnt, S, (int)C, (int)P, rpb, (int)H, (int)W, (int)h, (int)w, W1,
                             pre1_part, grid, (hipStream_t)stream))
        return rc;
    DENET_CHECK_LAUNCH();
    return DENET_OK;
}

extern "C" int denet_masked_pool_fwd(const void* Fs, int feat_dtype, const int64_t* Ys, int64_t B, int64_t C,
                                     int64_t H, int64_t W, int64_t h, int64_t w, float* mask, float* cnt, float* S,
                                     void* stream) {
    return denet_masked_pool_fwd_chain(Fs, feat_dtype, Ys, B, C, H, W, h, w, mask, cnt, S, nullptr, nullptr, stream);
}
