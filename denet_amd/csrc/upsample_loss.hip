// K4 / K4b / K5: bilinear up-sampling (align_corners=True) fused with the
// per-episode weighted cross entropy (forward + gradient w.r.t. the low-res
// logits) and with the binary focal loss (forward).  The [N,K,H,W] tensor the
// reference materialises (trainer/trainers.py:955-956; 8.7 MB per episode at
// K = 21) is never written.
//
// Decomposition: one block per (episode, low-res row py).  The block owns the
// band of label rows whose upper bilinear tap is py, so it needs only logits
// rows py and py+1 (2*K*w floats in LDS) and every label is read exactly once.
// A work item is (label row, low-res cell px): its <= 8 pixels share the four
// taps, so per class the item does 4 LDS reads + one vertical lerp, then 2 FMAs
// per pixel.  Gradient: the item reduces its pixels' contributions per tap in
// registers and issues 4 LDS float atomics per class; the block then writes two
// slabs (contributions to row py / row py+1) with plain stores, which K3b sums:
// no global atomics.
//
// Algorithmic bytes per episode: Q*8 (labels) + K*P*4 (logits) [+ 2*K*P*4 slabs].
#include "denet_common.hpp"

#include "dev_upsample_loss.hpp"
#include "dev_wce_chain.hpp"

// out[n,i] = gmul * scale[n] * (a[n,i] + b[n,i])   (dense gradient for the generic autograd path)
__global__ __launch_bounds__(256) void combine_slabs_kernel(const float* __restrict__ a, const float* __restrict__ b,
                                                           const float* __restrict__ scale,
                                                           const float* __restrict__ gmul, int64_t per,
                                                           float* __restrict__ out) {
    const int n = blockIdx.y;
    const float s = (scale ? scale[n] : 1.f) * (gmul ? gmul[0] : 1.f);
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < per; i += (int64_t)gridDim.x * 256) {
        const int64_t gi = (int64_t)n * per + i;
        out[gi] = s * (a[gi] + (b ? b[gi] : 0.f));
    }
}

// Plain up-sampling, ATen's operation order without FMA contraction:
//   out = h0*(w0*v00 + w1*v01) + h1*(w0*v10 + w1*v11)
// Write-bound (H*W*4 bytes per channel image): one thread per 4 consecutive output columns of one row,
// the row's vertical taps computed once per thread, 16-byte stores where the address allows.
typedef float f32x4_su __attribute__((ext_vector_type(4), aligned(4)));

__global__ __launch_bounds__(256) void upsample_bilinear_kernel(const float* __restrict__ x, int h, int w, int H,
                                                               int W, float* __restrict__ out) {
    const int nc = blockIdx.y;
    const float sy = bilinear_scale(h, H), sx = bilinear_scale(w, W);
    const float* src = x + (int64_t)nc * h * w;
    float* dst = out + (int64_t)nc * H * W;
    const int W4 = (W + 3) >> 2;
    for (int it = blockIdx.x * 256 + threadIdx.x; it < H * W4; it += gridDim.x * 256) {
        const int qy = it / W4, x4 = (it - qy * W4) * 4;
        int y0;
        float fy;
        bilinear_tap(qy, sy, h, y0, fy);
        const int y1 = y0 + (y0 < h - 1);
        const float h1 = fy, h0 = 1.f - fy;
        float v[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int qx = min(x4 + e, W - 1);
            int x0;
            float fx;
            bilinear_tap(qx, sx, w, x0, fx);
            const int x1 = x0 + (x0 < w - 1);
            const float w1 = fx, w0 = 1.f - fx;
            const float top = __fadd_rn(__fmul_rn(w0, src[y0 * w + x0]), __fmul_rn(w1, src[y0 * w + x1]));
            const float bot = __fadd_rn(__fmul_rn(w0, src[y1 * w + x0]), __fmul_rn(w1, src[y1 * w + x1]));
            v[e] = __fadd_rn(__fmul_rn(h0, top), __fmul_rn(h1, bot));
        }
        float* o = dst + (int64_t)qy * W + x4;
        if (x4 + 4 <= W) {
            f32x4_su vv = {v[0], v[1], v[2], v[3]};
            *reinterpret_cast<f32x4_su*>(o) = vv;
        } else {
            for (int e = 0; x4 + e < W; ++e) o[e] = v[e];
        }
    }
}

// ---------------------------------------------------------------------------
// C ABI
// ---------------------------------------------------------------------------
static inline bool pos(int64_t v) { return v > 0 && v < ((int64_t)1 << 30); }

template <typename KernelT>
static int set_lds(KernelT kern, size_t lds) {
    return denet_ensure_dynamic_lds(reinterpret_cast<const void*>(kern), lds);
}

// Launch geometry of the loss kernels, from the exact index rule (same float ops as the device).
struct LossGeom {
    int xsplit, rows_max, emax, xw_max, threads, ncmax, rsplit;
    size_t lds;          // upsample_loss_body<.,.,rsplit>
    // chain path of the weighted CE (dev_wce_chain.hpp): em = 0 if the geometry is not covered
    int em, kchunk;
    size_t lds_chain;    // max(chain layout, exact body with one row group: the in-kernel fallback)
};

static int max_span(int in_size, int out_size) {      // widest set of dst indices sharing one upper tap
    const float sc = bilinear_scale(in_size, out_size);
    int best = 0, prev = 0;
    for (int t = 1; t <= in_size; ++t) {
        const int q = first_dst_ge(t, sc, in_size, out_size);
        if (q - prev > best) best = q - prev;
        prev = q;
    }
    return best;
}

// LDS bytes of upsample_loss_body
static size_t exact_lds(int K, int ncmax, int rows_max, int emax, int xw_max, bool grad, int rsplit) {
    size_t off = (size_t)(2 + 4 * (grad ? rsplit : 0)) * K * ncmax + K + 8 + xw_max;
    off = (off + ncmax + 2 + 1) & ~(size_t)1;
    off += grad ? (size_t)2 * rows_max * emax * ncmax : 0;
    size_t lds = off * 4 + (size_t)rows_max * xw_max;
    return (lds + 15) & ~(size_t)15;
}

static bool loss_geom(int K, int h, int w, int H, int W, bool grad, LossGeom* g) {
    // up-sampling only, as the reference (loss/losses.py:29-37 raises otherwise): every low-res row and
    // column then owns at least one label row / column, which the kernel's band arithmetic relies on
    if (H < h || W < w) return false;
    const int rows_max = max_span(h, H), emax = max_span(w, W);
    const float sx = bilinear_scale(w, W);
    // aim at ~256 phase-1 items (rows x cells x 2 halves) per 256-thread block
    int want = 256 / (2 * (rows_max > 0 ? rows_max : 1)) - 1;
    if (want < 1) want = 1;
    bool found = false;
    for (int cw = want < w ? want : w; cw >= 1; --cw) {
        const int xsplit = (w + cw - 1) / cw;
        const int cwe = (w + xsplit - 1) / xsplit;          // the width the kernel derives from xsplit
        const int ncmax = cwe + (grad ? 2 : 1);           // must match the kernel's stride
        int xw_max = 0;
        for (int xsid = 0; xsid < xsplit; ++xsid) {
            const int c0 = xsid * cwe, c1 = c0 + cwe < w ? c0 + cwe : w;
            const int lo = (grad && c0 > 0) ? c0 - 1 : c0;
            const int span = first_dst_ge(c1, sx, w, W) - first_dst_ge(lo, sx, w, W);
            if (span > xw_max) xw_max = span;
        }
        // gradient pass items: (class, cell) x row groups.  Two row groups give K = 21 enough items for 256
        // threads; from ~500 (class, cell) pairs on, one group is enough and halves the accumulator arrays
        const int rsplit = grad ? (K * (cwe + 1) >= 512 ? 1 : 2) : 0;
        const size_t lds = exact_lds(K, ncmax, rows_max, emax, xw_max, grad, rsplit);
        if (lds <= 64 * 1024 || (cw == 1 && lds <= 150 * 1024)) {
            g->xsplit = xsplit; g->rows_max = rows_max; g->emax = emax; g->xw_max = xw_max; g->threads = UL_T;
            g->ncmax = ncmax;
            g->rsplit = rsplit;
            g->lds = lds;
            found = true;
            break;
        }
    }
    if (!found) return false;
    // chain path: cells of 8 or 9 label columns, the (row, cell) items of a block in one round of
    // thread pairs, and a table chunk that keeps the block within 40-52 KB of LDS (four / three blocks per CU)
    // if it can, within 64 KB otherwise
    g->em = 0;
    g->kchunk = 0;
    g->lds_chain = 0;
    const int em = (emax == 8 || emax == 9) ? emax : 0;
    if (em && 2 * rows_max * g->ncmax <= UL_T && rows_max * emax <= 500 /* 16-bit moment fields */) {
        const size_t fb = exact_lds(K, g->ncmax, rows_max, emax, g->xw_max, grad, 1);
        int best = 0;
        for (int pass = 0; pass < 2 && !best; ++pass) {
            // first try: 52 KB (three blocks per CU) for 8-column cells, 40 KB (four blocks) for 9-column cells.
            // Measured at K = 81: 41 -> 321 (C3) runs 5 chunks of 18 classes in 40 KB and 3 chunks of 28 in 52 KB,
            // F4 107 -> 87 us; 81 -> 642 (C5) 6 chunks of 14 against 4 of 22, 139 -> 144 us; 64 KB (two blocks per
            // CU) equals the 40 KB figure in both.  K <= 32 fits one chunk in 40 KB either way.
#ifndef DENET_WCE_CAP_KB
#define DENET_WCE_CAP_KB (emax <= 8 ? 52 : 40)
#endif
            const size_t cap = pass == 0 ? (size_t)DENET_WCE_CAP_KB * 1024 : 64 * 1024;
            for (int kc = K; kc >= (K < 8 ? K : 8); --kc) {
                if ((kc & 1) && kc != K) continue;          // class pairs must not straddle chunks
                const size_t b = (size_t)wc_lds_floats(K, g->ncmax, rows_max, kc, em, grad) * 4;
                if (b <= cap) { best = kc; break; }
            }
        }
        if (best && best < K) {
            // equal chunks of an even size (81 with room for 30 -> 3 chunks of 28, 28, 25)
            const int nchunk = (K + best - 1) / best;
            best = ((K + nchunk - 1) / nchunk + 1) & ~1;
        }
        if (best && fb <= 64 * 1024) {
            size_t b = (size_t)wc_lds_floats(K, g->ncmax, rows_max, best, em, grad) * 4;
            b = (b + 15) & ~(size_t)15;
            g->em = em;
            g->kchunk = best;
            g->lds_chain = b > fb ? b : fb;
        }
    }
    return true;
}

bool denet_loss_plan(int64_t K, int64_t h, int64_t w, int64_t H, int64_t W, bool grad, LossPlan* out) {
    LossGeom g;
    if (!loss_geom((int)K, (int)h, (int)w, (int)H, (int)W, grad, &g)) return false;
    *out = LossPlan{g.em, g.em ? g.kchunk : 0, g.xsplit, g.rsplit};
    return true;
}

extern "C" int64_t denet_upsample_loss_parts(int64_t K, int64_t h, int64_t w, int64_t H, int64_t W, int with_grad) {
    if (K <= 0 || h <= 0 || w <= 0 || H <= 0 || W <= 0) return DENET_E_BADARG;
    LossGeom g;
    if (!loss_geom((int)K, (int)h, (int)w, (int)H, (int)W, with_grad != 0, &g)) return DENET_E_UNSUPPORTED;
    return (int64_t)h * g.xsplit;
}

static int launch_loss(int mode, bool grad, const float* logits, const int64_t* Y, const float* wgt, int64_t N,
                       int64_t K, int64_t h, int64_t w, int64_t H, int64_t W, float alpha, float gamma, float* part,
                       float* da, float* db, hipStream_t st) {
    LossGeom g;
    if (!loss_geom((int)K, (int)h, (int)w, (int)H, (int)W, grad, &g)) return DENET_E_UNSUPPORTED;
    dim3 grid((unsigned)(h * g.xsplit), (unsigned)N), block((unsigned)g.threads);
    LossArgs a{logits, Y, wgt, (int)K, (int)h, (int)w, (int)H, (int)W, g.xsplit, g.rows_max, g.emax, g.xw_max,
               alpha, gamma, part, da, db, g.kchunk, bilinear_scale((int)h, (int)H), bilinear_scale((int)w, (int)W)};
#define UL_GO(KERN, LDS)                                                                                     \
    {                                                                                                        \
        int rc = set_lds(KERN, LDS);                                                                         \
        if (rc) return rc;                                                                                   \
        hipLaunchKernelGGL(KERN, grid, block, LDS, st, a);                                                   \
    }
    if (mode == 1) {
        hipLaunchKernelGGL(focal_kernel, grid, block, 0, st, a);
    }
    else if (g.em == 8 && grad) UL_GO((wce_chain_kernel<true, 8, 1>), g.lds_chain)
    else if (g.em == 9 && grad) UL_GO((wce_chain_kernel<true, 9, 1>), g.lds_chain)
    else if (g.em == 8) UL_GO((wce_chain_kernel<false, 8, 1>), g.lds_chain)
    else if (g.em == 9) UL_GO((wce_chain_kernel<false, 9, 1>), g.lds_chain)
    else if (!grad) UL_GO((upsample_loss_kernel<false, 2>), g.lds)
    else if (g.rsplit == 2) UL_GO((upsample_loss_kernel<true, 2>), g.lds)
    else UL_GO((upsample_loss_kernel<true, 1>), g.lds)
#undef UL_GO
    DENET_CHECK_LAUNCH();
    return DENET_OK;
}

// CE + focal in one launch (used by the fused step).  Same outputs as the two separate calls.
int denet_loss_pair_fwd(const float* logits, const int64_t* Yq_full, const float* wgt, const float* logits_binary,
                        const int64_t* Yq, int64_t N, int64_t NB, int64_t K, int64_t h, int64_t w, int64_t H,
                        int64_t W, float alpha, float gamma, float* part_k, float* part_p, float* da, float* db,
                        void* stream) {
    LossGeom g0, g1;
    const bool grad = da != nullptr;
    if (!loss_geom((int)K, (int)h, (int)w, (int)H, (int)W, grad, &g0)) return DENET_E_UNSUPPORTED;
    if (!loss_geom(2, (int)h, (int)w, (int)H, (int)W, false, &g1)) return DENET_E_UNSUPPORTED;
    if (N + NB > 65535) return DENET_E_UNSUPPORTED;
    const int nb0 = (int)h * g0.xsplit, nb1 = (int)h * g1.xsplit;
    const size_t lds = g0.em ? g0.lds_chain : g0.lds;       // (the focal blocks use 8 floats of it)
    LossArgs a0{logits, Yq_full, wgt, (int)K, (int)h, (int)w, (int)H, (int)W, g0.xsplit, g0.rows_max, g0.emax,
                g0.xw_max, 0.f, 0.f, part_k, da, db, g0.kchunk, bilinear_scale((int)h, (int)H), bilinear_scale((int)w, (int)W)};
    LossArgs a1{logits_binary, Yq, nullptr, 2, (int)h, (int)w, (int)H, (int)W, g1.xsplit, g1.rows_max, g1.emax,
                g1.xw_max, alpha, gamma, part_p, nullptr, nullptr, 0, bilinear_scale((int)h, (int)H), bilinear_scale((int)w, (int)W)};
    dim3 grid((unsigned)(nb0 > nb1 ? nb0 : nb1), (unsigned)(N + NB)), block(UL_T);
    hipStream_t st = (hipStream_t)stream;
#define UL_PAIR(KERN)                                                                                        \
    {                                                                                                        \
        int rc = set_lds(KERN, lds);                                                                         \
        if (rc) return rc;                                                                                   \
        hipLaunchKernelGGL(KERN, grid, block, lds, st, a0, a1, (int)N, nb0, nb1);                            \
    }
    if (g0.em == 8 && grad) UL_PAIR((wce_chain_pair_kernel<true, 8, 1>))
    else if (g0.em == 9 && grad) UL_PAIR((wce_chain_pair_kernel<true, 9, 1>))
    else if (g0.em == 8) UL_PAIR((wce_chain_pair_kernel<false, 8, 1>))
    else if (g0.em == 9) UL_PAIR((wce_chain_pair_kernel<false, 9, 1>))
    else if (!grad) UL_PAIR((upsample_loss_pair_kernel<false, 2>))
    else if (g0.rsplit == 2) UL_PAIR((upsample_loss_pair_kernel<true, 2>))
    else UL_PAIR((upsample_loss_pair_kernel<true, 1>))
#undef UL_PAIR
    DENET_CHECK_LAUNCH();
    return DENET_OK;
}

extern "C" int denet_upsample_wce_fwd(const float* logits, const int64_t* Yq_full, const float* wgt, int64_t N,
                                      int64_t K, int64_t h, int64_t w, int64_t H, int64_t W, float* part,
                                      float* dlog_a, float* dlog_b, void* stream) {
    if (!logits || !Yq_full || !part) return DENET_E_BADARG;
    if ((dlog_a == nullptr) != (dlog_b == nullptr)) return DENET_E_BADARG;
    if (!pos(N) || !pos(K) || !pos(h) || !pos(w) || !pos(H) || !pos(W) || !pos(H * W)) return DENET_E_BADARG;
    if (K > 128 || N > 65535) return DENET_E_UNSUPPORTED;
    return launch_loss(0, dlog_a != nullptr, logits, Yq_full, wgt, N, K, h, w, H, W, 0.f, 0.f, part, dlog_a, dlog_b,
                       (hipStream_t)stream);
}

extern "C" int denet_upsample_focal_fwd(const float* logits_binary, const int64_t* Yq, int64_t NB, int64_t h,
                                        int64_t w, int64_t H, int64_t W, float alpha, float gamma, float* part,
                                        void* stream) {
    if (!logits_binary || !Yq || !part) return DENET_E_BADARG;
    if (!pos(NB) || !pos(h) || !pos(w) || !pos(H) || !pos(W) || !pos(H * W)) return DENET_E_BADARG;
    if (NB > 65535) return DENET_E_UNSUPPORTED;
    return launch_loss(1, false, logits_binary, Yq, nullptr, NB, 2, h, w, H, W, alpha, gamma, part, nullptr,
                       nullptr, (hipStream_t)stream);
}

extern "C" int denet_combine_slabs(const float* a, const float* b, const float* scale, const float* gmul, int64_t N,
                                   int64_t per, float* out, void* stream) {
    if (!a || !out) return DENET_E_BADARG;
    if (!pos(N) || !pos(per)) return DENET_E_BADARG;
    if (N > 65535) return DENET_E_UNSUPPORTED;
    unsigned gx = (unsigned)((per + 255) / 256);
    if (gx > 256) gx = 256;
    hipLaunchKernelGGL(combine_slabs_kernel, dim3(gx, (unsigned)N), dim3(256), 0, (hipStream_t)stream, a, b, scale,
                       gmul, per, out);
    DENET_CHECK_LAUNCH();
    return DENET_OK;
}

extern "C" int denet_loss_finalize(const float* part_k, const float* part_p, int64_t N, int64_t NB,
                                   int64_t parts_k, int64_t parts_p, int pooled, float* out, float* per_episode,
                                   float* scale, void* stream) {
    if (!out || (!part_k && !part_p)) return DENET_E_BADARG;
    if ((part_k && (!pos(N) || !pos(parts_k))) || (part_p && (!pos(NB) || !pos(parts_p)))) return DENET_E_BADARG;
    hipLaunchKernelGGL(loss_finalize_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, part_k, part_p, (int)N,
                       (int)NB, (int)parts_k, (int)parts_p, pooled, out, per_episode, scale);
    DENET_CHECK_LAUNCH();
    return DENET_OK;
}

extern "C" int denet_upsample_bilinear(const float* x, int64_t NC, int64_t h, int64_t w, int64_t H, int64_t W,
                                       float* out, void* stream) {
    if (!x || !out) return DENET_E_BADARG;
    if (!pos(NC) || !pos(h) || !pos(w) || !pos(H) || !pos(W) || !pos(H * W)) return DENET_E_BADARG;
    if (NC > 65535) return DENET_E_UNSUPPORTED;
    unsigned gx = (unsigned)((H * ((W + 3) / 4) + 255) / 256);
    if (gx > 128) gx = 128;
    hipLaunchKernelGGL(upsample_bilinear_kernel, dim3(gx, (unsigned)NC), dim3(256), 0, (hipStream_t)stream, x,
                       (int)h, (int)w, (int)H, (int)W, out);
    DENET_CHECK_LAUNCH();
    return DENET_OK;
}
