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

#define UL_THREADS 256
#define UL_CH 8
#define UL_IGNORE 255

// smallest q in [0, out_size] with min(int(scale*q), in_size-1) >= target
__device__ __forceinline__ int first_dst_ge(int target, float scale, int in_size, int out_size) {
    int lo = 0, hi = out_size;
    while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        int i0 = (int)(scale * (float)mid);
        if (i0 > in_size - 1) i0 = in_size - 1;
        if (i0 >= target) hi = mid; else lo = mid + 1;
    }
    return lo;
}

__device__ __forceinline__ float block_sum_256(float v, float* red) {
    v = wave_sum(v);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    return (red[0] + red[1]) + (red[2] + red[3]);
}

// MODE 0: weighted CE (loss/losses.py:19-44 with per-episode class weights, trainer/utils.py:270-277)
// MODE 1: binary focal (loss/losses.py:47-79)
template <int MODE, bool GRAD>
__global__ __launch_bounds__(UL_THREADS) void upsample_loss_kernel(
    const float* __restrict__ logits, const int64_t* __restrict__ Y, const float* __restrict__ wgt, int K, int h,
    int w, int H, int W, int rch, float alpha, float gamma, float* __restrict__ part,
    float* __restrict__ dlog_a, float* __restrict__ dlog_b) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    const int n = blockIdx.y, py = blockIdx.x;
    const int py1 = py + 1 < h ? py + 1 : h - 1;
    const int tid = threadIdx.x;
    const int kw = K * w;
    float* lg0 = sm;                      // [K][w]   row py
    float* lg1 = lg0 + kw;                // [K][w]   row py1
    float* ga0 = lg1 + kw;                // [K][w]   grad wrt row py   (GRAD only)
    float* ga1 = ga0 + (GRAD ? kw : 0);
    float* wg = ga1 + (GRAD ? kw : 0);    // [K]
    float* red = wg + K;                  // [4]
    int* xs = reinterpret_cast<int*>(red + 4);              // [w+1]
    unsigned char* lab8 = reinterpret_cast<unsigned char*>(xs + w + 1);   // [rch][W]

    const float sy = bilinear_scale(h, H), sx = bilinear_scale(w, W);
    const int ys = first_dst_ge(py, sy, h, H), ye = first_dst_ge(py + 1, sy, h, H);

    const float* src = logits + (int64_t)n * K * h * w;
    for (int idx = tid; idx < kw; idx += UL_THREADS) {
        const int k = idx / w, x = idx - k * w;
        lg0[idx] = src[((int64_t)k * h + py) * w + x];
        lg1[idx] = src[((int64_t)k * h + py1) * w + x];
        if (GRAD) { ga0[idx] = 0.f; ga1[idx] = 0.f; }
    }
    for (int k = tid; k < K; k += UL_THREADS) wg[k] = (MODE == 0 && wgt) ? wgt[(int64_t)n * K + k] : 1.f;
    for (int px = tid; px <= w; px += UL_THREADS) xs[px] = first_dst_ge(px, sx, w, W);

    float num = 0.f, den = 0.f;
    const int64_t* Yn = Y + (int64_t)n * H * W;
    for (int r0 = ys; r0 < ye; r0 += rch) {
        const int nr = min(rch, ye - r0);
        __syncthreads();
        for (int i = tid; i < nr * W; i += UL_THREADS) {
            const int64_t v = Yn[(int64_t)r0 * W + i];
            lab8[i] = (v >= 0 && v < K) ? (unsigned char)v : (unsigned char)UL_IGNORE;
        }
        __syncthreads();
        for (int item = tid; item < nr * w; item += UL_THREADS) {
            const int r = item / w, px = item - r * w;
            const int px1 = px + 1 < w ? px + 1 : w - 1;
            const int qy = r0 + r;
            const float h1 = sy * (float)qy - (float)py, h0 = 1.f - h1;
            const int xend = xs[px + 1];
            for (int x0 = xs[px]; x0 < xend; x0 += UL_CH) {
                const int cntx = min(UL_CH, xend - x0);
                float w0[UL_CH], w1[UL_CH];
                int lb[UL_CH];
#pragma unroll
                for (int e = 0; e < UL_CH; ++e) {
                    const bool ok = e < cntx;
                    const int qx = x0 + e;
                    const float f = sx * (float)qx - (float)px;
                    w1[e] = ok ? f : 0.f;
                    w0[e] = 1.f - w1[e];
                    lb[e] = ok ? (int)lab8[r * W + qx] : UL_IGNORE;
                }
                // pass A: per-pixel max
                float mx[UL_CH];
#pragma unroll
                for (int e = 0; e < UL_CH; ++e) mx[e] = -INFINITY;
                for (int k = 0; k < K; ++k) {
                    const float a = h0 * lg0[k * w + px] + h1 * lg1[k * w + px];
                    const float b = h0 * lg0[k * w + px1] + h1 * lg1[k * w + px1];
#pragma unroll
                    for (int e = 0; e < UL_CH; ++e) mx[e] = fmaxf(mx[e], w0[e] * a + w1[e] * b);
                }
                // pass B: sum of exponentials and the label's logit
                float se[UL_CH], vl[UL_CH];
#pragma unroll
                for (int e = 0; e < UL_CH; ++e) { se[e] = 0.f; vl[e] = 0.f; }
                for (int k = 0; k < K; ++k) {
                    const float a = h0 * lg0[k * w + px] + h1 * lg1[k * w + px];
                    const float b = h0 * lg0[k * w + px1] + h1 * lg1[k * w + px1];
#pragma unroll
                    for (int e = 0; e < UL_CH; ++e) {
                        const float v = w0[e] * a + w1[e] * b;
                        se[e] += __expf(v - mx[e]);
                        vl[e] = (k == lb[e]) ? v : vl[e];
                    }
                }
                float ce[UL_CH];
#pragma unroll
                for (int e = 0; e < UL_CH; ++e) {
                    ce[e] = 0.f;
                    if (lb[e] != UL_IGNORE) {
                        if (MODE == 0) {
                            const float wy = wg[lb[e]];
                            num += wy * ((__logf(se[e]) + mx[e]) - vl[e]);
                            den += wy;
                            ce[e] = wy;
                        } else {
                            // log(softmax) as the reference computes it: -inf when the prob underflows
                            const float p = __expf(vl[e] - mx[e]) / se[e];
                            const float lp = logf(p);
                            const float om = 1.f - p;
                            const float mod = (gamma == 2.f) ? om * om : powf(om, gamma);
                            const float ay = lb[e] == 1 ? 1.f - alpha : alpha;
                            num -= ay * (mod * lp);
                            den += ay;
                        }
                    }
                }
                if (GRAD) {
                    float inv[UL_CH];
#pragma unroll
                    for (int e = 0; e < UL_CH; ++e) inv[e] = ce[e] / se[e];
                    for (int k = 0; k < K; ++k) {
                        const float a = h0 * lg0[k * w + px] + h1 * lg1[k * w + px];
                        const float b = h0 * lg0[k * w + px1] + h1 * lg1[k * w + px1];
                        float gA = 0.f, gB = 0.f;
#pragma unroll
                        for (int e = 0; e < UL_CH; ++e) {
                            const float v = w0[e] * a + w1[e] * b;
                            float g = __expf(v - mx[e]) * inv[e];
                            g -= (k == lb[e]) ? ce[e] : 0.f;
                            gA += g * w0[e];
                            gB += g * w1[e];
                        }
                        atomicAdd(&ga0[k * w + px], h0 * gA);
                        atomicAdd(&ga0[k * w + px1], h0 * gB);
                        atomicAdd(&ga1[k * w + px], h1 * gA);
                        atomicAdd(&ga1[k * w + px1], h1 * gB);
                    }
                }
            }
        }
    }
    const float tn = block_sum_256(num, red);
    const float td = block_sum_256(den, red);
    if (tid == 0) {
        part[((int64_t)n * h + py) * 2] = tn;
        part[((int64_t)n * h + py) * 2 + 1] = td;
    }
    if (GRAD) {
        __syncthreads();
        float* da = dlog_a + (int64_t)n * K * h * w;
        float* db = dlog_b + (int64_t)n * K * h * w;
        for (int idx = tid; idx < kw; idx += UL_THREADS) {
            const int k = idx / w, x = idx - k * w;
            float va = ga0[idx];
            if (py1 == py) va += ga1[idx];
            else db[((int64_t)k * h + py1) * w + x] = ga1[idx];
            da[((int64_t)k * h + py) * w + x] = va;
            if (py == 0) db[((int64_t)k * h) * w + x] = 0.f;
        }
    }
}

// loss_k = mean_n num_n/den_n ; loss_p = sum num / sum den ; scale[n] = 1/(N*den_n)
__global__ __launch_bounds__(256) void loss_finalize_kernel(const float* __restrict__ part_k,
                                                           const float* __restrict__ part_p, int N, int NB, int h,
                                                           int pooled, float* __restrict__ out,
                                                           float* __restrict__ per_episode,
                                                           float* __restrict__ scale) {
    __shared__ float red[4];
    const int tid = threadIdx.x;
    if (part_k && !pooled) {
        // trainer/utils.py:270-277: mean over episodes of per-episode weighted means
        float ls = 0.f;
        for (int n = tid; n < N; n += 256) {
            float nu = 0.f, de = 0.f;
            for (int y = 0; y < h; ++y) {
                nu += part_k[((int64_t)n * h + y) * 2];
                de += part_k[((int64_t)n * h + y) * 2 + 1];
            }
            const float l = nu / de;
            if (per_episode) per_episode[n] = l;
            if (scale) scale[n] = 1.f / ((float)N * de);
            ls += l;
        }
        ls = block_sum_256(ls, red);
        if (tid == 0) out[0] = ls / (float)N;
    }
    if (part_k && pooled) {
        // loss/losses.py:39-44 on a batch: one weighted mean over every pixel of every sample
        float nu = 0.f, de = 0.f;
        for (int i = tid; i < N * h; i += 256) {
            nu += part_k[(int64_t)i * 2];
            de += part_k[(int64_t)i * 2 + 1];
        }
        nu = block_sum_256(nu, red);
        de = block_sum_256(de, red);
        if (tid == 0) out[0] = nu / de;
        if (scale)
            for (int n = tid; n < N; n += 256) scale[n] = 1.f / de;
        if (per_episode)
            for (int n = tid; n < N; n += 256) per_episode[n] = nu / de;
    }
    if (part_p) {
        float nu = 0.f, de = 0.f;
        for (int i = tid; i < NB * h; i += 256) {
            nu += part_p[(int64_t)i * 2];
            de += part_p[(int64_t)i * 2 + 1];
        }
        nu = block_sum_256(nu, red);
        de = block_sum_256(de, red);
        if (tid == 0) out[1] = nu / de;
    }
}

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
__global__ __launch_bounds__(256) void upsample_bilinear_kernel(const float* __restrict__ x, int h, int w, int H,
                                                               int W, float* __restrict__ out) {
    const int nc = blockIdx.y;
    const float sy = bilinear_scale(h, H), sx = bilinear_scale(w, W);
    const float* src = x + (int64_t)nc * h * w;
    float* dst = out + (int64_t)nc * H * W;
    for (int q = blockIdx.x * 256 + threadIdx.x; q < H * W; q += gridDim.x * 256) {
        const int qy = q / W, qx = q - qy * W;
        int y0, x0;
        float fy, fx;
        bilinear_tap(qy, sy, h, y0, fy);
        bilinear_tap(qx, sx, w, x0, fx);
        const int y1 = y0 + (y0 < h - 1), x1 = x0 + (x0 < w - 1);
        const float h1 = fy, h0 = 1.f - fy, w1 = fx, w0 = 1.f - fx;
        const float top = __fadd_rn(__fmul_rn(w0, src[y0 * w + x0]), __fmul_rn(w1, src[y0 * w + x1]));
        const float bot = __fadd_rn(__fmul_rn(w0, src[y1 * w + x0]), __fmul_rn(w1, src[y1 * w + x1]));
        dst[q] = __fadd_rn(__fmul_rn(h0, top), __fmul_rn(h1, bot));
    }
}

// ---------------------------------------------------------------------------
// C ABI
// ---------------------------------------------------------------------------
static inline bool pos(int64_t v) { return v > 0 && v < ((int64_t)1 << 30); }

static int band_rows_bound(int64_t h, int64_t H) {
    if (h <= 1) return (int)H;
    return (int)((H - 1 + h - 2) / (h - 1)) + 1;
}

template <typename KernelT>
static int set_lds(KernelT kern, size_t lds) {
    return denet_ensure_dynamic_lds(reinterpret_cast<const void*>(kern), lds);
}

static int launch_loss(int mode, bool grad, const float* logits, const int64_t* Y, const float* wgt, int64_t N,
                       int64_t K, int64_t h, int64_t w, int64_t H, int64_t W, float alpha, float gamma, float* part,
                       float* da, float* db, hipStream_t st) {
    int rch = band_rows_bound(h, H);
    const int cap = (int)((16 * 1024) / W);
    if (rch > cap) rch = cap > 0 ? cap : 1;
    size_t lds = (size_t)(2 + (grad ? 2 : 0)) * K * w * 4 + (size_t)K * 4 + 16 + (size_t)(w + 1) * 4 +
                 (size_t)rch * W;
    lds = (lds + 15) & ~(size_t)15;
    if (lds > 160 * 1024) return DENET_E_UNSUPPORTED;
    dim3 grid((unsigned)h, (unsigned)N), block(UL_THREADS);
#define UL_GO(M, G)                                                                                          \
    {                                                                                                        \
        int rc = set_lds(upsample_loss_kernel<M, G>, lds);                                                   \
        if (rc) return rc;                                                                                   \
        hipLaunchKernelGGL((upsample_loss_kernel<M, G>), grid, block, lds, st, logits, Y, wgt, (int)K,       \
                           (int)h, (int)w, (int)H, (int)W, rch, alpha, gamma, part, da, db);                 \
    }
    if (mode == 0) {
        if (grad) UL_GO(0, true) else UL_GO(0, false)
    } else {
        UL_GO(1, false)
    }
#undef UL_GO
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

extern "C" int denet_loss_finalize(const float* part_k, const float* part_p, int64_t N, int64_t NB, int64_t h,
                                   int pooled, float* out, float* per_episode, float* scale, void* stream) {
    if (!out || (!part_k && !part_p)) return DENET_E_BADARG;
    if ((part_k && !pos(N)) || (part_p && !pos(NB)) || !pos(h)) return DENET_E_BADARG;
    hipLaunchKernelGGL(loss_finalize_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, part_k, part_p, (int)N,
                       (int)NB, (int)h, pooled, out, per_episode, scale);
    DENET_CHECK_LAUNCH();
    return DENET_OK;
}

extern "C" int denet_upsample_bilinear(const float* x, int64_t NC, int64_t h, int64_t w, int64_t H, int64_t W,
                                       float* out, void* stream) {
    if (!x || !out) return DENET_E_BADARG;
    if (!pos(NC) || !pos(h) || !pos(w) || !pos(H) || !pos(W) || !pos(H * W)) return DENET_E_BADARG;
    if (NC > 65535) return DENET_E_UNSUPPORTED;
    unsigned gx = (unsigned)((H * W + 255) / 256);
    if (gx > 64) gx = 64;
    hipLaunchKernelGGL(upsample_bilinear_kernel, dim3(gx, (unsigned)NC), dim3(256), 0, (hipStream_t)stream, x,
                       (int)h, (int)w, (int)H, (int)W, out);
    DENET_CHECK_LAUNCH();
    return DENET_OK;
}
