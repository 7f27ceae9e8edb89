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
#define UL_CH 4
#define UL_IGNORE 255

// smallest q in [0, out_size] with min(int(scale*q), in_size-1) >= target
__host__ __device__ __forceinline__ int first_dst_ge(int target, float scale, int in_size, int out_size) {
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
// any block size that is a multiple of 64 and <= 512 (red: 8 floats)
__device__ __forceinline__ float block_sum(float v, float* red) {
    v = wave_sum(v);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    float t = 0.f;
    const int nw = blockDim.x >> 6;
    for (int i = 0; i < nw; ++i) t += red[i];
    return t;
}

// MODE 0: weighted CE (loss/losses.py:19-44 with per-episode class weights, trainer/utils.py:270-277)
// MODE 1: binary focal (loss/losses.py:47-79)
//
// Block = (episode n, low-res row py, column range [c0,c1) of the low-res grid), 256 threads.
// All soft-max arithmetic is done in the log2 domain (the logit rows are multiplied by log2(e) when
// they are staged in LDS), so an exponential is one v_exp_f32.
// Phase 1, item = (label row r, cell px, half of the cell's pixels): per-pixel soft-max statistics.
//   The shift is the upper bound U = w0*max_k a_k + w1*max_k b_k of the blended logits (a, b: the
//   vertically blended left/right taps), which saves the per-pixel max pass; if every exponential
//   underflows against it (logit swings > 87 between neighbouring positions) the item falls back to
//   the exact per-pixel max.  Output: loss partial sums and (GRAD) (shift, weight/sum-exp) per pixel
//   in LDS as [r][e][cell] so phase 2 reads them with consecutive lanes on consecutive addresses.
// Phase 2 (GRAD), item = (class k, cell px, half of the band's rows): every (pixel, class) pair is
//   visited exactly once and its contribution to the cell's four taps is accumulated in registers
//   -- no atomics.  Left-tap and right-tap sums go to separate LDS arrays; output column x is
//   left[x] + right[x-1] (+ right[w-1] for the last column, whose right tap is clamped onto
//   itself).  The block owns output columns [c0,c1), so it also processes cell c0-1 (halo).
#define UL_T 256
#define UL_LOG2E 1.4426950408889634f
#define UL_LN2 0.6931471805599453f

template <int MODE, bool GRAD>
__global__ __launch_bounds__(UL_T) void upsample_loss_kernel(
    const float* __restrict__ logits, const int64_t* __restrict__ Y, const float* __restrict__ wgt, int K, int h,
    int w, int H, int W, int xsplit, int rows_max, int emax, int xw_max, float alpha, float gamma,
    float* __restrict__ part, float* __restrict__ dlog_a, float* __restrict__ dlog_b) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    const int n = blockIdx.y;
    const int py = blockIdx.x / xsplit, xsid = blockIdx.x - py * xsplit;
    const int py1 = py + 1 < h ? py + 1 : h - 1;
    const int tid = threadIdx.x;
    const int cw = (w + xsplit - 1) / xsplit;
    const int c0 = xsid * cw, c1 = min(w, c0 + cw);
    const int cell_lo = (GRAD && c0 > 0) ? c0 - 1 : c0;
    const int ncell = c1 - cell_lo;                       // cells processed
    const int ncmax = cw + (GRAD ? 2 : 1);                // LDS stride: cells + right tap (+ left halo cell)
    const int col_hi = min(c1, w - 1);                    // last logits column needed
    const int ncol = col_hi - cell_lo + 1;

    const int kc = K * ncmax;
    float* lg0 = sm;                                      // [K][ncmax]  row py * log2e, columns cell_lo..
    float* lg1 = lg0 + kc;                                // [K][ncmax]  row py1 * log2e
    float* gl0 = lg1 + kc;                                // GRAD: [2 row halves][K][ncmax] left tap, row-tap 0
    float* gl1 = gl0 + (GRAD ? 2 * kc : 0);
    float* gr0 = gl1 + (GRAD ? 2 * kc : 0);
    float* gr1 = gr0 + (GRAD ? 2 * kc : 0);
    float* wg = gr1 + (GRAD ? 2 * kc : 0);                // [K]
    float* red = wg + K;                                  // [8]
    float* w1t = red + 8;                                 // [xw_max] horizontal lerp weight of each label column
    int off = (2 + (GRAD ? 8 : 0)) * kc + K + 8 + xw_max;
    int* xs = reinterpret_cast<int*>(sm + off);           // [ncmax+1] first pixel of each cell
    off = (off + ncmax + 2 + 1) & ~1;
    f32x2* stats = reinterpret_cast<f32x2*>(sm + off);    // [rows_max][emax][ncmax] (shift, weight/sumexp), GRAD
    off += GRAD ? 2 * rows_max * emax * ncmax : 0;
    unsigned char* lab8 = reinterpret_cast<unsigned char*>(sm + off);   // [rows_max][xw_max]

    const float sy = bilinear_scale(h, H), sx = bilinear_scale(w, W);
    const int ys = first_dst_ge(py, sy, h, H), ye = first_dst_ge(py + 1, sy, h, H);
    const int rows = ye - ys;

    const float* src = logits + (int64_t)n * K * h * w;
#pragma unroll 4
    for (int idx = tid; idx < K * ncol; idx += UL_T) {
        const int k = idx / ncol, x = idx - k * ncol;
        lg0[k * ncmax + x] = src[((int64_t)k * h + py) * w + cell_lo + x] * UL_LOG2E;
        lg1[k * ncmax + x] = src[((int64_t)k * h + py1) * w + cell_lo + x] * UL_LOG2E;
    }
    for (int k = tid; k < K; k += UL_T) wg[k] = (MODE == 0 && wgt) ? wgt[(int64_t)n * K + k] : 1.f;
    for (int c = tid; c <= ncell; c += UL_T) xs[c] = first_dst_ge(cell_lo + c, sx, w, W);
    __syncthreads();
    const int xbase = xs[0], xw = xs[ncell] - xbase;      // label columns needed
    {
        const int64_t* Yn = Y + (int64_t)n * H * W + (int64_t)ys * W + xbase;
#pragma unroll 8
        for (int i = tid; i < rows * xw; i += UL_T) {
            const int r = i / xw, x = i - r * xw;
            const int64_t v = Yn[(int64_t)r * W + x];
            lab8[r * xw_max + x] = (v >= 0 && v < K) ? (unsigned char)v : (unsigned char)UL_IGNORE;
        }
        for (int x = tid; x < xw; x += UL_T) {
            int i0;
            float f;
            bilinear_tap(xbase + x, sx, w, i0, f);
            w1t[x] = f;
        }
    }
    __syncthreads();

    // ---------------- phase 1 ----------------
    float num = 0.f, den = 0.f;
    for (int item = tid; item < rows * ncell * 2; item += UL_T) {
        const int hf = item & 1;
        const int rc = item >> 1;
        const int r = rc / ncell, cl = rc - r * ncell;
        const int px = cell_lo + cl;
        const int cr = min(px + 1, w - 1) - cell_lo;      // local column of the right tap
        const bool owned = px >= c0;
        const int qy = ys + r;
        const float h1 = sy * (float)qy - (float)py, h0 = 1.f - h1;
        const int cx0 = xs[cl], cx1 = xs[cl + 1];
        const int mid = cx0 + ((cx1 - cx0 + 1) >> 1);
        const int xb = hf ? mid : cx0, xe = hf ? cx1 : mid;
        if (xb >= xe) continue;
        // max over classes of the two vertically blended taps
        float Ma = -INFINITY, Mb = -INFINITY;
#pragma unroll 3
        for (int k = 0; k < K; ++k) {
            Ma = fmaxf(Ma, h0 * lg0[k * ncmax + cl] + h1 * lg1[k * ncmax + cl]);
            Mb = fmaxf(Mb, h0 * lg0[k * ncmax + cr] + h1 * lg1[k * ncmax + cr]);
        }
        for (int x0 = xb; x0 < xe; x0 += UL_CH) {
            const int cntx = min(UL_CH, xe - x0);
            float w1[UL_CH], U[UL_CH], se[UL_CH];
            int lb[UL_CH];
#pragma unroll
            for (int e = 0; e < UL_CH; ++e) {
                const bool ok = e < cntx;
                w1[e] = ok ? w1t[x0 + e - xbase] : 0.f;
                lb[e] = ok ? (int)lab8[r * xw_max + x0 + e - xbase] : UL_IGNORE;
                U[e] = (1.f - w1[e]) * Ma + w1[e] * Mb;
                se[e] = 0.f;
            }
#pragma unroll 3
            for (int k = 0; k < K; ++k) {
                const float a = h0 * lg0[k * ncmax + cl] + h1 * lg1[k * ncmax + cl];
                const float d = (h0 * lg0[k * ncmax + cr] + h1 * lg1[k * ncmax + cr]) - a;
#pragma unroll
                for (int e = 0; e < UL_CH; ++e) se[e] += __builtin_amdgcn_exp2f(fmaf(w1[e], d, a) - U[e]);
            }
            bool under = false;
#pragma unroll
            for (int e = 0; e < UL_CH; ++e) under = under || !(se[e] > 1e-30f);
            if (under) {
                // exact per-pixel max (rare): every exponential underflowed against the bound
#pragma unroll
                for (int e = 0; e < UL_CH; ++e) U[e] = -INFINITY;
                for (int k = 0; k < K; ++k) {
                    const float a = h0 * lg0[k * ncmax + cl] + h1 * lg1[k * ncmax + cl];
                    const float d = (h0 * lg0[k * ncmax + cr] + h1 * lg1[k * ncmax + cr]) - a;
#pragma unroll
                    for (int e = 0; e < UL_CH; ++e) U[e] = fmaxf(U[e], fmaf(w1[e], d, a));
                }
#pragma unroll
                for (int e = 0; e < UL_CH; ++e) se[e] = 0.f;
                for (int k = 0; k < K; ++k) {
                    const float a = h0 * lg0[k * ncmax + cl] + h1 * lg1[k * ncmax + cl];
                    const float d = (h0 * lg0[k * ncmax + cr] + h1 * lg1[k * ncmax + cr]) - a;
#pragma unroll
                    for (int e = 0; e < UL_CH; ++e) se[e] += __builtin_amdgcn_exp2f(fmaf(w1[e], d, a) - U[e]);
                }
            }
#pragma unroll
            for (int e = 0; e < UL_CH; ++e) {
                float ce = 0.f;
                if (lb[e] != UL_IGNORE) {
                    const int k = lb[e];
                    const float a = h0 * lg0[k * ncmax + cl] + h1 * lg1[k * ncmax + cl];
                    const float d = (h0 * lg0[k * ncmax + cr] + h1 * lg1[k * ncmax + cr]) - a;
                    const float vl = fmaf(w1[e], d, a);            // label logit * log2e
                    if (MODE == 0) {
                        const float wy = wg[k];
                        ce = wy;
                        if (owned) {
                            num += wy * (((__log2f(se[e]) + U[e]) - vl) * UL_LN2);
                            den += wy;
                        }
                    } else if (owned) {
                        // log(softmax) as the reference computes it: -inf when the prob underflows
                        const float p = __builtin_amdgcn_exp2f(vl - U[e]) / se[e];
                        const float lp = logf(p);
                        const float om = 1.f - p;
                        const float mod = (gamma == 2.f) ? om * om : powf(om, gamma);
                        const float ay = k == 1 ? 1.f - alpha : alpha;
                        num -= ay * (mod * lp);
                        den += ay;
                    }
                }
                if (GRAD && e < cntx) {
                    f32x2 st;
                    st[0] = U[e];
                    st[1] = ce / se[e];
                    stats[(r * emax + (x0 - cx0) + e) * ncmax + cl] = st;
                }
            }
        }
    }
    const float tn = block_sum_256(num, red);
    const float td = block_sum_256(den, red);
    if (tid == 0) {
        const int64_t o = (((int64_t)n * h + py) * xsplit + xsid) * 2;
        part[o] = tn;
        part[o + 1] = td;
    }
    if (!GRAD) return;
    __syncthreads();

    // ---------------- phase 2 ----------------
    const int rmid = (rows + 1) >> 1;
    for (int item = tid; item < K * ncell * 2; item += UL_T) {
        const int rh = item / (K * ncell);
        const int kcx = item - rh * (K * ncell);
        const int k = kcx / ncell, cl = kcx - k * ncell;
        const int px = cell_lo + cl;
        const int cr = min(px + 1, w - 1) - cell_lo;
        const float t00 = lg0[k * ncmax + cl], t01 = lg0[k * ncmax + cr];
        const float t10 = lg1[k * ncmax + cl], t11 = lg1[k * ncmax + cr];
        const float wk = wg[k];
        const int x0 = xs[cl], nx = xs[cl + 1] - x0;
        const float* wrow = w1t + (x0 - xbase);
        float L0 = 0.f, L1 = 0.f, R0 = 0.f, R1 = 0.f;
        const int rb = rh ? rmid : 0, re = rh ? rows : rmid;
        for (int r = rb; r < re; ++r) {
            const int qy = ys + r;
            const float h1 = sy * (float)qy - (float)py, h0 = 1.f - h1;
            const float a = h0 * t00 + h1 * t10;
            const float d = (h0 * t01 + h1 * t11) - a;
            float gS = 0.f, gB = 0.f;
            const f32x2* strow = stats + (r * emax) * ncmax + cl;
            const unsigned char* lrow = lab8 + r * xw_max + (x0 - xbase);
#pragma unroll 4
            for (int e = 0; e < nx; ++e) {
                const float w1 = wrow[e];
                const f32x2 st = strow[e * ncmax];
                float g = __builtin_amdgcn_exp2f(fmaf(w1, d, a) - st[0]) * st[1];
                g -= ((int)lrow[e] == k) ? wk : 0.f;
                gS += g;
                gB += g * w1;
            }
            const float gA = gS - gB;                   // sum of g*(1-w1)
            L0 += h0 * gA; L1 += h1 * gA;
            R0 += h0 * gB; R1 += h1 * gB;
        }
        const int o = rh * kc + k * ncmax + cl;
        gl0[o] = L0; gl1[o] = L1; gr0[o] = R0; gr1[o] = R1;
    }
    __syncthreads();
    {
        float* da = dlog_a + (int64_t)n * K * h * w;
        float* db = dlog_b + (int64_t)n * K * h * w;
        const int nown = c1 - c0;
        for (int idx = tid; idx < K * nown; idx += UL_T) {
            const int k = idx / nown, x = c0 + idx - k * nown;
            const int o = k * ncmax + (x - cell_lo);
            float g0 = gl0[o] + gl0[kc + o], g1 = gl1[o] + gl1[kc + o];
            if (x >= 1) { g0 += gr0[o - 1] + gr0[kc + o - 1]; g1 += gr1[o - 1] + gr1[kc + o - 1]; }
            if (x == w - 1) { g0 += gr0[o] + gr0[kc + o]; g1 += gr1[o] + gr1[kc + o]; }
            // undo nothing: g is d(loss)/d(logit) already (the log2e factors cancel: p is a probability)
            if (py1 == py) {
                da[((int64_t)k * h + py) * w + x] = g0 + g1;
            } else {
                da[((int64_t)k * h + py) * w + x] = g0;
                db[((int64_t)k * h + py1) * w + x] = g1;
            }
            if (py == 0) db[((int64_t)k * h) * w + x] = 0.f;
        }
    }
}

// loss_k = mean_n num_n/den_n ; loss_p = sum num / sum den ; scale[n] = 1/(N*den_n)
// One wave per episode (lanes over the episode's partial sums), episodes round-robin over the waves.
__global__ __launch_bounds__(256) void loss_finalize_kernel(const float* __restrict__ part_k,
                                                           const float* __restrict__ part_p, int N, int NB,
                                                           int parts_k, int parts_p, int pooled,
                                                           float* __restrict__ out, float* __restrict__ per_episode,
                                                           float* __restrict__ scale) {
    __shared__ float red[4];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (part_k && !pooled) {
        // trainer/utils.py:270-277: mean over episodes of per-episode weighted means
        float ls = 0.f;
        for (int n = wave; n < N; n += 4) {
            float nu = 0.f, de = 0.f;
            for (int y = lane; y < parts_k; y += 64) {
                nu += part_k[((int64_t)n * parts_k + y) * 2];
                de += part_k[((int64_t)n * parts_k + y) * 2 + 1];
            }
            nu = wave_sum(nu);
            de = wave_sum(de);
            const float l = nu / de;
            if (lane == 0) {
                if (per_episode) per_episode[n] = l;
                if (scale) scale[n] = 1.f / ((float)N * de);
                ls += l;
            }
        }
        ls = block_sum_256(ls, red);
        if (tid == 0) out[0] = ls / (float)N;
    }
    if (part_k && pooled) {
        // loss/losses.py:39-44 on a batch: one weighted mean over every pixel of every sample
        float nu = 0.f, de = 0.f;
        for (int i = tid; i < N * parts_k; i += 256) {
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
        for (int i = tid; i < NB * parts_p; i += 256) {
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

template <typename KernelT>
static int set_lds(KernelT kern, size_t lds) {
    return denet_ensure_dynamic_lds(reinterpret_cast<const void*>(kern), lds);
}

// Launch geometry of upsample_loss_kernel, from the exact index rule (same float ops as the device).
struct LossGeom {
    int xsplit, rows_max, emax, xw_max, threads;
    size_t lds;
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

static bool loss_geom(int K, int h, int w, int H, int W, bool grad, LossGeom* g) {
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
        size_t off = (size_t)(2 + (grad ? 8 : 0)) * K * ncmax + K + 8 + xw_max;
        off = (off + ncmax + 2 + 1) & ~(size_t)1;
        off += grad ? (size_t)2 * rows_max * emax * ncmax : 0;
        size_t lds = off * 4 + (size_t)rows_max * xw_max;
        lds = (lds + 15) & ~(size_t)15;
        if (lds <= 64 * 1024 || (cw == 1 && lds <= 150 * 1024)) {
            g->xsplit = xsplit; g->rows_max = rows_max; g->emax = emax; g->xw_max = xw_max; g->threads = UL_T;
            g->lds = lds;
            found = true;
            break;
        }
    }
    return found;
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
#define UL_GO(M, G)                                                                                          \
    {                                                                                                        \
        int rc = set_lds(upsample_loss_kernel<M, G>, g.lds);                                                 \
        if (rc) return rc;                                                                                   \
        hipLaunchKernelGGL((upsample_loss_kernel<M, G>), grid, block, g.lds, st, logits, Y, wgt, (int)K,     \
                           (int)h, (int)w, (int)H, (int)W, g.xsplit, g.rows_max, g.emax, g.xw_max, alpha,    \
                           gamma, part, da, db);                                                             \
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
    unsigned gx = (unsigned)((H * W + 255) / 256);
    if (gx > 64) gx = 64;
    hipLaunchKernelGGL(upsample_bilinear_kernel, dim3(gx, (unsigned)NC), dim3(256), 0, (hipStream_t)stream, x,
                       (int)h, (int)w, (int)H, (int)W, out);
    DENET_CHECK_LAUNCH();
    return DENET_OK;
}
