// Device bodies of K4/K4b/K5 (fused bilinear up-sampling + losses) and of the loss reduction,
// shared by the stand-alone kernels (upsample_loss.hip) and the fused step kernels (head_step.hip).
#pragma once
#include "denet_common.hpp"

#define UL_THREADS 256
#define UL_CH 4
#define UL_IGNORE 255

// smallest q in [0, out_size] with min(int(scale*q), in_size-1) >= target: the first destination index whose
// upper bilinear tap is `target` or beyond.  The predicate is monotone in q, so a closed-form candidate plus
// a fix-up with the exact predicate (the float ops ATen uses) gives the same answer as a search.
__host__ __device__ __forceinline__ int first_dst_ge(int target, float scale, int in_size, int out_size) {
    if (target <= 0) return 0;
    if (target > in_size - 1 || !(scale > 0.f)) return out_size;
#if defined(__HIP_DEVICE_COMPILE__)
    const float qf = ceilf((float)target * __builtin_amdgcn_rcpf(scale));     // candidate only: fixed up below
#else
    const float qf = ceilf((float)target / scale);
#endif
    int q = qf < (float)out_size ? (int)qf : out_size;
    // the candidate is the answer unless target/scale sits within a rounding of an integer: verify it with the
    // exact predicate (two evaluations) and search only if that fails
    const bool low_ok = q == 0 || (int)(scale * (float)(q - 1)) < target;
    const bool high_ok = q == out_size || (int)(scale * (float)q) >= target;
    if (!(low_ok && high_ok)) {
        int lo = 0, hi = out_size;
        while (lo < hi) {
            const int mid = (lo + hi) >> 1;
            if ((int)(scale * (float)mid) >= target) hi = mid; else lo = mid + 1;
        }
        q = lo;
    }
    return q;
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

// Weighted CE (loss/losses.py:19-44 with per-episode class weights, trainer/utils.py:270-277), exact form.
// (The binary focal loss, loss/losses.py:47-79, has a body of its own: focal_body, dev_wce_chain.hpp.)
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

struct LossArgs {
    const float* logits;
    const int64_t* Y;
    const float* wgt;
    int K, h, w, H, W, xsplit, rows_max, emax, xw_max;
    float alpha, gamma;
    float *part, *dlog_a, *dlog_b;
    int kchunk;          // chain kernels (dev_wce_chain.hpp): classes per table chunk
    float sy, sx;        // bilinear_scale(h, H), bilinear_scale(w, W), computed once on the host
};

// Body for block (episode n, bx = py * xsplit + xsid); 256 threads; sm: the geometry's LDS bytes.
// This is the EXACT form: one exponential per (pixel, class) in either phase, the lerp weights exactly as
// ATen rounds them.  The weighted CE normally runs on the chain kernels of dev_wce_chain.hpp, which call
// this body for blocks whose logits leave their validity range and for geometries they do not cover
// (cells of other widths than 8 or 9 label columns).
// RS: row groups of the gradient pass (2 gives K = 21 enough (class, cell, group) items for 256 threads).
template <bool GRAD, int RS = 2>
__device__ __forceinline__ void upsample_loss_body(const LossArgs& g, int n, int bx, float* sm) {
    const float* __restrict__ logits = g.logits;
    const int64_t* __restrict__ Y = g.Y;
    const float* __restrict__ wgt = g.wgt;
    const int K = g.K, h = g.h, w = g.w, H = g.H, W = g.W, xsplit = g.xsplit, rows_max = g.rows_max, emax = g.emax,
              xw_max = g.xw_max;
    float* __restrict__ part = g.part;
    float* __restrict__ dlog_a = g.dlog_a;
    float* __restrict__ dlog_b = g.dlog_b;
    const int py = bx / xsplit, xsid = bx - py * xsplit;
    const int py1 = py + 1 < h ? py + 1 : h - 1;
    const int tid = threadIdx.x;
    const int cw = (w + xsplit - 1) / xsplit;
    const int c0 = xsid * cw, c1 = min(w, c0 + cw);
    const int cell_lo = (GRAD && c0 > 0) ? c0 - 1 : c0;
    const int ncell = c1 - cell_lo;                       // cells processed
    const int ncmax = cw + (GRAD ? 2 : 1);            // LDS stride: cells + right tap (+ left halo cell)
    const int col_hi = min(c1, w - 1);                    // last logits column needed
    const int ncol = col_hi - cell_lo + 1;

    const int kc = K * ncmax;
    float* lg0 = sm;                                      // [K][ncmax]  row py * log2e, columns cell_lo..
    float* lg1 = lg0 + kc;                                // [K][ncmax]  row py1 * log2e
    float* gl0 = lg1 + kc;                                // GRAD: [2 row halves][K][ncmax] left tap, row-tap 0
    constexpr int rsplit = GRAD ? RS : 0;                 // row groups of the gradient pass (2, or 1 for many classes)
    float* gl1 = gl0 + rsplit * kc;
    float* gr0 = gl1 + rsplit * kc;
    float* gr1 = gr0 + rsplit * kc;
    float* wg = gr1 + rsplit * kc;                        // [K]
    float* red = wg + K;                                  // [8]
    float* w1t = red + 8;                                 // [xw_max] horizontal lerp weight of each label column
    int off = (2 + 4 * rsplit) * kc + K + 8 + xw_max;
    int* xs = reinterpret_cast<int*>(sm + off);           // [ncmax+1] first pixel of each cell
    off = (off + ncmax + 2 + 1) & ~1;
    f32x2* stats = reinterpret_cast<f32x2*>(sm + off);    // [rows_max][emax][ncmax] (shift, weight/sumexp), GRAD
    off += GRAD ? 2 * rows_max * emax * ncmax : 0;
    unsigned char* lab8 = reinterpret_cast<unsigned char*>(sm + off);   // [rows_max][xw_max]

    const float sy = g.sy, sx = g.sx;
    const int ys = first_dst_ge(py, sy, h, H), ye = first_dst_ge(py + 1, sy, h, H);
    const int rows = ye - ys;

    STAMP(0);
    const float* src = logits + (int64_t)n * K * h * w;
    {
        // thread <-> (class, column) by shifts: the column count rounded up to a power of two (an integer
        // division per element cost more than the loads it addressed)
        const int xsh = 32 - __builtin_clz((unsigned)max(ncol - 1, 1));     // ncol <= 1 << xsh
        const int x = tid & ((1 << xsh) - 1), kstep = UL_T >> xsh;
        const int xc = min(x, ncol - 1);
        const float* s0 = src + (int64_t)py * w + cell_lo + xc;
        const float* s1 = src + (int64_t)py1 * w + cell_lo + xc;
        const int64_t hw = (int64_t)h * w;
        for (int k = tid >> xsh; k < K; k += kstep) {
            const float v0 = s0[(int64_t)k * hw], v1 = s1[(int64_t)k * hw];
            if (x < ncol) {
                lg0[k * ncmax + x] = v0 * UL_LOG2E;
                lg1[k * ncmax + x] = v1 * UL_LOG2E;
            }
        }
    }
    for (int k = tid; k < K; k += UL_T) wg[k] = wgt ? wgt[(int64_t)n * K + k] : 1.f;
    for (int c = tid; c <= ncell; c += UL_T) xs[c] = first_dst_ge(cell_lo + c, sx, w, W);
    __syncthreads();
    STAMP(1);
    const int xbase = xs[0], xw = xs[ncell] - xbase;      // label columns needed
    {
        const int64_t* Yn = Y + (int64_t)n * H * W + (int64_t)ys * W + xbase;
        if (xw <= UL_T) {
            // thread <-> (row, column) by shifts, the loads of up to four rows in flight together
            const int lsh = 32 - __builtin_clz((unsigned)max(xw - 1, 1));
            const int x = tid & ((1 << lsh) - 1), rstep = UL_T >> lsh;
            const int64_t* col = Yn + min(x, xw - 1);
            for (int r0 = tid >> lsh; r0 < rows; r0 += 4 * rstep) {
                int64_t v[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) v[u] = col[(int64_t)min(r0 + u * rstep, rows - 1) * W];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int r = r0 + u * rstep;
                    if (r < rows && x < xw)
                        lab8[r * xw_max + x] = (v[u] >= 0 && v[u] < K) ? (unsigned char)v[u] : (unsigned char)UL_IGNORE;
                }
            }
        } else {
#pragma unroll 8
            for (int i = tid; i < rows * xw; i += UL_T) {
                const int r = i / xw, x = i - r * xw;
                const int64_t v = Yn[(int64_t)r * W + x];
                lab8[r * xw_max + x] = (v >= 0 && v < K) ? (unsigned char)v : (unsigned char)UL_IGNORE;
            }
        }
        for (int x = tid; x < xw; x += UL_T) {
            int i0;
            float f;
            bilinear_tap(xbase + x, sx, w, i0, f);
            w1t[x] = f;
        }
    }
    __syncthreads();
    STAMP(2);

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
            bool under = false;
#pragma unroll 3
            for (int k = 0; k < K; ++k) {
                const float a = h0 * lg0[k * ncmax + cl] + h1 * lg1[k * ncmax + cl];
                const float d = (h0 * lg0[k * ncmax + cr] + h1 * lg1[k * ncmax + cr]) - a;
#pragma unroll
                for (int e = 0; e < UL_CH; ++e) se[e] += __builtin_amdgcn_exp2f(fmaf(w1[e], d, a) - U[e]);
            }
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
                    const float wy = wg[k];
                    ce = wy;
                    if (owned) {
                        num += wy * (((__log2f(se[e]) + U[e]) - vl) * UL_LN2);
                        den += wy;
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
    STAMP(3);
    const float tn = block_sum_256(num, red);
    const float td = block_sum_256(den, red);
    if (tid == 0) {
        const int64_t o = (((int64_t)n * h + py) * xsplit + xsid) * 2;
        part[o] = tn;
        part[o + 1] = td;
    }
    if (!GRAD) return;
    __syncthreads();
    STAMP(4);

    // ---------------- phase 2 ----------------
    const int rmid = rsplit == 2 ? (rows + 1) >> 1 : rows;
    for (int item = tid; item < K * ncell * rsplit; item += UL_T) {
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
    STAMP(5);
    __syncthreads();
    {
        float* da = dlog_a + (int64_t)n * K * h * w;
        float* db = dlog_b + (int64_t)n * K * h * w;
        const int nown = c1 - c0;
        for (int idx = tid; idx < K * nown; idx += UL_T) {
            const int k = idx / nown, x = c0 + idx - k * nown;
            const int o = k * ncmax + (x - cell_lo);
            // second row group (absent with rsplit == 1: its sums count as zero)
            auto two = [&](const float* arr, int i) { return rsplit == 2 ? arr[i] + arr[kc + i] : arr[i]; };
            float g0 = two(gl0, o), g1 = two(gl1, o);
            if (x >= 1) { g0 += two(gr0, o - 1); g1 += two(gr1, o - 1); }
            if (x == w - 1) { g0 += two(gr0, o); g1 += two(gr1, o); }
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
    STAMP(6);
}

// loss_k = mean_n num_n/den_n ; loss_p = sum num / sum den ; scale[n] = 1/(N*den_n)
// One wave per episode (lanes over the episode's partial sums), episodes round-robin over the waves.
__device__ __forceinline__ void loss_finalize_body(const float* __restrict__ part_k,
                                                   const float* __restrict__ part_p, int N, int NB, int parts_k,
                                                   int parts_p, int pooled, float* __restrict__ out,
                                                   float* __restrict__ per_episode, float* __restrict__ scale,
                                                   float* red /*LDS, 8 floats*/) {
    // any block size that is a multiple of 64, up to 512
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, nt = blockDim.x, nwv = nt >> 6;
    if (part_k && !pooled) {
        // trainer/utils.py:270-277: mean over episodes of per-episode weighted means
        float ls = 0.f;
        for (int n = wave; n < N; n += nwv) {
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
                if (scale) scale[n] = de > 0.f ? 1.f / ((float)N * de) : 0.f;     // (all pixels ignored: zero gradient, as the reference)
                ls += l;
            }
        }
        ls = block_sum(ls, red);
        if (tid == 0) out[0] = ls / (float)N;
    }
    if (part_k && pooled) {
        // loss/losses.py:39-44 on a batch: one weighted mean over every pixel of every sample
        float nu = 0.f, de = 0.f;
        for (int i = tid; i < N * parts_k; i += nt) {
            nu += part_k[(int64_t)i * 2];
            de += part_k[(int64_t)i * 2 + 1];
        }
        nu = block_sum(nu, red);
        de = block_sum(de, red);
        if (tid == 0) out[0] = nu / de;
        if (scale)
            for (int n = tid; n < N; n += nt) scale[n] = de > 0.f ? 1.f / de : 0.f;
        if (per_episode)
            for (int n = tid; n < N; n += nt) per_episode[n] = nu / de;
    }
    if (part_p) {
        float nu = 0.f, de = 0.f;
        for (int i = tid; i < NB * parts_p; i += nt) {
            nu += part_p[(int64_t)i * 2];
            de += part_p[(int64_t)i * 2 + 1];
        }
        nu = block_sum(nu, red);
        de = block_sum(de, red);
        if (tid == 0) out[1] = nu / de;
    }
}

static __global__ __launch_bounds__(256) void loss_finalize_kernel(const float* __restrict__ part_k,
                                                           const float* __restrict__ part_p, int N, int NB,
                                                           int parts_k, int parts_p, int pooled,
                                                           float* __restrict__ out, float* __restrict__ per_episode,
                                                           float* __restrict__ scale) {
    __shared__ float red[8];
    loss_finalize_body(part_k, part_p, N, NB, parts_k, parts_p, pooled, out, per_episode, scale, red);
}

