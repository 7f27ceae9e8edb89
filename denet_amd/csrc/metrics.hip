// M1 / M2: confusion matrix (FullIOU.update) and binary IoU counts
// (BinaryIOU.update) on the device.  Integer counts only: exact and
// independent of summation order, so the matrices are bit-identical to
// np.bincount for the same predictions and across any number of GPUs.
//
// reference: utils/metrics.py:24-31 (BinaryIOU.update), :59-67 (FullIOU.update)
#include "denet_common.hpp"

#define MT_THREADS 256

// torch.argmax semantics: first maximal index; a NaN beats everything (first NaN wins)
__device__ __forceinline__ bool argmax_better(float v, float best) {
    return (v > best) || (v != v && best == best);
}

// ---- API-level: preds already up-sampled [N,K,HW] ----------------------------
// HBM-bound (K*HW*4 bytes per sample).  Each thread owns 4 consecutive pixels and walks the classes
// with 16-byte loads (rows of different classes are only element-aligned: HW is odd for 321x321),
// four classes in flight at a time; counts go to an LDS-privatised histogram, then to the int64 matrix.
typedef float f32x4_mu __attribute__((ext_vector_type(4), aligned(4)));

__global__ __launch_bounds__(MT_THREADS) void confmat_kernel(const float* __restrict__ preds,
                                                            const int64_t* __restrict__ yt, int K, int64_t HW,
                                                            int chunk, unsigned long long* __restrict__ mat) {
    extern __shared__ unsigned int hist[];   // [K*K]
    const int n = blockIdx.y;
    for (int i = threadIdx.x; i < K * K; i += MT_THREADS) hist[i] = 0u;
    __syncthreads();
    const int64_t q0 = (int64_t)blockIdx.x * chunk;
    const int64_t q1 = q0 + chunk < HW ? q0 + chunk : HW;
    const float* pn = preds + (int64_t)n * K * HW;
    const int64_t* tn = yt + (int64_t)n * HW;
    for (int64_t q = q0 + 4 * threadIdx.x; q < q1; q += 4 * MT_THREADS) {
        if (q + 4 <= HW) {
            int64_t t[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) t[e] = tn[q + e];
            f32x4_mu best = *reinterpret_cast<const f32x4_mu*>(pn + q);
            int bi[4] = {0, 0, 0, 0};
#pragma unroll 4
            for (int k = 1; k < K; ++k) {
                const f32x4_mu v = *reinterpret_cast<const f32x4_mu*>(pn + (int64_t)k * HW + q);
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    if (argmax_better(v[e], best[e])) { best[e] = v[e]; bi[e] = k; }
            }
#pragma unroll
            for (int e = 0; e < 4; ++e)
                if (t[e] >= 0 && t[e] < K) atomicAdd(&hist[(int)t[e] * K + bi[e]], 1u);
        } else {
            for (int64_t qq = q; qq < HW; ++qq) {            // last (< 4) pixels of the sample
                const int64_t t = tn[qq];
                float best = pn[qq];
                int bi = 0;
                for (int k = 1; k < K; ++k) {
                    const float v = pn[(int64_t)k * HW + qq];
                    if (argmax_better(v, best)) { best = v; bi = k; }
                }
                if (t >= 0 && t < K) atomicAdd(&hist[(int)t * K + bi], 1u);
            }
        }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < K * K; i += MT_THREADS)
        if (hist[i]) atomicAdd(&mat[i], (unsigned long long)hist[i]);
}

__global__ __launch_bounds__(MT_THREADS) void binary_counts_kernel(const float* __restrict__ preds2,
                                                                  const int64_t* __restrict__ yt, int64_t HW,
                                                                  int chunk, unsigned long long* __restrict__ counts) {
    __shared__ unsigned long long red[3][4];
    const int n = blockIdx.y;
    const int64_t q0 = (int64_t)blockIdx.x * chunk;
    const int64_t q1 = q0 + chunk < HW ? q0 + chunk : HW;
    const float* p0 = preds2 + (int64_t)n * 2 * HW;
    const float* p1 = p0 + HW;
    const int64_t* tn = yt + (int64_t)n * HW;
    long long st = 0;
    unsigned int inter = 0, uni = 0;
    for (int64_t q = q0 + threadIdx.x; q < q1; q += MT_THREADS) {
        const int64_t t = tn[q];
        const int pred = argmax_better(p1[q], p0[q]) ? 1 : 0;
        const int64_t s = t + pred;
        st += t;
        inter += (s == 2);
        uni += (s == 1 || s == 2);
    }
    // block reduce (integers)
    long long a = st;
    unsigned int b = inter, c = uni;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        a += __shfl_xor(a, off, 64);
        b += __shfl_xor(b, off, 64);
        c += __shfl_xor(c, off, 64);
    }
    const int wv = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) { red[0][wv] = (unsigned long long)a; red[1][wv] = b; red[2][wv] = c; }
    __syncthreads();
    if (threadIdx.x < 3) {
        const unsigned long long v = red[threadIdx.x][0] + red[threadIdx.x][1] + red[threadIdx.x][2] + red[threadIdx.x][3];
        if (v) atomicAdd(&counts[(int64_t)n * 3 + threadIdx.x], v);
    }
}

// ---- fused: from low-res logits, up-sampling on the fly -----------------------
// One block per (episode, low-res row band) as in upsample_loss.hip.  The
// up-sampled value uses ATen's operation order without FMA contraction so that
// the arg-max agrees with F.interpolate + argmax wherever fp32 rounding is the
// only difference.
__device__ __forceinline__ int first_dst_ge_m(int target, float scale, int in_size, int out_size) {
    int lo = 0, hi = out_size;
    while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        int i0 = (int)(scale * (float)mid);
        if (i0 > in_size - 1) i0 = in_size - 1;
        if (i0 >= target) hi = mid; else lo = mid + 1;
    }
    return lo;
}

template <bool BINARY>
__global__ __launch_bounds__(MT_THREADS) void fused_metric_kernel(const float* __restrict__ logits,
                                                                 const int64_t* __restrict__ yt, int K, int h, int w,
                                                                 int H, int W, unsigned long long* __restrict__ out) {
    extern __shared__ __attribute__((aligned(16))) float smf[];
    const int n = blockIdx.y, py = blockIdx.x;
    const int py1 = py + 1 < h ? py + 1 : h - 1;
    const int tid = threadIdx.x;
    const int kw = K * w;
    float* lg0 = smf;
    float* lg1 = lg0 + kw;
    unsigned int* hist = reinterpret_cast<unsigned int*>(lg1 + kw);     // [K*K] (confmat) or [3*4] scratch
    const float sy = bilinear_scale(h, H), sx = bilinear_scale(w, W);
    const int ys = first_dst_ge_m(py, sy, h, H), ye = first_dst_ge_m(py + 1, sy, h, H);
    const float* src = logits + (int64_t)n * K * h * w;
#pragma unroll 4
    for (int idx = tid; idx < kw; idx += MT_THREADS) {
        const int k = idx / w, x = idx - k * w;
        lg0[idx] = src[((int64_t)k * h + py) * w + x];
        lg1[idx] = src[((int64_t)k * h + py1) * w + x];
    }
    if (!BINARY)
        for (int i = tid; i < K * K; i += MT_THREADS) hist[i] = 0u;
    __syncthreads();
    const int64_t* tn = yt + (int64_t)n * H * W;
    long long st = 0;
    unsigned int inter = 0, uni = 0;
    const int npix = (ye - ys) * W;
#pragma unroll 2
    for (int i = tid; i < npix; i += MT_THREADS) {
        const int r = i / W, qx = i - r * W;
        const int qy = ys + r;
        const int64_t t = tn[(int64_t)qy * W + qx];
        if (!BINARY && (t < 0 || t >= K)) continue;
        const float h1 = sy * (float)qy - (float)py, h0 = 1.f - h1;
        int x0;
        float w1;
        bilinear_tap(qx, sx, w, x0, w1);
        const float w0 = 1.f - w1;
        const int x1 = x0 + (x0 < w - 1);
        float best = 0.f;
        int bi = 0;
        for (int k = 0; k < K; ++k) {
            const float top = __fadd_rn(__fmul_rn(w0, lg0[k * w + x0]), __fmul_rn(w1, lg0[k * w + x1]));
            const float bot = __fadd_rn(__fmul_rn(w0, lg1[k * w + x0]), __fmul_rn(w1, lg1[k * w + x1]));
            const float v = __fadd_rn(__fmul_rn(h0, top), __fmul_rn(h1, bot));
            if (k == 0 || argmax_better(v, best)) { best = v; bi = k; }
        }
        if (BINARY) {
            const int64_t s = t + bi;
            st += t;
            inter += (s == 2);
            uni += (s == 1 || s == 2);
        } else {
            atomicAdd(&hist[(int)t * K + bi], 1u);
        }
    }
    if (BINARY) {
        long long a = st;
        unsigned int b = inter, c = uni;
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            a += __shfl_xor(a, off, 64);
            b += __shfl_xor(b, off, 64);
            c += __shfl_xor(c, off, 64);
        }
        unsigned long long* red = reinterpret_cast<unsigned long long*>(hist);   // 8-byte aligned: kw*8 bytes before
        __syncthreads();
        const int wv = tid >> 6;
        if ((tid & 63) == 0) { red[wv] = (unsigned long long)a; red[4 + wv] = b; red[8 + wv] = c; }
        __syncthreads();
        if (tid < 3) {
            const unsigned long long v = red[tid * 4] + red[tid * 4 + 1] + red[tid * 4 + 2] + red[tid * 4 + 3];
            if (v) atomicAdd(&out[(int64_t)n * 3 + tid], v);
        }
    } else {
        __syncthreads();
        for (int i = tid; i < K * K; i += MT_THREADS)
            if (hist[i]) atomicAdd(&out[i], (unsigned long long)hist[i]);
    }
}

// ---------------------------------------------------------------------------
// C ABI
// ---------------------------------------------------------------------------
static inline bool pos(int64_t v) { return v > 0 && v < ((int64_t)1 << 30); }

static int pick_chunk(int64_t HW, int64_t N, unsigned* gx) {
    // ~4096 blocks at most; each block at least 1024 pixels (4 per thread), in multiples of 1024
    int64_t per = (HW * N + 4095) / 4096;
    if (per < 1024) per = 1024;
    per = (per + 1023) / 1024 * 1024;
    *gx = (unsigned)((HW + per - 1) / per);
    return (int)per;
}

extern "C" int denet_confmat(const float* preds, const int64_t* label_true, int64_t N, int64_t K, int64_t HW,
                             int64_t* mat, void* stream) {
    if (!preds || !label_true || !mat) return DENET_E_BADARG;
    if (!pos(N) || !pos(K) || !pos(HW)) return DENET_E_BADARG;
    if (K > 128 || N > 65535) return DENET_E_UNSUPPORTED;
    unsigned gx;
    const int chunk = pick_chunk(HW, N, &gx);
    const size_t lds = (size_t)K * K * 4;
    if (denet_ensure_dynamic_lds(reinterpret_cast<const void*>(confmat_kernel), lds)) return DENET_E_LAUNCH;
    hipLaunchKernelGGL(confmat_kernel, dim3(gx, (unsigned)N), dim3(MT_THREADS), lds, (hipStream_t)stream, preds,
                       label_true, (int)K, HW, chunk, (unsigned long long*)mat);
    DENET_CHECK_LAUNCH();
    return DENET_OK;
}

extern "C" int denet_binary_iou_counts(const float* preds2, const int64_t* label_true, int64_t N, int64_t HW,
                                       int64_t* counts, void* stream) {
    if (!preds2 || !label_true || !counts) return DENET_E_BADARG;
    if (!pos(N) || !pos(HW)) return DENET_E_BADARG;
    if (N > 65535) return DENET_E_UNSUPPORTED;
    unsigned gx;
    const int chunk = pick_chunk(HW, N, &gx);
    hipLaunchKernelGGL(binary_counts_kernel, dim3(gx, (unsigned)N), dim3(MT_THREADS), 0, (hipStream_t)stream, preds2,
                       label_true, HW, chunk, (unsigned long long*)counts);
    DENET_CHECK_LAUNCH();
    return DENET_OK;
}

extern "C" int denet_confmat_fused(const float* logits, const int64_t* label_true, int64_t N, int64_t K, int64_t h,
                                   int64_t w, int64_t H, int64_t W, int64_t* mat, void* stream) {
    if (!logits || !label_true || !mat) return DENET_E_BADARG;
    if (!pos(N) || !pos(K) || !pos(h) || !pos(w) || !pos(H) || !pos(W) || !pos(H * W)) return DENET_E_BADARG;
    if (K > 128 || N > 65535 || H < h || W < w) return DENET_E_UNSUPPORTED;     // up-sampling only
    const size_t lds = (size_t)2 * K * w * 4 + (size_t)K * K * 4 + 128;
    if (lds > 160 * 1024) return DENET_E_UNSUPPORTED;
    if (denet_ensure_dynamic_lds(reinterpret_cast<const void*>(fused_metric_kernel<false>), lds))
        return DENET_E_LAUNCH;
    hipLaunchKernelGGL(fused_metric_kernel<false>, dim3((unsigned)h, (unsigned)N), dim3(MT_THREADS), lds,
                       (hipStream_t)stream, logits, label_true, (int)K, (int)h, (int)w, (int)H, (int)W,
                       (unsigned long long*)mat);
    DENET_CHECK_LAUNCH();
    return DENET_OK;
}

extern "C" int denet_binary_iou_counts_fused(const float* logits2, const int64_t* label_true, int64_t N, int64_t h,
                                             int64_t w, int64_t H, int64_t W, int64_t* counts, void* stream) {
    if (!logits2 || !label_true || !counts) return DENET_E_BADARG;
    if (!pos(N) || !pos(h) || !pos(w) || !pos(H) || !pos(W) || !pos(H * W)) return DENET_E_BADARG;
    if (N > 65535 || H < h || W < w) return DENET_E_UNSUPPORTED;                 // up-sampling only
    const size_t lds = (size_t)2 * 2 * w * 4 + 128;
    hipLaunchKernelGGL(fused_metric_kernel<true>, dim3((unsigned)h, (unsigned)N), dim3(MT_THREADS), lds,
                       (hipStream_t)stream, logits2, label_true, 2, (int)h, (int)w, (int)H, (int)W,
                       (unsigned long long*)counts);
    DENET_CHECK_LAUNCH();
    return DENET_OK;
}
