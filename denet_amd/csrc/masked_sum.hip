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
__global__ __launch_bounds__(256) void mask_resize_kernel(const int64_t* __restrict__ Ys, int H, int W,
                                                          int h, int w, float* __restrict__ mask,
                                                          float* __restrict__ cnt) {
    const int b = blockIdx.x;
    const int P = h * w;
    const float sy = nearest_scale(H, h), sx = nearest_scale(W, w);
    const int64_t* src = Ys + (int64_t)b * H * W;
    float acc = 0.f;
#pragma unroll 8
    for (int p = threadIdx.x; p < P; p += 256) {
        int y = p / w, x = p - y * w;
        int yy = nearest_src(y, sy, H), xx = nearest_src(x, sx, W);
        float m = (float)src[(int64_t)yy * W + xx];
        mask[(int64_t)b * P + p] = m;
        acc += m;
    }
    __shared__ float red[4];
    acc = wave_sum(acc);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) cnt[b] = (red[0] + red[1]) + (red[2] + red[3]);
}

// ---------------------------------------------------------------------------
// K1: S[b,c] = sum_p mask[b,p] * Fs[b,c,p]
// ---------------------------------------------------------------------------
template <typename T> struct Vec16;
template <> struct Vec16<float> {
    static constexpr int N = 4;
    typedef f32x4 type;
    static __device__ __forceinline__ float dot(const type& v, const float* m) {
        return v[0] * m[0] + v[1] * m[1] + v[2] * m[2] + v[3] * m[3];
    }
};
template <> struct Vec16<unsigned short> {
    static constexpr int N = 8;
    typedef u16x8 type;
    static __device__ __forceinline__ float dot(const type& v, const float* m) {
        float a = 0.f;
#pragma unroll
        for (int e = 0; e < 8; ++e) a += bf16_to_f32(v[e]) * m[e];
        return a;
    }
};

template <typename T>
__global__ __launch_bounds__(256) void masked_sum_fwd_kernel(const T* __restrict__ Fs,
                                                            const float* __restrict__ mask,
                                                            float* __restrict__ S, int C, int P,
                                                            int rows_per_block) {
    extern __shared__ __attribute__((aligned(16))) float sm[];   // [P]
    const int b = blockIdx.y;
    const int c0 = blockIdx.x * rows_per_block;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const float* mrow = mask + (int64_t)b * P;
    // batched: 8 independent loads in flight per thread (an un-unrolled copy loop pays the memory
    // latency once per iteration)
#pragma unroll 8
    for (int p = threadIdx.x; p < P; p += 256) sm[p] = mrow[p];
    __syncthreads();

    constexpr int V = Vec16<T>::N;
    typedef typename Vec16<T>::type vec_t;
    for (int r = wave; r < rows_per_block; r += 4) {
        const int c = c0 + r;
        if (c >= C) break;
        const T* row = Fs + ((int64_t)b * C + c) * P;
        const int mis = (int)(((uintptr_t)row / sizeof(T)) % V);
        int head = mis ? V - mis : 0;
        if (head > P) head = P;
        float acc = 0.f;
        if (lane < head) acc += FeatIO<T>::load(row + lane) * sm[lane];
        const int nvec = (P - head) / V;
        const vec_t* rv = reinterpret_cast<const vec_t*>(row + head);
#pragma unroll 4
        for (int i = lane; i < nvec; i += 64) {
            vec_t v = rv[i];
            acc += Vec16<T>::dot(v, sm + head + i * V);
        }
        const int t0 = head + nvec * V;
        if (t0 + lane < P) acc += FeatIO<T>::load(row + t0 + lane) * sm[t0 + lane];
        acc = wave_sum(acc);
        if (lane == 0) S[(int64_t)b * C + c] = acc;
    }
}

// ---------------------------------------------------------------------------
// K1b: dFs[b,c,p] = mask[b,p] * coef[b,c]
// ---------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void masked_sum_bwd_kernel(const float* __restrict__ mask,
                                                            const float* __restrict__ coef,
                                                            T* __restrict__ dFs, int C, int P,
                                                            int rows_per_block) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    const int b = blockIdx.y;
    const int c0 = blockIdx.x * rows_per_block;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const float* mrow = mask + (int64_t)b * P;
    // batched: 8 independent loads in flight per thread (an un-unrolled copy loop pays the memory
    // latency once per iteration)
#pragma unroll 8
    for (int p = threadIdx.x; p < P; p += 256) sm[p] = mrow[p];
    __syncthreads();

    constexpr int V = Vec16<T>::N;
    typedef typename Vec16<T>::type vec_t;
    for (int r = wave; r < rows_per_block; r += 4) {
        const int c = c0 + r;
        if (c >= C) break;
        const float k = coef[(int64_t)b * C + c];
        T* row = dFs + ((int64_t)b * C + c) * P;
        const int mis = (int)(((uintptr_t)row / sizeof(T)) % V);
        int head = mis ? V - mis : 0;
        if (head > P) head = P;
        if (lane < head) FeatIO<T>::store(row + lane, k * sm[lane]);
        const int nvec = (P - head) / V;
        vec_t* rv = reinterpret_cast<vec_t*>(row + head);
#pragma unroll 4
        for (int i = lane; i < nvec; i += 64) {
            const float* m = sm + head + i * V;
            vec_t v;
            if constexpr (V == 4) {
                v[0] = k * m[0]; v[1] = k * m[1]; v[2] = k * m[2]; v[3] = k * m[3];
            } else {
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] = f32_to_bf16(k * m[e]);
            }
            rv[i] = v;
        }
        const int t0 = head + nvec * V;
        if (t0 + lane < P) FeatIO<T>::store(row + t0 + lane, k * sm[t0 + lane]);
    }
}

// ---------------------------------------------------------------------------
// C ABI
// ---------------------------------------------------------------------------
static inline bool fits_int(int64_t v) { return v > 0 && v < (int64_t)1 << 30; }

extern "C" int denet_mask_resize(const int64_t* Ys, int64_t B, int64_t H, int64_t W, int64_t h, int64_t w,
                                 float* mask, float* cnt, void* stream) {
    if (!Ys || !mask || !cnt) return DENET_E_BADARG;
    if (!fits_int(B) || !fits_int(H) || !fits_int(W) || !fits_int(h) || !fits_int(w) || !fits_int(h * w))
        return DENET_E_BADARG;
    hipLaunchKernelGGL(mask_resize_kernel, dim3((unsigned)B), dim3(256), 0, (hipStream_t)stream, Ys, (int)H,
                       (int)W, (int)h, (int)w, mask, cnt);
    DENET_CHECK_LAUNCH();
    return DENET_OK;
}

static inline int pick_rows_per_block(int64_t C) { return C >= 16 ? 16 : (int)C; }

extern "C" int denet_masked_sum_fwd(const void* Fs, int feat_dtype, const float* mask, int64_t B, int64_t C,
                                    int64_t P, float* S, void* stream) {
    if (!Fs || !mask || !S) return DENET_E_BADARG;
    if (!fits_int(B) || !fits_int(C) || !fits_int(P)) return DENET_E_BADARG;
    if (P * 4 > 64 * 1024 || B > 65535) return DENET_E_UNSUPPORTED;
    const int rpb = pick_rows_per_block(C);
    dim3 grid((unsigned)((C + rpb - 1) / rpb), (unsigned)B);
    size_t lds = (size_t)P * 4;
    if (feat_dtype == DENET_F32)
        hipLaunchKernelGGL(masked_sum_fwd_kernel<float>, grid, dim3(256), lds, (hipStream_t)stream,
                           (const float*)Fs, mask, S, (int)C, (int)P, rpb);
    else if (feat_dtype == DENET_BF16)
        hipLaunchKernelGGL(masked_sum_fwd_kernel<unsigned short>, grid, dim3(256), lds, (hipStream_t)stream,
                           (const unsigned short*)Fs, mask, S, (int)C, (int)P, rpb);
    else
        return DENET_E_UNSUPPORTED;
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
    size_t lds = (size_t)P * 4;
    if (feat_dtype == DENET_F32)
        hipLaunchKernelGGL(masked_sum_bwd_kernel<float>, grid, dim3(256), lds, (hipStream_t)stream, mask, coef,
                           (float*)dFs, (int)C, (int)P, rpb);
    else if (feat_dtype == DENET_BF16)
        hipLaunchKernelGGL(masked_sum_bwd_kernel<unsigned short>, grid, dim3(256), lds, (hipStream_t)stream, mask,
                           coef, (unsigned short*)dFs, (int)C, (int)P, rpb);
    else
        return DENET_E_UNSUPPORTED;
    DENET_CHECK_LAUNCH();
    return DENET_OK;
}
