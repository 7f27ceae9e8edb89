// Shared device helpers for the DENet head kernels (gfx950 / CDNA4, wave64).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/denet_hip.h"

#define DENET_WAVE 64

// ---- launcher decisions shared between the translation units (reported by denet_head_step_plan) ----
struct PoolPlan { bool two_launch, long_rows, nt_loads; };
PoolPlan denet_pool_plan(int feat_dtype, int64_t B, int64_t C, int64_t P, bool fused);          // masked_sum.hip
int denet_classify_fwd_layout(int feat_dtype, int64_t N, int64_t K, int64_t P, int64_t* groups_out);            // classify.hip
struct LossPlan { int chain_cell, chunk, xsplit, rsplit; };
bool denet_loss_plan(int64_t K, int64_t h, int64_t w, int64_t H, int64_t W, bool grad, LossPlan* out);  // upsample_loss.hip

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned short u16x4 __attribute__((ext_vector_type(4)));
typedef unsigned short u16x8 __attribute__((ext_vector_type(8)));

#define DENET_CHECK_LAUNCH()                                   \
    do {                                                       \
        hipError_t e__ = hipGetLastError();                    \
        if (e__ != hipSuccess) return DENET_E_LAUNCH;          \
    } while (0)

// ---- index rules (must reproduce ATen bit for bit) -------------------------
// F.interpolate(mode='nearest'): src = min(int(floorf(dst * (float(in)/out))), in-1)
__device__ __forceinline__ int nearest_src(int dst, float scale, int in_size) {
    int s = (int)floorf((float)dst * scale);
    return s < in_size - 1 ? s : in_size - 1;
}
__host__ __device__ __forceinline__ float nearest_scale(int in_size, int out_size) {
    return (float)in_size / (float)out_size;
}
// bilinear, align_corners=True: scale = float(in-1)/(out-1) (0 if out<=1);
// src = scale*dst; i0 = int(src); i1 = i0 + (i0 < in-1); frac = src - i0
__host__ __device__ __forceinline__ float bilinear_scale(int in_size, int out_size) {
    return out_size > 1 ? (float)(in_size - 1) / (float)(out_size - 1) : 0.0f;
}
__device__ __forceinline__ void bilinear_tap(int dst, float scale, int in_size, int& i0, float& frac) {
    float src = scale * (float)dst;
    i0 = (int)src;
    if (i0 > in_size - 1) i0 = in_size - 1;
    frac = src - (float)i0;
}

// ---- wave / block reductions -------------------------------------------------
// DPP forms: the cross-lane operand is read by the VALU itself (no LDS-path ds_bpermute, which
// costs ~100+ cycles of dependent latency per step).
template <int CTRL>
__device__ __forceinline__ float dpp_f32(float v) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xF, 0xF, true));
}
// sum over the 16 lanes of a DPP row (lanes sharing lane>>4); every lane of the row gets the total
__device__ __forceinline__ float row16_sum(float v) {
    v += dpp_f32<0xB1>(v);     // quad_perm [1,0,3,2]  (xor 1)
    v += dpp_f32<0x4E>(v);     // quad_perm [2,3,0,1]  (xor 2)
    v += dpp_f32<0x141>(v);    // row_half_mirror      (other quad of the half row)
    v += dpp_f32<0x140>(v);    // row_mirror           (other half of the row)
    return v;
}
__device__ __forceinline__ float wave_sum(float v) {
    v = row16_sum(v);
    // combine the four rows: two readlane-free steps through ds_swizzle-less shuffles would need
    // LDS; use v_readlane (scalar broadcast) instead: 4 reads + 3 adds, all lanes get the total
    const float r0 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 0));
    const float r1 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 16));
    const float r2 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 32));
    const float r3 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 48));
    return (r0 + r1) + (r2 + r3);
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v = fmaxf(v, __shfl_xor(v, off, 64));
    return v;
}

// XCD-aware block mapping.  Workgroups are dealt to the 8 XCDs round-robin by linear block id, and each
// XCD has its own L2.  v = xcd_swizzle(id, total) renumbers the blocks so that CONSECUTIVE v land on the
// SAME XCD (v in [x*total/8, (x+1)*total/8) runs on XCD x): blocks that share read-only data (the 16
// channel blocks of one support image share its mask gather) then hit in one L2 instead of fetching
// the data once per XCD.  Speed only: any placement is correct.
__device__ __forceinline__ int xcd_swizzle(int id, int total) {
    const int per = total >> 3;
    if (id >= (per << 3)) return id;             // remainder blocks keep their id
    return (id & 7) * per + (id >> 3);
}

// bf16 <-> f32
__device__ __forceinline__ float bf16_to_f32(unsigned short u) {
    return __uint_as_float(((unsigned int)u) << 16);
}
__device__ __forceinline__ unsigned short f32_to_bf16(float f) {
    // round to nearest even; NaN stays NaN
    unsigned int u = __float_as_uint(f);
    if ((u & 0x7fffffffu) > 0x7f800000u) return (unsigned short)((u >> 16) | 0x40);
    return (unsigned short)((u + 0x7fffu + ((u >> 16) & 1u)) >> 16);
}

template <typename T> struct FeatIO;
template <> struct FeatIO<float> {
    static __device__ __forceinline__ float load(const float* p) { return *p; }
    static __device__ __forceinline__ void store(float* p, float v) { *p = v; }
};
template <> struct FeatIO<unsigned short> {
    static __device__ __forceinline__ float load(const unsigned short* p) { return bf16_to_f32(*p); }
    static __device__ __forceinline__ void store(unsigned short* p, float v) { *p = f32_to_bf16(v); }
};

// ---- train-mode dropout generated inside the kernels ---------------------------------------------
// F.dropout(p) of model/common.py:231 (gate) and :326 (class weights): multiplier 0 with probability p,
// 1/(1-p) otherwise.  Counter-based Philox4x32-10 (Salmon et al., SC'11): key = seed, counter =
// (element index / 4, stream id, 0, 0); one evaluation yields the multipliers of four consecutive
// elements, so forward and backward regenerate the same values from (seed, index) and nothing but the
// seed is kept between them.  A mask POINTER, where one is passed, always wins (injected-mask parity).
#define DENET_DROP_GATE 1u
#define DENET_DROP_WEIGHT 2u
struct DropGen {
    unsigned k0, k1;     // seed
    unsigned thresh;     // drop iff r < thresh, thresh = round(p * 2^32)
    float scale;         // 1 / (1 - p)
    int on;              // 0: multiplier 1
};
__host__ __forceinline__ DropGen make_dropgen(unsigned long long seed, float p) {
    DropGen g{(unsigned)seed, (unsigned)(seed >> 32), 0u, 1.f, 0};
    if (p > 0.f && p < 1.f) {
        const double t = (double)p * 4294967296.0 + 0.5;
        g.thresh = t >= 4294967295.0 ? 4294967295u : (unsigned)t;
        g.scale = 1.f / (1.f - p);
        g.on = 1;
    }
    return g;
}
__device__ __forceinline__ void philox4x32_10(unsigned c0, unsigned c1, unsigned k0, unsigned k1, unsigned r[4]) {
    unsigned c2 = 0u, c3 = 0u;
#pragma unroll
    for (int i = 0; i < 10; ++i) {
        const unsigned h0 = __umulhi(0xD2511F53u, c0), l0 = 0xD2511F53u * c0;
        const unsigned h1 = __umulhi(0xCD9E8D57u, c2), l1 = 0xCD9E8D57u * c2;
        c0 = h1 ^ c1 ^ k0; c1 = l1; c2 = h0 ^ c3 ^ k1; c3 = l0;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    r[0] = c0; r[1] = c1; r[2] = c2; r[3] = c3;
}
// multipliers of elements 4*i4 .. 4*i4+3 of stream `stream`
__device__ __forceinline__ f32x4 drop_mult4(const DropGen& g, unsigned stream, unsigned i4) {
    unsigned r[4];
    philox4x32_10(i4, stream, g.k0, g.k1, r);
    return f32x4{r[0] < g.thresh ? 0.f : g.scale, r[1] < g.thresh ? 0.f : g.scale,
                 r[2] < g.thresh ? 0.f : g.scale, r[3] < g.thresh ? 0.f : g.scale};
}
// multiplier of element idx
__device__ __forceinline__ float drop_mult1(const DropGen& g, unsigned stream, unsigned idx) {
    unsigned r[4];
    philox4x32_10(idx >> 2, stream, g.k0, g.k1, r);
    const unsigned lo = (idx & 1u) ? r[1] : r[0], hi = (idx & 1u) ? r[3] : r[2];
    return ((idx & 2u) ? hi : lo) < g.thresh ? 0.f : g.scale;
}

// Raise a kernel's dynamic-LDS limit above the 48 KiB default (host side, cached per kernel so
// steady-state launches make no runtime call besides the launch itself).  Defined in api.hip.
int denet_ensure_dynamic_lds(const void* kernel, size_t bytes);

// The three launches of K2b (proto_extend.hip), separately callable by the fused step.
int denet_pe_bwd_reduce(const float* dWn_part, int64_t n_part, int64_t N, int64_t K, int64_t C, float* dWn,
                        void* stream);
int denet_pe_bwd_chain(const float* dWn, int64_t n_part, const float* S, const float* cnt, const float* hidden, const float* gate,
                       const float* W1, const float* W2, const float* Wl, const int64_t* labels,
                       const float* gate_mask, int64_t N, int64_t way, int64_t shot, int64_t K, int64_t C,
                       int64_t P, float* u1, float* u2, float* u3, float* coef, void* stream);
int denet_pe_bwd_params(const float* u1, const float* u2, const float* u3, const float* S, const float* hidden,
                        const float* pool, const float* dWn, int64_t n_part, const int64_t* labels,
                        const float* weight_mask, int64_t N, int64_t way, int64_t shot, int64_t K, int64_t C, int64_t P, float* dW1,
                        float* db1, float* dW2, float* db2, float* dWl, float* dbl, float* dweight, void* stream);

// CE + focal in one launch (upsample_loss.hip), used by the fused step.
int denet_loss_pair_fwd(const float* logits, const int64_t* Yq_full, const float* wgt, const float* logits_binary,
                        const int64_t* Yq, int64_t N, int64_t NB, int64_t K, int64_t h, int64_t w, int64_t H,
                        int64_t W, float alpha, float gamma, float* part_k, float* part_p, float* da, float* db,
                        void* stream);

// Diagnostic builds only (scripts/diag): phase stamps of block 0 / thread 0 into a debug buffer.
#ifdef DENET_STAMPS
__device__ unsigned long long* g_stamps;
#define STAMP(i) do { if (blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) g_stamps[i] = __builtin_amdgcn_s_memrealtime(); } while (0)
#else
#define STAMP(i) do { } while (0)
#endif
// Diagnostic builds only: start / end time and CU of EVERY block (g_bstamps[3*block + {0,1,2}]).
#ifdef DENET_BLOCK_STAMPS
__device__ unsigned long long* g_bstamps;
#define BSTAMP(slot) do { if (threadIdx.x == 0) { \
        const size_t b__ = (size_t)blockIdx.y * gridDim.x + blockIdx.x; \
        g_bstamps[3 * b__ + (slot)] = __builtin_amdgcn_s_memrealtime(); \
        if ((slot) == 0) { unsigned id__, xc__; asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(id__)); asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xc__)); g_bstamps[3 * b__ + 2] = ((unsigned long long)xc__ << 32) | id__; } } } while (0)
#else
#define BSTAMP(slot) do { } while (0)
#endif
