// K4 / K4b, fast path: bilinear up-sampling fused with the weighted cross entropy and its gradient
// w.r.t. the low-res logits, with NO exponential, compare or label read per (pixel, class); and K5, the
// binary focal loss, as a body of its own.
// (trainer/trainers.py:954-956 + trainer/utils.py:264-281 + loss/losses.py:19-79 of the reference.)
//
// Same block decomposition as upsample_loss_body (dev_upsample_loss.hpp): block = (episode n, low-res
// row py, column range), band = the label rows whose upper bilinear tap is py, cell = the label columns
// whose left tap is one low-res column.  What is different is the arithmetic inside a cell:
//
//   * Inside a cell the up-sampled logit of class k is BILINEAR in (h1, w1), the vertical / horizontal
//     lerp weights, and those are arithmetic progressions over the band's rows r and the cell's pixels e:
//     h1 = hb + r*dh, w1 = wb + e*dw  (hb, wb: the weights ATen computes for the cell's first row /
//     pixel; dh, dw: the fp32 scales.  Exact when the scale is a power of two, 41 -> 321; otherwise
//     within one rounding of the source coordinate, <= 8e-6 at 81 -> 642, of the weights ATen rounds
//     per pixel).  So with y = x - U, U the same bilinear blend of the per-tap class maxima (U >= every
//     blended logit, y <= 0),
//         y_k(r,e) = y00 + r*yr + e*ye + r*e*yre,   exp2(y_k(r,e)) = E00 * sigma^r * rho0^e * tau^(r*e)
//     -- four exponentials per (class, cell) and then only multiplications:
//         E(r,0) = E(r-1,0)*sigma,  rho(r) = rho(r-1)*tau,  E(r,e) = E(r,e-1)*rho(r).
//   * Phase A, thread = (class pair, cell): chain starts {E(r,0), rho(r)} of every band row -> LDS table.
//   * Phase 1, two threads = (row, cell), each over half of the class pairs: sum-exp of the cell row's
//     pixels: one table read per class pair, then one packed multiplication + one packed addition per
//     (pixel, class pair).  Then per PIXEL (each thread of the pair takes half of the row's pixels, whose
//     labels it loaded itself at the start of the block): weight*log2(sum-exp), s = weight/sum-exp -> LDS, and
//     ONE 64-bit integer LDS atomic that adds the pixel's (1, e, r, r*e) to the moments of its label in the cell.
//     Everything that is linear in the label's logit -- the one-hot part of the gradient, the label-logit
//     term of the loss, the weight sum -- follows from those integer moments once per (class, cell).
//     (Integer sums are order independent: results stay bit-stable run to run.)
//   * Phase 2, thread = (class, cell): walks the same chain over (row pair, e) in packed arithmetic; per
//     (pixel, class) one LDS word of s, g = E*s, and two running sums.  The four taps' sums are linear in
//     the moments (sum g, sum e*g, sum r*g, sum r*e*g); sum e*g and sum r*g come from prefix sums
//     (sum_e e*g[e] = n*sum g - sum_e (g[0]+..+g[e])), so the inner loop is 3 packed instructions per
//     2 (pixel, class) visits.
//
// Why packed: measured on MI355X (scripts/diag/valu_rate.hip), a SIMD issues one wave64 VALU instruction
// per ~4.5 cycles whatever the wave count, v_pk_{mul,add,fma}_f32 cost the same slot as the plain forms and
// v_exp_f32 costs two -- the kernel is bound by the instruction count, so two lanes of work per
// instruction halve it.
//
// Validity: every chain element is a true value exp2(y) with y in [min tap, 0]; a block whose shifted tap
// logits all lie in [-120, 0] (log2 units: 83 nats below the tap's best class) can neither overflow nor lose
// an element to underflow.  Any other block (wild logits, NaN / Inf) runs the exact per-pixel arithmetic of
// upsample_loss_body -- block-uniform branch, same outputs.
#pragma once
#include "dev_upsample_loss.hpp"
#include <type_traits>

#define WC_TAME (-120.f)
// diagnostic builds only (scripts/diag/loss_bench.hip): leave the block after phase WC_STOP, to count the
// instructions of each phase with the SQ counters
#ifdef WC_STOP
#define WC_PHASE_END(i) do { if (WC_STOP == (i)) return; } while (0)
#else
#define WC_PHASE_END(i) do { } while (0)
#endif

// order-preserving float <-> int map (for the LDS integer max of the per-tap class maxima)
__device__ __forceinline__ int wc_enc(float f) {
    const int b = __float_as_int(f);
    return b >= 0 ? b : b ^ 0x7fffffff;
}
__device__ __forceinline__ float wc_dec(int b) { return __int_as_float(b >= 0 ? b : b ^ 0x7fffffff); }

// q = it / d, rem = it - q*d for 0 <= it < 2^20, 0 < d < 2^10, inv = 1/d (an integer division costs ~25 VALU slots)
__device__ __forceinline__ int wc_divmod(int it, int d, float inv, int& rem) {
    int q = (int)(((float)it + 0.5f) * inv);
    int r = it - q * d;
    if (r < 0) { --q; r += d; }
    if (r >= d) { ++q; r -= d; }
    rem = r;
    return q;
}

// exponents (log2 domain) of the chain of (class k, cell cl): y(r,e) = y00 + r*yr + e*ye + r*e*yre
struct ChainY {
    float y00, yr, ye, yre;
    bool tame;
};
__device__ __forceinline__ ChainY wc_chain_exponents(const float* raw0, const float* raw1, const float* Mf, int ncmax,
                                                     int k, int cl, int cr, float hb, float wb, float dh, float dw) {
    const float z00 = raw0[k * ncmax + cl] - Mf[cl], z01 = raw0[k * ncmax + cr] - Mf[cr];
    const float z10 = raw1[k * ncmax + cl] - Mf[ncmax + cl], z11 = raw1[k * ncmax + cr] - Mf[ncmax + cr];
    ChainY p;
    p.tame = (z00 >= WC_TAME) && (z01 >= WC_TAME) && (z10 >= WC_TAME) && (z11 >= WC_TAME);   // false on NaN
    const float V = z10 - z00, Hd = z01 - z00, c = (z11 - z10) - Hd;
    const float hc = fmaf(hb, c, Hd);                     // horizontal slope in the band's first row
    p.y00 = fmaf(wb, hc, fmaf(hb, V, z00));
    p.yr = dh * fmaf(wb, c, V);
    p.ye = dw * hc;
    p.yre = dh * dw * c;
    return p;
}

// s (weight / sum-exp) of a cell row pair: EM pixel slots of 8 bytes, padded to a cell stride of 4 m + 2 slots --
// phase 2 reads it with ds_read_b128, lanes = consecutive cells: at 8 slots (64 B) a 16-lane group falls on four
// of the sixteen 16-byte bank slots (3.5-way conflicts, 10.9 M of the kernel's 20.2 M LDS cycles at C3, PMC
// SQ_LDS_BANK_CONFLICT), at 10 slots (80 B) every cell of a group has a slot of its own
__host__ __device__ constexpr int wc_cell_stride(int EM) { return (((EM + 1) & ~1) % 4 == 0) ? ((EM + 1) & ~1) + 2 : ((EM + 1) & ~1); }
// LDS floats of the chain path (host and device must agree); KC = classes per table chunk (even, or K)
__host__ __device__ __forceinline__ int wc_lds_floats(int K, int ncmax, int rows_max, int KC, int EM, bool grad) {
    const int kcol = K * ncmax, CS = wc_cell_stride(EM), KC2 = (KC + 1) >> 1, RP = (rows_max + 1) >> 1;
    int off = 2 * kcol + 2 * ncmax + (K + 1) + 16 + ncmax + (ncmax + 1) + 1;
    off = (off + 3) & ~3;
    const int tsz = 4 * rows_max * KC2 * ncmax, asz = grad ? 4 * kcol : 0;
    off += tsz > asz ? tsz : asz;
    off = (off + 3) & ~3;
    off += grad ? 2 * RP * ncmax * CS : 0;
    off = (off + 3) & ~3;
    off += 2 * (kcol + UL_T / 2);
    return off;
}

// RSF: row groups of the exact fallback body (its LDS layout; the launcher sizes the allocation for both).
// EM: label columns per cell (the widest cell of the geometry: 8 at 41 -> 321, 9 at 81 -> 642).
template <bool GRAD, int EM, int RSF>
__device__ __forceinline__ void wce_chain_body(const LossArgs& g, int n, int bx, float* sm) {
    constexpr int EMP2 = (EM + 1) & ~1;                   // pixel slots read per cell row pair (16-byte reads of 2 pixels)
    constexpr int CS = wc_cell_stride(EM);                // cell stride of s in pixel slots (bank-conflict padding)
    const float* __restrict__ logits = g.logits;
    const int64_t* __restrict__ Y = g.Y;
    const float* __restrict__ wgt = g.wgt;
    const int K = g.K, h = g.h, w = g.w, H = g.H, W = g.W, xsplit = g.xsplit, rows_max = g.rows_max, KC = g.kchunk;
    const int py = bx / xsplit, xsid = bx - py * xsplit;
    const int py1 = py + 1 < h ? py + 1 : h - 1;
    const int tid = threadIdx.x;
    const int cw = (w + xsplit - 1) / xsplit;
    const int c0 = xsid * cw, c1 = min(w, c0 + cw);
    const int cell_lo = (GRAD && c0 > 0) ? c0 - 1 : c0;
    const int ncell = c1 - cell_lo;
    const int ncmax = cw + (GRAD ? 2 : 1);
    const int col_hi = min(c1, w - 1);
    const int ncol = col_hi - cell_lo + 1;
    const int kcol = K * ncmax;
    const int KC2 = (KC + 1) >> 1, RP = (rows_max + 1) >> 1;

    float* raw0 = sm;                                     // [K][ncmax] logits row py  * log2e
    float* raw1 = raw0 + kcol;                            // [K][ncmax] logits row py1 * log2e
    float* Mf = raw1 + kcol;                              // [2][ncmax] per-tap class maxima (ints while staged)
    float* wg = Mf + 2 * ncmax;                           // [K+1] class weights; [K] = 0 (ignored / padding pixels)
    float* red = wg + (K + 1);                            // [16]
    float* w1b = red + 16;                                // [ncmax] lerp weight of each cell's first pixel
    int* xs = reinterpret_cast<int*>(w1b + ncmax);        // [ncmax+1] first pixel of each cell
    int* flag = xs + ncmax + 1;                           // [1] some tap of the block is outside the tame range
    int off = 2 * kcol + 2 * ncmax + (K + 1) + 16 + ncmax + (ncmax + 1) + 1;
    off = (off + 3) & ~3;
    f32x4* T = reinterpret_cast<f32x4*>(sm + off);        // [rows_max][KC2][ncmax] {E_k, E_k', rho_k, rho_k'}(r)
    float* acc = sm + off;                                // GRAD: [4][K][ncmax] tap sums (after the table is dead)
    {
        const int tsz = 4 * rows_max * KC2 * ncmax, asz = GRAD ? 4 * kcol : 0;
        off += tsz > asz ? tsz : asz;
        off = (off + 3) & ~3;
    }
    f32x2* sbuf = reinterpret_cast<f32x2*>(sm + off);     // GRAD: [RP][ncmax][CS] weight / sum-exp of rows (2i, 2i+1)
    off += GRAD ? 2 * RP * ncmax * CS : 0;
    off = (off + 3) & ~3;
    // label moments [K][ncmax], then one dummy slot per phase-1 item (ignored / padding pixels add there)
    unsigned long long* oh = reinterpret_cast<unsigned long long*>(sm + off);
    off += 2 * (kcol + UL_T / 2);

    const float sy = g.sy, sx = g.sx;
    const int ys = first_dst_ge(py, sy, h, H), ye = first_dst_ge(py + 1, sy, h, H);
    const int rows = ye - ys;
    const float hb = sy * (float)ys - (float)py;          // h1 of the band's first row, as ATen rounds it
    const float dh = sy, dw = sx;
    const float inv_ncell = 1.f / (float)ncell;
    // the partial last round of the (class, cell) loops goes to a different wave in every block
    const int rot = ((bx + n) & 3) << 6;

    STAMP(0);
    // phase-1 thread pair (2i, 2i+1) = item i = (row r1, cell cl1).  Lane `half` owns the pixels
    // e = half*NPX + j of the cell row and sums the class pairs kp with (kp & 1) == half.  Its labels are
    // requested now, straight from the int64 map (no LDS staging, one memory round trip for labels and logits
    // together), and narrowed to bytes once the logits are in.
    constexpr int NPX = (EM + 1) / 2;
    const int item1 = tid >> 1, half = tid & 1;
    const bool p1 = item1 < rows * ncell;
    int cl1 = 0;
    const int r1 = p1 ? wc_divmod(item1, ncell, inv_ncell, cl1) : 0;
    int64_t labv[NPX];
    {
        const int x0 = first_dst_ge(cell_lo + cl1, sx, w, W) + half * NPX;
        const int64_t* lrow = Y + ((int64_t)n * H + ys + r1) * W;
#pragma unroll
        for (int j = 0; j < NPX; ++j) labv[j] = lrow[min(x0 + j, W - 1)];
    }
    int* Mi = reinterpret_cast<int*>(Mf);
    if (tid < 2 * ncmax) Mi[tid] = (int)0x80000000;
    if (tid == 0) flag[0] = 0;
    for (int i = tid; i < kcol; i += UL_T) oh[i] = 0ull;
    if (GRAD) {
        const f32x2 z2 = {0.f, 0.f};
        for (int i = tid; i < RP * ncmax * CS; i += UL_T) sbuf[i] = z2;
    }
    __syncthreads();

    const float* src = logits + (int64_t)n * K * h * w;
    // class weight of thread tid, requested with the logits (one round trip for all of the block's global operands)
    float wv0 = 1.f;
    if (wgt) wv0 = wgt[(int64_t)n * K + min(tid, K - 1)];
    {
        // thread <-> (class, column) by shifts (the column count rounded up to a power of two)
        const int xsh = 32 - __builtin_clz((unsigned)max(ncol - 1, 1));
        const int x = tid & ((1 << xsh) - 1), kstep = UL_T >> xsh;
        const int xc = min(x, ncol - 1);
        const float* s0 = src + (int64_t)py * w + cell_lo;       // block-uniform bases, 32-bit element offsets
        const float* s1 = src + (int64_t)py1 * w + cell_lo;
        const unsigned hw = (unsigned)(h * w);
        float m0 = -INFINITY, m1 = -INFINITY;
        bool nan = false;
        // bursts of NB row pairs, clamped and unconditional, all requested before the first is used: a loop of
        // load -> wait -> store costs one memory round trip per 256 >> xsh classes (six at K = 81)
        auto stage = [&](auto nb) {
            constexpr int NB = decltype(nb)::value;
            for (int kk = tid >> xsh; kk < K + (tid >> xsh); kk += NB * kstep) {
                float v0[NB], v1[NB];
#pragma unroll
                for (int u = 0; u < NB; ++u) {
                    const unsigned o = (unsigned)min(kk + u * kstep, K - 1) * hw + (unsigned)xc;
                    v0[u] = s0[o];
                    v1[u] = s1[o];
                }
#pragma unroll
                for (int u = 0; u < NB; ++u) {
                    const int k = kk + u * kstep;
                    const float a0 = v0[u] * UL_LOG2E, a1 = v1[u] * UL_LOG2E;
                    if (x < ncol && k < K) {
                        raw0[k * ncmax + x] = a0;
                        raw1[k * ncmax + x] = a1;
                    }
                    nan = nan || (a0 != a0) || (a1 != a1);
                    m0 = fmaxf(m0, a0);
                    m1 = fmaxf(m1, a1);
                }
            }
        };
        if (K <= 2 * kstep) stage(std::integral_constant<int, 2>{});
        else if (K <= 4 * kstep) stage(std::integral_constant<int, 4>{});
        else stage(std::integral_constant<int, 8>{});
        if (x < ncol && (tid >> xsh) < K) {
            atomicMax(&Mi[x], wc_enc(m0));
            atomicMax(&Mi[ncmax + x], wc_enc(m1));
        }
        if (nan) flag[0] = 1;
    }
    if (tid <= K) wg[tid] = tid < K ? wv0 : 0.f;
    for (int k = tid + UL_T; k <= K; k += UL_T) wg[k] = k < K ? (wgt ? wgt[(int64_t)n * K + k] : 1.f) : 0.f;
    for (int c = tid; c <= ncell; c += UL_T) {
        const int xq = first_dst_ge(cell_lo + c, sx, w, W);
        xs[c] = xq;
        int i0;
        float f;
        bilinear_tap(xq < W ? xq : W - 1, sx, w, i0, f);
        if (c < ncell) w1b[c] = f;
    }
    __syncthreads();
    STAMP(1);
    WC_PHASE_END(1);
    // labels -> class index, or K for an ignored label / a column past the cell (weight 0, dummy moments)
    unsigned lbytes[(NPX + 3) / 4];
    {
        const int nx = xs[cl1 + 1] - xs[cl1] - half * NPX;  // this lane's share of the cell's pixels
#pragma unroll
        for (int q = 0; q < (NPX + 3) / 4; ++q) lbytes[q] = 0;
#pragma unroll
        for (int j = 0; j < NPX; ++j) {
            const unsigned lb = ((unsigned long long)labv[j] < (unsigned long long)K && j < nx) ? (unsigned)labv[j] : (unsigned)K;
            lbytes[j >> 2] |= lb << (8 * (j & 3));
        }
    }
    if (tid < 2 * ncmax) Mf[tid] = wc_dec(Mi[tid]);
    __syncthreads();
    STAMP(2);
    WC_PHASE_END(2);

    // ---------------- phase A + phase 1, class chunk by class chunk ----------------
    f32x2 se2[EM];
#pragma unroll
    for (int e = 0; e < EM; ++e) se2[e] = {0.f, 0.f};
    int wild = 0;
    for (int kb = 0; kb < K; kb += KC) {
        const int kn = min(KC, K - kb), np = (kn + 1) >> 1;
        for (int it0 = 0; it0 < np * ncell; it0 += UL_T) {
            const int it = it0 + ((tid + rot) & (UL_T - 1));
            if (it < np * ncell) {
                int cl;
                const int kp = wc_divmod(it, ncell, inv_ncell, cl);
                const int cr = min(cell_lo + cl + 1, w - 1) - cell_lo;
                const int k0 = kb + 2 * kp;
                const bool has1 = 2 * kp + 1 < kn;
                const float wb = w1b[cl];
                const ChainY a = wc_chain_exponents(raw0, raw1, Mf, ncmax, k0, cl, cr, hb, wb, dh, dw);
                const ChainY b = wc_chain_exponents(raw0, raw1, Mf, ncmax, has1 ? k0 + 1 : k0, cl, cr, hb, wb, dh, dw);
                if (!(a.tame && b.tame)) flag[0] = 1;
                f32x2 E = {__builtin_amdgcn_exp2f(a.y00), has1 ? __builtin_amdgcn_exp2f(b.y00) : 0.f};
                f32x2 rho = {__builtin_amdgcn_exp2f(a.ye), has1 ? __builtin_amdgcn_exp2f(b.ye) : 0.f};
                const f32x2 sig = {__builtin_amdgcn_exp2f(a.yr), __builtin_amdgcn_exp2f(b.yr)};
                const f32x2 tau = {__builtin_amdgcn_exp2f(a.yre), __builtin_amdgcn_exp2f(b.yre)};
                f32x4* t = T + kp * ncmax + cl;
                for (int r = 0; r < rows; ++r) {
                    const f32x4 v = {E[0], E[1], rho[0], rho[1]};
                    t[r * KC2 * ncmax] = v;
                    E *= sig;
                    rho *= tau;
                }
            }
        }
        __syncthreads();
        wild = flag[0];
        if (p1) {
            const f32x4* t = T + (r1 * KC2) * ncmax + cl1;
#pragma unroll 2
            for (int kp = half; kp < np; kp += 2) {
                const f32x4 v = t[kp * ncmax];
                f32x2 E = {v[0], v[1]};
                const f32x2 rho = {v[2], v[3]};
#pragma unroll
                for (int e = 0; e < EM; ++e) {
                    se2[e] += E;
                    E *= rho;
                }
            }
        }
        if (kb + KC < K) __syncthreads();                 // the next chunk rewrites the table
    }
    if (wild) {
        // exact arithmetic for this block (every thread read the same flag value after the last phase-A barrier)
        __syncthreads();
        upsample_loss_body<GRAD, RSF>(g, n, bx, sm);
        return;
    }
    STAMP(3);
    WC_PHASE_END(3);

    // ---------------- phase 1, per pixel ----------------
    float numL = 0.f;                                     // sum weight * log2(sum-exp) over this thread's pixels
    if (p1) {
        float se[EM];
#pragma unroll
        for (int e = 0; e < EM; ++e) {
            se[e] = se2[e][0] + se2[e][1];
            se[e] += dpp_f32<0xB1>(se[e]);                // the other half of the class pairs
        }
        f32x2* srow = sbuf + ((r1 >> 1) * ncmax + cl1) * CS + half * NPX;
        const int rodd = r1 & 1;
        const unsigned hN = (unsigned)(half * NPX);
        const unsigned lo0 = 1u | (hN << 16), hi0 = (((unsigned)r1 * hN) << 16) | (unsigned)r1, hi1 = (unsigned)r1 << 16;
        const int hmask = -half;                          // bit select instead of ?: (an indexed select of se[]
                                                          // would put the array in scratch memory)
#pragma unroll
        for (int j = 0; j < NPX; ++j) {
            const int lb = (int)__builtin_amdgcn_ubfe(lbytes[j >> 2], 8 * (j & 3), 8);
            const float wy = wg[lb];
            const float shi = NPX + j < EM ? se[NPX + j < EM ? NPX + j : 0] : 1.f;
            const float sv = __int_as_float((__float_as_int(se[j]) & ~hmask) | (__float_as_int(shi) & hmask));
            numL = fmaf(wy, __log2f(sv), numL);
            // moments (1, e, r, r*e) of pixel e = hN + j, 16 bits each
            const unsigned lo = lo0 + ((unsigned)j << 16), hi = hi0 + hi1 * (unsigned)j;
            const int slot = lb < K ? lb * ncmax + cl1 : kcol + item1;
#ifndef WC_DIAG_NO_ATOMIC
            atomicAdd(&oh[slot], ((unsigned long long)hi << 32) | lo);
#else
            if (slot == 0x7fffffff) oh[0] = ((unsigned long long)hi << 32) | lo;
#endif
            if (GRAD) {
                const float s = wy * __builtin_amdgcn_rcpf(sv);
                reinterpret_cast<float*>(srow + j)[rodd] = s;
            }
        }
        if (cell_lo + cl1 < c0) numL = 0.f;               // halo cell: its pixels belong to the left neighbour block
    }
    __syncthreads();
    STAMP(4);
    WC_PHASE_END(4);

    // ---------------- phase 2 ----------------
    float* gl0 = acc;
    float* gl1 = gl0 + kcol;
    float* gr0 = gl1 + kcol;
    float* gr1 = gr0 + kcol;
    const int npair = (rows + 1) >> 1;
    const float fnp = (float)npair, fe = (float)EM;
    float numY = 0.f, den = 0.f;                          // sum weight * label exponent, sum weight (owned cells)
    for (int it0 = 0; it0 < K * ncell; it0 += UL_T) {
        const int it = it0 + ((tid + rot) & (UL_T - 1));
        if (it >= K * ncell) continue;
        int cl;
        const int k = wc_divmod(it, ncell, inv_ncell, cl);
        const int cr = min(cell_lo + cl + 1, w - 1) - cell_lo;
        const float wb = w1b[cl];
        const ChainY p = wc_chain_exponents(raw0, raw1, Mf, ncmax, k, cl, cr, hb, wb, dh, dw);
        const unsigned long long I = oh[k * ncmax + cl];
        const float wk = wg[k];
        const float I00 = (float)(unsigned)(I & 0xffffull), I01 = (float)(unsigned)((I >> 16) & 0xffffull);
        const float I10 = (float)(unsigned)((I >> 32) & 0xffffull), I11 = (float)(unsigned)(I >> 48);
        if (cell_lo + cl >= c0) {
            numY = fmaf(wk, fmaf(p.y00, I00, fmaf(p.yr, I10, fmaf(p.ye, I01, p.yre * I11))), numY);
            den = fmaf(wk, I00, den);
        }
        if (!GRAD) continue;
        const float sg = __builtin_amdgcn_exp2f(p.yr), ta = __builtin_amdgcn_exp2f(p.yre);
        const float e00 = __builtin_amdgcn_exp2f(p.y00), r00 = __builtin_amdgcn_exp2f(p.ye);
        f32x2 E0 = {e00, e00 * sg}, rho = {r00, r00 * ta};      // rows (2j, 2j+1)
        const f32x2 sg2 = {sg * sg, sg * sg}, ta2 = {ta * ta, ta * ta};
        f32x2 A = {0.f, 0.f}, PA = A, B = A, PB = A;
        const f32x2* srow = sbuf + cl * CS;
        for (int j = 0; j < npair; ++j) {
            f32x2 s[EMP2];
#pragma unroll
            for (int q = 0; q < EMP2 / 2; ++q) {
                const f32x4 v = *reinterpret_cast<const f32x4*>(srow + j * ncmax * CS + 2 * q);
                s[2 * q] = {v[0], v[1]};
                s[2 * q + 1] = {v[2], v[3]};
            }
            f32x2 E = E0, gS = {0.f, 0.f}, pre = gS;
#pragma unroll
            for (int e = 0; e < EM; ++e) {
                gS += E * s[e];
                pre += gS;
                E *= rho;
            }
            A += gS;
            PA += A;
            B += pre;
            PB += B;
            E0 *= sg2;
            rho *= ta2;
        }
        // moments over the cell: sum g, sum e*g, sum r*g, sum r*e*g, with r = 2j + c (c: packed component) and
        // prefix sums: sum_j j*x[j] = n*sum x - sum of prefixes
        const float S0 = A[0], S1 = A[1], Bs0 = B[0], Bs1 = B[1];
        const float J0 = fmaf(fnp, S0, -PA[0]), J1 = fmaf(fnp, S1, -PA[1]);          // sum_j j*gS
        const float JB0 = fmaf(fnp, Bs0, -PB[0]), JB1 = fmaf(fnp, Bs1, -PB[1]);      // sum_j j*pre
        float G00 = S0 + S1;
        float G10 = fmaf(2.f, J0 + J1, S1);
        float G01 = fmaf(fe, G00, -(Bs0 + Bs1));
        float G11 = fmaf(fe, G10, -fmaf(2.f, JB0 + JB1, Bs1));
        G00 = fmaf(-wk, I00, G00);
        G01 = fmaf(-wk, I01, G01);
        G10 = fmaf(-wk, I10, G10);
        G11 = fmaf(-wk, I11, G11);
        const float Gw = fmaf(dw, G01, wb * G00);                    // sum w1*g
        const float Gh = fmaf(dh, G10, hb * G00);                    // sum h1*g
        const float Ghw = fmaf(dh, fmaf(dw, G11, wb * G10), hb * Gw);   // sum h1*w1*g
        const int o = k * ncmax + cl;
        gr1[o] = Ghw;
        gr0[o] = Gw - Ghw;
        gl1[o] = Gh - Ghw;
        gl0[o] = ((G00 - Gw) - Gh) + Ghw;
    }
    STAMP(5);
    WC_PHASE_END(5);
    {
        const float a = wave_sum(numL), b = wave_sum(numY), c = wave_sum(den);
        if ((tid & 63) == 0) {
            red[(tid >> 6) * 3] = a;
            red[(tid >> 6) * 3 + 1] = b;
            red[(tid >> 6) * 3 + 2] = c;
        }
    }
    __syncthreads();
    if (tid == 0) {
        const float tl = (red[0] + red[3]) + (red[6] + red[9]), ty = (red[1] + red[4]) + (red[7] + red[10]);
        const int64_t o = (((int64_t)n * h + py) * xsplit + xsid) * 2;
        g.part[o] = (tl - ty) * UL_LN2;
        g.part[o + 1] = (red[2] + red[5]) + (red[8] + red[11]);
    }
    if (GRAD) {
        // block-uniform row bases, 32-bit element offsets
        float* da = g.dlog_a + (int64_t)n * K * h * w + (int64_t)py * w;
        float* db = g.dlog_b + (int64_t)n * K * h * w + (int64_t)py1 * w;
        float* db0 = g.dlog_b + (int64_t)n * K * h * w;
        const unsigned hw = (unsigned)(h * w);
        const int nown = c1 - c0;
        const float inv_nown = 1.f / (float)nown;
        for (int idx = tid; idx < K * nown; idx += UL_T) {
            int xo;
            const int k = wc_divmod(idx, nown, inv_nown, xo);
            const int x = c0 + xo;
            const int o = k * ncmax + (x - cell_lo);
            float g0 = gl0[o], g1 = gl1[o];
            if (x >= 1) { g0 += gr0[o - 1]; g1 += gr1[o - 1]; }
            if (x == w - 1) { g0 += gr0[o]; g1 += gr1[o]; }
            const unsigned go = (unsigned)k * hw + (unsigned)x;
            if (py1 == py) {
                da[go] = g0 + g1;
            } else {
                da[go] = g0;
                db[go] = g1;
            }
            if (py == 0) db0[go] = 0.f;
        }
    }
    STAMP(6);
}

// K5: binary focal loss on the up-sampled binary logits (loss/losses.py:47-79; alpha-balanced, log(softmax) as
// the reference computes it: -inf when the probability underflows).  Forward only.  No LDS staging: thread =
// (label column, row parity); the column's horizontal blend of the four taps of both channels is formed once
// (ATen's order: horizontal, then vertical), every label row of the band then costs two FMAs before the two-way
// soft-max.  Same block decomposition and partial-sum layout as upsample_loss_body<1, false>.
__device__ __forceinline__ void focal_body(const LossArgs& g, int n, int bx, float* sm) {
    const int h = g.h, w = g.w, H = g.H, W = g.W, xsplit = g.xsplit;
    const float alpha = g.alpha, gamma = g.gamma;
    const int py = bx / xsplit, xsid = bx - py * xsplit;
    const int py1 = py + 1 < h ? py + 1 : h - 1;
    const int tid = threadIdx.x;
    const int cw = (w + xsplit - 1) / xsplit;
    const int c0 = xsid * cw, c1 = min(w, c0 + cw);
    const float sy = g.sy, sx = g.sx;
    const int ys = first_dst_ge(py, sy, h, H), ye = first_dst_ge(py + 1, sy, h, H);
    const int xb = first_dst_ge(c0, sx, w, W), xe = first_dst_ge(c1, sx, w, W);
    const float* l0 = g.logits + (int64_t)n * 2 * h * w;
    const float* l1 = l0 + (int64_t)h * w;
    const int64_t* Yn = g.Y + (int64_t)n * H * W;
    float num = 0.f, den = 0.f;
    const int par = tid & 1;
    for (int x = xb + (tid >> 1); x < xe; x += UL_T / 2) {
        int i0;
        float w1;
        bilinear_tap(x, sx, w, i0, w1);
        const int i1 = i0 + (i0 < w - 1);
        const float w0 = 1.f - w1;
        // ATen: h0*(w0*v00 + w1*v01) + h1*(w0*v10 + w1*v11), in the log2 domain
        const float t0 = (w0 * l0[py * w + i0] + w1 * l0[py * w + i1]) * UL_LOG2E;
        const float b0 = (w0 * l0[py1 * w + i0] + w1 * l0[py1 * w + i1]) * UL_LOG2E;
        const float t1 = (w0 * l1[py * w + i0] + w1 * l1[py * w + i1]) * UL_LOG2E;
        const float b1 = (w0 * l1[py1 * w + i0] + w1 * l1[py1 * w + i1]) * UL_LOG2E;
        for (int qy0 = ys + par; qy0 < ye; qy0 += 8) {
            int64_t lab[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) lab[u] = Yn[(int64_t)min(qy0 + 2 * u, ye - 1) * W + x];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int qy = qy0 + 2 * u;
                const float h1 = sy * (float)qy - (float)py, h0 = 1.f - h1;
                const float x0 = h0 * t0 + h1 * b0, x1 = h0 * t1 + h1 * b1;
                const float m = fmaxf(x0, x1);
                const float e0 = __builtin_amdgcn_exp2f(x0 - m), e1 = __builtin_amdgcn_exp2f(x1 - m);
                const bool valid = qy < ye && (lab[u] == 0 || lab[u] == 1);
                const bool fg = lab[u] == 1;
                const float p = (fg ? e1 : e0) * __builtin_amdgcn_rcpf(e0 + e1);
                const float lp = __log2f(p) * UL_LN2;
                const float om = 1.f - p;
                const float mod = (gamma == 2.f) ? om * om : powf(om, gamma);
                const float ay = fg ? 1.f - alpha : alpha;
                if (valid) {
                    num -= ay * (mod * lp);
                    den += ay;
                }
            }
        }
    }
    float* red = sm;
    const float tn = block_sum_256(num, red);
    const float td = block_sum_256(den, red);
    if (tid == 0) {
        const int64_t o = (((int64_t)n * h + py) * xsplit + xsid) * 2;
        g.part[o] = tn;
        g.part[o + 1] = td;
    }
}

template <bool GRAD, int EM, int RSF>
__global__ __launch_bounds__(UL_T) void wce_chain_kernel(LossArgs g) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    wce_chain_body<GRAD, EM, RSF>(g, blockIdx.y, blockIdx.x, sm);
}

__global__ __launch_bounds__(UL_T) void focal_kernel(LossArgs g) {
    __shared__ float red[8];
    focal_body(g, blockIdx.y, blockIdx.x, red);
}

// exact weighted CE alone (geometries the chain path does not cover)
template <bool GRAD, int RS>
__global__ __launch_bounds__(UL_T) void upsample_loss_kernel(LossArgs g) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    upsample_loss_body<GRAD, RS>(g, blockIdx.y, blockIdx.x, sm);
}

// Weighted CE (blockIdx.y < N0) and binary focal (the rest) in ONE launch: the two losses read
// different logit tensors and label maps and are independent, so their blocks run side by side.
template <bool GRAD, int RS>
__global__ __launch_bounds__(UL_T) void upsample_loss_pair_kernel(LossArgs g0, LossArgs g1, int N0, int nb0, int nb1) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    if ((int)blockIdx.y < N0) {
        if ((int)blockIdx.x < nb0) upsample_loss_body<GRAD, RS>(g0, blockIdx.y, blockIdx.x, sm);
    } else {
        if ((int)blockIdx.x < nb1) focal_body(g1, blockIdx.y - N0, blockIdx.x, sm);
    }
}

// Weighted CE on the chain path (blockIdx.y < N0) and binary focal (the rest) in ONE launch.
template <bool GRAD, int EM, int RSF>
__global__ __launch_bounds__(UL_T) void wce_chain_pair_kernel(LossArgs g0, LossArgs g1, int N0, int nb0, int nb1) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    BSTAMP(0);
    if ((int)blockIdx.y < N0) {
        if ((int)blockIdx.x < nb0) wce_chain_body<GRAD, EM, RSF>(g0, blockIdx.y, blockIdx.x, sm);
    } else {
        if ((int)blockIdx.x < nb1) focal_body(g1, blockIdx.y - N0, blockIdx.x, sm);
    }
    BSTAMP(1);
}
