// Per-facet MLP cin -> hidden -> cout with bf16-STORED activations (FGC_CONV_BF16 companion of fgc_mlp.hip; the
// reference is fp32 only: /root/reference/Code/model.py:763-769,937-941, train.py:409-427).
//
// x (and dx in backward) are bf16 tensors; W1, b1, W2, b2 and every gradient of them stay fp32 (master weights).  The
// matrix products with the 1024-wide hidden layer run on v_mfma_f32_16x16x32_bf16 with fp32 accumulation; bias, leaky
// ReLU, the forward 1024 -> 3 layer and every sum over nodes stay fp32 on the vector ALU / in the accumulators.  In the
// backward pass the two K = 3 / N = 3 products of that layer ride on the matrix pipe too (g = dy W2^T with hi + lo
// operand pairs, dW2 = hact^T dy with hact rounded to bf16 and dy as a hi + lo pair): they were a third of the vector
// work of kernels that the vector ALU bounds.  The hidden activation is never stored (recomputed in backward), exactly
// as in the fp32 kernels.
//
// Fragment layouts of v_mfma_f32_16x16x32_bf16 (MI355X guide): lane l = (lr = l & 15, lq = l >> 4) holds
//   A[row lr][k = 8*lq + j],  B[k = 8*lq + j][col lr],  j = 0..7  (16 bytes each);  C/D: col = lr, row = 4*lq + reg.
#include <algorithm>

#include "fgc_reduce.h"
#include "fgc_mlp_split.h"
#include "fgc_pack.h"
#include "fgc_split.h"

namespace fgc {

constexpr int MB_THREADS = 256;
constexpr int MB_FWD_T = 64;       // rows per forward tile (as fgc_mlp.hip: fgc_mlp_num_partials counts these)
constexpr int MB_FWD_RT = MB_FWD_T / 16;
constexpr int MB_T = 32;           // rows per backward tile
constexpr int MB_RT = MB_T / 16;

__global__ void mlp_pack_bf16_kernel(const float* __restrict__ W1, unsigned short* __restrict__ Wp, int cin, int hidden) {
    mlp_pack_bf16_body(W1, Wp, cin, hidden, blockIdx.x, gridDim.x);
}

// (operand packs of the backward pass: fgc_pack.h)
__device__ __forceinline__ u32x4 mb_w2_frag(const u32x4* __restrict__ W2p, int ct, int lane) {
    const u32x4 v = W2p[ct * 32 + (lane & 31)];
    return lane < 32 ? v : u32x4{0u, 0u, 0u, 0u};
}
// the three operand packs of the backward pass in one launch: blocks [0, nb) W1 fragments, [nb, 2 nb) W1 in dx order, the rest W2
__global__ void mlp_pack_bwd_bf16_kernel(const float* __restrict__ W1, const float* __restrict__ W2, unsigned short* __restrict__ Wp,
                                         unsigned short* __restrict__ Wd, u32x4* __restrict__ W2p, int cin, int hidden, int cout,
                                         int nb) {
    const int b = blockIdx.x;
    if (b < nb) mlp_pack_bf16_body(W1, Wp, cin, hidden, b, nb);
    else if (b < 2 * nb) mlp_pack_w1dx_bf16_body(W1, Wd, cin, hidden, b - nb, nb);
    else mlp_pack_w2_bf16_body(W2, W2p, hidden, cout, b - 2 * nb);
}
// A fragment of that product from a lane's dy row (o = 0..2)
__device__ __forceinline__ u32x4 mb_dy_frag(float d0, float d1, float d2, int lq) {
    const unsigned h01 = f2_to_bf2(d0, d1), h2x = f2_to_bf2(d2, 0.f);
    const f32x2c f01 = bf2_to_f2(h01), f2x = bf2_to_f2(h2x);
    const unsigned l01 = f2_to_bf2(d0 - f01[0], d1 - f01[1]), l2x = f2_to_bf2(d2 - f2x[0], 0.f);
    // lq 0: {hi0 hi1 | hi2 lo0 | lo1 lo2 | 0}    lq 1: {hi0 hi1 | hi2 0 | 0 | 0}
    const unsigned w1 = lq == 0 ? ((h2x & 0xFFFFu) | (l01 << 16)) : (h2x & 0xFFFFu);
    const unsigned w2 = lq == 0 ? ((l01 >> 16) | (l2x << 16)) : 0u;
    return lq < 2 ? u32x4{h01, w1, w2, 0u} : u32x4{0u, 0u, 0u, 0u};
}

// ---------------------------------------------------------------------------------------------
// forward.  A workgroup = 64 rows; wave w computes the hidden column tiles w, w+4, ... for all four 16-row tiles and folds
// them into the second layer on the spot.  The x fragments (16 bytes per lane, k-step and row tile) come straight from
// global memory once and stay in registers.  KS = cin / 32.
// ---------------------------------------------------------------------------------------------
template <int KS, int CO>
__global__ __launch_bounds__(MB_THREADS, 4) void mlp_fwd_bf16_kernel(const unsigned short* __restrict__ x, int n, int hidden,
                                                                     int cout, const u32x4* __restrict__ Wp16,
                                                                     const float* __restrict__ b1,
                                                                     const float* __restrict__ W2,
                                                                     const float* __restrict__ b2, float alpha,
                                                                     float* __restrict__ y, float* __restrict__ abs_partial) {
    __shared__ float ypart[4 * MB_FWD_T * 4];
    __shared__ float red[4];
    constexpr int CIN = KS * 32;
    const int row0 = blockIdx.x * MB_FWD_T;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, lr = lane & 15, lq = lane >> 4;
    const int nct = hidden >> 4;
    u32x4 a[MB_FWD_RT][KS];
#pragma unroll
    for (int r = 0; r < MB_FWD_RT; ++r) {
        const int row = min(row0 + r * 16 + lr, n - 1);
#pragma unroll
        for (int ks = 0; ks < KS; ++ks)
            a[r][ks] = *reinterpret_cast<const u32x4*>(x + (size_t)row * CIN + ks * 32 + 8 * lq);
    }
    float yp[MB_FWD_RT][4][CO];
#pragma unroll
    for (int r = 0; r < MB_FWD_RT; ++r)
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int o = 0; o < CO; ++o) yp[r][t][o] = 0.f;

    u32x4 bnext[KS];
    float bbn = 0.f, w2n[CO];
    auto fetch_weights = [&](int ct) {   // (clamped: always a valid load)
        ct = min(ct, nct - 1);
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) bnext[ks] = Wp16[((size_t)ks * nct + ct) * 64 + lane];
        bbn = b1[ct * 16 + lr];
#pragma unroll
        for (int o = 0; o < CO; ++o) w2n[o] = W2[(size_t)(ct * 16 + lr) * cout + min(o, cout - 1)];
    };
    fetch_weights(wave);
    for (int ct = wave; ct < nct; ct += 4) {
        u32x4 bcur[KS];
        float w2[CO];
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) bcur[ks] = bnext[ks];
        const float bb = bbn;
#pragma unroll
        for (int o = 0; o < CO; ++o) w2[o] = o < cout ? w2n[o] : 0.f;
        fetch_weights(ct + 4);
        f32x4 h[MB_FWD_RT];
#pragma unroll
        for (int r = 0; r < MB_FWD_RT; ++r) h[r] = f32x4{bb, bb, bb, bb};   // the bias rides in the accumulator
#pragma unroll
        for (int ks = 0; ks < KS; ++ks)
#pragma unroll
            for (int r = 0; r < MB_FWD_RT; ++r)
                h[r] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a[r][ks]),
                                                              __builtin_bit_cast(bf16x8, bcur[ks]), h[r], 0, 0, 0);
#pragma unroll
        for (int r = 0; r < MB_FWD_RT; ++r)
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                float v = h[r][t];
                v = lrelu01(v, alpha);    // leaky ReLU for 0 <= alpha <= 1 (checked by the host)
#pragma unroll
                for (int o = 0; o < CO; ++o) yp[r][t][o] = fmaf(v, w2[o], yp[r][t][o]);
            }
    }
    // reduce over the 16 column lanes, then over the four waves (fixed order)
#pragma unroll
    for (int r = 0; r < MB_FWD_RT; ++r)
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int o = 0; o < CO; ++o) {
                float v = yp[r][t][o];
                FGC_ROW16_SUM(v);
                yp[r][t][o] = v;
            }
    if (lr == 0) {
#pragma unroll
        for (int r = 0; r < MB_FWD_RT; ++r)
#pragma unroll
            for (int t = 0; t < 4; ++t)
#pragma unroll
                for (int o = 0; o < CO; ++o) ypart[(wave * MB_FWD_T + r * 16 + lq * 4 + t) * 4 + o] = yp[r][t][o];
    }
    __syncthreads();
    float asum = 0.f;
    for (int t = threadIdx.x; t < MB_FWD_T * cout; t += MB_THREADS) {
        const int r = t / cout, o = t % cout;
        const int row = row0 + r;
        if (row < n) {
            float v = b2[o];
            v += ypart[(0 * MB_FWD_T + r) * 4 + o];
            v += ypart[(1 * MB_FWD_T + r) * 4 + o];
            v += ypart[(2 * MB_FWD_T + r) * 4 + o];
            v += ypart[(3 * MB_FWD_T + r) * 4 + o];
            y[(size_t)row * cout + o] = v;
            asum += fabsf(v);
        }
    }
    if (abs_partial) {
        for (int off = 32; off > 0; off >>= 1) asum += __shfl_xor(asum, off);
        if (lane == 0) red[wave] = asum;
        __syncthreads();
        if (threadIdx.x == 0) abs_partial[blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
    }
}

// ---------------------------------------------------------------------------------------------
// fp32 operands on the bf16 matrix pipe (the fp32 network's MLP).  v_mfma_f32_16x16x4_f32 runs at the vector FMA rate and
// shares the SIMD with the vector ALU; the bf16 pipe is 16 times faster and runs beside it.  An
// fp32 value is split into three bf16 terms v = v0 + v1 + v2 (v0 = bf16(v), v1 = bf16(v - v0), v2 = bf16(v - v0 - v1): 3 x 8
// significand bits >= the 24 of fp32, same exponent range), and a product a b is the six bf16 MFMAs a0 b2 + a2 b0 + a1 b1 +
// a0 b1 + a1 b0 + a0 b0 accumulated in fp32, smallest terms first.  Each bf16 x bf16 product is exact in fp32; the three
// terms left out (a1 b2, a2 b1, a2 b2) are below 2^-25 of the product: the result differs from the fp32-MFMA kernel's by
// summation order only (tests/test_gpu_ops.py compares the two).  FGC_NO_MLP_SPLIT=1 keeps the fp32 MFMA kernels.
// ---------------------------------------------------------------------------------------------
// W1 [cin, hidden] fp32 -> three planes of B fragments [plane][k-step][column tile][lane][8] bf16
__global__ void mlp_pack_split_kernel(const float* __restrict__ W1, unsigned short* __restrict__ Wp, int cin, int hidden) {
    mlp_pack_split_body(W1, Wp, cin, hidden, blockIdx.x, gridDim.x);
}

// forward: the bf16 kernel's structure (a workgroup = 64 rows, wave w walks the hidden column tiles w, w + 4, ...); x is
// fp32 and is split once per workgroup, the weights come split from mlp_pack_split_kernel.
template <int KS, int CO>
__global__ __launch_bounds__(MB_THREADS, 2) void mlp_fwd_split_kernel(const float* __restrict__ x, int n, int hidden, int cout,
                                                                      const u32x4* __restrict__ Wp16,
                                                                      const float* __restrict__ b1,
                                                                      const float* __restrict__ W2,
                                                                      const float* __restrict__ b2, float alpha,
                                                                      float* __restrict__ y, float* __restrict__ abs_partial) {
    __shared__ float ypart[4 * MB_FWD_T * 4];
    __shared__ float red[4];
    constexpr int CIN = KS * 32;
    const int row0 = blockIdx.x * MB_FWD_T;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, lr = lane & 15, lq = lane >> 4;
    const int nct = hidden >> 4;
    const size_t plane = (size_t)KS * nct * 64;            // u32x4 per plane
    u32x4 a[MB_FWD_RT][KS][3];
#pragma unroll
    for (int r = 0; r < MB_FWD_RT; ++r) {
        const int row = min(row0 + r * 16 + lr, n - 1);
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            const f32x4* src = reinterpret_cast<const f32x4*>(x + (size_t)row * CIN + ks * 32 + 8 * lq);
            u32x2 lo[3], hi[3];
            split3(src[0], lo[0], lo[1], lo[2]);
            split3(src[1], hi[0], hi[1], hi[2]);
#pragma unroll
            for (int p = 0; p < 3; ++p) a[r][ks][p] = u32x4{lo[p][0], lo[p][1], hi[p][0], hi[p][1]};
        }
    }
    float yp[MB_FWD_RT][4][CO];
#pragma unroll
    for (int r = 0; r < MB_FWD_RT; ++r)
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int o = 0; o < CO; ++o) yp[r][t][o] = 0.f;

    u32x4 bnext[KS][3];
    float bbn = 0.f, w2n[CO];
    auto fetch_weights = [&](int ct) {   // (clamped: always a valid load)
        ct = min(ct, nct - 1);
#pragma unroll
        for (int ks = 0; ks < KS; ++ks)
#pragma unroll
            for (int p = 0; p < 3; ++p) bnext[ks][p] = Wp16[p * plane + ((size_t)ks * nct + ct) * 64 + lane];
        bbn = b1[ct * 16 + lr];
#pragma unroll
        for (int o = 0; o < CO; ++o) w2n[o] = W2[(size_t)(ct * 16 + lr) * cout + min(o, cout - 1)];
    };
    fetch_weights(wave);
    for (int ct = wave; ct < nct; ct += 4) {
        u32x4 bcur[KS][3];
        float w2[CO];
#pragma unroll
        for (int ks = 0; ks < KS; ++ks)
#pragma unroll
            for (int p = 0; p < 3; ++p) bcur[ks][p] = bnext[ks][p];
        const float bb = bbn;
#pragma unroll
        for (int o = 0; o < CO; ++o) w2[o] = o < cout ? w2n[o] : 0.f;
        fetch_weights(ct + 4);
        f32x4 h[MB_FWD_RT];
#pragma unroll
        // (the bias rides in the accumulator: the small terms of the split product then lose what lies below the last place of
        //  max(|b1|, |x W1|) - below the last place of the result; one vector add per hidden activation less)
        for (int r = 0; r < MB_FWD_RT; ++r) h[r] = f32x4{bb, bb, bb, bb};
#pragma unroll
        for (int ks = 0; ks < KS; ++ks)
#pragma unroll
            for (int r = 0; r < MB_FWD_RT; ++r) h[r] = mfma_split(a[r][ks], bcur[ks], h[r]);
#pragma unroll
        for (int r = 0; r < MB_FWD_RT; ++r)
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                float v = h[r][t];
                v = lrelu01(v, alpha);
#pragma unroll
                for (int o = 0; o < CO; ++o) yp[r][t][o] = fmaf(v, w2[o], yp[r][t][o]);
            }
    }
    // reduce over the 16 column lanes, then over the four waves (fixed order)
#pragma unroll
    for (int r = 0; r < MB_FWD_RT; ++r)
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int o = 0; o < CO; ++o) {
                float v = yp[r][t][o];
                FGC_ROW16_SUM(v);
                yp[r][t][o] = v;
            }
    if (lr == 0) {
#pragma unroll
        for (int r = 0; r < MB_FWD_RT; ++r)
#pragma unroll
            for (int t = 0; t < 4; ++t)
#pragma unroll
                for (int o = 0; o < CO; ++o) ypart[(wave * MB_FWD_T + r * 16 + lq * 4 + t) * 4 + o] = yp[r][t][o];
    }
    __syncthreads();
    float asum = 0.f;
    for (int t = threadIdx.x; t < MB_FWD_T * cout; t += MB_THREADS) {
        const int r = t / cout, o = t % cout;
        const int row = row0 + r;
        if (row < n) {
            float v = b2[o];
            v += ypart[(0 * MB_FWD_T + r) * 4 + o];
            v += ypart[(1 * MB_FWD_T + r) * 4 + o];
            v += ypart[(2 * MB_FWD_T + r) * 4 + o];
            v += ypart[(3 * MB_FWD_T + r) * 4 + o];
            y[(size_t)row * cout + o] = v;
            asum += fabsf(v);
        }
    }
    if (abs_partial) {
        for (int off = 32; off > 0; off >>= 1) asum += __shfl_xor(asum, off);
        if (lane == 0) red[wave] = asum;
        __syncthreads();
        if (threadIdx.x == 0) abs_partial[blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
    }
}

// ---------------------------------------------------------------------------------------------
// backward = two kernels, because the two big products want opposite loop orders and the bf16 MFMA makes recomputing
// the hidden layer nearly free (a fused kernel with the fp32 kernel's structure kept ~200 registers of partial sums
// alive and spilled; giving the waves of a workgroup the same rows cost three barriers per 32-row tile):
//   mlp_bwd_dx_bf16_kernel   row major: a wave owns 32 rows and walks ALL hidden columns, 32 at a time:
//        h^T = W1^T x^T + b1 and g^T = W2 dy^T (MFMA, operands swapped), dh = g * lrelu'(h) (vector ALU, fp32),
//        dx += dh W1^T (MFMA; the transposed C layout of dh IS the A layout, no LDS).  dx leaves as bf16, complete: no
//        partial slabs.
//   mlp_bwd_w_bf16_kernel    column major: a wave owns 64 hidden columns (blockIdx.y) and walks its share of the
//        32-row tiles with dW1 / db1 / dW2 partial sums in registers:
//        dW1 += x^T dh: K = the 32 rows of the tile, ONE MFMA per (16 input channels, 16 hidden columns); B = dh
//        straight out of the registers (k order: row tile, then 4*lq + reg), A = x^T out of a wave-private LDS tile
//        read in that same k order.  One fp32 slab per wave, summed in fixed order by reduce_jobs.
// No barrier inside either loop: nothing is shared between the waves of a workgroup except read-only weights.
// MT = cin / 16.
// ---------------------------------------------------------------------------------------------
constexpr int MBB_THREADS = 256;
constexpr int MBB_WAVES = MBB_THREADS / 64;
constexpr int MBW_HCW = 64;               // hidden columns per wave of the parameter-gradient kernel
constexpr int MBW_CT = MBW_HCW / 16;

__device__ __forceinline__ void mb_load_rows(const unsigned short* __restrict__ x, const float* __restrict__ dy, int n,
                                             int cout, int row0, int cin, int lane, u32x4* ax /* [MB_RT][KS] */, int ks_n,
                                             float* dyw) {
    const int lr = lane & 15, lq = lane >> 4;
#pragma unroll
    for (int r = 0; r < MB_RT; ++r) {
        const int row = row0 + r * 16 + lr;
        for (int ks = 0; ks < ks_n; ++ks) {
            const u32x4 v = *reinterpret_cast<const u32x4*>(x + (size_t)min(row, n - 1) * cin + ks * 32 + 8 * lq);
            ax[r * ks_n + ks] = row < n ? v : u32x4{0u, 0u, 0u, 0u};
        }
    }
    // dy rows of the tile into the wave's LDS tile [32][4]: lane l < 32 owns row l
    const int rr = row0 + (lane & 31);
    f32x4 v = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int o = 0; o < 4; ++o) {
        const float g = dy[(size_t)min(rr, n - 1) * cout + min(o, cout - 1)];
        v[o] = (rr < n && o < cout) ? g : 0.f;
    }
    if (lane < 32) *reinterpret_cast<f32x4*>(dyw + lane * 4) = v;
}

template <int MT>
__global__ __launch_bounds__(MBB_THREADS, MT <= 2 ? 3 : 2) void mlp_bwd_dx_bf16_kernel(
    const unsigned short* __restrict__ x, const float* __restrict__ dy, int n, int hidden, int cout,
    const u32x4* __restrict__ Wp16, const u32x4* __restrict__ W1d /* mlp_pack_w1dx_bf16_kernel */,
    const float* __restrict__ b1, const u32x4* __restrict__ W2p /* mlp_pack_w2_bf16_kernel */, float alpha,
    unsigned short* __restrict__ dx) {
    constexpr int CIN = MT * 16;
    constexpr int KS = CIN / 32;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, lr = lane & 15, lq = lane >> 4;
    const int nct = hidden >> 4;
    const int tile = blockIdx.x * MBB_WAVES + wave;
    const int row0 = tile * MB_T;
    if (row0 >= n) return;                    // (whole wave; nothing below is shared between waves)
    u32x4 ax[MB_RT * KS], gA[MB_RT];
#pragma unroll
    for (int r = 0; r < MB_RT; ++r) {
        const int row = row0 + r * 16 + lr;
        const size_t rc = (size_t)min(row, n - 1);
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            const u32x4 v = *reinterpret_cast<const u32x4*>(x + rc * CIN + ks * 32 + 8 * lq);
            ax[r * KS + ks] = row < n ? v : u32x4{0u, 0u, 0u, 0u};
        }
        // this lane's dy row as the A fragment of g = dy W2^T (rows past n: zero, so their dh is zero)
        float d[3];
#pragma unroll
        for (int o = 0; o < 3; ++o) d[o] = (row < n && o < cout) ? dy[rc * cout + min(o, cout - 1)] : 0.f;
        gA[r] = mb_dy_frag(d[0], d[1], d[2], lq);
    }
    f32x4 dxacc[MB_RT][MT];
#pragma unroll
    for (int r = 0; r < MB_RT; ++r)
#pragma unroll
        for (int m = 0; m < MT; ++m) dxacc[r][m] = f32x4{0.f, 0.f, 0.f, 0.f};

    // weights of one pair of column tiles (W1 fragments, b1, W2 fragments, W1 rows for the dx product): clamped, so that the
    // pair after the last is a valid (unused) load; fetched one pair ahead.
    // The two products that make dh are computed TRANSPOSED (operands swapped: h^T = W1^T x^T, g^T = W2 dy^T): in the C
    // layout a lane then holds ONE row (its lr) and the hidden columns 4*lq + t of a tile - with the two tiles of a pair,
    // eight k values of the A operand of dx += dh W1^T, straight out of the registers.  k slot j = c2*4 + t of lane group
    // lq stands for hidden column c2*16 + 4*lq + t; the W1 rows (B operand) are loaded in that same order.  No LDS.
    struct PairW {
        u32x4 bw[2][KS];
        f32x4 bb[2];
        u32x4 bg[2];
        u32x4 bt[MT];
    };
    auto fetch = [&](int pp, PairW& w) {
        pp = min(pp, (nct >> 1) - 1);
#pragma unroll
        for (int c2 = 0; c2 < 2; ++c2) {
            const int ct = pp * 2 + c2;
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) w.bw[c2][ks] = Wp16[((size_t)ks * nct + ct) * 64 + lane];
            w.bb[c2] = *reinterpret_cast<const f32x4*>(b1 + ct * 16 + 4 * lq);
            w.bg[c2] = mb_w2_frag(W2p, ct, lane);
        }
#pragma unroll
        for (int m = 0; m < MT; ++m) w.bt[m] = W1d[((size_t)pp * MT + m) * 64 + lane];
    };
    auto pair = [&](const PairW& w) {
        f32x4 d[2][MB_RT];
#pragma unroll
        for (int c2 = 0; c2 < 2; ++c2) {
            f32x4 h[MB_RT], g[MB_RT];
#pragma unroll
            for (int r = 0; r < MB_RT; ++r) h[r] = w.bb[c2];
#pragma unroll
            for (int ks = 0; ks < KS; ++ks)
#pragma unroll
                for (int r = 0; r < MB_RT; ++r)
                    h[r] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, w.bw[c2][ks]),
                                                                  __builtin_bit_cast(bf16x8, ax[r * KS + ks]), h[r], 0, 0, 0);
#pragma unroll
            for (int r = 0; r < MB_RT; ++r)
                g[r] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, w.bg[c2]), __builtin_bit_cast(bf16x8, gA[r]),
                                                              f32x4{0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
#pragma unroll
            for (int r = 0; r < MB_RT; ++r)
#pragma unroll
                for (int t = 0; t < 4; ++t) d[c2][r][t] = g[r][t] * lrelu01_slope(h[r][t], alpha);
        }
#pragma unroll
        for (int r = 0; r < MB_RT; ++r) {
            const u32x4 ad = u32x4{f2_to_bf2(d[0][r][0], d[0][r][1]), f2_to_bf2(d[0][r][2], d[0][r][3]),
                                   f2_to_bf2(d[1][r][0], d[1][r][1]), f2_to_bf2(d[1][r][2], d[1][r][3])};
#pragma unroll
            for (int m = 0; m < MT; ++m)
                dxacc[r][m] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, ad), __builtin_bit_cast(bf16x8, w.bt[m]),
                                                                     dxacc[r][m], 0, 0, 0);
        }
    };
    // cout > 3 (a fourth output column) is folded in by a second sweep below: the network's heads have three
    PairW wa, wb;
    const int npairs = nct >> 1;
    fetch(0, wa);
    // sched_barrier: keep the program order "request the next pair, then work on this one" (left alone the scheduler sinks
    // every request to the end of the iteration and waits for all of them at the top of the next: no load ever overlaps
    // a product)
#pragma unroll 1
    for (int pp = 0; pp < npairs; pp += 2) {
        fetch(pp + 1, wb);
        __builtin_amdgcn_sched_barrier(0);
        pair(wa);
        __builtin_amdgcn_sched_barrier(0);
        fetch(pp + 2, wa);
        __builtin_amdgcn_sched_barrier(0);
        pair(wb);                 // (hidden % 256 == 0: the pair count is even; a conditional here lets the compiler sink
                                  //  the loads of wb into the branch, next to their use)
        __builtin_amdgcn_sched_barrier(0);
    }
    // C layout: column = input channel lr of tile m, rows 4*lq + t
#pragma unroll
    for (int r = 0; r < MB_RT; ++r)
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const int row = row0 + r * 16 + lq * 4 + t;
            if (row < n) {
#pragma unroll
                for (int m = 0; m < MT; ++m) dx[(size_t)row * CIN + m * 16 + lr] = f_to_bf(dxacc[r][m][t]);
            }
        }
}

template <int MT, int CO>
__global__ __launch_bounds__(MBB_THREADS, 2) void mlp_bwd_w_bf16_kernel(
    const unsigned short* __restrict__ x, const float* __restrict__ dy, int n, int hidden, int cout,
    const u32x4* __restrict__ Wp16, const float* __restrict__ b1, const u32x4* __restrict__ W2p, float alpha,
    float* __restrict__ dW1_slab /* [walkers][cin][hidden] */, float* __restrict__ db1_slab /* [walkers][hidden] */,
    float* __restrict__ dW2_slab /* [walkers][hidden][4] */, float* __restrict__ db2_slab /* [walkers][4] */) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    constexpr int CIN = MT * 16;
    constexpr int KS = CIN / 32;
    constexpr int XTS = MB_T * 2 + 8;         // bytes per row (= input channel) of a wave's transposed x tile
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, lr = lane & 15, lq = lane >> 4;
    char* xT = smem_raw + wave * (CIN * XTS + MB_T * 16);
    float* dyw = reinterpret_cast<float*>(xT + CIN * XTS);
    unsigned short* xT16 = reinterpret_cast<unsigned short*>(xT);
    const int hc0 = blockIdx.y * MBW_HCW;
    const int nct = hidden >> 4;
    const int ntiles = (n + MB_T - 1) / MB_T;
    const int walker = blockIdx.x * MBB_WAVES + wave, nwalkers = gridDim.x * MBB_WAVES;

    u32x4 bw[MBW_CT][KS], bg[MBW_CT];
    float bb[MBW_CT];
#pragma unroll
    for (int c = 0; c < MBW_CT; ++c) {
        const int ct = (hc0 >> 4) + c;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) bw[c][ks] = Wp16[((size_t)ks * nct + ct) * 64 + lane];
        bb[c] = b1[ct * 16 + lr];
        bg[c] = mb_w2_frag(W2p, ct, lane);
    }
    // dW1acc: C layout of x^T dh (column = hidden column lr, row = input channel 4*lq + t of tile m)
    // dW2acc: C layout of hact^T dy (column = output o = lr, row = hidden column 4*lq + t of tile c): lanes lr < cout count
    f32x4 dW1acc[MBW_CT][MT], dW2acc[MBW_CT];
    float db1acc[MBW_CT], db2acc = 0.f;
#pragma unroll
    for (int c = 0; c < MBW_CT; ++c) {
        db1acc[c] = 0.f;
        dW2acc[c] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int m = 0; m < MT; ++m) dW1acc[c][m] = f32x4{0.f, 0.f, 0.f, 0.f};
    }

#pragma unroll 1
    for (int tile = walker; tile < ntiles; tile += nwalkers) {
        const int row0 = tile * MB_T;
        u32x4 ax[MB_RT * KS];
        mb_load_rows(x, dy, n, cout, row0, CIN, lane, ax, KS, dyw);
        // x^T of the tile for the dW1 product
#pragma unroll
        for (int r = 0; r < MB_RT; ++r)
#pragma unroll
            for (int ks = 0; ks < KS; ++ks)
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int c = ks * 32 + 8 * lq + 2 * e;
                    xT16[(c * XTS) / 2 + r * 16 + lr] = (unsigned short)(ax[r * KS + ks][e] & 0xFFFFu);
                    xT16[((c + 1) * XTS) / 2 + r * 16 + lr] = (unsigned short)(ax[r * KS + ks][e] >> 16);
                }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        // dy of the tile in the two operand layouts it is needed in:
        //   gA[r]    A of g = dy W2^T: this lane's row r*16 + lr (mb_dy_frag)
        //   dyh/dyl  B of dW2 += hact^T dy: lane (o = lr, lq), k slot j <-> row (j >> 2) * 16 + 4*lq + (j & 3) - the row
        //            order in which a lane holds its hidden column in the C layout - as bf16 hi and lo parts
        u32x4 gA[MB_RT], dyh, dyl;
#pragma unroll
        for (int r = 0; r < MB_RT; ++r) {
            const f32x4 d = *reinterpret_cast<const f32x4*>(dyw + (r * 16 + lr) * 4);
            gA[r] = mb_dy_frag(d[0], d[1], d[2], lq);
        }
        {
            float v[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const float d = dyw[((j >> 2) * 16 + 4 * lq + (j & 3)) * 4 + (lr & 3)];
                v[j] = lr < CO ? d : 0.f;
            }
            if (blockIdx.y == 0) db2acc += ((v[0] + v[1]) + (v[2] + v[3])) + ((v[4] + v[5]) + (v[6] + v[7]));
            unsigned hh[4], ll[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                hh[q] = f2_to_bf2(v[2 * q], v[2 * q + 1]);
                const f32x2c f = bf2_to_f2(hh[q]);
                ll[q] = f2_to_bf2(v[2 * q] - f[0], v[2 * q + 1] - f[1]);
            }
            dyh = u32x4{hh[0], hh[1], hh[2], hh[3]};
            dyl = u32x4{ll[0], ll[1], ll[2], ll[3]};
        }
        u32x4 af[MT];
#pragma unroll
        for (int m = 0; m < MT; ++m) {
            const char* row = xT + (m * 16 + lr) * XTS;
            const u32x2 lo = *reinterpret_cast<const u32x2*>(row + lq * 8);          // rows 4*lq .. +3
            const u32x2 hi = *reinterpret_cast<const u32x2*>(row + 32 + lq * 8);     // rows 16 + 4*lq .. +3
            af[m] = u32x4{lo[0], lo[1], hi[0], hi[1]};
        }
#pragma unroll
        for (int c = 0; c < MBW_CT; ++c) {
            __builtin_amdgcn_sched_barrier(0);    // one column tile at a time
            f32x4 h[MB_RT], g[MB_RT];
#pragma unroll
            for (int r = 0; r < MB_RT; ++r) h[r] = f32x4{bb[c], bb[c], bb[c], bb[c]};
#pragma unroll
            for (int ks = 0; ks < KS; ++ks)
#pragma unroll
                for (int r = 0; r < MB_RT; ++r)
                    h[r] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, ax[r * KS + ks]),
                                                                  __builtin_bit_cast(bf16x8, bw[c][ks]), h[r], 0, 0, 0);
#pragma unroll
            for (int r = 0; r < MB_RT; ++r)
                g[r] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, gA[r]), __builtin_bit_cast(bf16x8, bg[c]),
                                                              f32x4{0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
            // lrelu(pre) = pre * lrelu'(pre): one multiply instead of a multiply and a max
            f32x4 ha[MB_RT];
#pragma unroll
            for (int r = 0; r < MB_RT; ++r)
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    const float slope = lrelu01_slope(h[r][t], alpha);
                    ha[r][t] = h[r][t] * slope;
                    g[r][t] *= slope;
                    db1acc[c] += g[r][t];
                }
            // k = r*16 + 4*lq + t: element j of the fragments is (r = j >> 2, t = j & 3)
            const u32x4 bf = u32x4{f2_to_bf2(g[0][0], g[0][1]), f2_to_bf2(g[0][2], g[0][3]), f2_to_bf2(g[1][0], g[1][1]),
                                   f2_to_bf2(g[1][2], g[1][3])};
            const u32x4 hf = u32x4{f2_to_bf2(ha[0][0], ha[0][1]), f2_to_bf2(ha[0][2], ha[0][3]), f2_to_bf2(ha[1][0], ha[1][1]),
                                   f2_to_bf2(ha[1][2], ha[1][3])};
#pragma unroll
            for (int m = 0; m < MT; ++m)
                dW1acc[c][m] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, af[m]), __builtin_bit_cast(bf16x8, bf),
                                                                      dW1acc[c][m], 0, 0, 0);
            // dW2 += hact^T dy (smaller term first)
            dW2acc[c] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, hf), __builtin_bit_cast(bf16x8, dyl),
                                                               dW2acc[c], 0, 0, 0);
            dW2acc[c] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, hf), __builtin_bit_cast(bf16x8, dyh),
                                                               dW2acc[c], 0, 0, 0);
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");   // the next tile overwrites the LDS tiles
    }
    // parameter-gradient slabs of this wave (slab index = its walker id)
    if (blockIdx.y == 0) {   // db2[o]: lanes (lr = o, lq) hold the sums of their rows
        float v = db2acc;
        v += __shfl_xor(v, 16);
        v += __shfl_xor(v, 32);
        if (lq == 0 && lr < 4) db2_slab[walker * 4 + lr] = lr < CO ? v : 0.f;
    }
#pragma unroll
    for (int c = 0; c < MBW_CT; ++c) {
        const int col = hc0 + c * 16 + lr;
        // dW1acc C layout: column = lr (hidden column), row = 4*lq + t (input channel within tile m)
#pragma unroll
        for (int m = 0; m < MT; ++m)
#pragma unroll
            for (int t = 0; t < 4; ++t)
                dW1_slab[((size_t)walker * CIN + m * 16 + lq * 4 + t) * hidden + col] = dW1acc[c][m][t];
        float v = db1acc[c];
        v += __shfl_xor(v, 16);
        v += __shfl_xor(v, 32);
        if (lq == 0) db1_slab[(size_t)walker * hidden + col] = v;
        // dW2acc C layout: column = lr (output o), row = 4*lq + t (hidden column within tile c)
        if (lr < 4) {
#pragma unroll
            for (int t = 0; t < 4; ++t)
                dW2_slab[((size_t)walker * hidden + hc0 + c * 16 + lq * 4 + t) * 4 + lr] = lr < CO ? dW2acc[c][t] : 0.f;
        }
    }
}

// ---------------------------------------------------------------------------------------------
// The fp32 network's MLP BACKWARD on the bf16 matrix pipe (round 5).  mlp_bwd_kernel (fgc_mlp.hip) spends 61 % of its SIMD
// cycles in v_mfma_f32_16x16x4_f32, which run at the vector FMA rate and keep the vector ALU from issuing (DESIGN.md
// section 3.1).  Here the two kernels above - the structure whose loop orders fit the two big products - take fp32 x and
// write fp32 dx, and every operand of a 1024-wide product is a three-term bf16 split (fgc_mlp_split.h): x once per row
// tile, W1 at pack time, dh = g * lrelu'(h) in registers right where it is produced (4.5 vector instructions per hidden
// activation against the 64 multiply-adds it then feeds).  g = dy W2^T (K = 3) is ONE MFMA whose 32 k slots hold the six
// significant term pairs of its three products (split_k3_frag); dW2 = hact^T dy and db2 stay fp32 on the vector ALU (three
// FMAs per hidden activation - cheaper than splitting hact).  The bias starts in the accumulator of the recomputed hidden
// layer (one vector add per activation less): the small terms of the split product are then added to a sum that already
// holds b1 - they lose what lies below the last place of max(|b1|, |x W1|), which is below the last place of the result.  Same results as the fp32-MFMA kernel up to summation order
// (tests/test_gpu_ops.py holds both against float64 at the same bound).  FGC_NO_MLP_BWD_SPLIT=1 keeps mlp_bwd_kernel.
// ---------------------------------------------------------------------------------------------
// (x tile: row = [plane 0 | plane 1 | plane 2 | 32 pad bytes], so that two workgroups of four waves fit a CU's 160 KB)
#define MBS_XTS(CIN_) (3 * (CIN_) * 2 + 32)
typedef short mbs_s16x4 __attribute__((ext_vector_type(4)));
// A fragment of x^T (rows = input channels col0 .. col0 + 15, k = the tile's rows in the order a lane holds its hidden
// column in the C layout: element j <-> row (j >> 2) * 16 + 4*lq + (j & 3)) out of a row-major plane, by two transposed reads
__device__ __forceinline__ u32x4 mbs_xT_frag(const char* plane, int stride, int col0, int lq, int lr) {
    const char* a = plane + (4 * lq + (lr >> 2)) * stride + (col0 + 4 * (lr & 3)) * 2;
    const mbs_s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) mbs_s16x4*)a);
    const mbs_s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) mbs_s16x4*)(a + 16 * stride));
    const u32x2 l2 = __builtin_bit_cast(u32x2, lo), h2 = __builtin_bit_cast(u32x2, hi);
    return u32x4{l2[0], l2[1], h2[0], h2[1]};
}
__device__ __forceinline__ float keep_if(float v, bool c) { return __uint_as_float(__float_as_uint(v) & (c ? ~0u : 0u)); }
__device__ __forceinline__ f32x4 keep_if(const f32x4& v, bool c) {
    return f32x4{keep_if(v[0], c), keep_if(v[1], c), keep_if(v[2], c), keep_if(v[3], c)};
}

// The weights of a pair of column tiles - 6 KS W1 fragments for h, 2 W2 operands for g, 3 MT W1 fragments for dx: 14 KB at
// cin = 32 - are the same for every wave: the workgroup stages them in LDS ONCE per pair (each thread moves four 16-byte
// pieces, requested one pair ahead through registers) and its four waves read them from there.  Loading them per wave, as
// the bf16 kernel does with a third of the bytes, kept the texture-address path busy with 2 GB per launch and the
// double-buffered fragments took 128 registers.  One barrier per pair.
template <int MT>
__global__ __launch_bounds__(MBB_THREADS, 3) void mlp_bwd_dx_split_kernel(
    const float* __restrict__ x, const float* __restrict__ dy, int n, int hidden, int cout,
    const u32x4* __restrict__ Wp16 /* mlp_pack_split_body: three planes */, const u32x4* __restrict__ W1d /* mlp_pack_w1dx_split_body */,
    const float* __restrict__ b1, const u32x4* __restrict__ W2s /* mlp_pack_w2_split_body */, float alpha, float* __restrict__ dx) {
    constexpr int CIN = MT * 16;
    constexpr int KS = CIN / 32;
    constexpr int NF = 6 * KS + 2 + 3 * MT;     // fragments per pair: bw[c2][ks][p], bg[c2], bt[m][p]
    constexpr int NI = (NF * 64 + MBB_THREADS - 1) / MBB_THREADS;   // 16-byte pieces per thread
    __shared__ u32x4 wl[2][NI * MBB_THREADS];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, lr = lane & 15, lq = lane >> 4;
    const int nct = hidden >> 4;
    const int tile = blockIdx.x * MBB_WAVES + wave;
    const int row0 = tile * MB_T;               // (a wave past the last row keeps walking: the barriers are the workgroup's)
    const size_t wplane = (size_t)KS * nct * 64;               // u32x4 per plane of Wp16
    const size_t dplane = (size_t)(nct >> 1) * MT * 64;        // ... of W1d
    // this thread's pieces of a pair: source of pair 0 and the step from pair to pair (in u32x4)
    const u32x4* src[NI];
    int step[NI];
#pragma unroll
    for (int i = 0; i < NI; ++i) {
        const int item = threadIdx.x + i * MBB_THREADS, f = min(item >> 6, NF - 1), l = item & 63;
        if (f < 6 * KS) {
            const int c2 = f / (3 * KS), ks = (f / 3) % KS, p = f % 3;
            src[i] = Wp16 + p * wplane + ((size_t)ks * nct + c2) * 64 + l;
            step[i] = 128;
        } else if (f < 6 * KS + 2) {
            src[i] = W2s + (f - 6 * KS) * 64 + l;
            step[i] = 128;
        } else {
            const int g = f - 6 * KS - 2, m = g / 3, p = g % 3;
            src[i] = W1d + p * dplane + (size_t)m * 64 + l;
            step[i] = MT * 64;
        }
    }
    u32x4 ax[MB_RT][KS][3], gA[MB_RT];
#pragma unroll
    for (int r = 0; r < MB_RT; ++r) {
        const int row = row0 + r * 16 + lr;
        const size_t rc = (size_t)min(row, n - 1);
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            const f32x4* s4 = reinterpret_cast<const f32x4*>(x + rc * CIN + ks * 32 + 8 * lq);
            // (unconditional loads from the clamped row, zeroed by a bit mask: no load under an exec mask)
            split3_frag(keep_if(s4[0], row < n), keep_if(s4[1], row < n), ax[r][ks]);
        }
        float d[3];
#pragma unroll
        for (int o = 0; o < 3; ++o) d[o] = keep_if(dy[rc * cout + min(o, cout - 1)], row < n && o < cout);
        gA[r] = split_k3_frag(d, lq, 0);      // rows past n: zero, so their dh is zero
    }
    f32x4 dxacc[MB_RT][MT], dxlo[MB_RT][MT];      // (large and small terms apart: mfma_split2)
#pragma unroll
    for (int r = 0; r < MB_RT; ++r)
#pragma unroll
        for (int m = 0; m < MT; ++m) dxacc[r][m] = dxlo[r][m] = f32x4{0.f, 0.f, 0.f, 0.f};

    const int npairs = nct >> 1;
    u32x4 stage[NI];
    f32x4 bbn[2];
    auto request = [&](int pp) {               // (clamped: the request behind the last pair is a valid, unused load)
        pp = min(pp, npairs - 1);
#pragma unroll
        for (int i = 0; i < NI; ++i) stage[i] = src[i][(size_t)pp * step[i]];
#pragma unroll
        for (int c2 = 0; c2 < 2; ++c2) bbn[c2] = *reinterpret_cast<const f32x4*>(b1 + (pp * 2 + c2) * 16 + 4 * lq);
    };
    auto park = [&](int buf) {
#pragma unroll
        for (int i = 0; i < NI; ++i) wl[buf][threadIdx.x + i * MBB_THREADS] = stage[i];
    };
    request(0);
    park(0);
    __syncthreads();
    // as mlp_bwd_dx_bf16_kernel: both products that make dh are computed transposed, so that dh leaves the accumulators in
    // the A layout of dx += dh W1^T
#pragma unroll 1
    for (int pp = 0; pp < npairs; ++pp) {
        const u32x4* w = wl[pp & 1] + lane;
        const f32x4 bb[2] = {bbn[0], bbn[1]};
        request(pp + 1);
        __builtin_amdgcn_sched_barrier(0);
        f32x4 d[2][MB_RT];
#pragma unroll
        for (int c2 = 0; c2 < 2; ++c2) {
            f32x4 h[MB_RT], g[MB_RT];
#pragma unroll
            for (int r = 0; r < MB_RT; ++r) h[r] = bb[c2];     // (the bias rides in the accumulator: below)
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                const u32x4 bw[3] = {w[((c2 * KS + ks) * 3 + 0) * 64], w[((c2 * KS + ks) * 3 + 1) * 64], w[((c2 * KS + ks) * 3 + 2) * 64]};
#pragma unroll
                for (int r = 0; r < MB_RT; ++r) h[r] = mfma_split(bw, ax[r][ks], h[r]);
            }
            const u32x4 bg = w[(6 * KS + c2) * 64];
#pragma unroll
            for (int r = 0; r < MB_RT; ++r)
                g[r] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, bg), __builtin_bit_cast(bf16x8, gA[r]),
                                                              f32x4{0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
#pragma unroll
            for (int r = 0; r < MB_RT; ++r)
#pragma unroll
                for (int t = 0; t < 4; ++t) d[c2][r][t] = g[r][t] * lrelu01_slope(h[r][t], alpha);
        }
        u32x4 ad[MB_RT][3];
#pragma unroll
        for (int r = 0; r < MB_RT; ++r) split3_frag(d[0][r], d[1][r], ad[r]);
#pragma unroll
        for (int m = 0; m < MT; ++m) {
            const u32x4 bt[3] = {w[(6 * KS + 2 + m * 3 + 0) * 64], w[(6 * KS + 2 + m * 3 + 1) * 64], w[(6 * KS + 2 + m * 3 + 2) * 64]};
#pragma unroll
            for (int r = 0; r < MB_RT; ++r) mfma_split2(ad[r], bt, dxacc[r][m], dxlo[r][m]);
        }
        __builtin_amdgcn_sched_barrier(0);
        park((pp + 1) & 1);                   // (buffer (pp + 1) & 1 was last read in iteration pp - 1, before its barrier)
        __syncthreads();
    }
    // C layout: column = input channel lr of tile m, rows 4*lq + t
#pragma unroll
    for (int r = 0; r < MB_RT; ++r)
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const int row = row0 + r * 16 + lq * 4 + t;
            if (row < n) {
#pragma unroll
                for (int m = 0; m < MT; ++m) dx[(size_t)row * CIN + m * 16 + lr] = dxacc[r][m][t] + dxlo[r][m][t];
            }
        }
}

// Parameter gradients.  A WORKGROUP walks the 32-row tiles; its four waves own 64 hidden columns each (256 per workgroup,
// blockIdx.y picks the quarter of the hidden layer) and share everything that belongs to the rows: the tile's x split into
// three planes, dy, and the dy operand of the g product are staged in LDS once per workgroup - each wave splits a
// quarter of the rows - one tile ahead (requested before the current tile's products, parked behind them, one barrier per
// tile).  The first version gave every wave its own tiles: 16 column slices each split the same x and built the same dy
// operand (270 of its 700 vector instructions per tile) and waited for its own loads at the top of every tile.
//   LDS: 2 x { x [32 rows][plane 0 | plane 1 | plane 2 | pad] bf16, dy [32][4] fp32, dy operand [32][3 lane groups] } +
//   per wave its W1 fragments [c][ks][plane][lane] (they do not change over the walk; as registers they were 48 of 256).
#define MBS_TILE_LDS(CIN_) (MB_T * MBS_XTS(CIN_) + MB_T * 16 + MB_T * 48)
#define MBS_WG_LDS(CIN_) (2 * MBS_TILE_LDS(CIN_))
// all nine dwords of the dy operand of one row (split_k3_frag, side 0): lane group lq takes dwords 4 lq .. 4 lq + 3
__device__ __forceinline__ void split_k3_row(const float (&v)[3], unsigned (&w)[9]) {
    unsigned short p[3][3];
#pragma unroll
    for (int o = 0; o < 3; ++o) split3_scalar(v[o], p[o]);
    constexpr int PD[6] = {0, 0, 1, 1, 0, 2};
#pragma unroll
    for (int i = 0; i < 9; ++i) {
        const int s0 = 2 * i, s1 = 2 * i + 1;
        w[i] = (unsigned)p[s0 % 3][PD[s0 / 3]] | ((unsigned)p[s1 % 3][PD[s1 / 3]] << 16);
    }
}

template <int MT>
__global__ __launch_bounds__(MBB_THREADS, 2) void mlp_bwd_w_split_kernel(
    const float* __restrict__ x, const float* __restrict__ dy, int n, int hidden, int cout,
    const u32x4* __restrict__ Wp16, const float* __restrict__ b1, const u32x4* __restrict__ W2s, float alpha,
    float* __restrict__ dW1_slab /* [walkers][cin][hidden] */, float* __restrict__ db1_slab /* [walkers][hidden] */,
    float* __restrict__ dW2_slab /* [walkers][hidden][4] */, float* __restrict__ db2_slab /* [walkers][4] */) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    constexpr int CIN = MT * 16;
    constexpr int KS = CIN / 32;
    static_assert(KS == 1, "staging below moves one 16-byte piece of a 32-channel row per lane");
    constexpr int XTS = MBS_XTS(CIN);         // row stride of the x tile, == 32 bytes mod 64 (conflict-free transposed reads)
    constexpr int XPL = CIN * 2;              // byte offset of a plane within a row
    constexpr int TILE = MBS_TILE_LDS(CIN);
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, lr = lane & 15, lq = lane >> 4;
    const int hc0 = blockIdx.y * (MBB_WAVES * MBW_HCW) + wave * MBW_HCW;
    const int nct = hidden >> 4;
    const int ntiles = (n + MB_T - 1) / MB_T;
    const int walker = blockIdx.x, nwalkers = gridDim.x;
    const size_t wplane = (size_t)KS * nct * 64;

    // the wave's weights do not change over its walk: W1 fragments (three planes: 48 registers - affordable since the file is
    // compiled without the SLP vectoriser, which had the kernel at 226 registers; in LDS they were 12 reads per tile), b1 and
    // the W2 operand
    u32x4 bw[MBW_CT][KS][3], bg[MBW_CT];
    float bb[MBW_CT];
#pragma unroll
    for (int c = 0; c < MBW_CT; ++c) {
        const int ct = (hc0 >> 4) + c;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks)
#pragma unroll
            for (int p = 0; p < 3; ++p) bw[c][ks][p] = Wp16[p * wplane + ((size_t)ks * nct + ct) * 64 + lane];
        bb[c] = b1[ct * 16 + lr];
        bg[c] = W2s[ct * 64 + lane];
    }
    // dW1acc: C layout of x^T dh (column = hidden column lr, row = input channel 4*lq + t of tile m)
    // dW2acc[c][o], db1acc[c]: this lane's hidden column lr of tile c, summed over the rows the lane holds (its lq)
    f32x4 dW1acc[MBW_CT][MT];
    float dW2acc[MBW_CT][3], db1acc[MBW_CT], db2acc[3] = {0.f, 0.f, 0.f};
#pragma unroll
    for (int c = 0; c < MBW_CT; ++c) {
        db1acc[c] = 0.f;
#pragma unroll
        for (int o = 0; o < 3; ++o) dW2acc[c][o] = 0.f;
#pragma unroll
        for (int m = 0; m < MT; ++m) dW1acc[c][m] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    // staging share of this lane: row 8 wave + (lane >> 3) of the tile, channels 4 (lane & 7) .. + 3; its dy row if lane < 8
    const int srow = wave * 8 + (lane >> 3), sch = (lane & 7) * 4, drow = wave * 8 + (lane & 7);
    f32x4 sx;
    float sd[3];
    auto request = [&](int tile) {             // (clamped rows: always valid loads; rows past n are zeroed by bit masks)
        const int row = tile * MB_T + srow, rd = tile * MB_T + drow;
        sx = keep_if(*reinterpret_cast<const f32x4*>(x + (size_t)min(row, n - 1) * CIN + sch), row < n);
#pragma unroll
        for (int o = 0; o < 3; ++o) sd[o] = keep_if(dy[(size_t)min(rd, n - 1) * cout + min(o, cout - 1)], rd < n && o < cout);
    };
    auto park = [&](int buf) {
        char* t = smem_raw + buf * TILE;
        u32x2 pl[3];
        split3(sx, pl[0], pl[1], pl[2]);
#pragma unroll
        for (int p = 0; p < 3; ++p) *reinterpret_cast<u32x2*>(t + srow * XTS + p * XPL + sch * 2) = pl[p];
        unsigned w9[9];
        split_k3_row(sd, w9);
        {   // (no `lane < 8`: the eight lanes that share a dy row store the same bytes - a branch here, with every accumulator
            //  live across it, made the register allocator spill 129 of them)
            *reinterpret_cast<f32x4*>(t + MB_T * XTS + drow * 16) = f32x4{sd[0], sd[1], sd[2], 0.f};
            u32x4* fr = reinterpret_cast<u32x4*>(t + MB_T * XTS + MB_T * 16 + drow * 48);
            fr[0] = u32x4{w9[0], w9[1], w9[2], w9[3]};
            fr[1] = u32x4{w9[4], w9[5], w9[6], w9[7]};
            fr[2] = u32x4{w9[8], 0u, 0u, 0u};
        }
    };
    const int my_tiles = walker < ntiles ? (ntiles - walker + nwalkers - 1) / nwalkers : 0;
    if (my_tiles > 0) {
        request(walker);
        park(0);
    }
    __syncthreads();
#pragma unroll 1
    for (int it = 0; it < my_tiles; ++it) {
        const int tile = walker + it * nwalkers;
        const char* t = smem_raw + (it & 1) * TILE;
        const float* dyw = reinterpret_cast<const float*>(t + MB_T * XTS);
        request(min(tile + nwalkers, ntiles - 1));      // (the tile behind the last: a valid load, parked and never read)
        __builtin_amdgcn_sched_barrier(0);
        // gA[r] = the A operand of g = dy W2^T (this lane's row r*16 + lr; lane groups 0-2 carry slots, group 3 zeros)
        u32x4 gA[MB_RT];
#pragma unroll
        for (int r = 0; r < MB_RT; ++r) {
            const u32x4 f = *reinterpret_cast<const u32x4*>(t + MB_T * XTS + MB_T * 16 + (r * 16 + lr) * 48 + min(lq, 2) * 16);
            const unsigned keep = lq < 3 ? ~0u : 0u;
            gA[r] = u32x4{f[0] & keep, f[1] & keep, f[2] & keep, f[3] & keep};
        }
        u32x4 ax[MB_RT][KS][3];       // x fragments: row r*16 + lr, channels ks*32 + 8*lq .. + 7, plane by plane (once per tile)
#pragma unroll
        for (int r = 0; r < MB_RT; ++r)
#pragma unroll
            for (int ks = 0; ks < KS; ++ks)
#pragma unroll
                for (int p = 0; p < 3; ++p)
                    ax[r][ks][p] = *reinterpret_cast<const u32x4*>(t + (r * 16 + lr) * XTS + p * XPL + (ks * 32 + 8 * lq) * 2);
        u32x4 af[MT][3];              // x^T fragments of the channel tiles (transposed reads, once per tile)
#pragma unroll
        for (int m = 0; m < MT; ++m)
#pragma unroll
            for (int p = 0; p < 3; ++p) af[m][p] = mbs_xT_frag(t + p * XPL, XTS, m * 16, lq, lr);
        if (blockIdx.y == 0 && wave == 0 && lr == 0) {
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const f32x4 d = *reinterpret_cast<const f32x4*>(dyw + ((j >> 2) * 16 + 4 * lq + (j & 3)) * 4);
#pragma unroll
                for (int o = 0; o < 3; ++o) db2acc[o] += d[o];
            }
        }
#pragma unroll
        for (int c = 0; c < MBW_CT; ++c) {
            __builtin_amdgcn_sched_barrier(0);    // one column tile at a time
            asm volatile("" ::: "memory");        // (and its LDS operands re-read: kept from the previous column tile - the
                                                  //  addresses are the same - they would be 60 registers held across the loop)
            f32x4 h[MB_RT], g[MB_RT];
#pragma unroll
            for (int r = 0; r < MB_RT; ++r) h[r] = f32x4{bb[c], bb[c], bb[c], bb[c]};   // (the bias rides in the accumulator)
#pragma unroll
            for (int ks = 0; ks < KS; ++ks)
#pragma unroll
                for (int r = 0; r < MB_RT; ++r) h[r] = mfma_split(ax[r][ks], bw[c][ks], h[r]);
#pragma unroll
            for (int r = 0; r < MB_RT; ++r)
                g[r] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, gA[r]), __builtin_bit_cast(bf16x8, bg[c]),
                                                              f32x4{0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);    // (phases kept apart: hoisted together their operands do not fit 256 registers)
#pragma unroll
            for (int r = 0; r < MB_RT; ++r)
#pragma unroll
                for (int t4 = 0; t4 < 4; ++t4) {
                    // (the row's dy: re-read per column tile, four addresses per wave - cheaper than 24 registers held)
                    const f32x4 dyv = *reinterpret_cast<const f32x4*>(dyw + (r * 16 + 4 * lq + t4) * 4);
                    const float pre = h[r][t4];
                    const float slope = lrelu01_slope(pre, alpha);
                    const float ha = pre * slope;            // lrelu(pre) = pre * lrelu'(pre)
                    g[r][t4] *= slope;
                    db1acc[c] += g[r][t4];
#pragma unroll
                    for (int o = 0; o < 3; ++o) dW2acc[c][o] = fmaf(ha, dyv[o], dW2acc[c][o]);
                }
            // k = r*16 + 4*lq + t: element j of the B fragment of dW1 += x^T dh is (r = j >> 2, t = j & 3)
            u32x4 bf[3];
            split3_frag(g[0], g[1], bf);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int m = 0; m < MT; ++m) {
                dW1acc[c][m] = mfma_split(af[m], bf, dW1acc[c][m]);
                // (volatile statements keep their order: this column tile's last product comes before the next one's first
                //  LDS read; left free, the compiler runs two column tiles' phases side by side and spills the accumulators)
                asm volatile("" : "+v"(dW1acc[c][m])::"memory");
            }
        }
        __builtin_amdgcn_sched_barrier(0);
        park((it + 1) & 1);                   // (that buffer was last read in iteration it - 1, before its barrier)
        __syncthreads();
    }
    // parameter-gradient slabs of this wave (slab index = the workgroup's walker id)
    if (blockIdx.y == 0 && wave == 0) {   // db2[o]: lanes (lr = 0, lq) hold the sums of their rows
#pragma unroll
        for (int o = 0; o < 3; ++o) {
            float v = db2acc[o];
            v += __shfl_xor(v, 16);
            v += __shfl_xor(v, 32);
            if (lane == 0) db2_slab[walker * 4 + o] = o < cout ? v : 0.f;
        }
        if (lane == 0) db2_slab[walker * 4 + 3] = 0.f;
    }
#pragma unroll
    for (int c = 0; c < MBW_CT; ++c) {
        const int col = hc0 + c * 16 + lr;
#pragma unroll
        for (int m = 0; m < MT; ++m)
#pragma unroll
            for (int t4 = 0; t4 < 4; ++t4)
                dW1_slab[((size_t)walker * CIN + m * 16 + lq * 4 + t4) * hidden + col] = dW1acc[c][m][t4];
        float v = db1acc[c];
        v += __shfl_xor(v, 16);
        v += __shfl_xor(v, 32);
        if (lq == 0) db1_slab[(size_t)walker * hidden + col] = v;
        float w3[3];
#pragma unroll
        for (int o = 0; o < 3; ++o) {
            float u = dW2acc[c][o];
            u += __shfl_xor(u, 16);
            u += __shfl_xor(u, 32);
            w3[o] = o < cout ? u : 0.f;
        }
        if (lq == 0) *reinterpret_cast<f32x4*>(dW2_slab + ((size_t)walker * hidden + col) * 4) = f32x4{w3[0], w3[1], w3[2], 0.f};
    }
}

}  // namespace fgc

using namespace fgc;

// row walkers (waves) of the parameter-gradient kernel: each keeps its partial sums in registers and writes one slab
static int mb_gx(int n) {
    const int ntiles = cdiv(n, MB_T);
    const int wg = cdiv(ntiles, 4);
    return (wg < 32 ? wg : 32) * 4;
}

namespace fgc {
// ---- the fp32 network's MLP through split operands (called by fgc_mlp_fwd in fgc_mlp.hip) ----------------
bool mlp_split_enabled() { return opt(OPT_NO_MLP_SPLIT) != 1; }
size_t mlp_split_pack_bytes(int cin, int hidden) { return align_up((size_t)3 * cin * hidden * 2, 256); }
bool mlp_fwd_split_ok(const float* x, int cin, int hidden, int cout) {
    return mlp_split_enabled() && (cin == 32 || cin == 64) && hidden % 256 == 0 && cout <= 4 && (uintptr_t)x % 16 == 0;
}
int launch_mlp_fwd_split(const float* x, int n, int cin, int hidden, int cout, const float* W1, const float* b1, const float* W2,
                         const float* b2, float alpha, float* y, float* abs_partial, void* workspace, bool packed, hipStream_t st) {
    unsigned short* Wp = (unsigned short*)workspace;
    if (!packed)
        FGC_LAUNCH("mlp_pack_kernel", st, mlp_pack_split_kernel, dim3(cdiv(cin * hidden, 1024)), dim3(256), 0, W1, Wp, cin, hidden);
    const u32x4* Wp16 = (const u32x4*)Wp;
    const dim3 grid(cdiv(n, MB_FWD_T));
#define FGC_MS_FWD(KS)                                                                                                       \
    do {                                                                                                                     \
        if (cout <= 3)                                                                                                       \
            FGC_LAUNCH("mlp_fwd_kernel", st, (mlp_fwd_split_kernel<KS, 3>), grid, dim3(MB_THREADS), 0, x, n, hidden, cout, Wp16, b1, \
                       W2, b2, alpha, y, abs_partial);                                                                       \
        else                                                                                                                 \
            FGC_LAUNCH("mlp_fwd_kernel", st, (mlp_fwd_split_kernel<KS, 4>), grid, dim3(MB_THREADS), 0, x, n, hidden, cout, Wp16, b1, \
                       W2, b2, alpha, y, abs_partial);                                                                       \
    } while (0)
    if (cin == 32) FGC_MS_FWD(1);
    else FGC_MS_FWD(2);
#undef FGC_MS_FWD
    FGC_CHECK_LAUNCH("fgc_mlp_fwd (split operands)");
    return FGC_OK;
}
}  // namespace fgc

namespace fgc {
// ---- the fp32 network's MLP backward through split operands (called by fgc_mlp_bwd in fgc_mlp.hip) -------------------
bool mlp_bwd_split_enabled() { return mlp_split_enabled() && opt(OPT_NO_MLP_BWD_SPLIT) != 1; }
bool mlp_bwd_split_ok(const float* x, const float* dx, int cin, int hidden, int cout) {
    return mlp_bwd_split_enabled() && cin == 32 && hidden % 256 == 0 && cout <= 3 && (uintptr_t)x % 16 == 0 &&
           (uintptr_t)dx % 16 == 0;
}
// workspace: [W1 fragments, 3 planes][W1 in dx order, 3 planes][dW1 slabs][db1 slabs][dW2 slabs][db2 partials][W2 operand][reduce]
struct SplitBwdWs {
    size_t wp, w1d, dW1, db1, dW2, db2, w2s, rtmp, total;
};
static SplitBwdWs split_bwd_plan(int n, int cin, int hidden) {
    const size_t gx = mb_gx(n);
    SplitBwdWs w;
    size_t o = 0;
    w.wp = o, o += align_up((size_t)3 * cin * hidden * 2, 256);
    w.w1d = o, o += align_up((size_t)3 * cin * hidden * 2, 256);
    w.dW1 = o, o += align_up(gx * (size_t)cin * hidden * 4, 256);
    w.db1 = o, o += align_up(gx * (size_t)hidden * 4, 256);
    w.dW2 = o, o += align_up(gx * (size_t)hidden * 4 * 4, 256);
    w.db2 = o, o += align_up((size_t)1024 * 4 * 4, 256);
    w.w2s = o, o += align_up((size_t)(hidden >> 4) * 64 * 16, 256);
    w.rtmp = o;
    o += align_up((reduce_tmp_floats(1024, 4) + reduce_tmp_floats((int)gx, (size_t)cin * hidden) +
                   reduce_tmp_floats((int)gx, (size_t)hidden * 5)) * 4 + 256, 256);
    w.total = o;
    return w;
}
size_t mlp_bwd_split_workspace_bytes(int n, int cin, int hidden) { return split_bwd_plan(n, cin, hidden).total; }
// the three operand packs as jobs of the step's housekeeping launch (fgc_conv_pack), laid out as launch_mlp_bwd_split reads them
int mlp_bwd_split_pack_jobs(const fgc_pack_extra* e, PackJob* jobs, size_t* totals) {
    const int cin = e->mlp_cin, hidden = e->mlp_hidden, cout = e->mlp_cout;
    if (e->mlp_n <= 0 || !e->mlp_W2 || (uintptr_t)e->mlp_bwd_ws % 16 != 0) return -1;
    const SplitBwdWs w = split_bwd_plan(e->mlp_n, cin, hidden);
    char* base = (char*)e->mlp_bwd_ws;
    jobs[0] = PackJob{e->mlp_W1, (float*)(base + w.wp), 10, cin, cout, 0, hidden, 0, 0, 0, 0, 0, 0};
    totals[0] = (size_t)cin * hidden;
    jobs[1] = PackJob{e->mlp_W1, (float*)(base + w.w1d), 15, cin, cout, 0, hidden, 0, 0, 0, 0, 0, 0};
    totals[1] = (size_t)cin * hidden;
    jobs[2] = PackJob{e->mlp_W2, (float*)(base + w.w2s), 16, cin, cout, 0, hidden, 0, 0, 0, 0, 0, 0};
    totals[2] = (size_t)cdiv((hidden >> 4) * 64, 256) * 1024;
    return 3;
}
__global__ void mlp_pack_bwd_split_kernel(const float* __restrict__ W1, const float* __restrict__ W2, unsigned short* __restrict__ Wp,
                                          unsigned short* __restrict__ Wd, u32x4* __restrict__ W2s, int cin, int hidden, int cout,
                                          int nb) {
    const int b = blockIdx.x;
    if (b < nb) mlp_pack_split_body(W1, Wp, cin, hidden, b, nb);
    else if (b < 2 * nb) mlp_pack_w1dx_split_body(W1, Wd, cin, hidden, b - nb, nb);
    else mlp_pack_w2_split_body(W2, W2s, hidden, cout, b - 2 * nb);
}
int launch_mlp_bwd_split(const float* x, const float* dy, int n, int cin, int hidden, int cout, const float* W1, const float* b1,
                         const float* W2, float alpha, float* dx, float* dW1, float* db1, float* dW2, float* db2, void* workspace,
                         bool packed, hipStream_t st) {
    const SplitBwdWs w = split_bwd_plan(n, cin, hidden);
    char* base = (char*)workspace;
    unsigned short* Wp = (unsigned short*)(base + w.wp);
    unsigned short* W1d = (unsigned short*)(base + w.w1d);
    float* dW1_slab = (float*)(base + w.dW1);
    float* db1_slab = (float*)(base + w.db1);
    float* dW2_slab = (float*)(base + w.dW2);
    float* db2_part = (float*)(base + w.db2);
    u32x4* W2s = (u32x4*)(base + w.w2s);
    float* rtmp = (float*)(base + w.rtmp);
    const int gx = mb_gx(n), gy = hidden / MBW_HCW;
    const int nbp = cdiv(cin * hidden, 1024);
    if (!packed)
        FGC_LAUNCH("mlp_pack_kernel", st, mlp_pack_bwd_split_kernel, dim3(2 * nbp + cdiv((hidden >> 4) * 64, 256)), dim3(256), 0, W1, W2,
                   Wp, W1d, W2s, cin, hidden, cout, nbp);
    const int tiles = cdiv(n, MB_T);
    constexpr int MT = 2;                      // cin == 32 (mlp_bwd_split_ok)
    FGC_LAUNCH("mlp_bwd_kernel<dx>", st, (mlp_bwd_dx_split_kernel<MT>), dim3(cdiv(tiles, MBB_WAVES)), dim3(MBB_THREADS), 0, x, dy, n,
               hidden, cout, (const u32x4*)Wp, (const u32x4*)W1d, b1, W2s, alpha, dx);
    const size_t smem_w = MBS_WG_LDS(MT * 16);
    hipFuncSetAttribute((const void*)mlp_bwd_w_split_kernel<MT>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem_w);
    FGC_LAUNCH("mlp_bwd_kernel<w>", st, (mlp_bwd_w_split_kernel<MT>), dim3(gx, hidden / (MBB_WAVES * MBW_HCW)), dim3(MBB_THREADS), smem_w,
               x, dy, n, hidden, cout, (const u32x4*)Wp, b1, W2s, alpha, dW1_slab, db1_slab, dW2_slab, db2_part);
    FGC_CHECK_LAUNCH("fgc_mlp_bwd (split operands)");
    const RedJob jobs[4] = {
        {dW1_slab, (size_t)cin * hidden, gx, cin * hidden, hidden, hidden, dW1},
        {db1_slab, (size_t)hidden, gx, hidden, hidden, hidden, db1},
        {dW2_slab, (size_t)hidden * 4, gx, hidden * 4, 4, cout, dW2},
        {db2_part, (size_t)4, gx, 4, 4, cout, db2},
    };
    return reduce_jobs("reduce:mlp", jobs, 4, rtmp, st);
}
}  // namespace fgc

extern "C" size_t fgc_mlp_bf16_workspace_bytes(int32_t cin, int32_t hidden, int32_t cout) {
    (void)cout;
    return align_up((size_t)cin * hidden * 2, 256);
}

extern "C" size_t fgc_mlp_bwd_bf16_workspace_bytes(int32_t n, int32_t cin, int32_t hidden, int32_t cout) {
    (void)cout;
    const size_t gx = mb_gx(n);
    size_t b = 2 * align_up((size_t)cin * hidden * 2, 256);   // W1 as MFMA fragments and as bf16 rows
    b += align_up(gx * (size_t)cin * hidden * 4, 256);         // dW1 slabs
    b += align_up(gx * (size_t)hidden * 4, 256);               // db1 slabs
    b += align_up(gx * (size_t)hidden * 4 * 4, 256);           // dW2 slabs
    b += align_up((size_t)1024 * 4 * 4, 256);                  // db2 partials
    b += align_up((size_t)(hidden >> 4) * 64 * 16, 256);       // W2 as fragments of the g product
    b += align_up((reduce_tmp_floats(1024, 4) + reduce_tmp_floats((int)gx, (size_t)cin * hidden) +
                   reduce_tmp_floats((int)gx, (size_t)hidden * 5)) * 4 + 256, 256);
    return b;
}

namespace fgc {
// byte offset of the W2 fragments in the backward workspace (behind the two W1 operands and the gradient slabs)
static size_t mb_w2p_offset(int n, int cin, int hidden) {
    const size_t gx = mb_gx(n);
    return 2 * align_up((size_t)cin * hidden * 2, 256) + align_up(gx * (size_t)cin * hidden * 4, 256) +
           align_up(gx * (size_t)hidden * 4, 256) + align_up(gx * (size_t)hidden * 4 * 4, 256) + align_up((size_t)1024 * 4 * 4, 256);
}
int mlp_pack_jobs_bf16(const fgc_pack_extra* e, PackJob* jobs, size_t* totals) {
    const int cin = e->mlp_cin, hidden = e->mlp_hidden, cout = e->mlp_cout;
    if (!(cin == 32 || cin == 64 || cin == 128) || hidden <= 0 || hidden % 256 != 0 || cout <= 0 || cout > 4) return -1;
    int nj = 0;
    if (e->mlp_fwd_ws) {
        if ((uintptr_t)e->mlp_fwd_ws % 16 != 0) return -1;
        jobs[nj] = PackJob{e->mlp_W1, (float*)e->mlp_fwd_ws, 11, cin, cout, 0, hidden, 0, 0, 0, 0, 0, 0};
        totals[nj++] = (size_t)cin * hidden;
    }
    if (e->mlp_bwd_ws) {
        if ((uintptr_t)e->mlp_bwd_ws % 16 != 0 || cin == 128 || cout > 3 || e->mlp_n <= 0 || !e->mlp_W2) return -1;
        char* w = (char*)e->mlp_bwd_ws;
        jobs[nj] = PackJob{e->mlp_W1, (float*)w, 11, cin, cout, 0, hidden, 0, 0, 0, 0, 0, 0};
        totals[nj++] = (size_t)cin * hidden;
        jobs[nj] = PackJob{e->mlp_W1, (float*)(w + align_up((size_t)cin * hidden * 2, 256)), 12, cin, cout, 0, hidden, 0, 0, 0, 0, 0, 0};
        totals[nj++] = (size_t)cin * hidden;
        jobs[nj] = PackJob{e->mlp_W2, (float*)(w + mb_w2p_offset(e->mlp_n, cin, hidden)), 13, cin, cout, 0, hidden, 0, 0, 0, 0, 0, 0};
        totals[nj++] = (size_t)cdiv((hidden >> 4) * 32, 256) * 1024;
    }
    return nj;
}
}  // namespace fgc

static int mlp_bf16_check(const char* who, const void* x, int n, int cin, int hidden, int cout) {
    FGC_CHECK_ARG(x && n > 0, "%s: null x / n=%d", who, n);
    FGC_CHECK_ARG(cin == 32 || cin == 64 || cin == 128, "%s: cin=%d (the bf16 MLP takes 32, 64 or 128 input channels)", who, cin);
    FGC_CHECK_ARG(hidden > 0 && hidden % 256 == 0, "%s: hidden=%d must be a multiple of 256", who, hidden);
    FGC_CHECK_ARG(cout > 0 && cout <= 4, "%s: cout=%d outside [1,4]", who, cout);
    FGC_CHECK_ARG((uintptr_t)x % 16 == 0, "%s: x needs 16-byte alignment", who);
    return FGC_OK;
}
static int mlp_bf16_alpha(const char* who, float alpha) {
    FGC_CHECK_ARG(alpha >= 0.f && alpha <= 1.f, "%s: alpha=%g outside [0,1] (the reference uses 0.1, model.py:846)", who, alpha);
    return FGC_OK;
}

extern "C" int fgc_mlp_fwd_bf16(const void* x, int32_t n, int32_t cin, int32_t hidden, int32_t cout, const float* W1,
                                const float* b1, const float* W2, const float* b2, float alpha, float* y,
                                float* abs_partial, int32_t flags, void* workspace, size_t workspace_bytes, void* stream) {
    int rc = mlp_bf16_check("fgc_mlp_fwd_bf16", x, n, cin, hidden, cout);
    if (rc) return rc;
    if ((rc = mlp_bf16_alpha("fgc_mlp_fwd_bf16", alpha))) return rc;
    FGC_CHECK_ARG(W1 && b1 && W2 && b2 && y, "fgc_mlp_fwd_bf16: null pointer");
    FGC_CHECK_ARG(!(flags & FGC_MLP_PACKED) || (flags >> 8) == 0 || (flags >> 8) == mlp_layout_id(cin, hidden, cout, true),
                  "fgc_mlp_fwd_bf16: FGC_MLP_PACKED, but the operands were packed in layout %d, not %d", flags >> 8,
                  mlp_layout_id(cin, hidden, cout, true));
    FGC_CHECK_ARG(workspace && workspace_bytes >= fgc_mlp_bf16_workspace_bytes(cin, hidden, cout) && (uintptr_t)workspace % 16 == 0,
                  "fgc_mlp_fwd_bf16: workspace too small or misaligned");
    hipStream_t st = (hipStream_t)stream;
    unsigned short* Wp = (unsigned short*)workspace;
    if (!(flags & FGC_MLP_PACKED))
        FGC_LAUNCH("mlp_pack_kernel", st, mlp_pack_bf16_kernel, dim3(cdiv(cin * hidden, 1024)), dim3(256), 0, W1, Wp, cin, hidden);
    const unsigned short* x16 = (const unsigned short*)x;
    const u32x4* Wp16 = (const u32x4*)Wp;
    const dim3 grid(cdiv(n, MB_FWD_T));
#define FGC_MB_FWD(KS)                                                                                                      \
    do {                                                                                                                    \
        if (cout <= 3)                                                                                                      \
            FGC_LAUNCH("mlp_fwd_kernel", st, (mlp_fwd_bf16_kernel<KS, 3>), grid, dim3(MB_THREADS), 0, x16, n, hidden, cout, Wp16, \
                       b1, W2, b2, alpha, y, abs_partial);                                                                  \
        else                                                                                                                \
            FGC_LAUNCH("mlp_fwd_kernel", st, (mlp_fwd_bf16_kernel<KS, 4>), grid, dim3(MB_THREADS), 0, x16, n, hidden, cout, Wp16, \
                       b1, W2, b2, alpha, y, abs_partial);                                                                  \
    } while (0)
    if (cin == 32) FGC_MB_FWD(1);
    else if (cin == 64) FGC_MB_FWD(2);
    else FGC_MB_FWD(4);
#undef FGC_MB_FWD
    FGC_CHECK_LAUNCH("fgc_mlp_fwd_bf16");
    return FGC_OK;
}

extern "C" int fgc_mlp_bwd_bf16(const void* x, const float* dy, int32_t n, int32_t cin, int32_t hidden, int32_t cout,
                                const float* W1, const float* b1, const float* W2, float alpha, void* dx, float* dW1,
                                float* db1, float* dW2, float* db2, int32_t flags, void* workspace, size_t workspace_bytes,
                                void* stream) {
    int rc = mlp_bf16_check("fgc_mlp_bwd_bf16", x, n, cin, hidden, cout);
    if (rc) return rc;
    if ((rc = mlp_bf16_alpha("fgc_mlp_bwd_bf16", alpha))) return rc;
    FGC_CHECK_ARG(dy && W1 && b1 && W2 && dx && dW1 && db1 && dW2 && db2, "fgc_mlp_bwd_bf16: null pointer");
    FGC_CHECK_ARG(!(flags & FGC_MLP_PACKED) || (flags >> 8) == 0 || (flags >> 8) == mlp_layout_id(cin, hidden, cout, true),
                  "fgc_mlp_bwd_bf16: FGC_MLP_PACKED, but the operands were packed in layout %d, not %d", flags >> 8,
                  mlp_layout_id(cin, hidden, cout, true));
    FGC_CHECK_ARG((cin == 32 || cin == 64) && cout <= 3, "fgc_mlp_bwd_bf16: cin=%d cout=%d (cin 32 or 64, cout <= 3)", cin, cout);
    FGC_CHECK_ARG(workspace && workspace_bytes >= fgc_mlp_bwd_bf16_workspace_bytes(n, cin, hidden, cout) &&
                      (uintptr_t)workspace % 16 == 0,
                  "fgc_mlp_bwd_bf16: workspace too small or misaligned");
    hipStream_t st = (hipStream_t)stream;
    const int gx = mb_gx(n), gy = hidden / MBW_HCW;
    char* w = (char*)workspace;
    unsigned short* Wp = (unsigned short*)w;
    w += align_up((size_t)cin * hidden * 2, 256);
    unsigned short* W1h = (unsigned short*)w;
    w += align_up((size_t)cin * hidden * 2, 256);
    float* dW1_slab = (float*)w;
    w += align_up((size_t)gx * cin * hidden * 4, 256);
    float* db1_slab = (float*)w;
    w += align_up((size_t)gx * hidden * 4, 256);
    float* dW2_slab = (float*)w;
    w += align_up((size_t)gx * hidden * 4 * 4, 256);
    float* db2_part = (float*)w;
    w += align_up((size_t)1024 * 4 * 4, 256);
    u32x4* W2p = (u32x4*)w;
    w += align_up((size_t)(hidden >> 4) * 64 * 16, 256);
    float* rtmp = (float*)w;

    const int nbp = cdiv(cin * hidden, 1024);
    if (!(flags & FGC_MLP_PACKED))
        FGC_LAUNCH("mlp_pack_kernel", st, mlp_pack_bwd_bf16_kernel, dim3(2 * nbp + cdiv((hidden >> 4) * 32, 256)), dim3(256), 0, W1, W2, Wp,
                   W1h, W2p, cin, hidden, cout, nbp);
    const unsigned short* x16 = (const unsigned short*)x;
    const u32x4* Wp16 = (const u32x4*)Wp;
    const int tiles = cdiv(n, MB_T);
#define FGC_MB_BWD(MT)                                                                                                      \
    do {                                                                                                                    \
        constexpr int CIN_ = MT * 16;                                                                                       \
        FGC_LAUNCH("mlp_bwd_kernel<dx>", st, (mlp_bwd_dx_bf16_kernel<MT>), dim3(cdiv(tiles, MBB_WAVES)), dim3(MBB_THREADS),  \
                   0, x16, dy, n, hidden, cout, Wp16, (const u32x4*)W1h, b1, W2p, alpha, (unsigned short*)dx);        \
        const size_t smem_w = (size_t)MBB_WAVES * ((size_t)CIN_ * (MB_T * 2 + 8) + MB_T * 16);                              \
        FGC_LAUNCH("mlp_bwd_kernel<w>", st, (mlp_bwd_w_bf16_kernel<MT, 3>), dim3(gx / 4, gy), dim3(MBB_THREADS), smem_w, x16, \
                   dy, n, hidden, cout, Wp16, b1, W2p, alpha, dW1_slab, db1_slab, dW2_slab, db2_part);                      \
    } while (0)
    if (cin == 32) FGC_MB_BWD(2);
    else FGC_MB_BWD(4);
#undef FGC_MB_BWD
    FGC_CHECK_LAUNCH("fgc_mlp_bwd_bf16");
    const RedJob jobs[4] = {
        {dW1_slab, (size_t)cin * hidden, gx, cin * hidden, hidden, hidden, dW1},
        {db1_slab, (size_t)hidden, gx, hidden, hidden, hidden, db1},
        {dW2_slab, (size_t)hidden * 4, gx, hidden * 4, 4, cout, dW2},
        {db2_part, (size_t)4, gx, 4, 4, cout, db2},
    };
    return reduce_jobs("reduce:mlp", jobs, 4, rtmp, st);
}
