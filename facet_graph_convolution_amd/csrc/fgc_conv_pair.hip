// Pair form of custom_conv2d (model.py:427-504) over a 4x-upsampled coarse tensor (custom_upsampling, model.py:817-825:
// the two up-convolutions of the network, model.py:905,926).
//
// With x_fine[j] = xc[j >> 2] every term of the convolution depends on the coarse rows only.  For a fine node i of block
// p = i >> 2 and its neighbours j with parents P = j >> 2:
//     l_ij = a_p + g_P,  q_ij = softmax(l_ij) = q_pP          (the same for all four siblings and every j under P)
//     y_i  = (1/deg_i) sum_P mult_iP sum_m q_pPm h_Pm + b,    h_P = W0 xc_P  (M*cout floats per COARSE row)
// where mult_iP counts the neighbours of i under P (fgc_pair_graph).  So instead of a gather, nine FMAs per channel and a
// [9 cin] x [9 cin, cout] product per FINE node and edge, the layer is
//     F1  pair_transform_kernel   hc = xc W0^T (fp32 MFMA, one product per COARSE row: 4x fewer FLOPs) and the logit table
//     F2  pair_fwd_kernel         per block: softmax per pair, t_pP = sum_m q_pPm h_Pm, y_i = sum_P mult_iP t_pP / deg_i + b
// and the backward pass (dt_pP = sum_i mult_iP dy_i lrelu'(y_i) / deg_i, the gradient of t_pP)
//     B1  pair_bwd_logits_kernel  dq_pPm = <dt_pP, h_Pm>, dl_pP = softmax backward, da_p, db / dc partials, dt per pair
//     B2  conv_w8_kernel<data>    over the TRANSPOSED pair graph with the dt rows as gathered operand (rows by pair id):
//                                 dh_P = sum_p q_pPm dt_pP (= r), dg_P, dxc_P = dh_P W0 + da_P u + dg_P v
//     B3  gemm_tn                 dW0 = dh^T xc, [du; dv] = [da | dg]^T xc with K = n/4 coarse rows
// Same sums as the fine form in another order (fp32 summation-order differences only).
#include <stdlib.h>

#include "fgc_conv_core.h"
#include "fgc_conv_pair.h"

namespace fgc {

constexpr int PQ_LD = 12;                 // floats per pair slot: q[0..8], [9] = coarse row, [10] = multiplicities
constexpr int PQ_SLOTS = 16;              // pairs per chunk
constexpr int PQ_STRIDE = PQ_SLOTS * PQ_LD + 4;   // (+4: consecutive blocks start on different banks)

// the graph-dependent limits (fgc_conv_pairs_allowed): in-pairs against the edge slots of the data-gradient kernel, hc and dt
// offsets against 32 bits, row ids of hc and dt against the 24-bit multiplies (dt also holds a facet-sharded rank's incoming
// cross pairs: a margin)
bool pairs_graph_allowed(int64_t rows, int64_t n_pairs, int max_in_deg, int cout) {
    if (rows < 0 || n_pairs < 0 || cout <= 0 || max_in_deg < 0 || max_in_deg > KMAX) return false;
    if ((uint64_t)rows * FGC_M * cout * 4 >= 0xFFFFFFFFull || (uint64_t)n_pairs * cout * 4 >= 0xFFFFFFFFull) return false;
    if (rows >= (1 << 24) || n_pairs >= (1 << 24) - (1 << 20)) return false;
    return true;
}

bool pairs_ok(const fgc_conv_desc* d) {
    if (opt(OPT_NO_PAIRS) == 1) return false;
    // (developer switches that take the eight-wave fast kernels away take the pair form's data-gradient kernel with them)
    if ((opt(OPT_NO_W8) == 1) || (opt(OPT_NO_W8FAST) == 1))
        return false;
    if (!d || !d->pair_rowptr || !d->pair_col || !d->pair_mul || !d->hc) return false;
    if (d->shift != 2 || d->c1 != 0 || d->x1 != nullptr || (d->n & 3)) return false;
    // (cin: the column count of the data-gradient product, whose kernel wants 2, 4 or 8 column tiles)
    if (!(d->c0 == 32 || d->c0 == 64 || d->c0 == 128) || !(d->cout == 32 || d->cout == 64)) return false;
    if (d->n_pairs <= 0 || d->max_pair_deg <= 0 || d->max_pair_in_deg < 0 || d->max_pair_in_deg > KMAX) return false;
    // partial forward calls of a facet-sharded caller: "transform + logits of these source rows only" (a tile list of no
    // tiles) and "the rest of the rows, then every block" (no tile list); there is no interior / boundary split of blocks
    if (d->tile_list && d->n_tiles != 0) return false;
    const size_t rows = d->src_rows > 0 ? (size_t)d->src_rows : (size_t)(d->n >> 2);
    if (!pairs_graph_allowed((int64_t)rows, d->n_pairs, d->max_pair_in_deg, d->cout)) return false;
    if (((uintptr_t)d->x0 | (uintptr_t)d->hc) % 16) return false;
    return true;
}

int pair_blocks_per_wg(int cout) { return 4 * (64 / (cout / 4)); }

// ---------------------------------------------------------------------------------------------
// F1: hc[rows, 9 cout] = xc[rows, cin] W0^T and ag[rows, 24] = xc [u | v]^T (+ c).  A wave owns 16 rows and a group of
// 16-column tiles.  The product is computed TRANSPOSED (D[column][row]): both operands are then read in their native
// layouts - W0 [9 cout, cin] and u / v [9, cin] are "one output column per row, k contiguous", like x - as one 16-byte
// load per lane and four k-steps, nothing is packed, and a lane ends up with four CONSECUTIVE columns of one row: one
// 16-byte store.
// ---------------------------------------------------------------------------------------------
// RT: 16-row tiles per wave (they share the weight fragments; RT independent accumulator chains)
#ifndef FGC_PT_RT
#define FGC_PT_RT 2
#endif
#ifndef FGC_PT_KO
#define FGC_PT_KO 0   // developer knock-outs: 1 = no stores, 2 = no MFMAs
#endif
#ifndef FGC_PT_CG
#define FGC_PT_CG 10  // column tiles per wave
#endif
template <int CIN, int RT>
__global__ __launch_bounds__(256) void pair_transform_kernel(const float* __restrict__ x, int rows,
                                                            const float* __restrict__ W0, const float* __restrict__ u,
                                                            const float* __restrict__ c, const float* __restrict__ v,
                                                            int cout, float* __restrict__ hc, float* __restrict__ ag,
                                                            int ngroups) {
    constexpr int KG = CIN / 16;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, lr = lane & 15, lq = lane >> 4;
    const int item = __builtin_amdgcn_readfirstlane(blockIdx.x * 4 + wave);
    const int rt = item / ngroups, cg = item - rt * ngroups;
    if (rt * (16 * RT) >= rows) return;
    const int th = (FGC_M * cout) >> 4;       // column tiles of h; then the a tile and the g tile
    const int tc = th + 2;
    const int ct0 = __builtin_amdgcn_readfirstlane(tc * cg / ngroups), ct1 = __builtin_amdgcn_readfirstlane(tc * (cg + 1) / ngroups);
    f32x4 xa[RT][KG];
#pragma unroll
    for (int r = 0; r < RT; ++r) {
        const float* xr = x + (size_t)min(rt * (16 * RT) + r * 16 + lr, rows - 1) * CIN + lq * 4;
#pragma unroll
        for (int g = 0; g < KG; ++g) xa[r][g] = *reinterpret_cast<const f32x4*>(xr + g * 16);
    }
    auto loadw = [&](int ct, f32x4 (&w)[KG]) {
        const int cc = min(ct, ct1 - 1);
        const float* wr = cc < th ? W0 + (size_t)(cc * 16 + lr) * CIN : (cc == th ? u : v) + (size_t)min(lr, FGC_M - 1) * CIN;
        wr += lq * 4;
#pragma unroll
        for (int g = 0; g < KG; ++g) w[g] = *reinterpret_cast<const f32x4*>(wr + g * 16);
    };
    const size_t ldh = (size_t)FGC_M * cout;
    // the logit bias of this lane's four columns of the a tile
    f32x4 cb = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int t = 0; t < 4; ++t) cb[t] = c[min(lq * 4 + t, FGC_M - 1)];
    auto tile = [&](int ct, const f32x4 (&w)[KG]) {
        f32x4 acc[RT];
#pragma unroll
        for (int r = 0; r < RT; ++r) acc[r] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int g = 0; g < KG; ++g)
#pragma unroll
            for (int t = 0; t < 4; ++t)
#pragma unroll
                for (int r = 0; r < RT; ++r) {
                    if (FGC_PT_KO & 2) asm volatile("" ::"v"(w[g][t]), "v"(xa[r][g][t]));
                    else acc[r] = __builtin_amdgcn_mfma_f32_16x16x4f32(w[g][t], xa[r][g][t], acc[r], 0, 0, 0);
                }
        // D[i = column 4 lq + t][j = row lr]
        if (FGC_PT_KO & 1) { if (acc[0][0] == 123.f) hc[0] = 1.f; return; }
#pragma unroll
        for (int r = 0; r < RT; ++r) {
            const int row = rt * (16 * RT) + r * 16 + lr;
            if (row >= rows) continue;
            if (ct < th) {
                *reinterpret_cast<f32x4*>(hc + (size_t)row * ldh + ct * 16 + lq * 4) = acc[r];
            } else if (lq < 3) {
                f32x4 o;
#pragma unroll
                for (int t = 0; t < 4; ++t) o[t] = lq * 4 + t < FGC_M ? acc[r][t] + (ct == th ? cb[t] : 0.f) : 0.f;
                *reinterpret_cast<f32x4*>(ag + (size_t)row * FGC_AG_LD + (ct == th ? 0 : 12) + lq * 4) = o;
            }
        }
    };
    f32x4 w0[KG], w1[KG];
    loadw(ct0, w0);
    for (int ct = ct0; ct < ct1; ct += 2) {
        loadw(ct + 1, w1);
        tile(ct, w0);
        loadw(ct + 2, w0);
        if (ct + 1 < ct1) tile(ct + 1, w1);
    }
}

// bf16 storage (FGC_CONV_BF16): x and hc are bf16, the weights a bf16 copy of W0 in its native [9 cout, cin] layout
// (pair_weight_bf16_floats of the forward workspace, written by the pack launch), the product on
// v_mfma_f32_16x16x32_bf16 with fp32 accumulation.  The two logit tiles stay on the fp32 MFMA with fp32 u / v against
// the widened x (the logit table is fp32 in the bf16 network, include/fgc.h): a lane's eight consecutive k of a 32-deep
// step are eight 4-deep fp32 steps whose k assignment (k = 32 ks + 8 lq + j) is the same in both operands.
template <int CIN, int RT>
__global__ __launch_bounds__(256) void pair_transform_bf16_kernel(const unsigned short* __restrict__ x, int rows,
                                                                 const unsigned short* __restrict__ Wb,
                                                                 const float* __restrict__ u, const float* __restrict__ c,
                                                                 const float* __restrict__ v, int cout,
                                                                 unsigned short* __restrict__ hc, float* __restrict__ ag,
                                                                 int ngroups) {
    constexpr int KS = CIN / 32;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, lr = lane & 15, lq = lane >> 4;
    const int item = __builtin_amdgcn_readfirstlane(blockIdx.x * 4 + wave);
    const int rt = item / ngroups, cg = item - rt * ngroups;
    if (rt * (16 * RT) >= rows) return;
    const int th = (FGC_M * cout) >> 4;
    const int tc = th + 2;
    const int ct0 = __builtin_amdgcn_readfirstlane(tc * cg / ngroups), ct1 = __builtin_amdgcn_readfirstlane(tc * (cg + 1) / ngroups);
    const int hend = min(ct1, th);            // h tiles of this wave: [ct0, hend)
    u32x4 xb[RT][KS];
#pragma unroll
    for (int r = 0; r < RT; ++r) {
        const unsigned short* xr = x + (size_t)min(rt * (16 * RT) + r * 16 + lr, rows - 1) * CIN + lq * 8;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) xb[r][ks] = *reinterpret_cast<const u32x4*>(xr + ks * 32);
    }
    auto loadw = [&](int ct, u32x4 (&w)[KS]) {
        const int cc = max(min(ct, hend - 1), 0);
        const unsigned short* wr = Wb + (size_t)(cc * 16 + lr) * CIN + lq * 8;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) w[ks] = *reinterpret_cast<const u32x4*>(wr + ks * 32);
    };
    const size_t ldh = (size_t)FGC_M * cout;
    auto tile = [&](int ct, const u32x4 (&w)[KS]) {
        f32x4 acc[RT];
#pragma unroll
        for (int r = 0; r < RT; ++r) acc[r] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ks = 0; ks < KS; ++ks)
#pragma unroll
            for (int r = 0; r < RT; ++r)
                acc[r] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, w[ks]), __builtin_bit_cast(bf16x8, xb[r][ks]),
                                                                acc[r], 0, 0, 0);
#pragma unroll
        for (int r = 0; r < RT; ++r) {
            const int row = rt * (16 * RT) + r * 16 + lr;
            if (row < rows) *reinterpret_cast<u32x2*>(hc + (size_t)row * ldh + ct * 16 + lq * 4) = f4_to_bf4(acc[r]);
        }
    };
    if (ct0 < hend) {
        u32x4 w0[KS], w1[KS];
        loadw(ct0, w0);
        for (int ct = ct0; ct < hend; ct += 2) {
            loadw(ct + 1, w1);
            tile(ct, w0);
            loadw(ct + 2, w0);
            if (ct + 1 < hend) tile(ct + 1, w1);
        }
    }
    // the logit tiles of this wave's column group (fp32 MFMA)
    for (int ct = max(ct0, th); ct < ct1; ++ct) {
        const float* wr = (ct == th ? u : v) + (size_t)min(lr, FGC_M - 1) * CIN + lq * 8;
        f32x4 acc[RT];
#pragma unroll
        for (int r = 0; r < RT; ++r) acc[r] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            const f32x4 wa = *reinterpret_cast<const f32x4*>(wr + ks * 32), wb = *reinterpret_cast<const f32x4*>(wr + ks * 32 + 4);
#pragma unroll
            for (int r = 0; r < RT; ++r) {
                const f32x4 xlo = bf4_to_f4(u32x2{xb[r][ks][0], xb[r][ks][1]}), xhi = bf4_to_f4(u32x2{xb[r][ks][2], xb[r][ks][3]});
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[r] = __builtin_amdgcn_mfma_f32_16x16x4f32(wa[j], xlo[j], acc[r], 0, 0, 0);
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[r] = __builtin_amdgcn_mfma_f32_16x16x4f32(wb[j], xhi[j], acc[r], 0, 0, 0);
            }
        }
        if (lq < 3) {
#pragma unroll
            for (int r = 0; r < RT; ++r) {
                const int row = rt * (16 * RT) + r * 16 + lr;
                if (row >= rows) continue;
                f32x4 o;
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    const int m = lq * 4 + t;
                    o[t] = m < FGC_M ? acc[r][t] + (ct == th ? c[min(m, FGC_M - 1)] : 0.f) : 0.f;
                }
                *reinterpret_cast<f32x4*>(ag + (size_t)row * FGC_AG_LD + (ct == th ? 0 : 12) + lq * 4) = o;
            }
        }
    }
}

// bf16 copy of W0 [9 cout, cin] for pair_transform_bf16_kernel (workgroup `bid` of `nb`, 256 threads each)
__global__ __launch_bounds__(256) void pair_weight_bf16_kernel(const float* __restrict__ W0, unsigned short* __restrict__ Wb, int count) {
    for (int i = blockIdx.x * 256 + threadIdx.x; i < count; i += gridDim.x * 256) Wb[i] = f_to_bf(W0[i]);
}

// ---------------------------------------------------------------------------------------------
// block kernels: LPB = cout / 4 lanes per block (four channels each), 64 / LPB blocks per wave, four waves
// ---------------------------------------------------------------------------------------------
struct PairParams {
    int nb;                  // blocks = n / 4
    const int* prow;
    const int* pcol;
    const unsigned* pmul;
    const float* ag;         // [rows, 24]
    const float* hc;         // [rows, 9 cout]
    const int* rowptr;       // fine CSR (degrees)
    // forward
    const float* bias;
    int bias_mask, act;
    float alpha;
    float* y;                // [n, cout]
    // backward
    const float* dy;         // [n, cout]
    const float* yact;       // forward output (activation slopes) or NULL
    float* dt;               // [n_pairs, cout]
    float* dl;               // [n_pairs, 12]
    float* dag;              // [n/4, 24]: writes 0..11
    float* db_part;          // [workgroups, cout]
    float* dc_part;          // [workgroups, 12]
    unsigned dt_bytes, dl_bytes;   // sizes of dt / dl (stores past them are dropped by the buffer descriptor)
};

__device__ __forceinline__ void pair_wave_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// soft assignment of the pairs [k0, k0 + 16) of this lane's block into its table: q, coarse row, multiplicities.  Slots
// past the block's degree get weight zero, multiplicity zero and a valid row (the block's own).
template <int LPB>
__device__ __forceinline__ void pair_softmax_chunk(const PairParams& p, float* qb, int kl, int e0, int d, int k0, int bc,
                                                   const float (&a)[FGC_M]) {
    constexpr int SPL = PQ_SLOTS / LPB;
    const __amdgpu_buffer_rsrc_t ag_rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.ag), 0, -1, 0x00020000);
    int P[SPL];
    unsigned mu[SPL];
#pragma unroll
    for (int t = 0; t < SPL; ++t) {
        const int kk = k0 + kl + LPB * t;
        const int ei = d > 0 ? e0 + min(kk, d - 1) : 0;
        const int pv = p.pcol[ei];
        const unsigned mv = p.pmul[ei];
        P[t] = kk < d ? pv : bc;
        mu[t] = kk < d ? mv : 0u;
    }
    f32x4 g0[SPL], g1[SPL];
    float g8[SPL];
#pragma unroll
    for (int t = 0; t < SPL; ++t) {
        const unsigned go = __umul24((unsigned)P[t], FGC_AG_LD * 4u) + 48u;
        g0[t] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(ag_rs, go, 0, 0));
        g1[t] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(ag_rs, go + 16u, 0, 0));
        g8[t] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(ag_rs, go + 32u, 0, 0));
    }
#pragma unroll
    for (int t = 0; t < SPL; ++t) {
        const int k = kl + LPB * t;
        float l[FGC_M];
        l[0] = a[0] + g0[t][0]; l[1] = a[1] + g0[t][1]; l[2] = a[2] + g0[t][2]; l[3] = a[3] + g0[t][3];
        l[4] = a[4] + g1[t][0]; l[5] = a[5] + g1[t][1]; l[6] = a[6] + g1[t][2]; l[7] = a[7] + g1[t][3];
        l[8] = a[8] + g8[t];
        float mx = l[0];
#pragma unroll
        for (int m = 1; m < FGC_M; ++m) mx = fmaxf(mx, l[m]);
        float sum = 0.f;
        const float nmx = -mx * 1.4426950408889634f;
#pragma unroll
        for (int m = 0; m < FGC_M; ++m) {
            l[m] = __builtin_amdgcn_exp2f(fmaf(l[m], 1.4426950408889634f, nmx));
            sum += l[m];
        }
        const float inv = (k0 + k < d) ? 1.0f / sum : 0.f;
        float* q = qb + k * PQ_LD;
        *reinterpret_cast<f32x4*>(q) = f32x4{l[0] * inv, l[1] * inv, l[2] * inv, l[3] * inv};
        *reinterpret_cast<f32x4*>(q + 4) = f32x4{l[4] * inv, l[5] * inv, l[6] * inv, l[7] * inv};
        *reinterpret_cast<f32x4*>(q + 8) = f32x4{l[8] * inv, __int_as_float(P[t]), __uint_as_float(mu[t]), 0.f};
    }
}

// the nine 16-byte pieces (four channels of every m) of coarse row P for this lane
// (BF: the table is bf16, FGC_CONV_BF16 - eight bytes per piece, kept as loaded and widened where they are used)
template <bool BF> struct PairH { typedef f32x4 T; };
template <> struct PairH<true> { typedef u32x2 T; };
template <int COUT, bool BF>
__device__ __forceinline__ void pair_load_h(__amdgpu_buffer_rsrc_t h_rs, const float* qk, unsigned laneoff,
                                            typename PairH<BF>::T (&h)[FGC_M]) {
    constexpr unsigned ESZ = BF ? 2u : 4u;
    const unsigned off = __umul24((unsigned)__float_as_int(qk[9]), (unsigned)(FGC_M * COUT) * ESZ) + laneoff;
#pragma unroll
    for (int m = 0; m < FGC_M; ++m) {
        if constexpr (BF) h[m] = __builtin_bit_cast(u32x2, __builtin_amdgcn_raw_buffer_load_b64(h_rs, off + (unsigned)(m * COUT) * ESZ, 0, 0));
        else h[m] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(h_rs, off + (unsigned)(m * COUT * 4), 0, 0));
    }
}
// piece m of the row whose byte offset (lane part included) is `off`
template <int COUT, bool BF>
__device__ __forceinline__ typename PairH<BF>::T pair_load_piece(__amdgpu_buffer_rsrc_t h_rs, unsigned off, int m) {
    if constexpr (BF) return __builtin_bit_cast(u32x2, __builtin_amdgcn_raw_buffer_load_b64(h_rs, off + (unsigned)(m * COUT * 2), 0, 0));
    else return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(h_rs, off + (unsigned)(m * COUT * 4), 0, 0));
}
template <int COUT, bool BF>
__device__ __forceinline__ unsigned pair_row_off(const float* qk, unsigned laneoff) {
    return __umul24((unsigned)__float_as_int(qk[9]), (unsigned)(FGC_M * COUT) * (BF ? 2u : 4u)) + laneoff;
}
__device__ __forceinline__ f32x4 pair_h(const f32x4& v) { return v; }
__device__ __forceinline__ f32x4 pair_h(const u32x2& v) { return bf4_to_f4(v); }
// four channels of an activation row: fp32 (16 bytes) or bf16 (8 bytes)
template <bool BF>
__device__ __forceinline__ f32x4 pair_ld4(const float* base, size_t idx) {
    if constexpr (BF) return bf4_to_f4(*reinterpret_cast<const u32x2*>(reinterpret_cast<const unsigned short*>(base) + idx));
    else return *reinterpret_cast<const f32x4*>(base + idx);
}
template <bool BF>
__device__ __forceinline__ void pair_st4(float* base, size_t idx, const f32x4& v) {
    if constexpr (BF) *reinterpret_cast<u32x2*>(reinterpret_cast<unsigned short*>(base) + idx) = f4_to_bf4(v);
    else *reinterpret_cast<f32x4*>(base + idx) = v;
}

__device__ __forceinline__ int wave_max_i32(int v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = max(v, __shfl_xor(v, o));
    return __builtin_amdgcn_readfirstlane(v);
}

#ifndef FGC_PAIR_UNROLL1
#define FGC_PAIR_UNROLL1 0
#endif
#if FGC_PAIR_UNROLL1
#define FGC_PAIR_KLOOP_PRAGMA _Pragma("unroll 1")
#else
#define FGC_PAIR_KLOOP_PRAGMA
#endif
#ifndef FGC_PAIR_LB_FWD
#define FGC_PAIR_LB_FWD 1
#endif
#ifndef FGC_PAIR_LB_BWD
#define FGC_PAIR_LB_BWD 1
#endif
template <int COUT, bool BF>
__global__ __launch_bounds__(256, FGC_PAIR_LB_FWD) void pair_fwd_kernel(PairParams p) {
    constexpr int LPB = COUT / 4, BPW = 64 / LPB, BPG = 4 * BPW;
    __shared__ __attribute__((aligned(16))) float qs[BPG * PQ_STRIDE];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, bl = lane / LPB, kl = lane % LPB;
    const int wg = xcd_tile(blockIdx.x, gridDim.x);
    const int b = (wg * 4 + wave) * BPW + bl;
    const int bc = min(b, p.nb - 1);
    const int e0 = p.prow[bc];
    const int d = b < p.nb ? p.prow[bc + 1] - e0 : 0;
    float a[FGC_M];
    {
        const float* ar = p.ag + (size_t)bc * FGC_AG_LD;
        const f32x4 a0 = *reinterpret_cast<const f32x4*>(ar), a1 = *reinterpret_cast<const f32x4*>(ar + 4);
        a[0] = a0[0]; a[1] = a0[1]; a[2] = a0[2]; a[3] = a0[3]; a[4] = a1[0]; a[5] = a1[1]; a[6] = a1[2]; a[7] = a1[3];
        a[8] = ar[8];
    }
    int deg[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) deg[i] = p.rowptr[4 * bc + i + 1] - p.rowptr[4 * bc + i];
    const int dmax = wave_max_i32(d);
    float* qb = qs + (wave * BPW + bl) * PQ_STRIDE;
    const __amdgpu_buffer_rsrc_t h_rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.hc), 0, -1, 0x00020000);
    const unsigned laneoff = (unsigned)kl * (BF ? 8u : 16u);
    typedef typename PairH<BF>::T HT;
    f32x4 yv[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) yv[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    // ONE buffer of nine pieces, refilled piece by piece: piece m of pair k + 1 is requested right behind the use of piece m
    // of pair k, so nine loads stay in flight at a constant wait count (8) with half the registers of two whole-row buffers
    for (int k0 = 0; k0 < dmax; k0 += PQ_SLOTS) {
        if (k0) pair_wave_sync();
        pair_softmax_chunk<LPB>(p, qb, kl, e0, d, k0, bc, a);
        pair_wave_sync();
        const int cnt = min(dmax - k0, PQ_SLOTS);
        HT h[FGC_M];
        pair_load_h<COUT, BF>(h_rs, qb, laneoff, h);
        FGC_PAIR_KLOOP_PRAGMA
        for (int k = 0; k < cnt; ++k) {
            const float* qk = qb + k * PQ_LD;
            const unsigned noff = pair_row_off<COUT, BF>(qb + min(k + 1, PQ_SLOTS - 1) * PQ_LD, laneoff);
            const f32x4 q0 = *reinterpret_cast<const f32x4*>(qk), q1 = *reinterpret_cast<const f32x4*>(qk + 4);
            const f32x2c q8m = *reinterpret_cast<const f32x2c*>(qk + 8), mm = *reinterpret_cast<const f32x2c*>(qk + 10);
            const float q[FGC_M] = {q0[0], q0[1], q0[2], q0[3], q1[0], q1[1], q1[2], q1[3], q8m[0]};
            const unsigned mu = __float_as_uint(mm[0]);
            f32x4 t = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int m = 0; m < FGC_M; ++m) {
                t += q[m] * pair_h(h[m]);
                h[m] = pair_load_piece<COUT, BF>(h_rs, noff, m);
            }
            yv[0] += (float)(mu & 0xffu) * t;
            yv[1] += (float)((mu >> 8) & 0xffu) * t;
            yv[2] += (float)((mu >> 16) & 0xffu) * t;
            yv[3] += (float)(mu >> 24) * t;
        }
    }
    if (b >= p.nb) return;
    const f32x4 bias = *reinterpret_cast<const f32x4*>(p.bias + kl * 4);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const float inv = deg[i] > 0 ? 1.0f / (float)deg[i] : 0.f;
        f32x4 o = yv[i] * inv;
        if (!p.bias_mask || deg[i] > 0) o += bias;
        if (p.act) {
#pragma unroll
            for (int t = 0; t < 4; ++t) o[t] = fmaxf(o[t], 0.f) - p.alpha * fmaxf(-o[t], 0.f);
        }
        pair_st4<BF>(p.y, (size_t)(4 * b + i) * COUT + kl * 4, o);
    }
}

// sum over the LPB lanes of a block, result in every lane (DPP: xor 1, xor 2, mirror within 8, mirror within 16)
template <int LPB>
__device__ __forceinline__ float pair_block_sum(float v) {
    v += fgc_dpp_c<0xB1>(v);
    v += fgc_dpp_c<0x4E>(v);
    v += fgc_dpp_c<0x141>(v);
    if constexpr (LPB == 16) v += fgc_dpp_c<0x140>(v);
    return v;
}

template <int COUT, bool BF>
__global__ __launch_bounds__(256, FGC_PAIR_LB_BWD) void pair_bwd_logits_kernel(PairParams p) {
    constexpr int LPB = COUT / 4, BPW = 64 / LPB, BPG = 4 * BPW;
    constexpr int RED_LD = COUT + 12;
    __shared__ __attribute__((aligned(16))) float qs[BPG * PQ_STRIDE];
    __shared__ __attribute__((aligned(16))) float red[BPG * RED_LD];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, bl = lane / LPB, kl = lane % LPB;
    const int wg = xcd_tile(blockIdx.x, gridDim.x);
    const int b = (wg * 4 + wave) * BPW + bl;
    const bool valid = b < p.nb;
    const int bc = min(b, p.nb - 1);
    const int e0 = p.prow[bc];
    const int d = valid ? p.prow[bc + 1] - e0 : 0;
    float a[FGC_M];
    {
        const float* ar = p.ag + (size_t)bc * FGC_AG_LD;
        const f32x4 a0 = *reinterpret_cast<const f32x4*>(ar), a1 = *reinterpret_cast<const f32x4*>(ar + 4);
        a[0] = a0[0]; a[1] = a0[1]; a[2] = a0[2]; a[3] = a0[3]; a[4] = a1[0]; a[5] = a1[1]; a[6] = a1[2]; a[7] = a1[3];
        a[8] = ar[8];
    }
    // s_i = dy_i lrelu'(y_i) / deg_i of the four children, db partial of this lane's four channels
    f32x4 sv[4];
    f32x4 dbv = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int dg = p.rowptr[4 * bc + i + 1] - p.rowptr[4 * bc + i];
        const size_t o = (size_t)(4 * bc + i) * COUT + kl * 4;
        f32x4 g = pair_ld4<BF>(p.dy, o);
        if (p.act) {
            const f32x4 yy = pair_ld4<BF>(p.yact, o);
#pragma unroll
            for (int t = 0; t < 4; ++t) g[t] *= yy[t] > 0.f ? 1.f : (yy[t] < 0.f ? p.alpha : 0.f);
        }
        if (!valid) g = f32x4{0.f, 0.f, 0.f, 0.f};
        if (!p.bias_mask || dg > 0) dbv += g;
        sv[i] = g * (dg > 0 ? 1.0f / (float)dg : 0.f);
    }
    const int dmax = wave_max_i32(d);
    float* qb = qs + (wave * BPW + bl) * PQ_STRIDE;
    const __amdgpu_buffer_rsrc_t h_rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.hc), 0, -1, 0x00020000);
    const unsigned laneoff = (unsigned)kl * (BF ? 8u : 16u);
    typedef typename PairH<BF>::T HT;
    float da[FGC_M];
#pragma unroll
    for (int m = 0; m < FGC_M; ++m) da[m] = 0.f;
    // dt / dl rows through bounded buffer descriptors: a slot past the block's degree stores at an offset beyond the
    // buffer, which the hardware drops - no exec-masked store in the loop
    const __amdgpu_buffer_rsrc_t dt_rs = __builtin_amdgcn_make_buffer_rsrc(p.dt, 0, p.dt_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t dl_rs = __builtin_amdgcn_make_buffer_rsrc(p.dl, 0, p.dl_bytes, 0x00020000);
    for (int k0 = 0; k0 < dmax; k0 += PQ_SLOTS) {
        if (k0) pair_wave_sync();
        pair_softmax_chunk<LPB>(p, qb, kl, e0, d, k0, bc, a);
        pair_wave_sync();
        const int cnt = min(dmax - k0, PQ_SLOTS);
        HT h[FGC_M];
        pair_load_h<COUT, BF>(h_rs, qb, laneoff, h);
        FGC_PAIR_KLOOP_PRAGMA
        for (int k = 0; k < cnt; ++k) {
            const float* qk = qb + k * PQ_LD;
            const int kk = k0 + k;
            const unsigned noff = pair_row_off<COUT, BF>(qb + min(k + 1, PQ_SLOTS - 1) * PQ_LD, laneoff);
            const f32x4 q0 = *reinterpret_cast<const f32x4*>(qk), q1 = *reinterpret_cast<const f32x4*>(qk + 4);
            const f32x2c q8m = *reinterpret_cast<const f32x2c*>(qk + 8), mm = *reinterpret_cast<const f32x2c*>(qk + 10);
            const float q[FGC_M] = {q0[0], q0[1], q0[2], q0[3], q1[0], q1[1], q1[2], q1[3], q8m[0]};
            const unsigned mu = __float_as_uint(mm[0]);
            f32x4 dt = (float)(mu & 0xffu) * sv[0];
            dt += (float)((mu >> 8) & 0xffu) * sv[1];
            dt += (float)((mu >> 16) & 0xffu) * sv[2];
            dt += (float)(mu >> 24) * sv[3];
            const bool live = kk < d;
            const unsigned erow = (unsigned)(e0 + kk);
            if constexpr (BF) {
                const u32x2 w = f4_to_bf4(dt);
                __builtin_amdgcn_raw_buffer_store_b64(w, dt_rs,
                                                      live ? erow * (unsigned)(COUT * 2) + (unsigned)kl * 8u : 0xFFFFFFF0u, 0, 0);
                dt = bf4_to_f4(w);      // (the data kernel gathers the stored, rounded rows)
            } else {
                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, dt), dt_rs,
                                                       live ? erow * (unsigned)(COUT * 4) + (unsigned)kl * 16u : 0xFFFFFFF0u, 0, 0);
            }
            float dq[FGC_M], sum = 0.f;
#pragma unroll
            for (int m = 0; m < FGC_M; ++m) {
                const f32x4 hv = pair_h(h[m]);
                h[m] = pair_load_piece<COUT, BF>(h_rs, noff, m);
                float v = dt[0] * hv[0];
                v = fmaf(dt[1], hv[1], v);
                v = fmaf(dt[2], hv[2], v);
                v = fmaf(dt[3], hv[3], v);
                dq[m] = pair_block_sum<LPB>(v);
                sum = fmaf(q[m], dq[m], sum);
            }
            float dlv[FGC_M];
#pragma unroll
            for (int m = 0; m < FGC_M; ++m) {
                dlv[m] = q[m] * (dq[m] - sum);
                da[m] += dlv[m];
            }
            // the block's lanes 0, 1, 2 store the three 16-byte pieces of the pair's d-logit row
            const f32x4 piece = kl == 0 ? f32x4{dlv[0], dlv[1], dlv[2], dlv[3]}
                                        : (kl == 1 ? f32x4{dlv[4], dlv[5], dlv[6], dlv[7]} : f32x4{dlv[8], 0.f, 0.f, 0.f});
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, piece), dl_rs,
                                                   (live && kl < 3) ? erow * (unsigned)(FGC_DL_LD * 4) + (unsigned)kl * 16u : 0xFFFFFFF0u, 0, 0);
        }
    }
    if (valid && kl == 0) {
        float* o = p.dag + (size_t)b * FGC_AG_LD;
        *reinterpret_cast<f32x4*>(o) = f32x4{da[0], da[1], da[2], da[3]};
        *reinterpret_cast<f32x4*>(o + 4) = f32x4{da[4], da[5], da[6], da[7]};
        *reinterpret_cast<f32x4*>(o + 8) = f32x4{da[8], 0.f, 0.f, 0.f};
    }
    // db / dc partials of the workgroup: fixed order over its blocks
    {
        float* rr = red + (wave * BPW + bl) * RED_LD;
        *reinterpret_cast<f32x4*>(rr + kl * 4) = dbv;
        if (kl == 0) {
#pragma unroll
            for (int m = 0; m < FGC_M; ++m) rr[COUT + m] = valid ? da[m] : 0.f;
        }
    }
    __syncthreads();
    const int tid = threadIdx.x;
    if (tid < COUT + 12) {
        float v = 0.f;
        if (tid < COUT + FGC_M)
            for (int t = 0; t < BPG; ++t) v += red[t * RED_LD + tid];
        if (tid < COUT) p.db_part[(size_t)wg * COUT + tid] = v;
        else p.dc_part[(size_t)wg * 12 + (tid - COUT)] = v;
    }
}

int launch_pair_fwd(const fgc_conv_desc* d, float* ag, float* y, void* workspace, hipStream_t st) {
    const int cin = d->c0, cout = d->cout;
    const int all_rows = d->src_rows > 0 ? d->src_rows : (d->n >> 2);
    // source rows this call transforms (fgc_conv_desc.proj_row0 / proj_rows, as for the logit table of the fine form)
    const int row0 = d->proj_rows ? d->proj_row0 : 0;
    const int rows = d->proj_rows ? (d->proj_rows < 0 ? 0 : d->proj_rows) : all_rows;
    FGC_CHECK_ARG(row0 >= 0 && row0 + rows <= all_rows, "fgc_conv_fwd: pair form: rows [%d, %d) outside the %d source rows", row0,
                  row0 + rows, all_rows);
    const bool blocks = !(d->tile_list && d->n_tiles == 0);
    const size_t esz = (d->flags & FGC_CONV_BF16) ? 2 : 4;
    const float* x0 = reinterpret_cast<const float*>(reinterpret_cast<const char*>(d->x0) + (size_t)row0 * cin * esz);
    float* hc0 = reinterpret_cast<float*>(reinterpret_cast<char*>(d->hc) + (size_t)row0 * FGC_M * cout * esz);
    float* ag0 = ag + (size_t)row0 * FGC_AG_LD;
    const int tc = (FGC_M * cout) / 16 + 2;
    const int ngroups = cdiv(tc, FGC_PT_CG);
    const int items = cdiv(rows, 16 * FGC_PT_RT) * ngroups;
    const bool bf16 = (d->flags & FGC_CONV_BF16) != 0;
    if (rows == 0) {
        // nothing to transform in this call
    } else if (bf16) {
        unsigned short* Wb = (unsigned short*)workspace;
        if (!(d->flags & FGC_CONV_PACKED)) {
            const int count = FGC_M * cout * cin;
            FGC_LAUNCH("pair_weight_bf16_kernel", st, pair_weight_bf16_kernel, dim3(cdiv(count, 1024)), dim3(256), 0, d->W0, Wb, count);
        }
#define FGC_PTB(CIN_)                                                                                                  \
    FGC_LAUNCH("pair_transform_kernel", st, (pair_transform_bf16_kernel<CIN_, FGC_PT_RT>), dim3(cdiv(items, 4)), dim3(256), 0,      \
               (const unsigned short*)x0, rows, Wb, d->u, d->c, d->v, cout, (unsigned short*)hc0, ag0, ngroups)
        if (cin == 32) FGC_PTB(32);
        else if (cin == 64) FGC_PTB(64);
        else FGC_PTB(128);
#undef FGC_PTB
    } else {
#define FGC_PT(CIN_)                                                                                                   \
    FGC_LAUNCH("pair_transform_kernel", st, (pair_transform_kernel<CIN_, FGC_PT_RT>), dim3(cdiv(items, 4)), dim3(256), 0, x0, rows,  \
               d->W0, d->u, d->c, d->v, cout, hc0, ag0, ngroups)
    if (cin == 32) FGC_PT(32);
    else if (cin == 64) FGC_PT(64);
    else FGC_PT(128);
#undef FGC_PT
    }
    FGC_CHECK_LAUNCH("fgc_conv_fwd/pair_transform");
    if (!blocks) return FGC_OK;
    PairParams p{};
    p.nb = d->n >> 2;
    p.prow = d->pair_rowptr;
    p.pcol = d->pair_col;
    p.pmul = d->pair_mul;
    p.ag = ag;
    p.hc = d->hc;
    p.rowptr = d->rowptr;
    p.bias = d->b;
    p.bias_mask = d->bias_mask;
    p.act = d->act;
    p.alpha = d->alpha;
    p.y = y;
    const int grid = pair_num_wgs(d);
    if (bf16) {
        if (cout == 32) FGC_LAUNCH("pair_fwd_kernel", st, (pair_fwd_kernel<32, true>), dim3(grid), dim3(256), 0, p);
        else FGC_LAUNCH("pair_fwd_kernel", st, (pair_fwd_kernel<64, true>), dim3(grid), dim3(256), 0, p);
    } else {
        if (cout == 32) FGC_LAUNCH("pair_fwd_kernel", st, (pair_fwd_kernel<32, false>), dim3(grid), dim3(256), 0, p);
        else FGC_LAUNCH("pair_fwd_kernel", st, (pair_fwd_kernel<64, false>), dim3(grid), dim3(256), 0, p);
    }
    FGC_CHECK_LAUNCH("fgc_conv_fwd/pair_fwd");
    return FGC_OK;
}

int launch_pair_bwd_logits(const fgc_conv_desc* d, const fgc_conv_bwd_io* io, float* db_part, float* dc_part, hipStream_t st) {
    PairParams p{};
    p.nb = d->n >> 2;
    p.prow = d->pair_rowptr;
    p.pcol = d->pair_col;
    p.pmul = d->pair_mul;
    p.ag = io->ag;
    p.hc = d->hc;
    p.rowptr = d->rowptr;
    p.bias_mask = d->bias_mask;
    p.act = d->act;
    p.alpha = d->alpha;
    p.dy = io->dy;
    p.yact = io->y;
    p.dt = io->dt;
    p.dl = io->dl;
    p.dag = io->dag;
    p.db_part = db_part;
    p.dc_part = dc_part;
    p.dt_bytes = (unsigned)((size_t)d->n_pairs * d->cout * ((d->flags & FGC_CONV_BF16) ? 2 : 4));
    p.dl_bytes = (unsigned)((size_t)d->n_pairs * FGC_DL_LD * 4);
    const int grid = pair_num_wgs(d);
    if (d->flags & FGC_CONV_BF16) {
        if (d->cout == 32) FGC_LAUNCH("pair_bwd_logits_kernel", st, (pair_bwd_logits_kernel<32, true>), dim3(grid), dim3(256), 0, p);
        else FGC_LAUNCH("pair_bwd_logits_kernel", st, (pair_bwd_logits_kernel<64, true>), dim3(grid), dim3(256), 0, p);
    } else {
        if (d->cout == 32) FGC_LAUNCH("pair_bwd_logits_kernel", st, (pair_bwd_logits_kernel<32, false>), dim3(grid), dim3(256), 0, p);
        else FGC_LAUNCH("pair_bwd_logits_kernel", st, (pair_bwd_logits_kernel<64, false>), dim3(grid), dim3(256), 0, p);
    }
    FGC_CHECK_LAUNCH("fgc_conv_bwd/pair_logits");
    return FGC_OK;
}

}  // namespace fgc

extern "C" int fgc_conv_uses_pairs(const fgc_conv_desc* d) {
    FGC_OPT_SCOPE(d);
    return fgc::pairs_ok(d) ? 1 : 0;
}
extern "C" int fgc_conv_pairs_allowed(int64_t rows, int64_t n_pairs, int32_t max_pair_in_deg, int32_t cout) {
    return fgc::pairs_graph_allowed(rows, n_pairs, max_pair_in_deg, cout) ? 1 : 0;
}
