// Backward of the graph convolution (gradient of custom_conv2d, /root/reference/Code/model.py:427-504;
// the reference gets it from tf.gradients over ~25 materialised ops, train.py:520).
//
// With s_i = dy_i * lrelu'(y_i) / deg_i  (the gradient w.r.t. the per-node sum), a_i/g_j/q_ik as in forward:
//   K1  logits kernel (node i centred, forward CSR)
//         dz_i   = W^T s_i                        f32 MFMA  [T,cout] x [cout, 9*cin]   -> LDS
//         dq_ikm = <dz_i[m,:], x_j>               per edge, 8 lanes x float4, butterfly reduce
//         dl_ikm = q_ikm (dq_ikm - sum_m' q dq)   -> dl[e, 12]   (softmax backward)
//         da_i   = sum_k dl_ik                    -> dag[i, 0..8];   dc partial per workgroup
//   K2  data kernel (node j centred, TRANSPOSED CSR): the same fused core as forward
//         r_j[m,:] = sum_{i->j} q_ijm s_i         -> r[j, 9*cout]  (also the dW reduction's A operand)
//         dg_j     = sum_{i->j} dl_(i->j)         -> dag[j, 12..20]
//         dx_j     = r_j W  (f32 MFMA) + da_j u + dg_j v ; 4:1 row sum when the input was upsampled
//   K3  reductions over nodes (f32 MFMA, K = nodes):  dW0 = r^T x,  [du; dv] = dag^T x ; db, dc column sums
// No float atomics anywhere: every sum has a fixed order, results are bitwise reproducible.
#include <stdlib.h>

#include <algorithm>

#include "fgc_conv_w8.h"
#include "fgc_conv_narrow.h"
#include "fgc_conv_pair.h"
#include "fgc_reduce.h"
#include "fgc_pack.h"
#include "fgc_split.h"

namespace fgc {

int validate_conv_desc(const fgc_conv_desc* d, const char* who);
bool conv_vec4_ok(const fgc_conv_desc* d);
void fill_core_params(CoreParams& p, const ConvGeom& g, int n, const int* rowptr, const int* col, const int* eid,
                      const float* s0, const float* s1, int c0, int c1, int shift, int nout, const float* ag,
                      int ag_shift, int ctr_off, int nbr_off, const float* Wp);
size_t conv_smem_bytes(const ConvGeom& g, size_t extra);
__global__ void pack_weight_kernel(const float* __restrict__ W0, float* __restrict__ Wp, int cin, int cout, int kdim,
                                   int ncols, int npad, int kc, int kpass, int passes, int transposed);

__device__ __forceinline__ float slope_from_y(float y, float alpha) { return y > 0.f ? 1.f : (y < 0.f ? alpha : 0.f); }

// What the 4:1 max pooling sends back to row r, column col of its input y: the pooled gradient split evenly over the rows
// of the group that equal the maximum (tf.reduce_max's gradient; fgc_pool4_bwd as a term of fgc_conv_bwd_io.pool_dy).
__device__ __forceinline__ float pool4_grad_term(const float* __restrict__ y, const float* __restrict__ pool_y,
                                                 const float* __restrict__ pool_dy, int r, int col, int cout, int bf16) {
    const size_t pi = (size_t)(r >> 2) * cout + col;
    const float m = ld_act(pool_y, pi, bf16);
    const size_t b = (size_t)(r & ~3) * cout + col;
    float ne = 0.f;
#pragma unroll
    for (int k = 0; k < 4; ++k) ne += ld_act(y, b + (size_t)k * cout, bf16) == m ? 1.f : 0.f;
    return ld_act(y, (size_t)r * cout + col, bf16) == m ? ld_act(pool_dy, pi, bf16) / ne : 0.f;
}

// ---------------------------------------------------------------------------------------------
// s = dy * lrelu'(y) / deg ; db partial column sums of dy * lrelu'(y) over rows that got the bias
// ---------------------------------------------------------------------------------------------
// workgroup = (256 / cp2) row lanes x cp2 columns (cp2 = cout rounded up to a power of two); coalesced over columns
// in_bf16: dy and y are bf16 tensors; out_bf16: so is ds (FGC_CONV_BF16; a narrow first layer keeps its ds in fp32)
__global__ __launch_bounds__(256) void ds_db_kernel(const float* __restrict__ dy, const float* __restrict__ y,
                                                    const int* __restrict__ rowptr, int n, int cout, int cp2, int act,
                                                    float alpha, int bias_mask, int rows_per_block,
                                                    float* __restrict__ ds, float* __restrict__ db_part, int in_bf16,
                                                    int out_bf16, const float* __restrict__ pool_y,
                                                    const float* __restrict__ pool_dy) {
    __shared__ float part[256];
    const int col = threadIdx.x % cp2, rl = threadIdx.x / cp2, nrl = 256 / cp2;
    const int r0 = blockIdx.x * rows_per_block;
    const int r1 = min(n, r0 + rows_per_block);
    float acc = 0.f;
    if (col < cout) {
        for (int r = r0 + rl; r < r1; r += nrl) {
            const int d = rowptr[r + 1] - rowptr[r];
            float g = ld_act(dy, (size_t)r * cout + col, in_bf16);
            if (pool_dy) g += pool4_grad_term(y, pool_y, pool_dy, r, col, cout, in_bf16);
            if (act) g *= slope_from_y(ld_act(y, (size_t)r * cout + col, in_bf16), alpha);
            if (!bias_mask || d > 0) acc += g;
            st_act(ds, (size_t)r * cout + col, d > 0 ? g / (float)d : 0.f, out_bf16);
        }
    }
    part[threadIdx.x] = acc;
    __syncthreads();
    if (rl == 0 && col < cout) {
        float v = 0.f;
        for (int t = 0; t < nrl; ++t) v += part[t * cp2 + col];
        db_part[(size_t)blockIdx.x * cout + col] = v;
    }
}

// Vector form for 16-byte aligned tensors whose width is a multiple of 8 (bf16 input) or 4 (fp32 input) and divides
// 256 chunks evenly: a thread owns one 16-byte chunk of dy per row (8 or 4 columns), so a row of the pooling term costs
// seven 16-byte loads instead of seven scalar ones per column (the scalar kernel: 23 us fp32 / 35 us bf16 on the first
// layer's 122k x 32 tensor, far above its 55 MB of traffic).  Same operations per element, so the same s bit for bit;
// the bias-gradient partials group their rows differently (256 / chunks-per-row row lanes).
template <bool BF_IN, bool BF_OUT>
__global__ __launch_bounds__(256) void ds_db_vec_kernel(const float* __restrict__ dy, const float* __restrict__ y,
                                                        const int* __restrict__ rowptr, int n, int cout, int act,
                                                        float alpha, int bias_mask, int rows_per_block,
                                                        float* __restrict__ ds, float* __restrict__ db_part,
                                                        const float* __restrict__ pool_y, const float* __restrict__ pool_dy) {
    constexpr int V = BF_IN ? 8 : 4;
    __shared__ float part[256 * V];
    const int cpr = cout / V, c = threadIdx.x % cpr, rl = threadIdx.x / cpr, nrl = 256 / cpr;
    const int r0 = blockIdx.x * rows_per_block;
    const int r1 = min(n, r0 + rows_per_block);
    auto ld = [&](const float* base, size_t chunk, float (&o)[V]) {
        if constexpr (BF_IN) {
            const u32x4 w = reinterpret_cast<const u32x4*>(base)[chunk];
            const f32x4 a = bf4_to_f4(u32x2{w[0], w[1]}), b = bf4_to_f4(u32x2{w[2], w[3]});
            o[0] = a[0]; o[1] = a[1]; o[2] = a[2]; o[3] = a[3]; o[4] = b[0]; o[5] = b[1]; o[6] = b[2]; o[7] = b[3];
        } else {
            const f32x4 a = reinterpret_cast<const f32x4*>(base)[chunk];
            o[0] = a[0]; o[1] = a[1]; o[2] = a[2]; o[3] = a[3];
        }
    };
    float acc[V];
#pragma unroll
    for (int k = 0; k < V; ++k) acc[k] = 0.f;
    for (int r = r0 + rl; r < r1; r += nrl) {
        const int d = rowptr[r + 1] - rowptr[r];
        float g[V], yv[V];
        ld(dy, (size_t)r * cpr + c, g);
        if (act || pool_dy) ld(y, (size_t)r * cpr + c, yv);
        if (pool_dy) {
            float m[V], gp[V], ne[V];
            const size_t pi = (size_t)(r >> 2) * cpr + c;
            ld(pool_y, pi, m);
            ld(pool_dy, pi, gp);
#pragma unroll
            for (int k = 0; k < V; ++k) ne[k] = 0.f;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                float yq[V];
                ld(y, (size_t)((r & ~3) + q) * cpr + c, yq);
#pragma unroll
                for (int k = 0; k < V; ++k) ne[k] += yq[k] == m[k] ? 1.f : 0.f;
            }
#pragma unroll
            for (int k = 0; k < V; ++k) g[k] += yv[k] == m[k] ? gp[k] / ne[k] : 0.f;
        }
        if (act) {
#pragma unroll
            for (int k = 0; k < V; ++k) g[k] *= slope_from_y(yv[k], alpha);
        }
        if (!bias_mask || d > 0) {
#pragma unroll
            for (int k = 0; k < V; ++k) acc[k] += g[k];
        }
        float sv[V];
#pragma unroll
        for (int k = 0; k < V; ++k) sv[k] = d > 0 ? g[k] / (float)d : 0.f;
        if constexpr (BF_OUT) {
            static_assert(!BF_OUT || BF_IN, "bf16 output comes with bf16 input");
            const u32x2 b0 = f4_to_bf4(f32x4{sv[0], sv[1], sv[2], sv[3]}), b1 = f4_to_bf4(f32x4{sv[4 % V], sv[5 % V], sv[6 % V], sv[7 % V]});
            reinterpret_cast<u32x4*>(ds)[(size_t)r * cpr + c] = u32x4{b0[0], b0[1], b1[0], b1[1]};
        } else {
#pragma unroll
            for (int k = 0; k < V; k += 4)
                reinterpret_cast<f32x4*>(ds)[((size_t)r * cout + c * V + k) >> 2] = f32x4{sv[k], sv[k + 1], sv[k + 2], sv[k + 3]};
        }
    }
#pragma unroll
    for (int k = 0; k < V; ++k) part[(rl * cpr + c) * V + k] = acc[k];   // = part[rl][column]
    __syncthreads();
    if ((int)threadIdx.x < cout) {
        float v = 0.f;
        for (int t = 0; t < nrl; ++t) v += part[t * cout + threadIdx.x];
        db_part[(size_t)blockIdx.x * cout + threadIdx.x] = v;
    }
}

// ---------------------------------------------------------------------------------------------
// K1 operand: Wq[pass][o/4][kk][o%4] = W0[m][o][pass*kc+cl], kk = m*kc+cl  (K = cout, N = kpass)
// ---------------------------------------------------------------------------------------------
__global__ void pack_logit_weight_kernel(const float* __restrict__ W0, float* __restrict__ Wq, int cin, int cout,
                                         int opad, int kc, int kpass, int passes) {
    pack_logit_weight_body(W0, Wq, cin, cout, opad, kc, kpass, passes, blockIdx.x, gridDim.x);
}

__global__ __launch_bounds__(256) void pack_many_kernel(PackJobs J) {
    int q = 0;
#pragma unroll
    for (int t = 1; t < PACK_MAX_JOBS; ++t)
        if (t < J.njobs && (int)blockIdx.x >= J.job[t].block0) q = t;
    const PackJob& j = J.job[q];
    const int bid = blockIdx.x - j.block0;
    const int nb = (q + 1 < J.njobs ? J.job[q + 1].block0 : J.nblocks) - j.block0;
    if (j.kind >= 8) {
        if (j.kind == 14) rotate_logits_body(j.W0, j.dst, j.kdim, j.cin, j.aux, j.lg_u, j.lg_c, j.lg_v, j.lg_ag, bid, nb);
        else if (j.kind == 8) rotate_rows_body(j.W0, j.dst, j.kdim, j.aux, bid, nb);
        else if (j.kind == 9) mlp_pack_body(j.W0, j.dst, j.cin, j.kdim, j.ncols, bid, nb);
        else if (j.kind == 10) mlp_pack_split_body(j.W0, (unsigned short*)j.dst, j.cin, j.ncols, bid, nb);
        else if (j.kind == 11) mlp_pack_bf16_body(j.W0, (unsigned short*)j.dst, j.cin, j.ncols, bid, nb);
        else if (j.kind == 12) mlp_pack_w1dx_bf16_body(j.W0, (unsigned short*)j.dst, j.cin, j.ncols, bid, nb);
        else if (j.kind == 15) mlp_pack_w1dx_split_body(j.W0, (unsigned short*)j.dst, j.cin, j.ncols, bid, nb);
        else if (j.kind == 16) mlp_pack_w2_split_body(j.W0, (u32x4*)j.dst, j.ncols, j.cout, bid);
        else if (j.kind == 17) pack_logit_weight_split_body(j.W0, (unsigned short*)j.dst, j.cin, j.cout, j.passes, bid, nb);
        else mlp_pack_w2_bf16_body(j.W0, (u32x4*)j.dst, j.ncols, j.cout, bid);
    } else if (j.kind == 7) pack_plain_bf16_body(j.W0, (unsigned short*)j.dst, j.kdim, bid, nb);
    else if (j.kind == 6) pack_logit_weight_bf16_body(j.W0, (unsigned short*)j.dst, j.cin, j.cout, j.passes, bid, nb);
    else if (j.kind >= 4) pack_weight_bf16_body(j.W0, (unsigned short*)j.dst, j.cin, j.cout, j.kdim, j.ncols, j.npad, j.passes,
                                                j.kind - 4, bid, nb);
    else if (j.kind == 2) pack_logit_weight_body(j.W0, j.dst, j.cin, j.cout, j.opad, j.kc, j.kpass, j.passes, bid, nb);
    else pack_weight_body(j.W0, j.dst, j.cin, j.cout, j.kdim, j.ncols, j.npad, j.kc, j.kpass, j.passes, j.kind, bid, nb);
}

struct LogitParams {
    const float* ds;     // [n, cout]
    int cout, opad, ostride;  // opad = roundup16(cout); LDS stride of the ds tile (== 8 mod 16)
    const float* Wq;
    float* dl;           // [nnz, 12]
    float* dag;          // [n, 24]  (writes 0..8)
    float* dc_part;      // [grid, 12]
    // fused s = dy * lrelu'(y) / deg (+ db partials): set when the d-logits kernel computes s itself (dy != NULL)
    const float* dy;
    const float* y;
    int act, bias_mask;
    float alpha;
    float* ds_out;       // [n, cout]
    float* db_part;      // [cdiv(n, TILE), cout]
    int a_global;        // LONG form, cout % 16 == 0: the dz GEMM reads its ds operand from global memory (the tile is
                         // L1-resident) instead of an LDS copy, which keeps two workgroups per CU for wide layers
    const float* pool_y;   // fused prologue only: fgc_conv_bwd_io.pool_y / pool_dy (NULL = no pooled gradient to fold in)
    const float* pool_dy;
};

constexpr int K1_CTW = 5;  // column tiles of dz per wave: kpass/16 <= 18 -> ceil(18/4)
// developer knock-outs of the deep d-logits kernel for phase timing (results are wrong with any bit set; never set in the
// shipped build): 1 = no dz-GEMM MFMAs, 2 = no packed-weight loads, 4 = no dz tile stores, 8 = no per-node product MFMAs,
// 16 = no neighbour-row gathers, 32 = no softmax backward / dl stores, 64 = no dl stores (softmax backward kept)
#ifndef FGC_KO1
#define FGC_KO1 0
#endif

template <int LPN, bool VEC4>
__global__ __launch_bounds__(NTHREADS) void conv_bwd_logits_kernel(CoreParams p, LogitParams lp) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    const Smem s = carve(smem_raw, ZSTRIDE);
    float* dst = s.extra;                       // ds tile [TILE][ostride]
    float* red = dst + TILE * lp.ostride;       // [4][12] block reduction scratch
    const int tile0 = xcd_tile(blockIdx.x, gridDim.x) * TILE;
    const int tid = threadIdx.x;
    const int wave = tid >> 6, lane = tid & 63, lr = lane & 15, lq = lane >> 4;
    constexpr int SLOTS = KMAX / LPN;           // edges owned per lane: k = slot*LPN + cl

    // ds tile -> LDS (zero padded)
    for (int t = tid; t < TILE * lp.opad; t += NTHREADS) {
        const int r = t / lp.opad, o = t % lp.opad;
        const int i = tile0 + r;
        dst[r * lp.ostride + o] = (i < p.n && o < lp.cout) ? lp.ds[(size_t)i * lp.cout + o] : 0.f;
    }
    const int dmine = softmax_phase<false>(p, s, tile0, 0, nullptr, nullptr);
    const int nchunks = edge_chunks(s, dmine);

    const int node = tid / LPN, cl = tid % LPN;
    const bool worker = node < TILE;
    float dcacc[FGC_M];
#pragma unroll
    for (int m = 0; m < FGC_M; ++m) dcacc[m] = 0.f;
    float daacc[FGC_M];
#pragma unroll
    for (int m = 0; m < FGC_M; ++m) daacc[m] = 0.f;

    const int nct = KPASS >> 4;
    const int okg = lp.opad >> 4;
    const f32x4* Wq4 = reinterpret_cast<const f32x4*>(lp.Wq);

    for (int ch = 0; ch < nchunks; ++ch) {
        const int kbase = ch * KMAX;
        if (ch > 0) {
            __syncthreads();
            softmax_phase<false>(p, s, tile0, kbase, nullptr, nullptr);
            __syncthreads();
        }
        float dq[SLOTS][FGC_M];
#pragma unroll
        for (int sl = 0; sl < SLOTS; ++sl)
#pragma unroll
            for (int m = 0; m < FGC_M; ++m) dq[sl][m] = 0.f;

        for (int pass = 0; pass < p.passes; ++pass) {
            // ---- dz tile = ds tile x Wq[pass]  (MFMA), C layout -> ztile
            f32x4 acc[RT][K1_CTW];
#pragma unroll
            for (int r = 0; r < RT; ++r)
#pragma unroll
                for (int c = 0; c < K1_CTW; ++c) acc[r][c] = f32x4{0.f, 0.f, 0.f, 0.f};
            for (int g = 0; g < okg; ++g) {
                f32x4 a[RT];
#pragma unroll
                for (int r = 0; r < RT; ++r)
                    a[r] = *reinterpret_cast<const f32x4*>(dst + (r * 16 + lr) * lp.ostride + g * 16 + lq * 4);
#pragma unroll
                for (int c = 0; c < K1_CTW; ++c) {
                    const int ct = wave + c * 4;
                    if (ct >= nct) continue;
                    const f32x4 b = Wq4[((size_t)pass * (lp.opad >> 2) + g * 4 + lq) * KPASS + ct * 16 + lr];
#pragma unroll
                    for (int r = 0; r < RT; ++r)
#pragma unroll
                        for (int t = 0; t < 4; ++t)
                            acc[r][c] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[r][t], b[t], acc[r][c], 0, 0, 0);
                }
            }
            __syncthreads();  // previous pass' readers of ztile are done
#pragma unroll
            for (int c = 0; c < K1_CTW; ++c) {
                const int ct = wave + c * 4;
                if (ct >= nct) continue;
#pragma unroll
                for (int r = 0; r < RT; ++r)
#pragma unroll
                    for (int t = 0; t < 4; ++t)
                        s.ztile[(size_t)(r * 16 + lq * 4 + t) * ZSTRIDE + ct * 16 + lr] = acc[r][c][t];
            }
            __syncthreads();
            // ---- per edge: dq[m] += <dz_i[m, chunk], x_j[chunk]>
            if (worker) {
                f32x4 dz[FGC_M];
                const float* zr = s.ztile + (size_t)node * ZSTRIDE + cl * 4;
#pragma unroll
                for (int m = 0; m < FGC_M; ++m) dz[m] = *reinterpret_cast<const f32x4*>(zr + m * KC);
                const int d = min(max(s.deg[node] - kbase, 0), KMAX);
                const int cbase = pass * KC + cl * 4;
                const float* qb = s.qbuf + (size_t)node * KMAX * QLD;
#pragma unroll
                for (int sl = 0; sl < SLOTS; ++sl) {
                    for (int kk = 0; kk < LPN; ++kk) {
                        const int k = sl * LPN + kk;
                        if (k >= d) break;
                        const f32x4 xv = load_chunk<VEC4>(p, __float_as_int(qb[k * QLD + 9]), cbase);
                        float part[FGC_M];
#pragma unroll
                        for (int m = 0; m < FGC_M; ++m) {
                            float v = dz[m][0] * xv[0];
                            v = fmaf(dz[m][1], xv[1], v);
                            v = fmaf(dz[m][2], xv[2], v);
                            v = fmaf(dz[m][3], xv[3], v);
#pragma unroll
                            for (int off = 1; off < LPN; off <<= 1) v += __shfl_xor(v, off);
                            part[m] = v;
                        }
                        if (kk == cl) {
#pragma unroll
                            for (int m = 0; m < FGC_M; ++m) dq[sl][m] += part[m];
                        }
                    }
                }
            }
        }
        // ---- softmax backward for the edges this lane owns
        if (worker) {
            const int i = tile0 + node;
            const int d = min(max(s.deg[node] - kbase, 0), KMAX);
            const float* qb = s.qbuf + (size_t)node * KMAX * QLD;
            const int e0 = i < p.n ? p.rowptr[i] + kbase : 0;
#pragma unroll
            for (int sl = 0; sl < SLOTS; ++sl) {
                const int k = sl * LPN + cl;
                if (k < d) {
                    float q[FGC_M];
                    const f32x4 q0 = *reinterpret_cast<const f32x4*>(qb + k * QLD);
                    const f32x4 q1 = *reinterpret_cast<const f32x4*>(qb + k * QLD + 4);
                    q[0] = q0[0]; q[1] = q0[1]; q[2] = q0[2]; q[3] = q0[3];
                    q[4] = q1[0]; q[5] = q1[1]; q[6] = q1[2]; q[7] = q1[3];
                    q[8] = qb[k * QLD + 8];
                    float dot = 0.f;
#pragma unroll
                    for (int m = 0; m < FGC_M; ++m) dot = fmaf(q[m], dq[sl][m], dot);
                    float dlv[FGC_M];
#pragma unroll
                    for (int m = 0; m < FGC_M; ++m) {
                        dlv[m] = q[m] * (dq[sl][m] - dot);
                        daacc[m] += dlv[m];
                        dcacc[m] += dlv[m];
                    }
                    float* o = lp.dl + (size_t)(e0 + k) * FGC_DL_LD;
                    *reinterpret_cast<f32x4*>(o) = f32x4{dlv[0], dlv[1], dlv[2], dlv[3]};
                    *reinterpret_cast<f32x4*>(o + 4) = f32x4{dlv[4], dlv[5], dlv[6], dlv[7]};
                    *reinterpret_cast<f32x4*>(o + 8) = f32x4{dlv[8], 0.f, 0.f, 0.f};
                }
            }
        }
    }
    // da_i: sum over the node's LPN lanes
#pragma unroll
    for (int m = 0; m < FGC_M; ++m) {
        float v = daacc[m];
#pragma unroll
        for (int off = 1; off < LPN; off <<= 1) v += __shfl_xor(v, off);
        daacc[m] = v;
    }
    if (worker && cl == 0 && tile0 + node < p.n) {
        float* o = lp.dag + (size_t)(tile0 + node) * FGC_AG_LD;
        *reinterpret_cast<f32x4*>(o) = f32x4{daacc[0], daacc[1], daacc[2], daacc[3]};
        *reinterpret_cast<f32x4*>(o + 4) = f32x4{daacc[4], daacc[5], daacc[6], daacc[7]};
        *reinterpret_cast<f32x4*>(o + 8) = f32x4{daacc[8], 0.f, 0.f, 0.f};
    }
    // dc partial of this workgroup (fixed order: wave butterfly, then 4 waves)
#pragma unroll
    for (int m = 0; m < FGC_M; ++m) {
        float v = dcacc[m];
        for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
        dcacc[m] = v;
    }
    __syncthreads();
    if (lane == 0) {
#pragma unroll
        for (int m = 0; m < FGC_M; ++m) red[wave * 12 + m] = dcacc[m];
    }
    __syncthreads();
    if (tid < 12) {
        const float v = tid < FGC_M ? (red[tid] + red[12 + tid]) + (red[24 + tid] + red[36 + tid]) : 0.f;
        lp.dc_part[(size_t)blockIdx.x * 12 + tid] = v;
    }
}

// ---------------------------------------------------------------------------------------------
// K1, matrix-core form (wide layers: 32-channel passes, float4 rows, degree <= 24).
// The per-edge products dq[m][k] = <dz_i[m,:], x_j(k)> of one node are a [9 x 32] x [32 x d] matrix product: the
// wave that owns the node issues it as 16x16x4 MFMAs (rows = m, columns = the node's edges, K = channels):
//   A fragment  dz_i[m = lane&15][cb + 4*(lane>>4) .. +3]        one ds_read_b128 from the dz tile
//   B fragment  x_j(k = lane&15)[cb + 4*(lane>>4) .. +3]         one global dwordx4 per lane (64 B per edge)
// so the VALU only does the softmax and its backward; no per-edge FMA chains, no butterfly per edge.
// ---------------------------------------------------------------------------------------------
constexpr int NPW = TILE / 4;  // nodes per wave

template <bool VEC4>
__global__ __launch_bounds__(NTHREADS) void conv_bwd_logits_mfma_kernel(CoreParams p, LogitParams lp) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    const Smem s = carve(smem_raw, ZSTRIDE);
    float* dst = s.extra;                       // ds tile [TILE][ostride]
    float* red = dst + TILE * lp.ostride;       // [4][12]
    const int tile0 = xcd_tile(blockIdx.x, gridDim.x) * TILE;
    const int tid = threadIdx.x;
    const int wave = tid >> 6, lane = tid & 63, lr = lane & 15, lq = lane >> 4;

    for (int t = tid; t < TILE * lp.opad; t += NTHREADS) {
        const int r = t / lp.opad, o = t % lp.opad;
        const int i = tile0 + r;
        dst[r * lp.ostride + o] = (i < p.n && o < lp.cout) ? lp.ds[(size_t)i * lp.cout + o] : 0.f;
    }
    softmax_phase<false>(p, s, tile0, 0, nullptr, nullptr);
    __syncthreads();

    const int nct = KPASS >> 4;   // 18
    const int okg = lp.opad >> 4;
    const f32x4* Wq4 = reinterpret_cast<const f32x4*>(lp.Wq);

    f32x4 dq[NPW][2];               // [node of this wave][edge tile 0..15 / 16..31]
#pragma unroll
    for (int nn = 0; nn < NPW; ++nn) {
        dq[nn][0] = f32x4{0.f, 0.f, 0.f, 0.f};
        dq[nn][1] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    const int mrow = lr < FGC_M ? lr : FGC_M - 1;  // rows 9..15 of the product are never read

    for (int pass = 0; pass < p.passes; ++pass) {
        // ---- dz tile = ds tile x Wq[pass] (f32 MFMA) -> LDS
        {
            f32x4 acc[RT][K1_CTW];
#pragma unroll
            for (int r = 0; r < RT; ++r)
#pragma unroll
                for (int c = 0; c < K1_CTW; ++c) acc[r][c] = f32x4{0.f, 0.f, 0.f, 0.f};
            for (int g = 0; g < okg; ++g) {
                f32x4 a[RT];
#pragma unroll
                for (int r = 0; r < RT; ++r)
                    a[r] = *reinterpret_cast<const f32x4*>(dst + (r * 16 + lr) * lp.ostride + g * 16 + lq * 4);
#pragma unroll
                for (int c = 0; c < K1_CTW; ++c) {
                    const int ct = min(wave + c * 4, nct - 1);
                    const f32x4 b = Wq4[((size_t)pass * (lp.opad >> 2) + g * 4 + lq) * KPASS + ct * 16 + lr];
#pragma unroll
                    for (int t = 0; t < 4; ++t)
#pragma unroll
                        for (int r = 0; r < RT; ++r)
                            acc[r][c] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[r][t], b[t], acc[r][c], 0, 0, 0);
                }
            }
            if (pass > 0) __syncthreads();  // the previous pass' readers of ztile are done
#pragma unroll
            for (int c = 0; c < K1_CTW; ++c) {
                const int ct = wave + c * 4;
                if (ct >= nct) continue;
#pragma unroll
                for (int r = 0; r < RT; ++r)
#pragma unroll
                    for (int t = 0; t < 4; ++t)
                        s.ztile[(size_t)(r * 16 + lq * 4 + t) * ZSTRIDE + ct * 16 + lr] = acc[r][c][t];
            }
        }
        __syncthreads();
        // ---- per node: dq += dz_i (9 x 32) . X_i (32 x d)
        const int cpass = pass * KC;
        auto rowof = [&](int node, int d, int et) {
            const int e = min(et * 16 + lr, d - 1);
            return __float_as_int(s.qbuf[((size_t)node * KMAX + e) * QLD + 9]);
        };
        f32x4 bc[2], bn[2];
        {
            const int node = wave * NPW;
            const int d = __builtin_amdgcn_readfirstlane(min(s.deg[node], KMAX));
            const int row = d > 0 ? rowof(node, d, 0) : 0;
            bc[0] = load_chunk<VEC4>(p, row, cpass + 4 * lq);
            bc[1] = load_chunk<VEC4>(p, row, cpass + 16 + 4 * lq);
        }
#pragma unroll
        for (int nn = 0; nn < NPW; ++nn) {
            const int node = wave * NPW + nn;
            const int d = __builtin_amdgcn_readfirstlane(min(s.deg[node], KMAX));
            if (nn + 1 < NPW) {  // next node's rows are requested before this node's MFMAs
                const int nd = __builtin_amdgcn_readfirstlane(min(s.deg[node + 1], KMAX));
                const int row = nd > 0 ? rowof(node + 1, nd, 0) : 0;
                bn[0] = load_chunk<VEC4>(p, row, cpass + 4 * lq);
                bn[1] = load_chunk<VEC4>(p, row, cpass + 16 + 4 * lq);
            }
            const float* zr = s.ztile + (size_t)node * ZSTRIDE + mrow * KC + 4 * lq;
            const f32x4 a0 = *reinterpret_cast<const f32x4*>(zr);
            const f32x4 a1 = *reinterpret_cast<const f32x4*>(zr + 16);
            if (d > 0) {
                f32x4 t0 = f32x4{0.f, 0.f, 0.f, 0.f}, t1 = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int t = 0; t < 4; ++t) {  // two independent accumulation chains
                    t0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a0[t], bc[0][t], t0, 0, 0, 0);
                    t1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a1[t], bc[1][t], t1, 0, 0, 0);
                }
                dq[nn][0] += t0 + t1;
                if (d > 16) {  // rare: 17..24 neighbours
                    const int row = rowof(node, d, 1);
                    const f32x4 x0 = load_chunk<VEC4>(p, row, cpass + 4 * lq);
                    const f32x4 x1 = load_chunk<VEC4>(p, row, cpass + 16 + 4 * lq);
                    f32x4 u0 = f32x4{0.f, 0.f, 0.f, 0.f}, u1 = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                    for (int t = 0; t < 4; ++t) {
                        u0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a0[t], x0[t], u0, 0, 0, 0);
                        u1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a1[t], x1[t], u1, 0, 0, 0);
                    }
                    dq[nn][1] += u0 + u1;
                }
            }
            bc[0] = bn[0];
            bc[1] = bn[1];
        }
    }

    // ---- softmax backward: lane (edge = lr, m0 = 4*lq) holds dq[m0..m0+3][edge]
    f32x4 dcacc = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int nn = 0; nn < NPW; ++nn) {
        const int node = wave * NPW + nn;
        const int i = tile0 + node;
        const int d = __builtin_amdgcn_readfirstlane(min(s.deg[node], KMAX));
        f32x4 da = f32x4{0.f, 0.f, 0.f, 0.f};
        if (d > 0 && i < p.n) {
            const int e0 = p.rowptr[i];
            const int ntile = d > 16 ? 2 : 1;
            for (int et = 0; et < ntile; ++et) {
                const int edge = et * 16 + lr;
                const bool ok = edge < d;
                const float* qr = s.qbuf + ((size_t)node * KMAX + min(edge, d - 1)) * QLD;
                f32x4 q = f32x4{0.f, 0.f, 0.f, 0.f};
                if (lq < 2) q = *reinterpret_cast<const f32x4*>(qr + 4 * lq);
                else if (lq == 2) q[0] = qr[8];
                f32x4 g = et == 0 ? dq[nn][0] : dq[nn][1];
                if (lq == 2) { g[1] = 0.f; g[2] = 0.f; g[3] = 0.f; }
                if (lq == 3) g = f32x4{0.f, 0.f, 0.f, 0.f};
                float dot = q[0] * g[0] + q[1] * g[1] + q[2] * g[2] + q[3] * g[3];
                dot += __shfl_xor(dot, 16);
                dot += __shfl_xor(dot, 32);
                f32x4 dl;
#pragma unroll
                for (int t = 0; t < 4; ++t) dl[t] = ok ? q[t] * (g[t] - dot) : 0.f;
                if (ok && lq < 3) *reinterpret_cast<f32x4*>(lp.dl + (size_t)(e0 + edge) * FGC_DL_LD + 4 * lq) = dl;
                da += dl;
            }
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                float v = da[t];
                v += __shfl_xor(v, 1);
                v += __shfl_xor(v, 2);
                v += __shfl_xor(v, 4);
                v += __shfl_xor(v, 8);
                da[t] = v;
            }
        }
        if (i < p.n && lr == 0 && lq < 3) *reinterpret_cast<f32x4*>(lp.dag + (size_t)i * FGC_AG_LD + 4 * lq) = da;
        dcacc += da;
    }
    // dc partial of this workgroup: lanes lr == 0 hold the per-wave sums (m0 = 4*lq)
    __syncthreads();
    if (lr == 0 && lq < 3) {
#pragma unroll
        for (int t = 0; t < 4; ++t) red[wave * 12 + 4 * lq + t] = dcacc[t];
    }
    __syncthreads();
    if (tid < 12) {
        const float v = tid < FGC_M ? (red[tid] + red[12 + tid]) + (red[24 + tid] + red[36 + tid]) : 0.f;
        lp.dc_part[(size_t)blockIdx.x * 12 + tid] = v;
    }
}

// ---------------------------------------------------------------------------------------------
// K1, matrix-core form with deep gathers (every 32-channel pass comes from ONE source: cg % 32 == 0 and the concat
// boundary on a pass boundary; degree <= 16 handled on the fast path, 17..24 by a second edge tile).
// Same math and summation order as conv_bwd_logits_mfma_kernel.  What changes is when memory is asked for: the
// neighbour rows of ALL 8 nodes of a wave (the B fragments of the per-edge products) are requested with buffer
// loads at the top of the pass, so their latency runs under the dz GEMM instead of once per node, and the packed-
// weight fragments of the dz GEMM are requested one k-group ahead.  LDS limits residency to 2 waves per SIMD, so
// the 256-VGPR budget is there to be used.
// ---------------------------------------------------------------------------------------------
// LONG (some node of the graph has 17..24 edges: irregular meshes, where nearly every tile holds such a node): one sweep
// that also carries the accumulators of the second edge tile (slots 16..23), paid for with the second gather register
// set: the neighbour rows of nodes 4..7 are requested after the products of nodes 0..3, those of the second edge tile
// on demand.  Without it such tiles took the two-sweep path below, i.e. the dz GEMM twice.
// OKG = cout / 16 for cout = 32 and 64 (strides of the ds tile become compile-time, its load one coalesced dwordx4
// stream issued up front), 0 = any cout (measured faster than OKG = 8 for the 128-wide layers of the coarsest level)
// NT_ nodes per workgroup, NPW_ nodes per wave.  (32, 8): the form described above, four waves, 68 KB of LDS, 188 registers:
// two workgroups = two waves per SIMD per CU.  (16, 4), regular graphs only: four waves on HALF a tile - half the per-wave
// state (gathers of 4 nodes instead of 8), 34 KB of LDS: four workgroups per CU if the registers stay under 128.
// SPLIT (half tiles, OKG = 2): the dz GEMM on the bf16 matrix pipe with three-term operand splits (fgc_split.h) - the s
// tile is the A operand of 18 column tiles per pass, so splitting it costs 44 vector instructions per wave and k-step and
// replaces 8 v_mfma_f32_16x16x4_f32 (32 cycles each) per column tile by 6 v_mfma_f32_16x16x32_bf16 (16 cycles each); the
// weights come as three planes of B fragments (pack_logit_weight_split_body).  Same sums as the fp32 form up to the order
// of the additions.  The per-edge products stay on the fp32 MFMA: both of their operands are used once.  Measured (round 5,
// same-box alternating runs): dconv1 at 100k facets 122.9 -> 116.6 us.  The 64-wide layers were tried and gained nothing
// (77.5 -> 76.0 / 77.6 us): a half tile re-reads the whole packed operand from L2 - 446 MB per level-0 launch as fp32, half
// as much again as three bf16 planes - and that stream, not the matrix pipe, is what those launches wait for.
// QSR: edge slots per node of the soft-assignment table for the regular (!LONG) forms - 16, or 14 where the host knows that no
// node has more edges (a closed triangle mesh's facet graph: 13): 1.5 KB less LDS per half tile, 32.6 KB - a FIFTH workgroup
// per CU if the registers stay under 97 (the launch bound asks for five waves per SIMD then).  Round-5 review, candidate
// "14-slot d-logits table"; measured in DESIGN.md section 10.
template <bool LONG, int OKG, int NT_ = 32, int NPW_ = 8, bool SPLIT = false, int QSR = 16>
__global__ __launch_bounds__((NT_ / NPW_) * 64, NT_ == 32 ? 2 : (QSR == 14 ? 5 : 4)) void conv_bwd_logits_deep_kernel(CoreParams p, LogitParams lp) {
    static_assert(QSR == 16 || (QSR == 14 && NT_ == 16), "14 slots: the half-tile form");
    static_assert((NT_ == 32 && NPW_ == 8) || (NT_ == 16 && NPW_ == 4 && !LONG), "tile shapes");
    static_assert(!SPLIT || (NT_ == 16 && OKG == 2), "split dz GEMM: half tiles of the 32-wide layers");
    // (shadow the 32-node constants of the file)
    constexpr int TILE = NT_, NPW = NPW_, NWV = NT_ / NPW_, NTHREADS = NWV * 64, RT = NT_ / 16, LPN = NTHREADS / NT_;
    static_assert(NWV == 4, "four waves either way: the column-tile split and the dc sums below assume it");
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    constexpr int QS = LONG ? KMAX : QSR;       // the host sends graphs with a degree above 16 to the LONG form
    const Smem s = carve(smem_raw, ZSTRIDE, QS, NT_);
    const int opad = OKG ? OKG * 16 : lp.opad;
    const int ostride = OKG ? OKG * 16 + 8 : lp.ostride;
    float* dst = s.extra;                       // ds tile [TILE][ostride]
    float* red = dst + ((LONG && lp.a_global) ? 0 : TILE * ostride);          // [4][12]
    const int tile0 = xcd_tile(blockIdx.x, gridDim.x) * TILE;
    const int tid = threadIdx.x;
    const int wave = tid >> 6, lane = tid & 63, lr = lane & 15, lq = lane >> 4;

    const bool a_global = LONG && lp.a_global;      // block-uniform
    if (a_global) {
        // no LDS copy of the ds tile
    } else if constexpr (OKG > 0) {
        // rows tile0 .. tile0+31 of ds are one contiguous run of 32 * cout floats (cout == opad)
        constexpr int V4 = TILE * OKG * 16 / 4;          // float4s in the tile: 128 * OKG (half tiles: 64 * OKG)
        constexpr int PER = (V4 + NTHREADS - 1) / NTHREADS;
        static_assert(V4 % NTHREADS == 0 || V4 < NTHREADS, "whole float4s per thread, or fewer float4s than threads");
        const f32x4* src = reinterpret_cast<const f32x4*>(lp.ds + (size_t)tile0 * (OKG * 16));
        const int vmax = (min(p.n - tile0, TILE) * OKG * 16) / 4 - 1;     // last valid float4 (n > tile0)
        f32x4 v[PER];
        if (lp.dy) {
            // s = dy * lrelu'(y) / deg computed here instead of by a launch of its own: the tile goes to LDS and to ds
            // (the data kernel gathers it), the bias-gradient partial of the tile's 32 rows to db_part
            const f32x4* dy4 = reinterpret_cast<const f32x4*>(lp.dy + (size_t)tile0 * (OKG * 16));
            const f32x4* y4 = reinterpret_cast<const f32x4*>(lp.y + (size_t)tile0 * (OKG * 16));
            f32x4* out4 = reinterpret_cast<f32x4*>(lp.ds_out + (size_t)tile0 * (OKG * 16));
            f32x4 gy[PER], yy[PER];
            int dg[PER];
#pragma unroll
            for (int k = 0; k < PER; ++k) {
                const int t = min(tid + k * NTHREADS, vmax);
                const int i = tile0 + t / (OKG * 4);
                gy[k] = dy4[t];
                if (lp.act || lp.pool_dy) yy[k] = y4[t];
                dg[k] = p.rowptr[i + 1] - p.rowptr[i];
            }
            if (lp.pool_dy) {
                // the gradient of the 4:1 max pooling of this layer's output, folded in here instead of a pass of its own
                // over dy: the four rows of a pooling group sit in the same tile (tile0 is a multiple of 32)
#pragma unroll
                for (int k = 0; k < PER; ++k) {
                    const int t = min(tid + k * NTHREADS, vmax);
                    const int rl = t / (OKG * 4), c4 = t % (OKG * 4);
                    const size_t pi = (size_t)((tile0 + rl) >> 2) * (OKG * 4) + c4;
                    const f32x4 m = reinterpret_cast<const f32x4*>(lp.pool_y)[pi];
                    const f32x4 gp = reinterpret_cast<const f32x4*>(lp.pool_dy)[pi];
                    f32x4 ne = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const int tq = min(((rl & ~3) + q) * (OKG * 4) + c4, vmax);
                        const f32x4 yq = y4[tq];
#pragma unroll
                        for (int c = 0; c < 4; ++c) ne[c] += yq[c] == m[c] ? 1.f : 0.f;
                    }
#pragma unroll
                    for (int c = 0; c < 4; ++c) gy[k][c] += yy[k][c] == m[c] ? gp[c] / ne[c] : 0.f;
                }
            }
#pragma unroll
            for (int k = 0; k < PER; ++k) {
                const int t = tid + k * NTHREADS;
                const bool ok = t <= vmax;
                f32x4 g = gy[k];
                if (lp.act) {
#pragma unroll
                    for (int c = 0; c < 4; ++c) g[c] *= slope_from_y(yy[k][c], lp.alpha);
                }
                f32x4 sv;
#pragma unroll
                for (int c = 0; c < 4; ++c) sv[c] = (ok && dg[k] > 0) ? g[c] / (float)dg[k] : 0.f;
                v[k] = sv;
                if (ok) out4[t] = sv;
                const bool counts = ok && (!lp.bias_mask || dg[k] > 0);
                const int r = t / (OKG * 4), o4 = t % (OKG * 4);
                if (V4 % NTHREADS == 0 || t < V4)
                    *reinterpret_cast<f32x4*>(dst + r * ostride + o4 * 4) = counts ? g : f32x4{0.f, 0.f, 0.f, 0.f};
            }
            __syncthreads();
            {
                // column sums of the tile with ALL threads: P = 256 / columns adjacent lanes share a column (rows part,
                // part + P, ...) and add up on the DPP crossbar in a fixed order.  (One thread per column walking the 32
                // rows was a chain of 32 dependent LDS reads in front of a barrier, once per tile.)
                constexpr int C = OKG * 16, P = NTHREADS / C;
                static_assert(P == 8 || P == 4 || P == 2, "one, two or three DPP steps");
                const int col = tid / P, part = tid % P;
                float acc = 0.f;
#pragma unroll
                for (int j = 0; j < TILE / P; ++j) acc += dst[(part + P * j) * ostride + col];
                acc += fgc_dpp_c<0xB1>(acc);
                if (P >= 4) acc += fgc_dpp_c<0x4E>(acc);
                if (P == 8) acc += fgc_dpp_c<0x141>(acc);
                if (part == 0) lp.db_part[(size_t)(tile0 / TILE) * C + col] = acc;
            }
            __syncthreads();
        } else {
#pragma unroll
            for (int k = 0; k < PER; ++k) v[k] = src[min(tid + k * NTHREADS, vmax)];
        }
#pragma unroll
        for (int k = 0; k < PER; ++k) {
            const int t = tid + k * NTHREADS;
            if (V4 % NTHREADS == 0 || t < V4) {
                const int r = t / (OKG * 4), o4 = t % (OKG * 4);
                *reinterpret_cast<f32x4*>(dst + r * ostride + o4 * 4) = t <= vmax ? v[k] : f32x4{0.f, 0.f, 0.f, 0.f};
            }
        }
    } else {
        for (int t = tid; t < TILE * opad; t += NTHREADS) {
            const int r = t / opad, o = t % opad;
            const int i = tile0 + r;
            dst[r * ostride + o] = (i < p.n && o < lp.cout) ? lp.ds[(size_t)i * lp.cout + o] : 0.f;
        }
    }
    const int dmine = softmax_phase<false, QS, NT_, LPN>(p, s, tile0, 0, nullptr, nullptr);
    // edges 0..15 of every node in sweep 0; a second sweep (block-uniform, rare) for nodes with 17..24 edges.  The
    // per-edge work is independent across edges, so a sweep is the whole computation for its 16 edge slots.
    (void)dmine;
    __syncthreads();
    constexpr int nsweeps = 1;   // 16 edge slots per sweep: the non-LONG form only sees degrees <= 16, LONG carries 17..24 along

    const int nct = KPASS >> 4;   // 18
    const int okg = opad >> 4;
    const __amdgpu_buffer_rsrc_t wq_rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(lp.Wq), 0, -1, 0x00020000);
    int dn[NPW];
#pragma unroll
    for (int nn = 0; nn < NPW; ++nn) dn[nn] = __builtin_amdgcn_readfirstlane(min(s.deg[wave * NPW + nn], KMAX));
    const int mrow = lr < FGC_M ? lr : FGC_M - 1;  // rows 9..15 of the product are never read
    f32x4 dcacc = f32x4{0.f, 0.f, 0.f, 0.f};

    for (int sweep = 0; sweep < nsweeps; ++sweep) {
        const int ebase = sweep * 16;
        // row id of this lane's edge slot (clamped into the node's list) for each node of the wave
        int rowid[NPW];
        f32x4 dq[NPW];
        f32x4 dq_hi[LONG ? NPW : 1];    // LONG: edge slots 16..23
#pragma unroll
        for (int nn = 0; nn < (LONG ? NPW : 1); ++nn) dq_hi[nn] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int nn = 0; nn < NPW; ++nn) {
            const int node = wave * NPW + nn;
            const int e = max(min(ebase + lr, dn[nn] - 1), 0);
            rowid[nn] = dn[nn] > 0 ? __float_as_int(s.qbuf[(size_t)node * qnode_stride(QS) + e * QLD + 9]) : 0;
            dq[nn] = f32x4{0.f, 0.f, 0.f, 0.f};
        }
        for (int pass = 0; pass < p.passes; ++pass) {
            const int cpass = pass * KC;
            const bool first = cpass < p.c0;                                            // block-uniform
            const float* base = first ? p.src0 : p.src1;
            const unsigned rowbytes = (unsigned)(first ? p.c0 : p.c1) * 4u;
            const unsigned laneoff = (unsigned)((first ? cpass : cpass - p.c0) + 4 * lq) * 4u;
            const __amdgpu_buffer_rsrc_t rsrc =
                __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(base), 0, -1, 0x00020000);
            // ---- B fragments (neighbour rows) of the wave's nodes, in two halves of 4 nodes = 8 x dwordx4 per lane:
            // the first half is requested here and lands under the dz GEMM, the second half after the GEMM and
            // lands under the barrier, the dz store and the first half's MFMAs
            constexpr int H = NPW / 2;
            f32x4 bxa[H][2], bxb[H][2];
            auto gather = [&](int n0, f32x4 (&bx)[H][2]) {
#pragma unroll
                for (int nn = 0; nn < H; ++nn) {
                    const unsigned off = __umul24((unsigned)rowid[n0 + nn], rowbytes) + laneoff;
                    if (FGC_KO1 & 16) { bx[nn][0] = bx[nn][1] = f32x4{__uint_as_float(off), 1.f, 2.f, 3.f}; continue; }
                    bx[nn][0] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc, off, 0, 0));
                    bx[nn][1] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc, off + 64u, 0, 0));
                }
            };
            gather(0, bxa);
            // ---- dz tile = ds tile x Wq[pass] (f32 MFMA) -> LDS
            {
                f32x4 acc[RT][K1_CTW];
#pragma unroll
                for (int r = 0; r < RT; ++r)
#pragma unroll
                    for (int c = 0; c < K1_CTW; ++c) acc[r][c] = f32x4{0.f, 0.f, 0.f, 0.f};
                if constexpr (SPLIT) {
                    // one chunk per column tile (K = cout = 32 is one k-step); the three planes of chunk c + 2 are requested
                    // before the six MFMAs of chunk c (a ring of three 12-register slots)
                    u32x4 wring[3][3], a3[3];
                    auto loadq = [&](int c, u32x4 (&b)[3]) {
                        const int ct = min(wave + c * 4, nct - 1);
                        const unsigned soff = (unsigned)(((pass * 18 + ct) * 3) * 1024);
#pragma unroll
                        for (int pl = 0; pl < 3; ++pl)
                            b[pl] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(wq_rs, (unsigned)(lane * 16 + pl * 1024), soff, 0));
                    };
                    loadq(0, wring[0]);
                    loadq(1, wring[1]);
                    {
                        const float* ar = dst + lr * ostride + 8 * lq;
                        split3_frag(*reinterpret_cast<const f32x4*>(ar), *reinterpret_cast<const f32x4*>(ar + 4), a3);
                    }
#pragma unroll
                    for (int c = 0; c < K1_CTW; ++c) {
                        if (c + 2 < K1_CTW) loadq(c + 2, wring[(c + 2) % 3]);
                        acc[0][c] = mfma_split(a3, wring[c % 3], acc[0][c]);
                    }
                } else {
                // (buffer loads: the lane's part of the offset is a loop invariant, the k-group's part scalar - the indexed form
                // spent a 64-bit multiply-add chain per fragment on the vector ALU, which the fp32 MFMA shares)
                auto loadw = [&](int g, f32x4 (&b)[K1_CTW]) {
                    const int gg = min(g, okg - 1);
                    const unsigned soff = (unsigned)((pass * (opad >> 2) + gg * 4) * KPASS * 16);
#pragma unroll
                    for (int c = 0; c < K1_CTW; ++c) {
                        const int ct = min(wave + c * 4, nct - 1);
                        if (FGC_KO1 & 2) { b[c] = f32x4{(float)ct, (float)soff, 1.f, 2.f}; continue; }
                        b[c] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(
                                                             wq_rs, (unsigned)((lq * KPASS + ct * 16 + lr) * 16), soff, 0));
                    }
                };
                auto mmw = [&](int g, const f32x4 (&b)[K1_CTW]) {
                    f32x4 a[RT];
#pragma unroll
                    for (int r = 0; r < RT; ++r) {
                        if (a_global)
                            a[r] = *reinterpret_cast<const f32x4*>(lp.ds + (size_t)min(tile0 + r * 16 + lr, p.n - 1) * lp.cout +
                                                                   g * 16 + lq * 4);
                        else
                            a[r] = *reinterpret_cast<const f32x4*>(dst + (r * 16 + lr) * ostride + g * 16 + lq * 4);
                    }
#pragma unroll
                    for (int c = 0; c < K1_CTW; ++c) {
                        // 18 column tiles over 4 waves = 5, 5, 4, 4: waves 2 and 3 skip their (duplicate) fifth tile.  Only
                        // where it was measured to pay (64-wide layers: -5 %); the branch costs the other forms 10 - 90 %
                        if (OKG == 4 && wave + c * 4 >= nct) continue;
#pragma unroll
                        for (int t = 0; t < 4; ++t)
#pragma unroll
                            for (int r = 0; r < RT; ++r) {
                                if (FGC_KO1 & 1) asm volatile("" ::"v"(a[r][t]), "v"(b[c][t]));
                                else acc[r][c] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[r][t], b[c][t], acc[r][c], 0, 0, 0);
                            }
                    }
                };
                if constexpr (OKG == 4) {      // (OKG = 8, the 128-wide layers: the plain loop was measured faster)
                    // two weight buffers with fixed roles: the fragments of group g + 1 are in flight under the MFMAs of g
                    f32x4 wA[K1_CTW], wB[K1_CTW];
                    loadw(0, wA);
#pragma unroll 1
                    for (int g = 0; g < OKG; g += 2) {
                        loadw(g + 1, wB);
                        mmw(g, wA);
                        loadw(g + 2, wA);       // (clamped to the last group at the end)
                        mmw(g + 1, wB);
                    }
                } else {
#pragma unroll OKG == 2 ? 2 : 1
                    for (int g = 0; g < okg; ++g) {
                        f32x4 w0[K1_CTW];
                        loadw(g, w0);
                        mmw(g, w0);
                    }
                }
                }
                if (!LONG) gather(H, bxb);
                if (pass > 0 || sweep > 0) __syncthreads();  // the previous readers of ztile are done
#pragma unroll
                for (int c = 0; c < K1_CTW; ++c) {
                    const int ct = wave + c * 4;
                    if (ct >= nct) continue;
                    if (FGC_KO1 & 4) { if (acc[0][c][0] == 123.f) s.ztile[tid] = acc[1][c][3]; continue; }
#pragma unroll
                    for (int r = 0; r < RT; ++r)
#pragma unroll
                        for (int t = 0; t < 4; ++t)
                            s.ztile[(size_t)(r * 16 + lq * 4 + t) * ZSTRIDE + ct * 16 + lr] = acc[r][c][t];
                }
            }
            __syncthreads();
            // ---- per node: dq += dz_i (9 x 32) . X_i (32 x 16 edge slots)
            auto products = [&](int n0, const f32x4 (&bx)[H][2]) {
#pragma unroll
                for (int nn = 0; nn < H; ++nn) {
                    const int node = wave * NPW + n0 + nn;
                    const float* zr = s.ztile + (size_t)node * ZSTRIDE + mrow * KC + 4 * lq;
                    const f32x4 a0 = *reinterpret_cast<const f32x4*>(zr);
                    const f32x4 a1 = *reinterpret_cast<const f32x4*>(zr + 16);
                    f32x4 t0 = f32x4{0.f, 0.f, 0.f, 0.f}, t1 = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                    for (int t = 0; t < 4; ++t) {  // two independent accumulation chains
                        if (FGC_KO1 & 8) { t0[t] += a0[t] * bx[nn][0][t]; t1[t] += a1[t] * bx[nn][1][t]; continue; }
                        t0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a0[t], bx[nn][0][t], t0, 0, 0, 0);
                        t1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a1[t], bx[nn][1][t], t1, 0, 0, 0);
                    }
                    dq[n0 + nn] += t0 + t1;
                }
            };
            products(0, bxa);
            if (!LONG) {
                products(H, bxb);
            } else {
                gather(H, bxa);
                products(H, bxa);
#pragma unroll
                for (int nn = 0; nn < NPW; ++nn) {
                    if (dn[nn] <= 16) continue;            // wave-uniform
                    const int node = wave * NPW + nn;
                    const int e = min(16 + lr, dn[nn] - 1);
                    const int row = __float_as_int(s.qbuf[(size_t)node * qnode_stride(QS) + e * QLD + 9]);
                    const unsigned off = __umul24((unsigned)row, rowbytes) + laneoff;
                    const f32x4 x0 = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc, off, 0, 0));
                    const f32x4 x1 = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc, off + 64u, 0, 0));
                    const float* zr = s.ztile + (size_t)node * ZSTRIDE + mrow * KC + 4 * lq;
                    const f32x4 a0 = *reinterpret_cast<const f32x4*>(zr);
                    const f32x4 a1 = *reinterpret_cast<const f32x4*>(zr + 16);
                    f32x4 u0 = f32x4{0.f, 0.f, 0.f, 0.f}, u1 = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                    for (int t = 0; t < 4; ++t) {
                        u0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a0[t], x0[t], u0, 0, 0, 0);
                        u1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a1[t], x1[t], u1, 0, 0, 0);
                    }
                    dq_hi[nn] += u0 + u1;
                }
            }
        }

        // ---- softmax backward of this sweep's edge slots: lane (edge = ebase + lr, m0 = 4*lq) holds dq[m0..m0+3]
#pragma unroll
        for (int nn = 0; nn < NPW; ++nn) {
            const int node = wave * NPW + nn;
            const int i = tile0 + node;
            const int d = dn[nn];
            if (i >= p.n) continue;                    // wave-uniform
            if (d <= ebase) {                          // wave-uniform; an isolated node still owns a dag row
                if (sweep == 0 && lr == 0 && lq < 3)
                    *reinterpret_cast<f32x4*>(lp.dag + (size_t)i * FGC_AG_LD + 4 * lq) = f32x4{0.f, 0.f, 0.f, 0.f};
                continue;
            }
            const int e0 = s.deg[TILE + 4 + node];     // first edge id, left in LDS by the softmax phase
            f32x4 da = f32x4{0.f, 0.f, 0.f, 0.f};
            const int ntile = (FGC_KO1 & 32) ? 0 : ((LONG && d > 16) ? 2 : 1);
            for (int et = 0; et < ntile; ++et) {
                const int edge = ebase + 16 * et + lr;
                const bool ok = edge < d;
                const float* qr = s.qbuf + (size_t)node * qnode_stride(QS) + min(edge, d - 1) * QLD;
                f32x4 q = f32x4{0.f, 0.f, 0.f, 0.f};
                if (lq < 2) q = *reinterpret_cast<const f32x4*>(qr + 4 * lq);
                else if (lq == 2) q[0] = qr[8];
                f32x4 g = (LONG && et == 1) ? dq_hi[LONG ? nn : 0] : dq[nn];
                if (lq == 2) { g[1] = 0.f; g[2] = 0.f; g[3] = 0.f; }
                if (lq == 3) g = f32x4{0.f, 0.f, 0.f, 0.f};
                float dot = q[0] * g[0] + q[1] * g[1] + q[2] * g[2] + q[3] * g[3];
                dot += __shfl_xor(dot, 16);
                dot += __shfl_xor(dot, 32);
                f32x4 dl;
#pragma unroll
                for (int t = 0; t < 4; ++t) dl[t] = ok ? q[t] * (g[t] - dot) : 0.f;
                if (!(FGC_KO1 & 64) && ok && lq < 3) *reinterpret_cast<f32x4*>(lp.dl + (size_t)(e0 + edge) * FGC_DL_LD + 4 * lq) = dl;
                da += dl;
            }
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                float v = da[t];
                FGC_ROW16_SUM(v);
                da[t] = v;
            }
            dcacc += da;                               // dc = sum over nodes and edges of dl
            if (lr == 0 && lq < 3) {
                float* o = lp.dag + (size_t)i * FGC_AG_LD + 4 * lq;
                // second sweep: add to what this same thread stored in the first
                if (sweep > 0) da += *reinterpret_cast<const f32x4*>(o);
                *reinterpret_cast<f32x4*>(o) = da;
            }
        }
    }
    // dc partial of this workgroup: lanes lr == 0 hold the per-wave sums (m0 = 4*lq)
    __syncthreads();
    if (lr == 0 && lq < 3) {
#pragma unroll
        for (int t = 0; t < 4; ++t) red[wave * 12 + 4 * lq + t] = dcacc[t];
    }
    __syncthreads();
    if (tid < 12) {
        const float v = tid < FGC_M ? (red[tid] + red[12 + tid]) + (red[24 + tid] + red[36 + tid]) : 0.f;
        lp.dc_part[(size_t)blockIdx.x * 12 + tid] = v;
    }
}

// ---------------------------------------------------------------------------------------------
// K1, bf16 storage (FGC_CONV_BF16).  Same decomposition and summation structure as conv_bwd_logits_deep_kernel; the two
// matrix products run on v_mfma_f32_16x16x32_bf16:
//   dz tile [32 x 288] = s tile [32 x cout] (bf16, LDS) x Wq (bf16, packed fragments)     K = cout, 32 per k-step
//   dq_i [9 x 16 edges] = dz_i [9 x 32] (bf16, LDS) x X_i [32 x 16 edges]                 K = the pass' 32 channels:
//       one MFMA per node and pass; its B fragment is ONE 16-byte load per lane from the neighbour's bf16 row
// s = dy * lrelu'(y) / deg comes from ds_db_kernel (bf16), the soft assignment and its backward stay fp32.
// ---------------------------------------------------------------------------------------------
// NT_ / NPW_: nodes per workgroup / per wave, (32, 8) or - regular graphs - the half tile (16, 4) of
// conv_bwd_logits_deep_kernel: half the per-wave state and half the LDS, more workgroups resident per CU.
template <bool LONG, int NT_ = 32, int NPW_ = 8>
__global__ __launch_bounds__(256, NT_ == 32 ? 2 : 4) void conv_bwd_logits_bf16_kernel(CoreParams p, LogitParams lp) {
    static_assert((NT_ == 32 && NPW_ == 8) || (NT_ == 16 && NPW_ == 4 && !LONG), "tile shapes");
    constexpr int TILE = NT_, NPW = NPW_, NTHREADS = 256, RT = NT_ / 16, LPN = NTHREADS / NT_;   // (shadow the 32-node constants)
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    constexpr int QS = LONG ? KMAX : 16;
    const Smem s = carve(smem_raw, ZSTRIDE_BF / 2, QS, NT_);
    const int cout = lp.cout;                       // a multiple of 32
    const int obytes = cout * 2 + 32;               // LDS row stride of the s tile (== 32 mod 64)
    char* dst = reinterpret_cast<char*>(s.extra);   // s tile [TILE][obytes]
    float* red = reinterpret_cast<float*>(dst + TILE * obytes);   // [4][12]
    const int tile0 = xcd_tile(blockIdx.x, gridDim.x) * TILE;
    const int tid = threadIdx.x;
    const int wave = tid >> 6, lane = tid & 63, lr = lane & 15, lq = lane >> 4;
    {
        // rows tile0 .. tile0+31 of s are one contiguous run of 32 * cout bf16
        const int cpr = cout >> 3;                   // 16-byte chunks per row
        const int vmax = min(p.n - tile0, TILE) * cpr - 1;
        if (lp.dy) {
            // s = dy * lrelu'(y) / deg computed here instead of by ds_db_kernel (same operations in the same order on the
            // same bf16 inputs: the same s, bit for bit): the tile goes to LDS and to ds (the data kernel gathers it), the
            // fp32 values that count for the bias gradient to a staging tile in the (still unused) dz tile for the column sums
            const u32x4* dy8 = reinterpret_cast<const u32x4*>(reinterpret_cast<const unsigned short*>(lp.dy) + (size_t)tile0 * cout);
            const u32x4* y8 = reinterpret_cast<const u32x4*>(reinterpret_cast<const unsigned short*>(lp.y) + (size_t)tile0 * cout);
            u32x4* out8 = reinterpret_cast<u32x4*>(reinterpret_cast<unsigned short*>(lp.ds_out) + (size_t)tile0 * cout);
            float* gst = s.ztile;                    // [TILE][cout + 8] fp32
            const int gs = cout + 8;
            for (int t = tid; t < TILE * cpr; t += NTHREADS) {
                const int tt = min(t, vmax);
                const int r = tt / cpr, c8 = tt % cpr;
                const int i = tile0 + r;
                const u32x4 gy = dy8[tt];
                u32x4 yy = u32x4{0u, 0u, 0u, 0u};
                if (lp.act || lp.pool_dy) yy = y8[tt];
                const int dg = p.rowptr[i + 1] - p.rowptr[i];
                f32x4 g0 = bf4_to_f4(u32x2{gy[0], gy[1]}), g1 = bf4_to_f4(u32x2{gy[2], gy[3]});
                const f32x4 y0 = bf4_to_f4(u32x2{yy[0], yy[1]}), y1 = bf4_to_f4(u32x2{yy[2], yy[3]});
                if (lp.pool_dy) {
                    // the gradient of the 4:1 max pooling of this layer's output (the four rows of a pooling group sit in
                    // the same tile: tile0 is a multiple of 16)
                    const size_t pi = (size_t)(i >> 2) * cpr + c8;
                    const u32x4 mm = reinterpret_cast<const u32x4*>(lp.pool_y)[pi];
                    const u32x4 gp = reinterpret_cast<const u32x4*>(lp.pool_dy)[pi];
                    const f32x4 m0 = bf4_to_f4(u32x2{mm[0], mm[1]}), m1 = bf4_to_f4(u32x2{mm[2], mm[3]});
                    const f32x4 p0 = bf4_to_f4(u32x2{gp[0], gp[1]}), p1 = bf4_to_f4(u32x2{gp[2], gp[3]});
                    f32x4 n0 = f32x4{0.f, 0.f, 0.f, 0.f}, n1 = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const u32x4 yq = y8[min(((r & ~3) + q) * cpr + c8, vmax)];
                        const f32x4 q0 = bf4_to_f4(u32x2{yq[0], yq[1]}), q1 = bf4_to_f4(u32x2{yq[2], yq[3]});
#pragma unroll
                        for (int c = 0; c < 4; ++c) {
                            n0[c] += q0[c] == m0[c] ? 1.f : 0.f;
                            n1[c] += q1[c] == m1[c] ? 1.f : 0.f;
                        }
                    }
#pragma unroll
                    for (int c = 0; c < 4; ++c) {
                        g0[c] += y0[c] == m0[c] ? p0[c] / n0[c] : 0.f;
                        g1[c] += y1[c] == m1[c] ? p1[c] / n1[c] : 0.f;
                    }
                }
                if (lp.act) {
#pragma unroll
                    for (int c = 0; c < 4; ++c) {
                        g0[c] *= slope_from_y(y0[c], lp.alpha);
                        g1[c] *= slope_from_y(y1[c], lp.alpha);
                    }
                }
                const bool ok = t <= vmax;
                const bool counts = ok && (!lp.bias_mask || dg > 0);
                const f32x4 z4 = f32x4{0.f, 0.f, 0.f, 0.f};
                const int rr = t / cpr, cc = t % cpr;
                *reinterpret_cast<f32x4*>(gst + rr * gs + cc * 8) = counts ? g0 : z4;
                *reinterpret_cast<f32x4*>(gst + rr * gs + cc * 8 + 4) = counts ? g1 : z4;
                f32x4 s0, s1;
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    s0[c] = (ok && dg > 0) ? g0[c] / (float)dg : 0.f;
                    s1[c] = (ok && dg > 0) ? g1[c] / (float)dg : 0.f;
                }
                const u32x2 b0 = f4_to_bf4(s0), b1 = f4_to_bf4(s1);
                const u32x4 sv = u32x4{b0[0], b0[1], b1[0], b1[1]};
                if (ok) out8[t] = sv;
                *reinterpret_cast<u32x4*>(dst + rr * obytes + cc * 16) = sv;
            }
            __syncthreads();
            // column sums of the staging tile with all threads: P = 256 / cout adjacent lanes share a column and add up on
            // the DPP crossbar in a fixed order
            const int P = NTHREADS / cout;           // 8, 4 or 2
            const int col = tid / P, part = tid % P;
            float acc = 0.f;
            for (int j = part; j < TILE; j += P) acc += gst[j * gs + col];
            acc += fgc_dpp_c<0xB1>(acc);
            if (P >= 4) acc += fgc_dpp_c<0x4E>(acc);
            if (P == 8) acc += fgc_dpp_c<0x141>(acc);
            if (part == 0) lp.db_part[(size_t)(tile0 / TILE) * cout + col] = acc;
            // (the barrier behind the softmax phase orders these reads before the first dz tile store)
        } else {
            const u32x4* src = reinterpret_cast<const u32x4*>(reinterpret_cast<const unsigned short*>(lp.ds) + (size_t)tile0 * cout);
            for (int t = tid; t < TILE * cpr; t += NTHREADS) {
                const int r = t / cpr, c8 = t % cpr;
                const u32x4 v = src[min(t, vmax)];
                *reinterpret_cast<u32x4*>(dst + r * obytes + c8 * 16) = t <= vmax ? v : u32x4{0u, 0u, 0u, 0u};
            }
        }
    }
    softmax_phase<false, QS, NT_, LPN>(p, s, tile0, 0, nullptr, nullptr);
    __syncthreads();

    constexpr int nct = KPASS >> 4;   // 18
    const int kso = cout >> 5;
    const __amdgpu_buffer_rsrc_t wq_rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(lp.Wq), 0, -1, 0x00020000);
    unsigned short* zt16 = reinterpret_cast<unsigned short*>(s.ztile);
    const char* ztb = reinterpret_cast<const char*>(s.ztile);
    int dn[NPW], rowid[NPW];
    f32x4 dq[NPW];
    f32x4 dq_hi[LONG ? NPW : 1];
#pragma unroll
    for (int nn = 0; nn < (LONG ? NPW : 1); ++nn) dq_hi[nn] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int nn = 0; nn < NPW; ++nn) {
        const int node = wave * NPW + nn;
        dn[nn] = __builtin_amdgcn_readfirstlane(min(s.deg[node], KMAX));
        const int e = max(min(lr, dn[nn] - 1), 0);
        rowid[nn] = dn[nn] > 0 ? __float_as_int(s.qbuf[(size_t)node * qnode_stride(QS) + e * QLD + 9]) : 0;
        dq[nn] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    const int mrow = lr < FGC_M ? lr : FGC_M - 1;  // rows 9..15 of the product are never read
    f32x4 dcacc = f32x4{0.f, 0.f, 0.f, 0.f};

    for (int pass = 0; pass < p.passes; ++pass) {
        const int cpass = pass * KC;
        const bool first = cpass < p.c0;                                            // block-uniform
        const float* base = first ? p.src0 : p.src1;
        const unsigned rowbytes = (unsigned)(first ? p.c0 : p.c1) * 2u;
        const unsigned laneoff = (unsigned)((first ? cpass : cpass - p.c0) + 8 * lq) * 2u;
        const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(base), 0, -1, 0x00020000);
        // B fragments of the per-node products: the neighbour rows of all 8 nodes of the wave, requested up front
        u32x4 bx[NPW];
#pragma unroll
        for (int nn = 0; nn < NPW; ++nn)
            bx[nn] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(
                                                   rsrc, __umul24((unsigned)rowid[nn], rowbytes) + laneoff, 0, 0));
        // ---- dz tile = s tile x Wq[pass] -> LDS (bf16)
        {
            f32x4 acc[RT][K1_CTW];
#pragma unroll
            for (int r = 0; r < RT; ++r)
#pragma unroll
                for (int c = 0; c < K1_CTW; ++c) acc[r][c] = f32x4{0.f, 0.f, 0.f, 0.f};
            for (int ks = 0; ks < kso; ++ks) {
                u32x4 b[K1_CTW];
#pragma unroll
                for (int c = 0; c < K1_CTW; ++c)
                    b[c] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(
                                                         wq_rs, (unsigned)((min(wave + c * 4, nct - 1) * 64 + lane) * 16),
                                                         (unsigned)((pass * kso + ks) * nct * 1024), 0));
                u32x4 a[RT];
#pragma unroll
                for (int r = 0; r < RT; ++r) a[r] = *reinterpret_cast<const u32x4*>(dst + (r * 16 + lr) * obytes + ks * 64 + lq * 16);
#pragma unroll
                for (int c = 0; c < K1_CTW; ++c)
#pragma unroll
                    for (int r = 0; r < RT; ++r)
                        acc[r][c] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a[r]),
                                                                           __builtin_bit_cast(bf16x8, b[c]), acc[r][c], 0, 0, 0);
            }
            if (pass > 0) __syncthreads();  // the previous pass' readers of the dz tile are done
#pragma unroll
            for (int c = 0; c < K1_CTW; ++c) {
                const int ct = wave + c * 4;
                if (ct >= nct) continue;
#pragma unroll
                for (int r = 0; r < RT; ++r)
#pragma unroll
                    for (int t = 0; t < 4; ++t)
                        zt16[(size_t)(r * 16 + lq * 4 + t) * ZSTRIDE_BF + ct * 16 + lr] = f_to_bf(acc[r][c][t]);
            }
        }
        __syncthreads();
        // ---- per node: dq += dz_i (9 x 32) . X_i (32 x 16 edge slots), one MFMA
#pragma unroll
        for (int nn = 0; nn < NPW; ++nn) {
            const int node = wave * NPW + nn;
            const u32x4 a = *reinterpret_cast<const u32x4*>(ztb + (size_t)node * (ZSTRIDE_BF * 2) + (mrow * 32 + 8 * lq) * 2);
            dq[nn] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, bx[nn]),
                                                            dq[nn], 0, 0, 0);
            if constexpr (LONG) {
                if (dn[nn] > 16) {                      // wave-uniform: edge slots 16..23
                    const int e = min(16 + lr, dn[nn] - 1);
                    const int row = __float_as_int(s.qbuf[(size_t)node * qnode_stride(QS) + e * QLD + 9]);
                    const u32x4 x1 = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(
                                                                   rsrc, __umul24((unsigned)row, rowbytes) + laneoff, 0, 0));
                    dq_hi[nn] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a),
                                                                       __builtin_bit_cast(bf16x8, x1), dq_hi[nn], 0, 0, 0);
                }
            }
        }
    }

    // ---- softmax backward: lane (edge = lr, m0 = 4*lq) holds dq[m0..m0+3][edge]  (as conv_bwd_logits_deep_kernel)
#pragma unroll
    for (int nn = 0; nn < NPW; ++nn) {
        const int node = wave * NPW + nn;
        const int i = tile0 + node;
        const int d = dn[nn];
        if (i >= p.n) continue;                    // wave-uniform
        if (d <= 0) {                              // an isolated node still owns a dag row
            if (lr == 0 && lq < 3)
                *reinterpret_cast<f32x4*>(lp.dag + (size_t)i * FGC_AG_LD + 4 * lq) = f32x4{0.f, 0.f, 0.f, 0.f};
            continue;
        }
        const int e0 = s.deg[TILE + 4 + node];     // first edge id, left in LDS by the softmax phase
        f32x4 da = f32x4{0.f, 0.f, 0.f, 0.f};
        const int ntile = (LONG && d > 16) ? 2 : 1;
        for (int et = 0; et < ntile; ++et) {
            const int edge = 16 * et + lr;
            const bool ok = edge < d;
            const float* qr = s.qbuf + (size_t)node * qnode_stride(QS) + min(edge, d - 1) * QLD;
            f32x4 q = f32x4{0.f, 0.f, 0.f, 0.f};
            if (lq < 2) q = *reinterpret_cast<const f32x4*>(qr + 4 * lq);
            else if (lq == 2) q[0] = qr[8];
            f32x4 g = (LONG && et == 1) ? dq_hi[LONG ? nn : 0] : dq[nn];
            if (lq == 2) { g[1] = 0.f; g[2] = 0.f; g[3] = 0.f; }
            if (lq == 3) g = f32x4{0.f, 0.f, 0.f, 0.f};
            float dot = q[0] * g[0] + q[1] * g[1] + q[2] * g[2] + q[3] * g[3];
            dot += __shfl_xor(dot, 16);
            dot += __shfl_xor(dot, 32);
            f32x4 dl;
#pragma unroll
            for (int t = 0; t < 4; ++t) dl[t] = ok ? q[t] * (g[t] - dot) : 0.f;
            if (ok && lq < 3) *reinterpret_cast<f32x4*>(lp.dl + (size_t)(e0 + edge) * FGC_DL_LD + 4 * lq) = dl;
            da += dl;
        }
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            float v = da[t];
            FGC_ROW16_SUM(v);
            da[t] = v;
        }
        dcacc += da;
        if (lr == 0 && lq < 3) *reinterpret_cast<f32x4*>(lp.dag + (size_t)i * FGC_AG_LD + 4 * lq) = da;
    }
    __syncthreads();
    if (lr == 0 && lq < 3) {
#pragma unroll
        for (int t = 0; t < 4; ++t) red[wave * 12 + 4 * lq + t] = dcacc[t];
    }
    __syncthreads();
    if (tid < 12) {
        const float v = tid < FGC_M ? (red[tid] + red[12 + tid]) + (red[24 + tid] + red[36 + tid]) : 0.f;
        lp.dc_part[(size_t)blockIdx.x * 12 + tid] = v;
    }
}

// ---------------------------------------------------------------------------------------------
// K2: data gradient = forward core over the transposed graph
// ---------------------------------------------------------------------------------------------
template <int LPN, bool VEC4>
__global__ __launch_bounds__(NTHREADS) void conv_bwd_data_kernel(CoreParams p, DataEpilogue ep) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    const Smem s = carve(smem_raw, ZSTRIDE);
    float* dagt = s.extra;  // [TILE][24]: da | dg of the tile's nodes
    const int tile0 = block_tile0(p);
    const int tid = threadIdx.x;
    const WaveTiling wt = wave_tiling(p.npad, threadIdx.x >> 6);

    float dgsum[FGC_M];
#pragma unroll
    for (int m = 0; m < FGC_M; ++m) dgsum[m] = 0.f;
    const int dmine = softmax_phase<true>(p, s, tile0, 0, ep.dl, dgsum);
    zero_zpad(p, s);
    const int nchunks = edge_chunks(s, dmine);

    f32x4 acc[RT][CTW];
#pragma unroll
    for (int r = 0; r < RT; ++r)
#pragma unroll
        for (int c = 0; c < CTW; ++c) acc[r][c] = f32x4{0.f, 0.f, 0.f, 0.f};

    const bool want_dx = ep.dx0 != nullptr;
    for (int pass = 0; pass < p.passes; ++pass) {
        f32x4 z[FGC_M];
#pragma unroll
        for (int m = 0; m < FGC_M; ++m) z[m] = f32x4{0.f, 0.f, 0.f, 0.f};
        aggregate_pass<LPN, VEC4>(p, s, pass, 0, z);
        for (int ch = 1; ch < nchunks; ++ch) {
            __syncthreads();
            float dummy[FGC_M];
#pragma unroll
            for (int m = 0; m < FGC_M; ++m) dummy[m] = 0.f;
            if (pass == 0) softmax_phase<true>(p, s, tile0, ch * KMAX, ep.dl, dgsum);
            else softmax_phase<false>(p, s, tile0, ch * KMAX, nullptr, dummy);
            __syncthreads();
            aggregate_pass<LPN, VEC4>(p, s, pass, ch * KMAX, z);
        }
        if (nchunks > 1 && pass + 1 < p.passes) {
            __syncthreads();
            softmax_phase<false>(p, s, tile0, 0, nullptr, nullptr);
            __syncthreads();
        }
        // r[j, m*cout + channel] straight from the accumulators
        {
            const int node = tid / LPN, cl = tid % LPN;
            const int j = tile0 + node;
            const int ch0 = pass * KC + cl * 4;
            if (node < TILE && j < p.n && ch0 < p.cg) {
                float* rr = ep.r + (size_t)j * ep.rld + ch0;
                if (VEC4) {
#pragma unroll
                    for (int m = 0; m < FGC_M; ++m) *reinterpret_cast<f32x4*>(rr + m * p.cg) = z[m];
                } else {
#pragma unroll
                    for (int m = 0; m < FGC_M; ++m)
#pragma unroll
                        for (int t = 0; t < 4; ++t)
                            if (ch0 + t < p.cg) rr[m * p.cg + t] = z[m][t];
                }
            }
        }
        if (!want_dx) continue;
        if (pass > 0) __syncthreads();
        store_ztile<LPN>(p, s, z);
        __syncthreads();
        gemm_pass(p, s, pass, wt, acc);
    }
    // dg_j = sum over in-edges of dl: reduce the 8 softmax lanes of each node
    {
        const int node = tid >> 3, kl = tid & 7;
        const int j = tile0 + node;
#pragma unroll
        for (int m = 0; m < FGC_M; ++m) {
            float v = dgsum[m];
            v += __shfl_xor(v, 1);
            v += __shfl_xor(v, 2);
            v += __shfl_xor(v, 4);
            dgsum[m] = v;
        }
        if (kl == 0) {
            float* t = dagt + node * 24;
            if (j < p.n) {
                const float* da = ep.dag + (size_t)j * FGC_AG_LD;
#pragma unroll
                for (int m = 0; m < FGC_M; ++m) {
                    t[m] = da[m];
                    t[12 + m] = dgsum[m];
                }
                float* o = ep.dag + (size_t)j * FGC_AG_LD + 12;
                *reinterpret_cast<f32x4*>(o) = f32x4{dgsum[0], dgsum[1], dgsum[2], dgsum[3]};
                *reinterpret_cast<f32x4*>(o + 4) = f32x4{dgsum[4], dgsum[5], dgsum[6], dgsum[7]};
                *reinterpret_cast<f32x4*>(o + 8) = f32x4{dgsum[8], 0.f, 0.f, 0.f};
                // da | dg behind the node's r row (scalar stores: rld need not be a multiple of 4 on this path)
                float* rt = ep.r + (size_t)j * ep.rld + (FGC_M * p.cg);
#pragma unroll
                for (int m = 0; m < 12; ++m) {
                    rt[m] = m < FGC_M ? da[m] : 0.f;
                    rt[12 + m] = m < FGC_M ? dgsum[m] : 0.f;
                }
            } else {
#pragma unroll
                for (int m = 0; m < 24; ++m) t[m] = 0.f;
            }
        }
    }
    if (!want_dx) return;
    __syncthreads();
    const int oldd = p.npad + 4;
    float* otile = s.ztile;
    store_acc(otile, oldd, wt, p.npad, acc);
    __syncthreads();

    // dx rows: GEMM part + da.u + dg.v ; optional 4:1 row sum (input was an upsampled coarse tensor)
    const int kparts = wt.kparts;
    const int group = 1 << ep.shiftf;  // 1 or 4 tile rows per source row
    const int nsrc = TILE / group;
    for (int t = tid; t < nsrc * ep.cin; t += NTHREADS) {
        const int sr = t / ep.cin, c = t % ep.cin;
        float val = 0.f;
        bool any = false;
        for (int q = 0; q < group; ++q) {
            const int row = sr * group + q;
            if (tile0 + row >= p.n) continue;
            any = true;
            float g = 0.f;
            for (int kp = 0; kp < kparts; ++kp) g += otile[((size_t)kp * TILE + row) * oldd + c];
            const float* dg = dagt + row * 24;
#pragma unroll
            for (int m = 0; m < FGC_M; ++m) {
                g = fmaf(dg[m], ep.u[m * ep.cin + c], g);
                g = fmaf(dg[12 + m], ep.v[m * ep.cin + c], g);
            }
            val += g;
        }
        if (!any) continue;
        const size_t srow = (size_t)((tile0 >> ep.shiftf) + sr);
        if (c < ep.c0f) {
            float* o = ep.dx0 + srow * ep.c0f + c;
            *o = ep.acc0 ? *o + val : val;
        } else if (ep.dx1) {
            float* o = ep.dx1 + srow * ep.c1f + (c - ep.c0f);
            *o = ep.acc1 ? *o + val : val;
        }
    }
}

// ---------------------------------------------------------------------------------------------
// K3: C[P,Q] = sum_rows A[row,P] * X[row >> shift, Q]   (X = [x0 | x1]); f32 MFMA with K = rows.
// Both operands are row-major with the reduction index as the slow dimension, so a lane's 16-byte load of
// A[row, p0+4*lr .. +3] holds the SAME k (row) for 4 different output rows: MFMA number e takes element e,
// i.e. MFMA e owns output rows p0 + 4*i + e (i = MFMA row index).  One dwordx4 of A and one of X per lane feed
// 16 MFMAs (a 64 x 64 tile per wave, 4 rows of K per step); no LDS staging, no barrier in the loop.
// grid (P tiles * Q tiles, row splits); the 4 waves of a workgroup interleave the k-steps of their split and are
// summed through LDS in a fixed order; partials go to slab[split][P][Q], reduced by reduce_jobs.
// ---------------------------------------------------------------------------------------------
template <bool VEC4>
__device__ __forceinline__ void tn_load(const float* __restrict__ A, int lda, int P, const float* __restrict__ x0,
                                        const float* __restrict__ x1, int c0, int c1, int shift, int row, bool valid,
                                        int pbase, int qbase, f32x4& a, f32x4& b) {
    a = f32x4{0.f, 0.f, 0.f, 0.f};
    b = f32x4{0.f, 0.f, 0.f, 0.f};
    if (!valid) return;
    const int Q = c0 + c1;
    const size_t sr = (size_t)(row >> shift);
    if (VEC4) {
        if (pbase < P) a = *reinterpret_cast<const f32x4*>(A + (size_t)row * lda + pbase);
        if (qbase < c0) b = *reinterpret_cast<const f32x4*>(x0 + sr * c0 + qbase);
        else if (qbase < Q) b = *reinterpret_cast<const f32x4*>(x1 + sr * c1 + (qbase - c0));
    } else {
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            if (pbase + t < P) a[t] = A[(size_t)row * lda + pbase + t];
            const int q = qbase + t;
            if (q < c0) b[t] = x0[sr * c0 + q];
            else if (q < Q) b[t] = x1[sr * c1 + (q - c0)];
        }
    }
}

// Workgroup -> (output tile, node-range split) of the weight-gradient GEMMs.  Workgroups are dealt round-robin over the
// 8 XCDs, each with a private L2.  With the plain (tile, split) grid the tiles of one split - which read the SAME rows of r
// and x - land on different XCDs and every L2 fetches those rows from HBM again (dconv2: x came in ten times).  Here all
// tiles of a split run on one XCD, next to each other in dispatch order: the rows are fetched once and the other tiles
// hit in L2.  The grid is padded to 8 * ceil(splits / 8) splits; workgroups of a padding split return at once.
// Same work per (tile, split), same slabs, same sums: results are unchanged bit for bit.
__device__ __forceinline__ bool tn_block(int ntiles, int nsplits, int& tile, int& split, int vblock = -1) {
    // (vblock: the workgroup's index within ITS job of a grouped launch; jobs start at multiples of 8, so vblock & 7 is
    //  still the XCD the hardware dealt this workgroup to)
    const int L = vblock >= 0 ? vblock : (int)blockIdx.x, xcd = L & 7, idx = L >> 3;
    tile = idx % ntiles;
    split = (idx / ntiles) * 8 + xcd;
    return split < nsplits;
}
static inline dim3 tn_grid(int ntiles, int nsplits) { return dim3((unsigned)(ntiles * 8 * cdiv(nsplits, 8))); }

// The weight-gradient GEMMs of several layers in ONE launch (fgc_conv_bwd_reduce with FGC_CONV_DEFER_DW): every job is what
// one launch of the kernel would be - same tiles, same slabs, same sums, bit-identical gradients -, its workgroups are the
// range [block0, block0 + tn_grid) of the grid.  On the bf16 network a layer's GEMM is 5-15 us of ramp and tail around a few
// microseconds of streaming: eight of them back to back cost four times what their work takes.
struct TnArgs {
    const void* A;
    const void* x0;
    const void* x1;
    float* slab;
    int lda, P, c0, c1, shift, rows, rps;
    int block0;
};
constexpr int TN_MAX_JOBS = 8;
struct TnJobs {
    TnArgs job[TN_MAX_JOBS];
    int njobs;
};
__device__ __forceinline__ int tn_job_of(const TnJobs& J) {
    int q = 0;
#pragma unroll
    for (int t = 1; t < TN_MAX_JOBS; ++t)
        if (t < J.njobs && (int)blockIdx.x >= J.job[t].block0) q = t;
    return q;
}

template <bool VEC4>
__global__ __launch_bounds__(256) void gemm_tn_kernel(const float* __restrict__ A, int lda, int P,
                                                      const float* __restrict__ x0, const float* __restrict__ x1,
                                                      int c0, int c1, int shift, int rows, int rows_per_split,
                                                      float* __restrict__ slab) {
    __shared__ float red[4][64][65];
    const int Q = c0 + c1;
    const int npt = (P + 63) >> 6;
    int tile_id, split_id;
    if (!tn_block(npt * ((Q + 63) >> 6), (rows + rows_per_split - 1) / rows_per_split, tile_id, split_id)) return;
    const int pt = tile_id % npt, qt = tile_id / npt;
    const int p0 = pt * 64, q0 = qt * 64;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, lr = lane & 15, lq = lane >> 4;
    const int r_begin = split_id * rows_per_split;
    const int r_end = min(rows, r_begin + rows_per_split);
    const int nsteps = (r_end - r_begin + 3) >> 2;
    const int pbase = p0 + 4 * lr, qbase = q0 + 4 * lr;
    f32x4 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    // two operand register sets with fixed roles (unrolled by 2, no copies of in-flight loads)
    f32x4 a0, b0, a1, b1;
    auto ld = [&](int step, f32x4& a, f32x4& b) {
        const int row = r_begin + 4 * step + lq;
        tn_load<VEC4>(A, lda, P, x0, x1, c0, c1, shift, row, step < nsteps && row < r_end, pbase, qbase, a, b);
    };
    auto mm = [&](const f32x4& a, const f32x4& b) {
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i], b[j], acc[i][j], 0, 0, 0);
    };
    ld(wave, a0, b0);
    ld(wave + 4, a1, b1);
    for (int s = wave; s < nsteps; s += 8) {
        mm(a0, b0);
        ld(s + 8, a0, b0);
        if (s + 4 < nsteps) mm(a1, b1);
        ld(s + 12, a1, b1);
    }
    // C layout of acc[i][j]: column index lr -> q = 4*lr + j ; row index lq*4+reg -> p = 4*(lq*4+reg) + i
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int t = 0; t < 4; ++t) red[wave][4 * (lq * 4 + t) + i][4 * lr + j] = acc[i][j][t];
    __syncthreads();
    float* out = slab + (size_t)split_id * P * Q;
    for (int t = tid; t < 64 * 64; t += 256) {
        const int pp = t >> 6, qq = t & 63;
        if (p0 + pp < P && q0 + qq < Q)
            out[(size_t)(p0 + pp) * Q + q0 + qq] = (red[0][pp][qq] + red[1][pp][qq]) + (red[2][pp][qq] + red[3][pp][qq]);
    }
}

// Streaming form of gemm_tn_kernel for 16-byte aligned operands whose widths are multiples of 4 (every layer but
// conv1).  Same tiling, same summation order, bit-identical results; what differs is how memory is asked for:
//   * loads are UNCONDITIONAL (row and column indices clamped into the operands, out-of-range rows zeroed by a
//     select on the A fragment): no exec-masked branch around a load, so hipcc counts its s_waitcnt instead of
//     draining everything with vmcnt(0) in front of every MFMA group;
//   * four operand register sets with fixed roles (loop unrolled by 4): each load has three MFMA groups = 48
//     matrix instructions to land;
//   * the 4-wave sum goes through 2 x 16 KB of LDS instead of 4, so four workgroups are resident per CU.
// NJ = 4: 64 x 64 output tile (lane lr owns columns 4*lr .. 4*lr+3); NJ = 2: 64 x 32 for operands only 32 wide (columns
// 2*lr, 2*lr+1: half the MFMAs instead of multiplying clamped duplicates)
// BF: both operands are bf16 tensors (FGC_CONV_BF16); they are widened on load and multiplied on the fp32 MFMA: the
// products are exact and the sum over the nodes stays an fp32 chain, as in the fp32 network
template <int NJ, bool BF>
__device__ __forceinline__ void tn_stream_body(const float* __restrict__ A, int lda, int P, const float* __restrict__ x0,
                                               const float* __restrict__ x1, int c0, int c1, int shift, int rows,
                                               int rows_per_split, float* __restrict__ slab, int vblock) {
    __shared__ float red[2][64][65];
    const int Q = c0 + c1;
    const int npt = (P + 63) >> 6;
    int tile_id, split_id;
    if (!tn_block(npt * ((Q + 16 * NJ - 1) / (16 * NJ)), (rows + rows_per_split - 1) / rows_per_split, tile_id, split_id, vblock))
        return;
    const int pt = tile_id % npt, qt = tile_id / npt;
    const int p0 = pt * 64, q0 = qt * (16 * NJ);
    const int tid = threadIdx.x, lane = tid & 63, lr = lane & 15, lq = lane >> 4;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r_begin = split_id * rows_per_split;
    const int r_end = min(rows, r_begin + rows_per_split);
    const int nsteps = (r_end - r_begin + 3) >> 2;
    // column quads of this lane, clamped into the operands (results of clamped columns are never stored)
    const int pc = min(p0 + 4 * lr, P - 4);
    const int qc = min(q0 + NJ * lr, Q - NJ);
    const float* bsrc = BF ? reinterpret_cast<const float*>(reinterpret_cast<const unsigned short*>(qc < c0 ? x0 : x1) +
                                                            (qc < c0 ? qc : qc - c0))
                           : (qc < c0 ? x0 + qc : x1 + (qc - c0));
    const int bld = qc < c0 ? c0 : c1;
    f32x4 acc[4][NJ];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < NJ; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    // A load only requests memory; the zeroing of out-of-range rows happens where the fragment is consumed (a select
    // right behind the load would make the compiler wait for it on the spot)
    auto ld = [&](int step, f32x4& a, f32x4& b, int& row) {
        row = r_begin + 4 * step + lq;
        const int rc = min(row, r_end - 1);
        if constexpr (BF) {
            const unsigned short* A16 = reinterpret_cast<const unsigned short*>(A);
            const unsigned short* b16 = reinterpret_cast<const unsigned short*>(bsrc);
            a = bf4_to_f4(*reinterpret_cast<const u32x2*>(A16 + (size_t)rc * lda + pc));
            if constexpr (NJ == 4) {
                b = bf4_to_f4(*reinterpret_cast<const u32x2*>(b16 + (size_t)(rc >> shift) * bld));
            } else {
                const f32x2c b2 = bf2_to_f2(*reinterpret_cast<const unsigned*>(b16 + (size_t)(rc >> shift) * bld));
                b = f32x4{b2[0], b2[1], 0.f, 0.f};
            }
            return;
        }
        a = *reinterpret_cast<const f32x4*>(A + (size_t)rc * lda + pc);
        if constexpr (NJ == 4) {
            b = *reinterpret_cast<const f32x4*>(bsrc + (size_t)(rc >> shift) * bld);
        } else {
            const f32x2c b2 = *reinterpret_cast<const f32x2c*>(bsrc + (size_t)(rc >> shift) * bld);
            b = f32x4{b2[0], b2[1], 0.f, 0.f};
        }
    };
    auto mm = [&](f32x4 a, const f32x4& b, int row) {
        if (row >= r_end) a = f32x4{0.f, 0.f, 0.f, 0.f};   // a select, not a branch
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < NJ; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i], b[j], acc[i][j], 0, 0, 0);
    };
    f32x4 a0, b0, a1, b1, a2, b2, a3, b3;
    int r0, r1, r2, r3;
    ld(wave, a0, b0, r0);
    ld(wave + 4, a1, b1, r1);
    ld(wave + 8, a2, b2, r2);
    ld(wave + 12, a3, b3, r3);
    // sched_barrier: keep the program order "16 MFMAs, then the refill of the set they consumed" (left alone the
    // scheduler sinks every refill to just in front of its use and the prefetch distance collapses to zero)
#define FGC_TN_STEP(A_, B_, R_, NEXT_)          \
    mm(A_, B_, R_);                             \
    __builtin_amdgcn_sched_barrier(0);          \
    ld(NEXT_, A_, B_, R_);                      \
    __builtin_amdgcn_sched_barrier(0);
    for (int s = wave; s < nsteps; s += 16) {   // steps past the end load a clamped row and multiply by zero
        FGC_TN_STEP(a0, b0, r0, s + 16)
        FGC_TN_STEP(a1, b1, r1, s + 20)
        FGC_TN_STEP(a2, b2, r2, s + 24)
        FGC_TN_STEP(a3, b3, r3, s + 28)
    }
#undef FGC_TN_STEP
    // (w0 + w1) + (w2 + w3), as gemm_tn_kernel sums them.  C layout of acc[i][j]: column lr -> q = 4*lr + j,
    // row lq*4+reg -> p = 4*(lq*4+reg) + i
    auto put = [&](int slot) {
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < NJ; ++j)
#pragma unroll
                for (int t = 0; t < 4; ++t) red[slot][4 * (lq * 4 + t) + i][NJ * lr + j] = acc[i][j][t];
    };
    auto add = [&](int slot) {
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < NJ; ++j)
#pragma unroll
                for (int t = 0; t < 4; ++t) acc[i][j][t] += red[slot][4 * (lq * 4 + t) + i][NJ * lr + j];
    };
    if (wave == 1) put(0);
    if (wave == 3) put(1);
    __syncthreads();
    if (wave == 0) add(0);
    if (wave == 2) add(1);
    __syncthreads();
    if (wave == 0) put(0);
    if (wave == 2) put(1);
    __syncthreads();
    float* out = slab + (size_t)split_id * P * Q;
    for (int t = tid; t < 64 * 16 * NJ; t += 256) {
        const int pp = t / (16 * NJ), qq = t % (16 * NJ);
        if (p0 + pp < P && q0 + qq < Q) out[(size_t)(p0 + pp) * Q + q0 + qq] = red[0][pp][qq] + red[1][pp][qq];
    }
}
template <int NJ, bool BF = false>
__global__ __launch_bounds__(256, 4) void gemm_tn_stream_kernel(const float* __restrict__ A, int lda, int P,
                                                                const float* __restrict__ x0,
                                                                const float* __restrict__ x1, int c0, int c1, int shift,
                                                                int rows, int rows_per_split, float* __restrict__ slab) {
    tn_stream_body<NJ, BF>(A, lda, P, x0, x1, c0, c1, shift, rows, rows_per_split, slab, -1);
}
template <int NJ, bool BF = false>
__global__ __launch_bounds__(256, 4) void gemm_tn_stream_group_kernel(TnJobs J) {
    const TnArgs& a = J.job[tn_job_of(J)];
    tn_stream_body<NJ, BF>((const float*)a.A, a.lda, a.P, (const float*)a.x0, (const float*)a.x1, a.c0, a.c1, a.shift, a.rows,
                           a.rps, a.slab, (int)blockIdx.x - a.block0);
}

// ---------------------------------------------------------------------------------------------
// K3 for bf16-stored operands ON the bf16 matrix cores (FGC_CONV_BF16).  C[P,Q] = sum_rows A[row,P] * X[row >> shift, Q]
// reduces over the rows, the slow index of both operands, while a v_mfma_f32_16x16x32_bf16 fragment wants 8 consecutive k
// of ONE output row / column in a lane.  The transposition is done by the LDS read: a chunk of 32 rows of A (up to 320
// columns) and of X (QT * 16 columns) is staged row-major, as it lies in memory (16-byte pieces, coalesced), and read back
// with ds_read_b64_tr_b16: per 16-lane group a 4 row x 16 column block comes back column-major, lane i holding column i
// of the four rows.  Two such reads (rows 4*lq .. +3 and 16 + 4*lq .. +3) are a whole fragment; A and X use the same row
// order, so the permuted k is consistent.  Row strides == 32 bytes mod 256, an odd multiple of 32: the eight rows a
// 32-lane half touches land on eight disjoint 32-byte bank spans.  Wave w owns row tiles 5w .. 5w+4 of the product
// (columns of A) and all column tiles: 18 transposed reads per 20 MFMAs.  The kernel streams: what bounds it is how fast
// the rows of r arrive, so the next chunk travels through registers under the current one.  fp32 accumulators; slabs and
// their fixed-order sum as for the fp32 kernels.
// ---------------------------------------------------------------------------------------------
constexpr int TNB_THREADS = 256;
constexpr int TNB_PC = 320;               // columns of A per workgroup
constexpr int TNB_AS = TNB_PC * 2 + 32;   // LDS row strides in bytes
typedef short s16x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ u32x4 tnb_frag(const char* tile, int stride, int col0, int lq, int lr) {
    // block rows 4*lq + q (then 16 + 4*lq + q), columns col0 + 4*p .. +3 for lane 4*q + p of the group
    const char* a = tile + (4 * lq + (lr >> 2)) * stride + (col0 + 4 * (lr & 3)) * 2;
    const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)a);
    const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(a + 16 * stride));
    const u32x2 l2 = __builtin_bit_cast(u32x2, lo), h2 = __builtin_bit_cast(u32x2, hi);
    return u32x4{l2[0], l2[1], h2[0], h2[1]};
}

template <int QT>
__device__ __forceinline__ void tn_bf16_body(const unsigned short* __restrict__ A, int lda, int P,
                                             const unsigned short* __restrict__ x0, const unsigned short* __restrict__ x1, int c0,
                                             int c1, int shift, int rows, int rows_per_split, float* __restrict__ slab,
                                             int vblock) {
    constexpr int QC = QT * 16;
    constexpr int XS = QC * 2 + 32;
    constexpr int APC = TNB_PC / 8;                                  // 16-byte pieces per row of the A chunk
    constexpr int NA = 32 * APC / TNB_THREADS;                       // pieces per thread: 5
    static_assert(32 * APC % TNB_THREADS == 0 && 32 * (QC / 8) <= TNB_THREADS, "staging shape");
    __shared__ __attribute__((aligned(16))) char As[2][32 * TNB_AS];
    __shared__ __attribute__((aligned(16))) char Xs[2][32 * XS];
    const int Q = c0 + c1;
    const int npc = (P + TNB_PC - 1) / TNB_PC;
    int tile_id, split_id;
    if (!tn_block(npc * (Q / QC), (rows + rows_per_split - 1) / rows_per_split, tile_id, split_id, vblock)) return;
    const int pc = tile_id % npc, qc = tile_id / npc;
    const int p0 = pc * TNB_PC, q0 = qc * QC;
    const int pw = min(P - p0, TNB_PC);                              // valid columns of A here (a multiple of 8)
    const int tid = threadIdx.x, lane = tid & 63, lr = lane & 15, lq = lane >> 4;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r_begin = split_id * rows_per_split;
    const int r_end = min(rows, r_begin + rows_per_split);
    const int nchunks = (r_end - r_begin + 31) >> 5;

    u32x4 ra[NA], rx;
    const int xrow = min(tid, 32 * (QC / 8) - 1) / (QC / 8), xcol = (min(tid, 32 * (QC / 8) - 1) % (QC / 8)) * 8;
    const int qcol = q0 + xcol;
    const bool x_first = qcol < c0;
    const unsigned short* xsrc = x_first ? x0 : x1;
    const int xld = x_first ? c0 : c1;
    const int xoff = min(x_first ? qcol : qcol - c0, xld - 8);
    auto fetch = [&](int ch) {
        const int rb = r_begin + ch * 32;
#pragma unroll
        for (int i = 0; i < NA; ++i) {
            const int t = tid + i * TNB_THREADS;
            const int row = min(rb + t / APC, r_end - 1);
            const int col = p0 + min((t % APC) * 8, pw - 8);
            ra[i] = *reinterpret_cast<const u32x4*>(A + (size_t)row * lda + col);
        }
        rx = *reinterpret_cast<const u32x4*>(xsrc + (size_t)(min(rb + xrow, r_end - 1) >> shift) * xld + xoff);
    };
    auto stage = [&](int ch, int buf) {
        const int rb = r_begin + ch * 32;
#pragma unroll
        for (int i = 0; i < NA; ++i) {
            const int t = tid + i * TNB_THREADS;
            const int r = t / APC, c = (t % APC) * 8;
            const bool ok = rb + r < r_end && c < pw;
            *reinterpret_cast<u32x4*>(&As[buf][r * TNB_AS + c * 2]) = ok ? ra[i] : u32x4{0u, 0u, 0u, 0u};
        }
        if (tid < 32 * (QC / 8)) {
            const bool ok = rb + xrow < r_end && qcol < Q;
            *reinterpret_cast<u32x4*>(&Xs[buf][xrow * XS + xcol * 2]) = ok ? rx : u32x4{0u, 0u, 0u, 0u};
        }
    };
    f32x4 acc[5][QT];
#pragma unroll
    for (int i = 0; i < 5; ++i)
#pragma unroll
        for (int j = 0; j < QT; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    if (nchunks > 0) {
        fetch(0);
        stage(0, 0);
    }
    __syncthreads();
    for (int ch = 0; ch < nchunks; ++ch) {
        const int buf = ch & 1;
        if (ch + 1 < nchunks) fetch(ch + 1);
        u32x4 af[5], bq[QT];
#pragma unroll
        for (int i = 0; i < 5; ++i) af[i] = tnb_frag(As[buf], TNB_AS, (wave * 5 + i) * 16, lq, lr);
#pragma unroll
        for (int j = 0; j < QT; ++j) bq[j] = tnb_frag(Xs[buf], XS, j * 16, lq, lr);
#pragma unroll
        for (int i = 0; i < 5; ++i)
#pragma unroll
            for (int j = 0; j < QT; ++j)
                acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, af[i]), __builtin_bit_cast(bf16x8, bq[j]),
                                                                   acc[i][j], 0, 0, 0);
        if (ch + 1 < nchunks) stage(ch + 1, buf ^ 1);     // (the other buffer: its readers finished before the last barrier)
        __syncthreads();
    }
    // C layout: column = lr -> q, row = 4*lq + reg -> p
    float* out = slab + (size_t)split_id * P * Q;
#pragma unroll
    for (int i = 0; i < 5; ++i)
#pragma unroll
        for (int j = 0; j < QT; ++j)
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const int pp = p0 + (wave * 5 + i) * 16 + lq * 4 + t, qq = q0 + j * 16 + lr;
                if (pp < p0 + pw && qq < Q) out[(size_t)pp * Q + qq] = acc[i][j][t];
            }
}
template <int QT>
__global__ __launch_bounds__(TNB_THREADS, 2) void gemm_tn_bf16_kernel(const unsigned short* __restrict__ A, int lda, int P,
                                                                      const unsigned short* __restrict__ x0,
                                                                      const unsigned short* __restrict__ x1, int c0, int c1,
                                                                      int shift, int rows, int rows_per_split,
                                                                      float* __restrict__ slab) {
    tn_bf16_body<QT>(A, lda, P, x0, x1, c0, c1, shift, rows, rows_per_split, slab, -1);
}
template <int QT>
__global__ __launch_bounds__(TNB_THREADS, 2) void gemm_tn_bf16_group_kernel(TnJobs J) {
    const TnArgs& a = J.job[tn_job_of(J)];
    tn_bf16_body<QT>((const unsigned short*)a.A, a.lda, a.P, (const unsigned short*)a.x0, (const unsigned short*)a.x1, a.c0, a.c1,
                     a.shift, a.rows, a.rps, a.slab, (int)blockIdx.x - a.block0);
}

static bool tn_bf16_ok(int P, int c0, int c1) {
    if (opt(OPT_NO_TNBF16) == 1) return false;
    const int Q = c0 + c1;
    return P % 8 == 0 && c0 % 8 == 0 && c1 % 8 == 0 && Q % 32 == 0 && (c1 == 0 || c0 % 16 == 0) && c0 >= 8 && (c1 == 0 || c1 >= 8);
}

int launch_gemm_tn_stream(const char* tag, const float* A, int lda, int P, const float* x0, int c0, int rows,
                          int rows_per_split, int nsplits, float* slab, hipStream_t st) {
    if (c0 <= 32 && c0 % 2 == 0) {
        const dim3 grid = tn_grid(cdiv(P, 64), nsplits);
        FGC_LAUNCH(tag, st, gemm_tn_stream_kernel<2>, grid, dim3(256), 0, A, lda, P, x0, (const float*)nullptr, c0, 0, 0, rows,
                   rows_per_split, slab);
    } else {
        const dim3 grid = tn_grid(cdiv(P, 64) * cdiv(c0, 64), nsplits);
        FGC_LAUNCH(tag, st, gemm_tn_stream_kernel<4>, grid, dim3(256), 0, A, lda, P, x0, (const float*)nullptr, c0, 0, 0, rows,
                   rows_per_split, slab);
    }
    FGC_CHECK_LAUNCH("gemm_tn_stream_kernel");
    return FGC_OK;
}

static int tn_rows_per_slab(int n, int splits) { return cdiv(cdiv(n, splits), 4) * 4; }

// A split count near `desired` (at most `maxs`) whose EFFECTIVE number of slabs is a multiple of 8: tn_block gives every
// XCD the slabs s = xcd, xcd + 8, ...; with 27 slabs two XCDs would work through four of them and six through three.
int tn_balanced_splits(int desired, int maxs, int rows) {
    desired = std::max(1, std::min(desired, maxs));
    for (int delta = 0; delta < 24; ++delta)
        for (int sgn = 1; sgn >= -1; sgn -= 2) {
            const int s = desired + sgn * delta;
            if (s >= 8 && s <= maxs && cdiv(rows, tn_rows_per_slab(rows, s)) % 8 == 0) return s;
        }
    return desired;
}

// An XCD has 32 CUs x 4 resident workgroups of these kernels = 128 slots and is given tiles x (slabs / 8) workgroups: the
// slab count fills a whole number of slots per CU exactly once (a count just above a multiple of 32 leaves a few CUs with one
// workgroup more than the rest, and the launch waits for them).
static int tn_splits(int P, int Q, int rows) {
    const int tiles = cdiv(P, 64) * cdiv(Q, 64);
    // Two workgroups per CU (64 slots per XCD), not the four that fit: the kernel is bound by the matrix pipe and by HBM,
    // which eight waves per CU keep as busy as sixteen, and every workgroup less is a 16 KB slab less to write and to sum
    // (measured over 32 ... 256 slots: 64 is the minimum of the step, 2.192 -> 2.179 ms; the GEMMs 1-2 us faster each, the
    // sums 17 -> 12 us per launch).  Layers with more than 32 output tiles (the 128 -> 128 layer of the coarsest level)
    // would get one slab per XCD that way and keep 128 slots (32 -> 39 us otherwise).  FGC_TN_SLOTS: developer knob.
    const int slots = (int)opt(OPT_TN_SLOTS);
    int per = slots / tiles;
    if (per < 2) per = std::max(1, 2 * slots / tiles);
    return tn_balanced_splits(8 * per, cdiv(rows, 128), rows);
}

// Nodes per workgroup of the deep d-logits kernel: half tiles (16 nodes, four workgroups per CU) for the fp32 network on
// regular graphs.  A function of the descriptor alone: the partial-sum slots (dc per workgroup, db per tile of the fused
// prologue and of ds_db_kernel) are counted in these units by the workspace plan, the launches and the reductions.
// FGC_K1_NT16=0: 32-node tiles everywhere.
static int k1_nodes(const fgc_conv_desc* d) {
    // (read on every call, like the launch code reads FGC_NO_K1M / FGC_NO_K1DEEP: a process that changes a switch between
    //  two calls gets slot counts and kernels that agree)
    const bool on = !(opt(OPT_K1_NT16) == 0);
    const bool k1m = !(opt(OPT_NO_K1M) == 1);
    const bool k1deep = !(opt(OPT_NO_K1DEEP) == 1);
    if (!on || !k1m || !k1deep) return TILE;
    const int cin = d->c0 + d->c1;
    const ConvGeom g1 = conv_geom(cin, d->cout);
    if ((d->flags & FGC_CONV_BF16) && d->cout % 32 != 0) return TILE;
    const bool deep = g1.lpn == 8 && d->max_deg > 0 && d->max_deg <= 16 && conv_vec4_ok(d) && cin % 32 == 0 &&
                      (d->c1 == 0 || d->c0 % 32 == 0) && (size_t)d->n * 4 * 128 < 0xFFFFFFFFull;
    return deep ? 16 : TILE;
}

// The dz GEMM of the half-tile d-logits kernel on split bf16 operands (conv_bwd_logits_deep_kernel<.., SPLIT>): the fp32
// network's 32-wide layers on regular graphs.  A function of the descriptor and the options alone: it decides the
// layout (and size) of the packed operand Wq, which fgc_conv_pack may write long before the launch.  NO_K1_SPLIT=1: fp32 MFMA.
static bool k1_split(const fgc_conv_desc* d) {
    if (opt(OPT_NO_K1_SPLIT) == 1 || (d->flags & FGC_CONV_BF16)) return false;
    return k1_nodes(d) == 16 && d->cout == 32 && !pairs_ok(d);
}
// Everything fgc_conv_pack branches on when it chooses what to write into a layer's workspaces, as one number: a caller keeps
// it with the packed operands (fgc_conv_desc.packed_layout) and the FGC_CONV_PACKED calls compare.
uint64_t conv_layout_id(const fgc_conv_desc* d) {
    const uint64_t bf16 = (d->flags & FGC_CONV_BF16) ? 1 : 0;
    return 1ull | (uint64_t)narrow_supported(d) << 1 | (uint64_t)pairs_ok(d) << 2 | (uint64_t)k1_split(d) << 3 | bf16 << 4;
}
// (room for either layout wherever the shape allows the split one: the workspace a caller sized before changing NO_K1_SPLIT
//  stays large enough; the operand itself must be packed again after such a change, like every packed operand)
static size_t k1_wq_floats(const fgc_conv_desc* d) {
    const ConvGeom g1 = conv_geom(d->c0 + d->c1, d->cout);
    const int opad = (d->cout + 15) / 16 * 16;
    const size_t f32 = (size_t)g1.passes * opad * g1.kpass;
    const bool maybe = !(d->flags & FGC_CONV_BF16) && d->cout == 32;
    return maybe ? std::max(f32, (size_t)g1.passes * (d->cout >> 5) * 18 * 3 * 256) : f32;
}

// ---- which kernel computes a layer's weight gradient, and with what arguments: shared by the per-layer launch (stage 8) and
// ---- the grouped launch of fgc_conv_bwd_reduce (FGC_CONV_DEFER_DW)
enum TnVariant { TN_STREAM2 = 0, TN_STREAM4, TN_STREAM2_BF, TN_STREAM4_BF, TN_BF16_4, TN_BF16_2, TN_PLAIN_V4, TN_PLAIN, TN_NVARIANTS };
struct TnPlan {
    int variant;
    TnArgs a;
    int ntiles, nsplits;
};
static bool tn_groupable(int v) { return v <= TN_BF16_2; }
// rows x [PL columns of A] against [c0 + c1 columns of x0 | x1]
static TnPlan tn_plan_of(bool bf16, bool vec4, bool stream_ok, const void* A, int PL, const void* x0, const void* x1, int c0, int c1,
                         int shift, int rows, int rps, float* slab, int lda = 0) {
    TnPlan pl;
    const int cin = c0 + c1, ns = cdiv(rows, rps);
    pl.a = TnArgs{A, x0, x1, slab, lda ? lda : PL, PL, c0, c1, shift, rows, rps, 0};   // (lda: row stride of A, >= its PL columns)
    pl.nsplits = ns;
    if (bf16 && tn_bf16_ok(PL, c0, c1)) {
        pl.variant = cin % 64 == 0 ? TN_BF16_4 : TN_BF16_2;
        pl.ntiles = cdiv(PL, TNB_PC) * (cin % 64 == 0 ? cin / 64 : cin / 32);
    } else if (bf16 && cin <= 32 && c1 == 0) {
        pl.variant = TN_STREAM2_BF;
        pl.ntiles = cdiv(PL, 64);
    } else if (bf16) {
        pl.variant = TN_STREAM4_BF;
        pl.ntiles = cdiv(PL, 64) * cdiv(cin, 64);
    } else if (stream_ok && cin <= 32 && c1 == 0 && cin % 2 == 0) {
        pl.variant = TN_STREAM2;
        pl.ntiles = cdiv(PL, 64);
    } else if (stream_ok) {
        pl.variant = TN_STREAM4;
        pl.ntiles = cdiv(PL, 64) * cdiv(cin, 64);
    } else {
        pl.variant = vec4 ? TN_PLAIN_V4 : TN_PLAIN;
        pl.ntiles = cdiv(PL, 64) * cdiv(cin, 64);
    }
    return pl;
}
static int tn_launch_one(const TnPlan& pl, const char* tag, hipStream_t st) {
    const TnArgs& a = pl.a;
    const dim3 grid = tn_grid(pl.ntiles, pl.nsplits);
    const float *A = (const float*)a.A, *x0 = (const float*)a.x0, *x1 = (const float*)a.x1;
    const unsigned short *A16 = (const unsigned short*)a.A, *h0 = (const unsigned short*)a.x0, *h1 = (const unsigned short*)a.x1;
    switch (pl.variant) {
        case TN_BF16_4: FGC_LAUNCH(tag, st, (gemm_tn_bf16_kernel<4>), grid, dim3(TNB_THREADS), 0, A16, a.lda, a.P, h0, h1, a.c0, a.c1, a.shift, a.rows, a.rps, a.slab); break;
        case TN_BF16_2: FGC_LAUNCH(tag, st, (gemm_tn_bf16_kernel<2>), grid, dim3(TNB_THREADS), 0, A16, a.lda, a.P, h0, h1, a.c0, a.c1, a.shift, a.rows, a.rps, a.slab); break;
        case TN_STREAM2_BF: FGC_LAUNCH(tag, st, (gemm_tn_stream_kernel<2, true>), grid, dim3(256), 0, A, a.lda, a.P, x0, x1, a.c0, a.c1, a.shift, a.rows, a.rps, a.slab); break;
        case TN_STREAM4_BF: FGC_LAUNCH(tag, st, (gemm_tn_stream_kernel<4, true>), grid, dim3(256), 0, A, a.lda, a.P, x0, x1, a.c0, a.c1, a.shift, a.rows, a.rps, a.slab); break;
        case TN_STREAM2: FGC_LAUNCH(tag, st, gemm_tn_stream_kernel<2>, grid, dim3(256), 0, A, a.lda, a.P, x0, x1, a.c0, a.c1, a.shift, a.rows, a.rps, a.slab); break;
        case TN_STREAM4: FGC_LAUNCH(tag, st, gemm_tn_stream_kernel<4>, grid, dim3(256), 0, A, a.lda, a.P, x0, x1, a.c0, a.c1, a.shift, a.rows, a.rps, a.slab); break;
        case TN_PLAIN_V4: FGC_LAUNCH(tag, st, (gemm_tn_kernel<true>), grid, dim3(256), 0, A, a.lda, a.P, x0, x1, a.c0, a.c1, a.shift, a.rows, a.rps, a.slab); break;
        default: FGC_LAUNCH(tag, st, (gemm_tn_kernel<false>), grid, dim3(256), 0, A, a.lda, a.P, x0, x1, a.c0, a.c1, a.shift, a.rows, a.rps, a.slab); break;
    }
    FGC_CHECK_LAUNCH("fgc_conv_bwd/dW");
    return FGC_OK;
}
// jobs of one variant in one launch (block ranges in job order; every tn_grid is a multiple of 8 workgroups)
static int tn_launch_group(int variant, TnJobs& J, int nblocks, const char* tag, hipStream_t st) {
    if (J.njobs == 0) return FGC_OK;
    switch (variant) {
        case TN_BF16_4: FGC_LAUNCH(tag, st, (gemm_tn_bf16_group_kernel<4>), dim3(nblocks), dim3(TNB_THREADS), 0, J); break;
        case TN_BF16_2: FGC_LAUNCH(tag, st, (gemm_tn_bf16_group_kernel<2>), dim3(nblocks), dim3(TNB_THREADS), 0, J); break;
        case TN_STREAM2_BF: FGC_LAUNCH(tag, st, (gemm_tn_stream_group_kernel<2, true>), dim3(nblocks), dim3(256), 0, J); break;
        case TN_STREAM4_BF: FGC_LAUNCH(tag, st, (gemm_tn_stream_group_kernel<4, true>), dim3(nblocks), dim3(256), 0, J); break;
        case TN_STREAM2: FGC_LAUNCH(tag, st, (gemm_tn_stream_group_kernel<2>), dim3(nblocks), dim3(256), 0, J); break;
        default: FGC_LAUNCH(tag, st, (gemm_tn_stream_group_kernel<4>), dim3(nblocks), dim3(256), 0, J); break;
    }
    FGC_CHECK_LAUNCH("fgc_conv_bwd_reduce/dW");
    J.njobs = 0;
    return FGC_OK;
}

struct BwdWorkspace {
    float* Wq;        // logits operand
    float* Wpt;       // data-gradient operand
    float* db_part;   // [nb][cout]
    float* dc_part;   // [tiles][12]
    float* slab;      // gemm_tn partials of [dW0; du; dv]
    float* rtmp;      // scratch of the fixed-order reductions
    float* narrow;    // first-layer path (cin <= 8): z buffer, partial slabs (fgc_conv_narrow.hip)
    size_t bytes;
    int nb_db, rows_per_db;
    int splitW;
};

static BwdWorkspace plan_bwd(const fgc_conv_desc* d, char* base) {
    BwdWorkspace w;
    const int cin = d->c0 + d->c1;
    const ConvGeom g1 = conv_geom(cin, d->cout);
    const ConvGeom g2 = conv_geom(d->cout, cin);
    const int opad = (d->cout + 15) / 16 * 16;
    size_t off = 0;
    auto take = [&](size_t nfloats) {
        float* ptr = base ? (float*)(base + off) : nullptr;
        off += align_up(nfloats * 4, 256);
        return ptr;
    };
    w.Wq = take(k1_wq_floats(d));
    w.Wpt = take((size_t)g2.passes * g2.kpass * g2.npad);
    // pair form (fgc_conv_pair.hip): one db / dc partial per workgroup of its d-logits kernel (k1n fine nodes each), and the
    // weight-gradient GEMM reduces over the n / 4 coarse rows
    const bool pairs = pairs_ok(d);
    const int k1n = pairs ? 4 * pair_blocks_per_wg(d->cout) : k1_nodes(d);
    const int nred = pairs ? (d->n >> 2) : d->n;
    // one bias-gradient partial per d-logits tile at every size: the fused prologue of the d-logits kernel (which needs
    // exactly that) then also serves meshes beyond 131k nodes (it used to stop there: 4096 partials, ds_db launches)
    w.nb_db = cdiv(d->n, k1n);
    w.rows_per_db = cdiv(d->n, w.nb_db);
    w.nb_db = cdiv(d->n, w.rows_per_db);
    w.db_part = take((size_t)w.nb_db * d->cout);
    w.dc_part = take((size_t)cdiv(d->n, k1n) * 12);
    w.splitW = tn_splits(FGC_M * d->cout + 24, cin, nred);
    if ((d->flags & FGC_CONV_BF16) && tn_bf16_ok(FGC_M * d->cout + 24, d->c0, d->c1)) {
        // the bf16 kernel's workgroups own up to 320 x 64 of the product: one or two per CU in all
        const int target = (int)opt(OPT_TNB_WGS);   // (developer knob)
        const int tiles = cdiv(FGC_M * d->cout + 24, TNB_PC) * cdiv(cin, 64);
        w.splitW = tn_balanced_splits(target / tiles, cdiv(nred, 256), nred);
    }
    w.slab = take((size_t)w.splitW * (FGC_M * d->cout + 24) * cin);
    w.rtmp = take(reduce_tmp_floats(w.splitW, (size_t)FGC_M * d->cout * cin) + 2 * reduce_tmp_floats(w.splitW, (size_t)FGC_M * cin) +
                  reduce_tmp_floats(cdiv(d->n, k1n), 12) + reduce_tmp_floats(w.nb_db, d->cout) + 64);
    // (the first layer's scratch ends with the scratch of its fixed-order sums; the db partials of stage 1 are summed with
    //  them and need theirs behind it - it used to be missing: 64 groups x cout floats written past the workspace)
    w.narrow = narrow_supported(d) ? take(narrow_bwd_floats(d) + reduce_tmp_floats(w.nb_db, d->cout) + 64) : nullptr;
    w.bytes = off;
    return w;
}

// the weight-gradient GEMM of a layer in the fine or in the pair form (not the narrow first layer: narrow_tn_operands)
static TnPlan layer_tn_plan(const fgc_conv_desc* d, const fgc_conv_bwd_io* io, const BwdWorkspace& w) {
    const int cin = d->c0 + d->c1, cout = d->cout;
    const int PL = FGC_M * cout + 24;
    const bool bf16 = (d->flags & FGC_CONV_BF16) != 0;
    if (pairs_ok(d)) {      // K = the n / 4 coarse rows, one source
        const int nc = d->n >> 2;
        return tn_plan_of(bf16, true, true, io->r, PL, d->x0, nullptr, d->c0, 0, 0, nc, tn_rows_per_slab(nc, w.splitW), w.slab,
                          io_r_ld(io, cout, bf16));
    }
    const bool v4 = conv_vec4_ok(d) && (cout % 4 == 0) && ((uintptr_t)io->r % 16 == 0);
    const bool stream_ok = v4 && !(opt(OPT_NO_TNSTREAM) == 1);
    (void)cin;
    return tn_plan_of(bf16, v4, stream_ok, io->r, PL, d->x0, d->x1, d->c0, d->c1, d->shift, d->n, tn_rows_per_slab(d->n, w.splitW),
                      w.slab, io_r_ld(io, cout, bf16));
}

// the five fixed-order sums behind a layer's parameter gradients (slabs of the weight-gradient GEMM, db and dc partials)
static void conv_param_jobs(const fgc_conv_desc* d, const fgc_conv_bwd_io* io, const BwdWorkspace& w, RedJob* jobs) {
    const int cin = d->c0 + d->c1, cout = d->cout;
    const int P = FGC_M * cout, PL = P + 24;
    const bool pairs = pairs_ok(d);
    const int nred = pairs ? (d->n >> 2) : d->n;
    const int ns = cdiv(nred, tn_rows_per_slab(nred, w.splitW));
    const size_t sst = (size_t)PL * cin;
    jobs[0] = RedJob{w.slab, sst, ns, P * cin, cin, cin, io->dW0, w.rtmp};
    jobs[1] = RedJob{w.slab + (size_t)P * cin, sst, ns, FGC_M * cin, cin, cin, io->du};
    jobs[2] = RedJob{w.slab + (size_t)(P + 12) * cin, sst, ns, FGC_M * cin, cin, cin, io->dv};
    jobs[3] = RedJob{w.db_part, (size_t)cout, w.nb_db, cout, cout, cout, io->db};
    jobs[4] = RedJob{w.dc_part, (size_t)12, cdiv(d->n, pairs ? 4 * pair_blocks_per_wg(cout) : k1_nodes(d)), 12, 12, FGC_M, io->dc};
}

}  // namespace fgc

using namespace fgc;

extern "C" size_t fgc_conv_bwd_workspace_bytes(const fgc_conv_desc* d) {
    FGC_OPT_SCOPE(d);
    if (!d) return 0;
    return plan_bwd(d, nullptr).bytes;
}

extern "C" uint64_t fgc_conv_layout_id(const fgc_conv_desc* d) {
    FGC_OPT_SCOPE(d);
    return d ? conv_layout_id(d) : 1;
}

extern "C" int32_t fgc_conv_r_ld(int32_t cout, int32_t padded, int32_t bf16) {
    return conv_r_ld(cout, padded ? FGC_CONV_R_PAD : 0, bf16 != 0);
}

extern "C" int fgc_conv_bwd_needs_exchange(const fgc_conv_desc* d, const fgc_conv_bwd_io* io) {
    if (!d || !io) return 1;
    return (io->dx0 == nullptr && narrow_supported(d)) ? 0 : 1;
}

template <int LPN>
static int launch_logits(const CoreParams& p, const LogitParams& lp, bool vec4, size_t smem, hipStream_t st) {
    const int grid = cdiv(p.n, TILE);
    if (vec4) {
        hipFuncSetAttribute((const void*)conv_bwd_logits_kernel<LPN, true>, hipFuncAttributeMaxDynamicSharedMemorySize,
                            (int)smem);
        FGC_LAUNCH("conv_bwd_logits_kernel<LPN, true>", st, (conv_bwd_logits_kernel<LPN, true>), dim3(grid), dim3(NTHREADS), smem, p, lp);
    } else {
        hipFuncSetAttribute((const void*)conv_bwd_logits_kernel<LPN, false>, hipFuncAttributeMaxDynamicSharedMemorySize,
                            (int)smem);
        FGC_LAUNCH("conv_bwd_logits_kernel<LPN, false>", st, (conv_bwd_logits_kernel<LPN, false>), dim3(grid), dim3(NTHREADS), smem, p, lp);
    }
    FGC_CHECK_LAUNCH("fgc_conv_bwd/logits");
    return FGC_OK;
}

template <int LPN>
static int launch_data(const CoreParams& p, const DataEpilogue& ep, bool vec4, size_t smem, hipStream_t st) {
    const int grid = core_grid(p);
    if (vec4) {
        hipFuncSetAttribute((const void*)conv_bwd_data_kernel<LPN, true>, hipFuncAttributeMaxDynamicSharedMemorySize,
                            (int)smem);
        FGC_LAUNCH("conv_bwd_data_kernel<LPN, true>", st, (conv_bwd_data_kernel<LPN, true>), dim3(grid), dim3(NTHREADS), smem, p, ep);
    } else {
        hipFuncSetAttribute((const void*)conv_bwd_data_kernel<LPN, false>, hipFuncAttributeMaxDynamicSharedMemorySize,
                            (int)smem);
        FGC_LAUNCH("conv_bwd_data_kernel<LPN, false>", st, (conv_bwd_data_kernel<LPN, false>), dim3(grid), dim3(NTHREADS), smem, p, ep);
    }
    FGC_CHECK_LAUNCH("fgc_conv_bwd/data");
    return FGC_OK;
}

extern "C" int fgc_conv_bwd(const fgc_conv_desc* d, const fgc_conv_bwd_io* io, void* workspace,
                            size_t workspace_bytes, void* stream) {
    FGC_OPT_SCOPE(d);
    int rc = validate_conv_desc(d, "fgc_conv_bwd");
    if (rc) return rc;
    FGC_CHECK_ARG(io != nullptr, "fgc_conv_bwd: null io");
    FGC_CHECK_ARG(io->trowptr && io->tcol && io->tedge, "fgc_conv_bwd: transposed CSR missing");
    FGC_CHECK_ARG(io->ag && io->dy && io->ds && io->dl && io->dag && io->r, "fgc_conv_bwd: null buffer");
    FGC_CHECK_ARG(!d->act || io->y, "fgc_conv_bwd: y required when an activation was applied");
    FGC_CHECK_ARG(io->dW0 && io->db && io->du && io->dc && io->dv, "fgc_conv_bwd: null parameter-gradient pointer");
    FGC_CHECK_ARG(io->dx0 != nullptr || io->dx1 == nullptr, "fgc_conv_bwd: dx1 without dx0");
    FGC_CHECK_ARG(io_r_ld_ok(io, d->cout, (d->flags & FGC_CONV_BF16) != 0),
                  "fgc_conv_bwd: r_ld = %d is not a row stride of r for cout = %d (0, or >= %d and congruent to it modulo %d)",
                  io->r_ld, d->cout, FGC_M * d->cout + 24, (d->flags & FGC_CONV_BF16) ? 8 : 4);
    FGC_CHECK_ARG(!(io->flags & FGC_CONV_PACKED) || d->packed_layout == 0 || d->packed_layout == conv_layout_id(d),
                  "fgc_conv_bwd: FGC_CONV_PACKED, but the operands were packed in layout %llu and the options now select %llu "
                  "(an option changed between fgc_conv_pack and this call)", (unsigned long long)d->packed_layout,
                  (unsigned long long)conv_layout_id(d));
    FGC_CHECK_ARG(workspace && workspace_bytes >= fgc_conv_bwd_workspace_bytes(d) && (uintptr_t)workspace % 16 == 0,
                  "fgc_conv_bwd: workspace too small or misaligned (%zu < %zu)", workspace_bytes,
                  fgc_conv_bwd_workspace_bytes(d));
    hipStream_t st = (hipStream_t)stream;
    const int cin = d->c0 + d->c1;
    const int cout = d->cout;
    const BwdWorkspace w = plan_bwd(d, (char*)workspace);
    const ConvGeom g1 = conv_geom(cin, cout);   // gathers x (cin wide)
    const ConvGeom g2 = conv_geom(cout, cin);   // gathers s (cout wide), GEMM N = cin
    const int opad = (cout + 15) / 16 * 16;
    const int ostride = opad + 8;

    const int stages = io->stages ? io->stages : 15;
    const bool bf16 = (d->flags & FGC_CONV_BF16) != 0;
    if (pairs_ok(d)) {
        // pair form: B1 (s, db, dt, dl, da, dc) on the pair graph, then the data kernel and the weight-gradient GEMM of a
        // convolution over the n / 4 coarse rows whose gathered operand is dt (one row per pair)
        FGC_CHECK_ARG(io->tpair_rowptr && io->tpair_col && io->tpair_edge && io->dt && d->max_pair_in_deg > 0,
                      "fgc_conv_bwd: the pair form needs the transposed pair graph, dt and max_pair_in_deg");
        FGC_CHECK_ARG(!io->pool_dy, "fgc_conv_bwd: the pair form has no pooled output");
        // (a tile list names 32-row tiles of the n / 4 COARSE rows the data kernel runs over: a facet-sharded caller computes the
        //  tiles whose in-pairs are all its own while the dt / d-logit rows of the others travel)
        FGC_CHECK_ARG(io->data_tile_list == nullptr || (io->n_data_tiles >= 0 && io->n_data_tiles <= cdiv(d->n >> 2, TILE)),
                      "fgc_conv_bwd: pair form: n_data_tiles=%d outside [0, %d]", io->n_data_tiles, cdiv(d->n >> 2, TILE));
        FGC_CHECK_ARG(((uintptr_t)io->dt | (uintptr_t)io->dy | (uintptr_t)io->dl | (uintptr_t)io->dag | (uintptr_t)io->r) % 16 == 0,
                      "fgc_conv_bwd: the pair form needs 16-byte aligned buffers");
        const int nc = d->n >> 2;
        if ((stages & 4) && !(io->flags & FGC_CONV_PACKED) && bf16) {
            PackJobs J;
            J.njobs = 1;
            const size_t t2 = (size_t)g2.passes * 9 * (g2.npad >> 4) * 512;
            J.job[0] = PackJob{d->W0, w.Wpt, 5, cin, cout, cout, cin, g2.npad, g2.kc, g2.kpass, g2.passes, 0, 0};
            J.nblocks = cdiv((int)t2, 1024);
            FGC_LAUNCH("pack_many_kernel", st, pack_many_kernel, dim3(J.nblocks), dim3(256), 0, J);
            FGC_CHECK_LAUNCH("fgc_conv_bwd/pack");
        } else if ((stages & 4) && !(io->flags & FGC_CONV_PACKED)) {
            const size_t tot2 = (size_t)g2.passes * g2.kpass * g2.npad;
            FGC_LAUNCH("pack_weight_kernel", st, pack_weight_kernel, dim3(cdiv((int)tot2, 1024)), dim3(256), 0, d->W0, w.Wpt, cin, cout, cout,
                       cin, g2.npad, g2.kc, g2.kpass, g2.passes, 1);
            FGC_CHECK_LAUNCH("fgc_conv_bwd/pack");
        }
        if (stages & 3) {
            rc = launch_pair_bwd_logits(d, io, w.db_part, w.dc_part, st);
            if (rc) return rc;
        }
        if (stages & 4) {
            CoreParams p;
            fill_core_params(p, g2, nc, io->tpair_rowptr, io->tpair_col, io->tpair_edge, io->dt, nullptr, cout, 0, 0, cin, io->ag,
                             0, 12, 0, w.Wpt);
            DataEpilogue ep{io->dl, io->dag, io->r, io_r_ld(io, cout, bf16), d->u, d->v, cin, d->c0, 0, 0,
                            io->dx0, nullptr, io->accumulate0, 0};
            const size_t smem = conv_smem_bytes(g2, (size_t)TILE * 24 * 4);
            FGC_CHECK_ARG(w8_erow_supported(p, d->max_pair_in_deg) && (!bf16 || w8_bf16_supported(p, d->max_pair_in_deg)),
                          "fgc_conv_bwd: pair form: unsupported shape (cin=%d cout=%d max_pair_in_deg=%d)", cin, cout,
                          d->max_pair_in_deg);
            p.tile_list = io->data_tile_list;
            p.n_tiles = io->n_data_tiles;
            if (!p.tile_list || p.n_tiles > 0) {
                rc = launch_data_w8_erow(p, ep, smem, d->max_pair_in_deg, st, bf16);
                if (rc) return rc;
            }
        }
        if (stages & 8) {
            if (!(io->flags & FGC_CONV_DEFER_DW)) {
                rc = tn_launch_one(layer_tn_plan(d, io, w), "gemm_tn_kernel:dW", st);
                if (rc) return rc;
            }
            if (!(io->flags & FGC_CONV_DEFER_REDUCE)) {
                RedJob jobs[5];
                conv_param_jobs(d, io, w, jobs);
                rc = reduce_jobs("reduce:params", jobs, 5, nullptr, st);
                if (rc) return rc;
            }
        }
        return FGC_OK;
    }
    FGC_CHECK_ARG(io->data_tile_list == nullptr || (io->n_data_tiles >= 0 && io->n_data_tiles <= cdiv(d->n, TILE)),
                  "fgc_conv_bwd: n_data_tiles=%d outside [0, %d]", io->n_data_tiles, cdiv(d->n, TILE));
    // The deep d-logits kernel of the 32- and 64-wide layers can compute s (and the db partials) in its prologue: one
    // launch and one pass over dy / y less, when stages 1 and 2 come in the same call (a facet-sharded caller runs stage
    // 1 on its own: the halo rows of s travel under the d-logits kernel)
    const bool narrow_path = io->dx0 == nullptr && w.narrow;
    const bool deep_ok = !narrow_path && g1.lpn == 8 && d->max_deg > 0 && d->max_deg <= KMAX && conv_vec4_ok(d) &&
                         cin % 32 == 0 && (d->c1 == 0 || d->c0 % 32 == 0) && (size_t)d->n * 4 * 128 < 0xFFFFFFFFull &&
                         !(opt(OPT_NO_K1M) == 1) &&
                         !(opt(OPT_NO_K1DEEP) == 1);
    FGC_CHECK_ARG(!bf16 || narrow_path || (deep_ok && cout % 32 == 0),
                  "fgc_conv_bwd: FGC_CONV_BF16 needs widths that are multiples of 32, 16-byte aligned tensors and degrees <= %d "
                  "(cin=%d cout=%d max_deg=%d)", KMAX, cin, cout, d->max_deg);
    FGC_CHECK_ARG(!io->pool_dy || (io->pool_y && io->y && d->n % 4 == 0),
                  "fgc_conv_bwd: pool_dy needs pool_y, y and a row count that is a multiple of 4 (n=%d)", d->n);
    // (the bf16 kernel: any width it supports - 32, 64, 128 - and both degree forms keep the LDS copy)
    const bool fuse_ds = (stages & 3) == 3 && deep_ok && w.nb_db == cdiv(d->n, k1_nodes(d)) &&
                         (bf16 ? (!narrow_path && (cout == 32 || cout == 64 || cout == 128) &&
                                  !(opt(OPT_NO_FUSED_DS_BF16) == 1))
                               : ((cout == 32 || cout == 64 || (cout == 128 && k1_nodes(d) == 16 &&
                                                                   !(opt(OPT_NO_FUSED_DS128) == 1))) &&
                                  !(d->max_deg > 16 && cout > 32))) &&   // that form keeps no LDS copy of the tile (a_global)
                         ((uintptr_t)io->ds % 16) == 0 && ((uintptr_t)io->dy % 16) == 0 &&
                         (!d->act || ((uintptr_t)io->y % 16) == 0) &&
                         (!io->pool_dy || (((uintptr_t)io->y | (uintptr_t)io->pool_y | (uintptr_t)io->pool_dy) % 16) == 0) &&
                         !(opt(OPT_NO_FUSED_DS) == 1);
    // s = dy*lrelu'(y)/deg, db partials
    if ((stages & 1) && !fuse_ds && !(narrow_path && narrow_fuses_ds(d, io))) {
        int cp2 = 1;
        while (cp2 < cout) cp2 <<= 1;
        const int vw = bf16 ? 8 : 4;
        const float* yy = io->y ? io->y : io->dy;
        const bool vec = cout % vw == 0 && 256 % (cout / vw) == 0 && cout <= 256 &&
                         (((uintptr_t)io->dy | (uintptr_t)yy | (uintptr_t)io->ds) % 16) == 0 &&
                         (!io->pool_dy || (((uintptr_t)io->pool_y | (uintptr_t)io->pool_dy) % 16) == 0) &&
                         !(opt(OPT_NO_DS_VEC) == 1);
#define FGC_DS_VEC(BI, BO)                                                                                                  \
    FGC_LAUNCH("ds_db_kernel", st, (ds_db_vec_kernel<BI, BO>), dim3(w.nb_db), dim3(256), 0, io->dy, yy, d->rowptr, d->n, cout,   \
               d->act, d->alpha, d->bias_mask, w.rows_per_db, io->ds, w.db_part, io->pool_dy ? io->pool_y : nullptr, io->pool_dy)
        if (vec && bf16 && !narrow_path) FGC_DS_VEC(true, true);
        else if (vec && bf16) FGC_DS_VEC(true, false);
        else if (vec) FGC_DS_VEC(false, false);
        else
        FGC_LAUNCH("ds_db_kernel", st, ds_db_kernel, dim3(w.nb_db), dim3(256), 0, io->dy, io->y, d->rowptr, d->n, cout, cp2,
                   d->act, d->alpha, d->bias_mask, w.rows_per_db, io->ds, w.db_part, bf16 ? 1 : 0,
                   (bf16 && !narrow_path) ? 1 : 0, io->pool_dy ? io->pool_y : nullptr, io->pool_dy);
#undef FGC_DS_VEC
        FGC_CHECK_LAUNCH("fgc_conv_bwd/ds");   // db partials are summed with the other parameter gradients (stage 8)
    }
    // first layer over a narrow input (no input gradient wanted): vector-ALU path, no transposed graph, no r buffer
    if (io->dx0 == nullptr && w.narrow) {
        if (stages & 2) {
            rc = narrow_bwd_logits(d, io, w.narrow, w.db_part, st);
            if (rc) return rc;
        }
        if (stages & 8) {
            rc = narrow_bwd_params(d, io, w.narrow, w.db_part, narrow_db_partials(d, io, w.nb_db),
                                   (io->flags & FGC_CONV_DEFER_REDUCE) ? 1 : 3, nullptr, st);
            if (rc) return rc;
        }
        return FGC_OK;
    }
    // operand packing
    if ((stages & 6) && !(io->flags & FGC_CONV_PACKED) && bf16) {
        PackJobs J;
        J.njobs = 2;
        const size_t t1 = (size_t)g1.passes * (cout >> 5) * 18 * 512, t2 = (size_t)g2.passes * 9 * (g2.npad >> 4) * 512;
        J.job[0] = PackJob{d->W0, w.Wq, 6, cin, cout, 0, 0, 0, g1.kc, g1.kpass, g1.passes, opad, 0};
        J.job[1] = PackJob{d->W0, w.Wpt, 5, cin, cout, cout, cin, g2.npad, g2.kc, g2.kpass, g2.passes, 0, cdiv((int)t1, 1024)};
        J.nblocks = cdiv((int)t1, 1024) + cdiv((int)t2, 1024);
        FGC_LAUNCH("pack_many_kernel", st, pack_many_kernel, dim3(J.nblocks), dim3(256), 0, J);
        FGC_CHECK_LAUNCH("fgc_conv_bwd/pack");
    } else if ((stages & 6) && !(io->flags & FGC_CONV_PACKED)) {
        const size_t tot = (size_t)g1.passes * opad * g1.kpass;
        if (k1_split(d)) {
            PackJobs J;
            J.njobs = 1;
            J.job[0] = PackJob{d->W0, w.Wq, 17, cin, cout, 0, 0, 0, g1.kc, g1.kpass, g1.passes, opad, 0};
            J.nblocks = cdiv((int)((size_t)g1.passes * (cout >> 5) * 18 * 3 * 512), 1024);
            FGC_LAUNCH("pack_many_kernel", st, pack_many_kernel, dim3(J.nblocks), dim3(256), 0, J);
        } else
        FGC_LAUNCH("pack_logit_weight_kernel", st, pack_logit_weight_kernel, dim3(cdiv((int)tot, 1024)), dim3(256), 0, d->W0, w.Wq, cin, cout,
                           opad, g1.kc, g1.kpass, g1.passes);
        const size_t tot2 = (size_t)g2.passes * g2.kpass * g2.npad;
        FGC_LAUNCH("pack_weight_kernel", st, pack_weight_kernel, dim3(cdiv((int)tot2, 1024)), dim3(256), 0, d->W0, w.Wpt, cin, cout, cout,
                           cin, g2.npad, g2.kc, g2.kpass, g2.passes, 1);
        FGC_CHECK_LAUNCH("fgc_conv_bwd/pack");
    }
    // K1
    if (stages & 2) {
        CoreParams p;
        fill_core_params(p, g1, d->n, d->rowptr, d->col, nullptr, d->x0, d->x1, d->c0, d->c1, d->shift, cout, io->ag,
                         d->shift, 0, 12, nullptr);
        LogitParams lp{io->ds, cout, opad, ostride, w.Wq, io->dl, io->dag, w.dc_part};
        if (fuse_ds) {
            lp.dy = io->dy;
            lp.y = io->y ? io->y : io->dy;
            lp.act = d->act;
            lp.bias_mask = d->bias_mask;
            lp.alpha = d->alpha;
            lp.ds_out = io->ds;
            lp.db_part = w.db_part;
            lp.pool_y = io->pool_dy ? io->pool_y : nullptr;
            lp.pool_dy = io->pool_dy;
        }
        size_t smem = smem_core_bytes(g1.zstride) + (size_t)(TILE * ostride + 48) * 4;
        const bool vec4 = conv_vec4_ok(d);
        if (bf16) {
            const bool lng = d->max_deg > 16;
            smem = smem_core_bytes(ZSTRIDE_BF / 2, lng ? KMAX : 16) + (size_t)TILE * (cout * 2 + 32) + 48 * 4;
            if (lng) {
                hipFuncSetAttribute((const void*)conv_bwd_logits_bf16_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                    (int)smem);
                FGC_LAUNCH("conv_bwd_logits_bf16_kernel", st, (conv_bwd_logits_bf16_kernel<true>), dim3(cdiv(d->n, TILE)),
                           dim3(NTHREADS), smem, p, lp);
            } else if (k1_nodes(d) == 16) {
                constexpr int NT = 16;
                const size_t smem16 = (size_t)NT * (ZSTRIDE_BF / 2) * 4 + (size_t)NT * qnode_stride(16) * 4 + (2 * NT + 4) * 4 +
                                      (size_t)NT * (cout * 2 + 32) + 48 * 4;
                hipFuncSetAttribute((const void*)conv_bwd_logits_bf16_kernel<false, NT, 4>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                    (int)smem16);
                FGC_LAUNCH("conv_bwd_logits_bf16_kernel", st, (conv_bwd_logits_bf16_kernel<false, NT, 4>), dim3(cdiv(d->n, NT)),
                           dim3(256), smem16, p, lp);
            } else {
                hipFuncSetAttribute((const void*)conv_bwd_logits_bf16_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                    (int)smem);
                FGC_LAUNCH("conv_bwd_logits_bf16_kernel", st, (conv_bwd_logits_bf16_kernel<false>), dim3(cdiv(d->n, TILE)),
                           dim3(NTHREADS), smem, p, lp);
            }
            FGC_CHECK_LAUNCH("fgc_conv_bwd/logits_bf16");
            rc = 0;
        } else if (g1.lpn == 8 && d->max_deg > 0 && d->max_deg <= KMAX &&
            !(opt(OPT_NO_K1M) == 1)) {
            static bool attr = false;
            if (!attr) {
                hipFuncSetAttribute((const void*)conv_bwd_logits_mfma_kernel<true>,
                                    hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
                hipFuncSetAttribute((const void*)conv_bwd_logits_mfma_kernel<false>,
                                    hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
                attr = true;
            }
            const bool deep = vec4 && cin % 32 == 0 && (d->c1 == 0 || d->c0 % 32 == 0) &&
                              (size_t)d->n * 4 * 128 < 0xFFFFFFFFull &&
                              !(opt(OPT_NO_K1DEEP) == 1);
            if (deep) {
                // (__syncthreads_or owns 256 B of static LDS: ask for exactly what this launch needs)
#define FGC_DEEP_LAUNCH(LONG_, OKG_)                                                                                  \
    do {                                                                                                             \
        hipFuncSetAttribute((const void*)conv_bwd_logits_deep_kernel<LONG_, OKG_>,                                   \
                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);                                  \
        FGC_LAUNCH("conv_bwd_logits_deep_kernel", st, (conv_bwd_logits_deep_kernel<LONG_, OKG_>),                    \
                   dim3(cdiv(d->n, TILE)), dim3(NTHREADS), smem, p, lp);                                             \
    } while (0)
                const bool lng = d->max_deg > 16;
                if (!lng) smem = smem_core_bytes(g1.zstride, 16) + (size_t)(TILE * ostride + 48) * 4;
                if (k1_nodes(d) == 16) {      // half tiles (implies !lng)
                    constexpr int NT = 16;
                    const size_t smem16 = (size_t)NT * g1.zstride * 4 + (size_t)NT * qnode_stride(16) * 4 + (2 * NT + 4) * 4 +
                                          (size_t)(NT * ostride + 48) * 4;
#define FGC_DEEP_HALF(OKG_)                                                                                          \
    do {                                                                                                             \
        hipFuncSetAttribute((const void*)conv_bwd_logits_deep_kernel<false, OKG_, NT, 4>,                            \
                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem16);                                \
        FGC_LAUNCH("conv_bwd_logits_deep_kernel", st, (conv_bwd_logits_deep_kernel<false, OKG_, NT, 4>),             \
                   dim3(cdiv(d->n, NT)), dim3(256), smem16, p, lp);                                                  \
    } while (0)
                    const bool al = ((uintptr_t)io->ds % 16) == 0;
#define FGC_DEEP_HALF_SPLIT(OKG_)                                                                                    \
    do {                                                                                                             \
        hipFuncSetAttribute((const void*)conv_bwd_logits_deep_kernel<false, OKG_, NT, 4, true>,                      \
                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem16);                                \
        FGC_LAUNCH("conv_bwd_logits_deep_kernel", st, (conv_bwd_logits_deep_kernel<false, OKG_, NT, 4, true>),       \
                   dim3(cdiv(d->n, NT)), dim3(256), smem16, p, lp);                                                  \
    } while (0)
                    // 14-slot table (option K1_QS14, degrees <= 14): a fifth workgroup per CU
                    const bool qs14 = opt(OPT_K1_QS14) == 1 && d->max_deg > 0 && d->max_deg <= 14;
                    const size_t smem14 = smem16 - (size_t)NT * (qnode_stride(16) - qnode_stride(14)) * 4;
#define FGC_DEEP_HALF14(OKG_, SPLIT_)                                                                                \
    do {                                                                                                             \
        hipFuncSetAttribute((const void*)conv_bwd_logits_deep_kernel<false, OKG_, NT, 4, SPLIT_, 14>,                \
                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem14);                                \
        FGC_LAUNCH("conv_bwd_logits_deep_kernel", st, (conv_bwd_logits_deep_kernel<false, OKG_, NT, 4, SPLIT_, 14>), \
                   dim3(cdiv(d->n, NT)), dim3(256), smem14, p, lp);                                                  \
    } while (0)
                    if (k1_split(d)) {
                        // (the packed operand is in the split layout whatever the pointers: no other kernel can take it)
                        FGC_CHECK_ARG(al, "fgc_conv_bwd: ds must be 16-byte aligned (cout=%d)", cout);
                        if (qs14) FGC_DEEP_HALF14(2, true);
                        else FGC_DEEP_HALF_SPLIT(2);
                    } else
                    if (cout == 32 && al) FGC_DEEP_HALF(2);
                    else if (cout == 64 && al && qs14) FGC_DEEP_HALF14(4, false);
                    else if (cout == 64 && al) FGC_DEEP_HALF(4);
                    else if (cout == 128 && al && fuse_ds) FGC_DEEP_HALF(8);   // (only for its prologue: s and db in this launch)
                    else FGC_DEEP_HALF(0);
#undef FGC_DEEP_HALF
#undef FGC_DEEP_HALF_SPLIT
                } else {
                // 24 edge slots + the ds tile of a 64- or 128-wide layer do not fit twice into a CU's LDS
                if (lng && cout > 32 && cout % 16 == 0 && ((uintptr_t)io->ds % 16) == 0) {
                    lp.a_global = 1;
                    smem = smem_core_bytes(g1.zstride) + (size_t)48 * 4;
                }
                const bool al16 = ((uintptr_t)io->ds % 16) == 0;
                if (cout == 32 && al16) { if (lng) FGC_DEEP_LAUNCH(true, 2); else FGC_DEEP_LAUNCH(false, 2); }
                else if (cout == 64 && al16) { if (lng) FGC_DEEP_LAUNCH(true, 4); else FGC_DEEP_LAUNCH(false, 4); }
                else { if (lng) FGC_DEEP_LAUNCH(true, 0); else FGC_DEEP_LAUNCH(false, 0); }
                }
#undef FGC_DEEP_LAUNCH
            } else if (vec4)
                FGC_LAUNCH("conv_bwd_logits_mfma_kernel", st, (conv_bwd_logits_mfma_kernel<true>),
                           dim3(cdiv(d->n, TILE)), dim3(NTHREADS), smem, p, lp);
            else
                FGC_LAUNCH("conv_bwd_logits_mfma_kernel", st, (conv_bwd_logits_mfma_kernel<false>),
                           dim3(cdiv(d->n, TILE)), dim3(NTHREADS), smem, p, lp);
            FGC_CHECK_LAUNCH("fgc_conv_bwd/logits_mfma");
            rc = 0;
        } else
            rc = launch_logits<8>(p, lp, vec4, smem, st);
        if (rc) return rc;   // dc partials: stage 8
    }
    // K2
    if ((stages & 4) && !(io->data_tile_list && io->n_data_tiles == 0)) {
        CoreParams p;
        fill_core_params(p, g2, d->n, io->trowptr, io->tcol, io->tedge, io->ds, nullptr, cout, 0, 0, cin, io->ag,
                         d->shift, 12, 0, w.Wpt);
        p.tile_list = io->data_tile_list;
        p.n_tiles = io->n_data_tiles;
        DataEpilogue ep{io->dl, io->dag, io->r, io_r_ld(io, cout, bf16), d->u, d->v, cin, d->c0, d->c1, d->shift,
                        io->dx0, io->dx1, io->accumulate0, io->accumulate1};
        const size_t smem = conv_smem_bytes(g2, (size_t)TILE * 24 * 4);
        const bool vec4 = (cout % 4 == 0) && ((uintptr_t)io->ds % 16 == 0) && ((uintptr_t)io->r % 16 == 0);
        if (bf16) {
            FGC_CHECK_ARG(w8_bf16_supported(p, io->max_in_deg), "fgc_conv_bwd: FGC_CONV_BF16: unsupported shape for the data "
                          "gradient (cin=%d cout=%d max_in_deg=%d)", cin, cout, io->max_in_deg);
            rc = launch_data_w8(p, ep, smem, io->max_in_deg, st, true);
            if (rc) return rc;
        } else if (g2.lpn == 8 && w8_supported(p, io->max_in_deg)) {
            rc = launch_data_w8(p, ep, smem, io->max_in_deg, st);
            if (rc) return rc;
        } else
            rc = launch_data<8>(p, ep, vec4, smem, st);
        if (rc) return rc;
    }
    // K3: dW0 = r^T x ; [du; dv] = dag^T x
    if (stages & 8) {
        // one GEMM over the rows of r = [9*cout aggregate columns | da | dg]: rows 0..P-1 of the product are dW0^T
        // blocks, rows P..P+8 du, rows P+12..P+20 dv (layer_tn_plan); FGC_CONV_DEFER_DW: launched by fgc_conv_bwd_reduce
        if (!(io->flags & FGC_CONV_DEFER_DW)) {
            rc = tn_launch_one(layer_tn_plan(d, io, w), "gemm_tn_kernel:dW", st);
            if (rc) return rc;
        }
        // every parameter gradient of the layer in two launches (fixed summation order).  The db / dc partials were
        // left in the workspace by stages 1 and 2: a staged caller keeps the workspace untouched between its calls;
        // with FGC_CONV_DEFER_REDUCE also until fgc_conv_bwd_reduce sums the layers of the whole network at once.
        if (!(io->flags & FGC_CONV_DEFER_REDUCE)) {
            RedJob jobs[5];
            conv_param_jobs(d, io, w, jobs);
            rc = reduce_jobs("reduce:params", jobs, 5, nullptr, st);
            if (rc) return rc;
        }
    }
    return FGC_OK;
}


// ---------------------------------------------------------------------------------------------
// whole-network helpers: one launch where every layer used to bring its own
// ---------------------------------------------------------------------------------------------
extern "C" int fgc_conv_pack(const fgc_conv_desc* const* descs, const fgc_conv_bwd_io* const* ios, void* const* fwd_ws,
                             void* const* bwd_ws, int32_t count, const fgc_pack_extra* extra, void* stream) {
    FGC_CHECK_ARG((descs || count == 0) && count >= 0, "fgc_conv_pack: bad arguments");
    hipStream_t st = (hipStream_t)stream;
    PackJobs J;
    J.njobs = 0;
    J.nblocks = 0;
    auto flush = [&]() {
        if (J.njobs == 0) return;
        FGC_LAUNCH("pack_many_kernel", st, pack_many_kernel, dim3(J.nblocks), dim3(256), 0, J);
        J.njobs = 0;
        J.nblocks = 0;
    };
    auto add = [&](const PackJob& j, size_t total) {
        if (J.njobs == PACK_MAX_JOBS) flush();
        PackJob& q = J.job[J.njobs++];
        q = j;
        q.block0 = J.nblocks;
        J.nblocks += cdiv((int)total, 1024);
    };
    for (int i = 0; i < count; ++i) {
        const fgc_conv_desc* d = descs[i];
        FGC_OPT_SCOPE(d);      // (this layer's own option values, if its descriptor carries any)
        int rc = validate_conv_desc(d, "fgc_conv_pack");
        if (rc) return rc;
        const int cin = d->c0 + d->c1, cout = d->cout;
        const bool narrow = narrow_supported(d);
        const bool bf16 = (d->flags & FGC_CONV_BF16) != 0;
        const bool pairs = pairs_ok(d);     // reads W0 / u / v in place (bf16 storage: a bf16 copy of W0); backward packs only
                                            // the data-gradient operand
        if (fwd_ws && fwd_ws[i] && pairs && bf16)
            add(PackJob{d->W0, (float*)fwd_ws[i], 7, cin, cout, FGC_M * cout * cin, 0, 0, 0, 0, 0, 0, 0}, (size_t)FGC_M * cout * cin);
        if (fwd_ws && fwd_ws[i] && !narrow && !pairs) {
            const ConvGeom g = conv_geom(cin, cout);
            FGC_CHECK_ARG((uintptr_t)fwd_ws[i] % 16 == 0, "fgc_conv_pack: workspace %d misaligned", i);
            add(PackJob{d->W0, (float*)fwd_ws[i], bf16 ? 4 : 0, cin, cout, cin, cout, g.npad, g.kc, g.kpass, g.passes, 0, 0},
                (size_t)g.passes * g.kpass * g.npad);
        }
        const bool narrow_bwd = narrow && ios && ios[i] && ios[i]->dx0 == nullptr;   // vector-ALU path: nothing to pack
        if (bwd_ws && bwd_ws[i] && !narrow_bwd) {
            FGC_CHECK_ARG((uintptr_t)bwd_ws[i] % 16 == 0, "fgc_conv_pack: workspace %d misaligned", i);
            const BwdWorkspace w = plan_bwd(d, (char*)bwd_ws[i]);
            const ConvGeom g1 = conv_geom(cin, cout), g2 = conv_geom(cout, cin);
            const int opad = (cout + 15) / 16 * 16;
            if (bf16) {
                if (!pairs)
                add(PackJob{d->W0, w.Wq, 6, cin, cout, 0, 0, 0, g1.kc, g1.kpass, g1.passes, opad, 0},
                    (size_t)g1.passes * (cout >> 5) * 18 * 512);
                add(PackJob{d->W0, w.Wpt, 5, cin, cout, cout, cin, g2.npad, g2.kc, g2.kpass, g2.passes, 0, 0},
                    (size_t)g2.passes * 9 * (g2.npad >> 4) * 512);
            } else {
            if (!pairs && k1_split(d))
            add(PackJob{d->W0, w.Wq, 17, cin, cout, 0, 0, 0, g1.kc, g1.kpass, g1.passes, opad, 0},
                (size_t)g1.passes * (cout >> 5) * 18 * 3 * 512);
            else if (!pairs)
            add(PackJob{d->W0, w.Wq, 2, cin, cout, 0, 0, 0, g1.kc, g1.kpass, g1.passes, opad, 0},
                (size_t)g1.passes * opad * g1.kpass);
            add(PackJob{d->W0, w.Wpt, 1, cin, cout, cout, cin, g2.npad, g2.kc, g2.kpass, g2.passes, 0, 0},
                (size_t)g2.passes * g2.kpass * g2.npad);
            }
        }
    }
    if (extra && extra->rot_x) {
        const int64_t nvec = (int64_t)extra->rot_rows * extra->rot_vecs;
        FGC_CHECK_ARG(extra->rot_y && extra->rot_R && extra->rot_rows > 0 && extra->rot_vecs > 0 && nvec < (1ll << 31),
                      "fgc_conv_pack: extra: bad rotation (rows=%lld vecs=%d)", (long long)extra->rot_rows, extra->rot_vecs);
        if (extra->rot_ag) {
            FGC_CHECK_ARG(extra->rot_u && extra->rot_c && extra->rot_v && extra->rot_vecs <= 2 && (uintptr_t)extra->rot_ag % 16 == 0,
                          "fgc_conv_pack: extra: the first layer's logit table needs u, c, v and at most 6 input channels");
            PackJob j{extra->rot_x, extra->rot_y, 14, extra->rot_vecs, 0, (int)extra->rot_rows, 0, 0, 0, 0, 0, 0, 0, extra->rot_R,
                      extra->rot_u, extra->rot_c, extra->rot_v, extra->rot_ag};
            add(j, (size_t)extra->rot_rows * 4);     // one row per thread
        } else {
            PackJob j{extra->rot_x, extra->rot_y, 8, 0, 0, (int)nvec, 0, 0, 0, 0, 0, 0, 0, extra->rot_R};
            add(j, (size_t)nvec * 2);     // two 3-vectors per thread-iteration share
        }
    }
    if (extra && extra->mlp_W1) {
        PackJob mj[4];
        size_t tot[4];
        const int nj = extra->mlp_bf16 ? mlp_pack_jobs_bf16(extra, mj, tot) : mlp_pack_jobs_f32(extra, mj, tot);
        FGC_CHECK_ARG(nj >= 0, "fgc_conv_pack: extra: MLP shape cin=%d hidden=%d cout=%d n=%d not served%s", extra->mlp_cin,
                      extra->mlp_hidden, extra->mlp_cout, extra->mlp_n, extra->mlp_bf16 ? " (bf16)" : "");
        for (int i = 0; i < nj; ++i) add(mj[i], tot[i]);
    }
    flush();
    FGC_CHECK_LAUNCH("fgc_conv_pack");
    return FGC_OK;
}

extern "C" int fgc_conv_bwd_reduce(const fgc_conv_desc* const* descs, const fgc_conv_bwd_io* const* ios,
                                   void* const* bwd_ws, int32_t count, void* stream) {
    FGC_CHECK_ARG(descs && ios && bwd_ws && count >= 0, "fgc_conv_bwd_reduce: bad arguments");
    hipStream_t st = (hipStream_t)stream;
    RedJob jobs[RED_MAX_JOBS];
    int nj = 0, rc = 0;
    auto flush = [&]() {
        const int r = nj ? reduce_jobs("reduce:params", jobs, nj, nullptr, st) : FGC_OK;
        nj = 0;
        return r;
    };
    // the weight-gradient GEMMs of the layers that deferred them (FGC_CONV_DEFER_DW): one launch per kernel form
    {
        TnJobs G[TN_NVARIANTS];
        int nb[TN_NVARIANTS];
        for (int v = 0; v < TN_NVARIANTS; ++v) G[v].njobs = 0, nb[v] = 0;
        for (int i = 0; i < count; ++i) {
            const fgc_conv_desc* d = descs[i];
            const fgc_conv_bwd_io* io = ios[i];
            if (!io || !(io->flags & FGC_CONV_DEFER_DW)) continue;
            FGC_OPT_SCOPE(d);
            rc = validate_conv_desc(d, "fgc_conv_bwd_reduce");
            if (rc) return rc;
            FGC_CHECK_ARG(bwd_ws[i], "fgc_conv_bwd_reduce: layer %d: null workspace", i);
            const BwdWorkspace w = plan_bwd(d, (char*)bwd_ws[i]);
            TnPlan pl;
            if (io->dx0 == nullptr && w.narrow) {
                const float* A;
                float* slab;
                int zld, rps;
                narrow_tn_operands(d, io, w.narrow, &A, &zld, &slab, &rps);
                pl = tn_plan_of(false, true, true, A, zld, io->ds, nullptr, d->cout, 0, 0, d->n, rps, slab);
            } else {
                FGC_CHECK_ARG(io->r, "fgc_conv_bwd_reduce: layer %d: FGC_CONV_DEFER_DW without r", i);
                FGC_CHECK_ARG(io_r_ld_ok(io, d->cout, (d->flags & FGC_CONV_BF16) != 0),
                              "fgc_conv_bwd_reduce: layer %d: r_ld = %d is not a row stride of r for cout = %d", i, io->r_ld, d->cout);
                pl = layer_tn_plan(d, io, w);
            }
            if (!tn_groupable(pl.variant)) {
                rc = tn_launch_one(pl, "gemm_tn_kernel:dW", st);
                if (rc) return rc;
                continue;
            }
            TnJobs& J = G[pl.variant];
            if (J.njobs == TN_MAX_JOBS) {
                rc = tn_launch_group(pl.variant, J, nb[pl.variant], "gemm_tn_kernel:dW", st);
                if (rc) return rc;
                nb[pl.variant] = 0;
            }
            pl.a.block0 = nb[pl.variant];
            J.job[J.njobs++] = pl.a;
            nb[pl.variant] += (int)tn_grid(pl.ntiles, pl.nsplits).x;
        }
        for (int v = 0; v < TN_NVARIANTS; ++v) {
            rc = tn_launch_group(v, G[v], nb[v], "gemm_tn_kernel:dW", st);
            if (rc) return rc;
        }
    }
    for (int i = 0; i < count; ++i) {
        const fgc_conv_desc* d = descs[i];
        const fgc_conv_bwd_io* io = ios[i];
        FGC_OPT_SCOPE(d);
        rc = validate_conv_desc(d, "fgc_conv_bwd_reduce");
        if (rc) return rc;
        FGC_CHECK_ARG(io && bwd_ws[i] && io->dW0 && io->db && io->du && io->dc && io->dv,
                      "fgc_conv_bwd_reduce: layer %d: null io / workspace / gradient pointer", i);
        const BwdWorkspace w = plan_bwd(d, (char*)bwd_ws[i]);
        if (nj + 5 > RED_MAX_JOBS && (rc = flush())) return rc;
        if (io->dx0 == nullptr && w.narrow) {
            rc = narrow_bwd_params(d, io, w.narrow, w.db_part, narrow_db_partials(d, io, w.nb_db), 0, jobs + nj, st);
            if (rc) return rc;
            nj += NARROW_RED_JOBS;
        } else {
            conv_param_jobs(d, io, w, jobs + nj);
            nj += 5;
        }
    }
    if ((rc = flush())) return rc;
    return FGC_OK;
}
