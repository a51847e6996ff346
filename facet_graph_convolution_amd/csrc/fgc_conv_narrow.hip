// Graph convolution over NARROW inputs (cin <= 8, one source, no upsampling): the network's first layer (6 -> 32).
//
// The tiled kernels pad every gathered row to 32 channels so that the weight transform fills the matrix cores; with
// 6 input channels that is 5x wasted MFMA and gather work (conv1 used 13 % of the training step for 1 % of its FLOPs).
// Here the arithmetic is small enough for the vector ALU and the problem is purely a gather:
//   forward   one NODE PER LANE: the lane walks its edges (4 at a time: neighbour ids, then their logit and feature
//             rows, all requested together), keeps z_i[m][c] = sum_k q_ikm x_j[c] (9*cin registers) and applies the
//             weights from scalar registers (the weight index is wave-uniform, hipcc turns it into s_load + v_fma);
//             outputs leave through an LDS tile so that rows are written coalesced (+ activation, 4:1 max-pool).
//   backward  (first layer only: no input gradient wanted)  same walk, per lane: dz_i = W^T s_i, per edge the softmax
//             backward, da_i, and the outer products that the tiled path gets from two extra kernels and a GEMM:
//               du = sum_i da_i (x) x_i     dv = sum_i sum_k dl_ik (x) x_j(i,k)     dc = sum_i da_i
//             block-reduced in a fixed order and written as slabs.  dW0 = sum_i s_i (x) z_i is the only dense
//             contraction left: z is written out by the forward kernel in "z only" mode and contracted with s by the
//             streaming TN GEMM.  No transposed graph, no per-edge buffer, no r buffer.
// Reference: custom_conv2d, model.py:427-504 (forward); its gradient as derived in SURVEY.md Appendix A.
#include <stdlib.h>

#include "fgc_conv_narrow.h"
#include <type_traits>

#include "fgc_reduce.h"

namespace fgc {

typedef float f32x2n __attribute__((ext_vector_type(2)));
constexpr int NB = 256;          // nodes per workgroup (= 8 tiles of 32)
constexpr int EB_FWD = 8;        // edges requested together (forward; 4 -> 8: two dependent batches per node instead of four, -1.5 us)
constexpr int EB_BWD = 2;        // (backward: twice the per-lane state, half the batch)

struct NarrowFwd {
    int n;
    const int* rowptr;
    const int* col;
    const float* x;        // [rows, cin]
    const float* ag;       // [rows, 24]
    const float* W0;       // [9, cout, cin]
    const float* bias;
    int cin, cout, bias_mask, act;
    float alpha;
    float* y;
    float* y_pool;
    float* z;              // ZONLY: [n, zld], z[i][m * CIN + c]
    int zld;
    const int* tile_list;  // 32-row tiles (NULL = all); a workgroup takes 8 consecutive entries
    int n_tiles;
    int out_bf16;          // y / y_pool are bf16 tensors
};

__device__ __forceinline__ int narrow_node(const int* tile_list, int n_tiles, int n, int& active) {
    const int t = blockIdx.x * (NB / TILE) + (threadIdx.x >> 5);
    if (tile_list) {
        active = t < n_tiles;
        const int i = (active ? tile_list[t] : 0) * TILE + (threadIdx.x & 31);
        active = active && i < n;
        return i;
    }
    const int i = t * TILE + (threadIdx.x & 31);
    active = i < n;
    return i;
}

// softmax over the 9 logits a + g
__device__ __forceinline__ void softmax9(const float (&a)[FGC_M], const float (&g)[FGC_M], float (&q)[FGC_M]) {
    float mx = a[0] + g[0];
#pragma unroll
    for (int m = 0; m < FGC_M; ++m) {
        q[m] = a[m] + g[m];
        mx = fmaxf(mx, q[m]);
    }
    float sum = 0.f;
#pragma unroll
    for (int m = 0; m < FGC_M; ++m) {
        q[m] = __expf(q[m] - mx);
        sum += q[m];
    }
    const float inv = 1.0f / sum;
#pragma unroll
    for (int m = 0; m < FGC_M; ++m) q[m] *= inv;
}

// rows of EB edges of this lane's node: ids first, then every logit / feature row, so that the two memory round
// trips of an edge are shared by EB edges.  Slots past the node's last edge repeat it (their weight is dropped by the
// caller), so every load is unconditional.
template <int CIN, int EB>
__device__ __forceinline__ void fetch_edges(const int* __restrict__ col, const float* __restrict__ ag,
                                            const float* __restrict__ x, int cin, int e, int e1,
                                            float (&g)[EB][FGC_M], float (&xj)[EB][CIN]) {
    int j[EB];
#pragma unroll
    for (int t = 0; t < EB; ++t) j[t] = col[min(e + t, e1 - 1)];
#pragma unroll
    for (int t = 0; t < EB; ++t) {
        const float* gr = ag + (size_t)j[t] * FGC_AG_LD + 12;
        const f32x4 g0 = *reinterpret_cast<const f32x4*>(gr);
        const f32x4 g1 = *reinterpret_cast<const f32x4*>(gr + 4);
        g[t][0] = g0[0]; g[t][1] = g0[1]; g[t][2] = g0[2]; g[t][3] = g0[3];
        g[t][4] = g1[0]; g[t][5] = g1[1]; g[t][6] = g1[2]; g[t][7] = g1[3];
        g[t][8] = gr[8];
        const float* xr = x + (size_t)j[t] * CIN;
        if (CIN % 2 == 0) {   // rows are 8-byte aligned when the width is even
#pragma unroll
            for (int c = 0; c < CIN; c += 2) {
                const f32x2n v = *reinterpret_cast<const f32x2n*>(xr + c);
                xj[t][c] = v[0];
                xj[t][c + 1] = v[1];
            }
        } else {
#pragma unroll
            for (int c = 0; c < CIN; ++c) xj[t][c] = xr[c];
        }
    }
}

template <int CIN, bool ZONLY>
__global__ __launch_bounds__(NB) void conv_narrow_fwd_kernel(NarrowFwd p) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    float* yt = reinterpret_cast<float*>(smem_raw);   // [NB][cout + 1]
    int active;
    const int i = narrow_node(p.tile_list, p.n_tiles, p.n, active);
    int e0 = 0, e1 = 0;
    float a[FGC_M];
#pragma unroll
    for (int m = 0; m < FGC_M; ++m) a[m] = 0.f;
    if (active) {
        e0 = p.rowptr[i];
        e1 = p.rowptr[i + 1];
        const float* ar = p.ag + (size_t)i * FGC_AG_LD;
        const f32x4 a0 = *reinterpret_cast<const f32x4*>(ar);
        const f32x4 a1 = *reinterpret_cast<const f32x4*>(ar + 4);
        a[0] = a0[0]; a[1] = a0[1]; a[2] = a0[2]; a[3] = a0[3];
        a[4] = a1[0]; a[5] = a1[1]; a[6] = a1[2]; a[7] = a1[3];
        a[8] = ar[8];
    }
    float z[FGC_M][CIN];
#pragma unroll
    for (int m = 0; m < FGC_M; ++m)
#pragma unroll
        for (int c = 0; c < CIN; ++c) z[m][c] = 0.f;
    for (int e = e0; e < e1; e += EB_FWD) {
        float g[EB_FWD][FGC_M], xj[EB_FWD][CIN];
        fetch_edges<CIN, EB_FWD>(p.col, p.ag, p.x, p.cin, e, e1, g, xj);
#pragma unroll
        for (int t = 0; t < EB_FWD; ++t) {
            float q[FGC_M];
            softmax9(a, g[t], q);
            const float w = e + t < e1 ? 1.f : 0.f;
#pragma unroll
            for (int m = 0; m < FGC_M; ++m) {
                const float qm = q[m] * w;
#pragma unroll
                for (int c = 0; c < CIN; ++c) z[m][c] = fmaf(qm, xj[t][c], z[m][c]);
            }
        }
    }
    auto store_z = [&]() {     // the aggregates themselves: operand of the backward pass' weight-gradient GEMM
        if (active) {
            float* zr = p.z + (size_t)i * p.zld;
#pragma unroll
            for (int m = 0; m < FGC_M; ++m)
#pragma unroll
                for (int c = 0; c < CIN; ++c) zr[m * CIN + c] = z[m][c];
            for (int k = FGC_M * CIN; k < p.zld; ++k) zr[k] = 0.f;
        }
    };
    if (ZONLY) {
        store_z();
        return;
    }
    // y_i = (1/d) W~ z_i + b [d > 0]; the weight address depends on loop counters only (scalar loads)
    const int d = e1 - e0;
    const float inv = d > 0 ? 1.0f / (float)d : 0.f;
    const int ys = p.cout + 1;
    for (int o = 0; o < p.cout; ++o) {
        float acc = 0.f;
#pragma unroll
        for (int m = 0; m < FGC_M; ++m) {
            const float* w = p.W0 + ((size_t)m * p.cout + o) * CIN;
#pragma unroll
            for (int c = 0; c < CIN; ++c) acc = fmaf(w[c], z[m][c], acc);
        }
        float val = acc * inv;
        if (!p.bias_mask || d > 0) val += p.bias[o];
        if (p.act) val = fmaxf(val, 0.f) - p.alpha * fmaxf(-val, 0.f);
        yt[threadIdx.x * ys + o] = val;
    }
    // (stored only now: a global store ahead of the loop above would make the compiler treat W0 as possibly clobbered
    // and load the wave-uniform weights through the vector memory path instead of scalar loads)
    if (p.z != nullptr) store_z();
    __syncthreads();
    // coalesced row writes, one 32-row tile (= 32 * cout contiguous floats) at a time; 4:1 max-pool from the tile
    for (int tl = 0; tl < NB / TILE; ++tl) {
        const int t = blockIdx.x * (NB / TILE) + tl;
        if (p.tile_list ? t >= p.n_tiles : t * TILE >= p.n) break;
        const int row0 = (p.tile_list ? p.tile_list[t] : t) * TILE;
        for (int k = threadIdx.x; k < TILE * p.cout; k += NB) {
            const int r = k / p.cout, o = k % p.cout;
            if (row0 + r < p.n) st_act(p.y, (size_t)(row0 + r) * p.cout + o, yt[(tl * TILE + r) * ys + o], p.out_bf16);
        }
        if (p.y_pool) {
            for (int k = threadIdx.x; k < (TILE / 4) * p.cout; k += NB) {
                const int pr = k / p.cout, o = k % p.cout;
                float mx = -INFINITY;
                bool any = false;
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int r = pr * 4 + q;
                    if (row0 + r < p.n) {
                        mx = fmaxf(mx, yt[(tl * TILE + r) * ys + o]);
                        any = true;
                    }
                }
                if (any) st_act(p.y_pool, (size_t)((row0 >> 2) + pr) * p.cout + o, mx, p.out_bf16);
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------
// The same forward pass with the per-node product y_i = W~ z_i on the matrix cores (cout a multiple of 16): every lane
// parks the aggregates of its node in LDS ([node][k], k = m * CIN + c, zero padded to a multiple of 16), from where
// (a) they are copied to global memory as whole rows for the backward pass (FGC_CONV_SAVE_Z; the per-lane stores of the
// vector form write 4 bytes per lane with a 224-byte stride), (b) they are the A operand of [64 nodes x K] x [K x cout]
// per wave, B = the weights re-laid as [k][o] in LDS once per workgroup.  The C layout puts four consecutive nodes of one
// output channel in one lane: bias, activation, the row stores (64-byte segments) and the 4:1 max-pool need no second
// trip through LDS.
// ---------------------------------------------------------------------------------------------
template <int CIN, int OT>
__global__ __launch_bounds__(NB) void conv_narrow_fwd_mma_kernel(NarrowFwd p) {
    constexpr int K9 = FGC_M * CIN;
    constexpr int KZ = (K9 + 15) / 16 * 16;     // padded k extent
    constexpr int ZS = KZ + 4;                  // LDS row stride of the aggregates (== 4 mod 8)
    constexpr int COUT = OT * 16;
    constexpr int WS = COUT + 4;                // row stride of the weights (== 4 mod 8)
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    float* zt = reinterpret_cast<float*>(smem_raw);            // [NB][ZS]
    float* Wk = zt + NB * ZS;                                   // [KZ][WS]
    int* nodes = reinterpret_cast<int*>(Wk + KZ * WS);          // [NB] node of every row, -1 = none
    int* degs = nodes + NB;                                     // [NB]
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, lr = lane & 15, lq = lane >> 4;
    int active;
    const int i = narrow_node(p.tile_list, p.n_tiles, p.n, active);
    // weights as [k][o] (zero rows past 9 * CIN)
    for (int t = tid; t < KZ * COUT; t += NB) {
        const int k = t / COUT, o = t % COUT;
        Wk[k * WS + o] = k < K9 ? p.W0[((size_t)(k / CIN) * COUT + o) * CIN + k % CIN] : 0.f;
    }
    int e0 = 0, e1 = 0;
    float a[FGC_M];
#pragma unroll
    for (int m = 0; m < FGC_M; ++m) a[m] = 0.f;
    if (active) {
        e0 = p.rowptr[i];
        e1 = p.rowptr[i + 1];
        const float* ar = p.ag + (size_t)i * FGC_AG_LD;
        const f32x4 a0 = *reinterpret_cast<const f32x4*>(ar);
        const f32x4 a1 = *reinterpret_cast<const f32x4*>(ar + 4);
        a[0] = a0[0]; a[1] = a0[1]; a[2] = a0[2]; a[3] = a0[3];
        a[4] = a1[0]; a[5] = a1[1]; a[6] = a1[2]; a[7] = a1[3];
        a[8] = ar[8];
    }
    float z[KZ];      // flat [m * CIN + c], zero tail
#pragma unroll
    for (int k = 0; k < KZ; ++k) z[k] = 0.f;
    for (int e = e0; e < e1; e += EB_FWD) {
        float g[EB_FWD][FGC_M], xj[EB_FWD][CIN];
        fetch_edges<CIN, EB_FWD>(p.col, p.ag, p.x, p.cin, e, e1, g, xj);
#pragma unroll
        for (int t = 0; t < EB_FWD; ++t) {
            float q[FGC_M];
            softmax9(a, g[t], q);
            const float w = e + t < e1 ? 1.f : 0.f;
#pragma unroll
            for (int m = 0; m < FGC_M; ++m) {
                const float qm = q[m] * w;
#pragma unroll
                for (int c = 0; c < CIN; ++c) z[m * CIN + c] = fmaf(qm, xj[t][c], z[m * CIN + c]);
            }
        }
    }
    {
        float* zr = zt + tid * ZS;
#pragma unroll
        for (int k = 0; k < KZ; k += 4) *reinterpret_cast<f32x4*>(zr + k) = f32x4{z[k], z[k + 1], z[k + 2], z[k + 3]};
        nodes[tid] = active ? i : -1;
        degs[tid] = e1 - e0;
    }
    __syncthreads();       // Wk is shared by the waves; everything else below is private to a wave (its 64 rows)
    const int w0 = wave * 64;
    if (p.z != nullptr) {
        // rows of the aggregates to global memory, 16 bytes per lane, consecutive lanes consecutive addresses
        const int q4 = p.zld >> 2;                        // float4s per row (zld = roundup4(9 * CIN) <= KZ)
        for (int f = lane; f < 64 * q4; f += 64) {
            const int r = f / q4, c4 = f % q4;
            const int node = nodes[w0 + r];
            if (node >= 0)
                *reinterpret_cast<f32x4*>(p.z + (size_t)node * p.zld + c4 * 4) =
                    *reinterpret_cast<const f32x4*>(zt + (w0 + r) * ZS + c4 * 4);
        }
    }
    float bias_o[OT];
#pragma unroll
    for (int ot = 0; ot < OT; ++ot) bias_o[ot] = p.bias[ot * 16 + lr];
#pragma unroll 1
    for (int nt = 0; nt < 4; ++nt) {
        f32x4 acc[OT];
#pragma unroll
        for (int ot = 0; ot < OT; ++ot) acc[ot] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int g = 0; g < KZ / 16; ++g) {
            const f32x4 av = *reinterpret_cast<const f32x4*>(zt + (w0 + nt * 16 + lr) * ZS + g * 16 + lq * 4);
#pragma unroll
            for (int t = 0; t < 4; ++t)
#pragma unroll
                for (int ot = 0; ot < OT; ++ot)
                    acc[ot] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[t], Wk[(g * 16 + lq * 4 + t) * WS + ot * 16 + lr],
                                                                   acc[ot], 0, 0, 0);
        }
        // C layout: column = lr (output channel within the tile), rows = nt*16 + lq*4 + t: one 4:1 pooling group per lane
        int nd[4], dd[4];
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            nd[t] = nodes[w0 + nt * 16 + lq * 4 + t];
            dd[t] = degs[w0 + nt * 16 + lq * 4 + t];
        }
#pragma unroll
        for (int ot = 0; ot < OT; ++ot) {
            const int o = ot * 16 + lr;
            float mx = -INFINITY;
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                if (nd[t] < 0) continue;
                float val = acc[ot][t] * (dd[t] > 0 ? 1.0f / (float)dd[t] : 0.f);
                if (!p.bias_mask || dd[t] > 0) val += bias_o[ot];
                if (p.act) val = fmaxf(val, 0.f) - p.alpha * fmaxf(-val, 0.f);
                st_act(p.y, (size_t)nd[t] * COUT + o, val, p.out_bf16);
                mx = fmaxf(mx, val);
            }
            if (p.y_pool && nd[0] >= 0) st_act(p.y_pool, (size_t)(nd[0] >> 2) * COUT + o, mx, p.out_bf16);
        }
    }
}

// ---------------------------------------------------------------------------------------------
// backward of the first layer (no input gradient): du, dv, dc partial sums per workgroup
// ---------------------------------------------------------------------------------------------
struct NarrowBwd {
    int n;
    const int* rowptr;
    const int* col;
    const float* x;
    const float* ag;
    const float* W0;
    const float* ds;       // [n, cout]  s = dy * lrelu'(y) / deg
    int cin, cout;
    float* part;           // [grid][NARROW_PART]: du [9*CIN] | dv [9*CIN] | dc [9], zero padded
    // FUSE (stage 1 folded in, cout == 32): s is computed here from dy / y (and the pooled gradient), written to ds_out for
    // the weight-gradient GEMM; one db partial per workgroup
    const float* dy;
    const float* y;
    const float* pool_y;
    const float* pool_dy;
    float* ds_out;
    float* db_part;        // [grid][32]
    int act, bias_mask, in_bf16;
    float alpha;
};
__device__ __forceinline__ float narrow_slope(float y, float alpha) { return y > 0.f ? 1.f : (y < 0.f ? alpha : 0.f); }
// lane (lane & ~3) + Q of every quad of lanes
template <int Q>
__device__ __forceinline__ float quad_bcast(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), Q * 0x55, 0xf, 0xf, false));
}
constexpr int NARROW_PART = 160;   // >= 2 * 9 * 8 + 9

// MMA (cout == 32): dz = W^T s of the workgroup's 256 nodes as [64 x 32] x [32 x K] per wave on the matrix cores, handed
// to the node-per-lane part through LDS (the vector form spends 1 728 FMAs per lane and 32 strided 4-byte loads of s)
template <int CIN, bool MMA, bool FUSE = false>
__global__ __launch_bounds__(NB) void conv_narrow_bwd_kernel(NarrowBwd p) {
    static_assert(MMA || !FUSE, "the fused prologue belongs to the matrix-core form");
    __shared__ float red[NB / 64][NARROW_PART];
    __shared__ float reddb[FUSE ? NB / 64 : 1][32];
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    const int i = blockIdx.x * NB + threadIdx.x;
    const bool active = i < p.n;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    int e0 = 0, e1 = 0;
    float a[FGC_M], xi[CIN];
#pragma unroll
    for (int m = 0; m < FGC_M; ++m) a[m] = 0.f;
#pragma unroll
    for (int c = 0; c < CIN; ++c) xi[c] = 0.f;
    // dz_i[m][c] = sum_o W0[m][o][c] s_i[o]   (scalar weight loads)
    float dz[FGC_M][CIN];
#pragma unroll
    for (int m = 0; m < FGC_M; ++m)
#pragma unroll
        for (int c = 0; c < CIN; ++c) dz[m][c] = 0.f;
    if (active) {
        e0 = p.rowptr[i];
        e1 = p.rowptr[i + 1];
        const float* ar = p.ag + (size_t)i * FGC_AG_LD;
#pragma unroll
        for (int m = 0; m < FGC_M; ++m) a[m] = ar[m];
#pragma unroll
        for (int c = 0; c < CIN; ++c) xi[c] = p.x[(size_t)i * CIN + c];
    }
    if constexpr (MMA) {
        constexpr int K9 = FGC_M * CIN, KZ = (K9 + 15) / 16 * 16, ZS = KZ + 4;
        float* dzt = reinterpret_cast<float*>(smem_raw);      // [NB][ZS]
        float* Wt = dzt + NB * ZS;                             // [32][ZS]: Wt[o][m * CIN + c] = W0[m][o][c]
        const int lr = lane & 15, lq = lane >> 4;
        for (int t = threadIdx.x; t < 32 * KZ; t += NB) {
            const int o = t / KZ, k = t % KZ;
            Wt[o * ZS + k] = k < K9 ? p.W0[((size_t)(k / CIN) * 32 + o) * CIN + k % CIN] : 0.f;
        }
        __syncthreads();
        const int w0 = wave * 64;
        const int node0 = blockIdx.x * NB + w0;
        f32x4 dbacc[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
        // FUSE: every operand of s for the wave's four row tiles is requested up front (32 loads in flight per lane; the
        // kernel runs two waves per SIMD on its LDS footprint, registers are free)
        f32x4 Gd[FUSE ? 4 : 1][2], Yd[FUSE ? 4 : 1][2], Md[FUSE ? 4 : 1][2], Pd[FUSE ? 4 : 1][2];
        int Dg[FUSE ? 4 : 1];
        if constexpr (FUSE) {
            auto load_all = [&](auto bf_tag) {
                constexpr bool BF = decltype(bf_tag)::value;
                auto ld4 = [&](const float* base, size_t chunk) {     // four columns of a 32-wide row
                    if constexpr (BF) return bf4_to_f4(reinterpret_cast<const u32x2*>(base)[chunk]);
                    else return reinterpret_cast<const f32x4*>(base)[chunk];
                };
#pragma unroll
                for (int nt = 0; nt < 4; ++nt) {
                    const int row = min(node0 + nt * 16 + lr, p.n - 1);
                    Dg[nt] = p.rowptr[row + 1] - p.rowptr[row];
#pragma unroll
                    for (int g = 0; g < 2; ++g) {
                        Gd[nt][g] = ld4(p.dy, (size_t)row * 8 + g * 4 + lq);
                        Yd[nt][g] = ld4(p.y, (size_t)row * 8 + g * 4 + lq);
                    }
                }
                if (p.pool_dy) {
#pragma unroll
                    for (int nt = 0; nt < 4; ++nt) {
                        const int row = min(node0 + nt * 16 + lr, p.n - 1);
#pragma unroll
                        for (int g = 0; g < 2; ++g) {
                            Md[nt][g] = ld4(p.pool_y, (size_t)(row >> 2) * 8 + g * 4 + lq);
                            Pd[nt][g] = ld4(p.pool_dy, (size_t)(row >> 2) * 8 + g * 4 + lq);
                        }
                    }
                }
            };
            if (p.in_bf16) load_all(std::true_type{});
            else load_all(std::false_type{});
        }
        auto row_tile = [&](int nt, const int ntc) {      // ntc: nt as an index of the register arrays (FUSE: unrolled)
            const int rown = node0 + nt * 16 + lr;
            const int row = min(rown, p.n - 1);      // rows past n: never used (no edges there)
            f32x4 av[2];
            if constexpr (FUSE) {
                // s = (dy + pooled gradient) * lrelu'(y) / deg of this lane's row, columns g * 16 + lq * 4 .. + 3: the same
                // operations in the same order as ds_db_vec_kernel (the same s bit for bit; reciprocal multiplies in place of
                // the divisions did not change the launch time: it is the 55 MB this prologue moves).  The rows of a 4:1
                // pooling group sit in the four lanes of a quad.
                const int dg = Dg[ntc];
                const bool counts = rown < p.n && (!p.bias_mask || dg > 0);
#pragma unroll
                for (int g = 0; g < 2; ++g) {
                    f32x4 gv = Gd[ntc][g];
                    const f32x4 yv = Yd[ntc][g];
                    if (p.pool_dy) {
                        const f32x4 m = Md[ntc][g], gp = Pd[ntc][g];
#pragma unroll
                        for (int k = 0; k < 4; ++k) {
                            float ne = 0.f;
                            ne += quad_bcast<0>(yv[k]) == m[k] ? 1.f : 0.f;
                            ne += quad_bcast<1>(yv[k]) == m[k] ? 1.f : 0.f;
                            ne += quad_bcast<2>(yv[k]) == m[k] ? 1.f : 0.f;
                            ne += quad_bcast<3>(yv[k]) == m[k] ? 1.f : 0.f;
                            gv[k] += yv[k] == m[k] ? gp[k] / ne : 0.f;
                        }
                    }
                    if (p.act) {
#pragma unroll
                        for (int k = 0; k < 4; ++k) gv[k] *= narrow_slope(yv[k], p.alpha);
                    }
                    if (counts) dbacc[g] += gv;
#pragma unroll
                    for (int k = 0; k < 4; ++k) av[g][k] = dg > 0 ? gv[k] / (float)dg : 0.f;
                    if (rown < p.n) *reinterpret_cast<f32x4*>(p.ds_out + (size_t)row * 32 + g * 16 + lq * 4) = av[g];
                }
            } else {
#pragma unroll
                for (int g = 0; g < 2; ++g) av[g] = *reinterpret_cast<const f32x4*>(p.ds + (size_t)row * 32 + g * 16 + lq * 4);
            }
#pragma unroll
            for (int kt = 0; kt < KZ / 16; ++kt) {
                f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int g = 0; g < 2; ++g)
#pragma unroll
                    for (int t = 0; t < 4; ++t)
                        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av[g][t], Wt[(g * 16 + lq * 4 + t) * ZS + kt * 16 + lr], acc,
                                                                   0, 0, 0);
#pragma unroll
                for (int t = 0; t < 4; ++t) dzt[(w0 + nt * 16 + lq * 4 + t) * ZS + kt * 16 + lr] = acc[t];
            }
        };
        if constexpr (FUSE) {
            row_tile(0, 0);
            row_tile(1, 1);
            row_tile(2, 2);
            row_tile(3, 3);
        } else {
#pragma unroll 1
            for (int nt = 0; nt < 4; ++nt) row_tile(nt, 0);
        }
        if constexpr (FUSE) {
            // column sums over the wave's 64 rows: the 16 lanes of a row group hold the same columns
#pragma unroll
            for (int g = 0; g < 2; ++g)
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    float v = dbacc[g][k];
                    v += __shfl_xor(v, 1);
                    v += __shfl_xor(v, 2);
                    v += __shfl_xor(v, 4);
                    v += __shfl_xor(v, 8);
                    if (lr == 0) reddb[wave][g * 16 + lq * 4 + k] = v;
                }
        }
        // the wave wrote the rows of its own 64 nodes: each lane takes its node's row back
        float dzf[KZ];
#pragma unroll
        for (int k = 0; k < KZ; k += 4) {
            const f32x4 v = *reinterpret_cast<const f32x4*>(dzt + threadIdx.x * ZS + k);
            dzf[k] = v[0]; dzf[k + 1] = v[1]; dzf[k + 2] = v[2]; dzf[k + 3] = v[3];
        }
#pragma unroll
        for (int m = 0; m < FGC_M; ++m)
#pragma unroll
            for (int c = 0; c < CIN; ++c) dz[m][c] = active ? dzf[m * CIN + c] : 0.f;
    } else {
        for (int o = 0; o < p.cout; ++o) {
            const float so = active ? p.ds[(size_t)i * p.cout + o] : 0.f;
#pragma unroll
            for (int m = 0; m < FGC_M; ++m) {
                const float* w = p.W0 + ((size_t)m * p.cout + o) * CIN;
#pragma unroll
                for (int c = 0; c < CIN; ++c) dz[m][c] = fmaf(w[c], so, dz[m][c]);
            }
        }
    }
    float da[FGC_M], wv[FGC_M][CIN];
#pragma unroll
    for (int m = 0; m < FGC_M; ++m) {
        da[m] = 0.f;
#pragma unroll
        for (int c = 0; c < CIN; ++c) wv[m][c] = 0.f;
    }
    for (int e = e0; e < e1; e += EB_BWD) {
        float g[EB_BWD][FGC_M], xj[EB_BWD][CIN];
        fetch_edges<CIN, EB_BWD>(p.col, p.ag, p.x, p.cin, e, e1, g, xj);
#pragma unroll
        for (int t = 0; t < EB_BWD; ++t) {
            float q[FGC_M];
            softmax9(a, g[t], q);
            float dq[FGC_M], dot = 0.f;
#pragma unroll
            for (int m = 0; m < FGC_M; ++m) {
                float v = 0.f;
#pragma unroll
                for (int c = 0; c < CIN; ++c) v = fmaf(dz[m][c], xj[t][c], v);
                dq[m] = v;
                dot = fmaf(q[m], v, dot);
            }
            const float w = e + t < e1 ? 1.f : 0.f;
#pragma unroll
            for (int m = 0; m < FGC_M; ++m) {
                const float dl = q[m] * (dq[m] - dot) * w;
                da[m] += dl;
#pragma unroll
                for (int c = 0; c < CIN; ++c) wv[m][c] = fmaf(dl, xj[t][c], wv[m][c]);
            }
        }
    }
    // block sums in a fixed order: wave butterfly, then the 4 waves
    auto wave_sum = [](float v) {
        for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
        return v;
    };
#pragma unroll
    for (int m = 0; m < FGC_M; ++m) {
#pragma unroll
        for (int c = 0; c < CIN; ++c) {
            const float u = wave_sum(da[m] * xi[c]);
            const float v = wave_sum(wv[m][c]);
            if (lane == 0) {
                red[wave][m * CIN + c] = u;
                red[wave][FGC_M * CIN + m * CIN + c] = v;
            }
        }
        const float cc = wave_sum(da[m]);
        if (lane == 0) red[wave][2 * FGC_M * CIN + m] = cc;
    }
    __syncthreads();
    for (int k = threadIdx.x; k < NARROW_PART; k += NB) {
        const float v = k < 2 * FGC_M * CIN + FGC_M ? (red[0][k] + red[1][k]) + (red[2][k] + red[3][k]) : 0.f;
        p.part[(size_t)blockIdx.x * NARROW_PART + k] = v;
    }
    if constexpr (FUSE) {
        if (threadIdx.x < 32)
            p.db_part[(size_t)blockIdx.x * 32 + threadIdx.x] =
                (reddb[0][threadIdx.x] + reddb[1][threadIdx.x]) + (reddb[2][threadIdx.x] + reddb[3][threadIdx.x]);
    }
}

// ---------------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------------
static int narrow_cin_pad(int cin) { return cin; }   // kernels are instantiated for the exact width 1..8

bool narrow_supported(const fgc_conv_desc* d) {
    if (opt(OPT_NO_NARROW) == 1) return false;
    const int cin = d->c0 + d->c1;
    return d->c1 == 0 && d->x1 == nullptr && d->shift == 0 && cin <= 8 && d->cout <= 64 && d->cout % 4 == 0 &&
           (size_t)d->n * 4 * 128 < 0xFFFFFFFFull;
}

template <int CIN, bool ZONLY>
static int launch_narrow_fwd_t(const NarrowFwd& p, hipStream_t st) {
    const int tiles = p.tile_list ? p.n_tiles : cdiv(p.n, TILE);
    if (tiles == 0) return FGC_OK;
    const size_t smem = ZONLY ? 0 : (size_t)NB * (p.cout + 1) * 4;
    hipFuncSetAttribute((const void*)conv_narrow_fwd_kernel<CIN, ZONLY>, hipFuncAttributeMaxDynamicSharedMemorySize,
                        (int)smem);
    FGC_LAUNCH(ZONLY ? "conv_narrow_kernel<z>" : "conv_narrow_kernel<fwd>", st, (conv_narrow_fwd_kernel<CIN, ZONLY>),
               dim3(cdiv(tiles, NB / TILE)), dim3(NB), smem, p);
    FGC_CHECK_LAUNCH("conv_narrow_fwd_kernel");
    return FGC_OK;
}

template <bool ZONLY>
static int launch_narrow_fwd_z(const NarrowFwd& p, hipStream_t st) {
    switch (p.cin) {
        case 1: return launch_narrow_fwd_t<1, ZONLY>(p, st);
        case 2: return launch_narrow_fwd_t<2, ZONLY>(p, st);
        case 3: return launch_narrow_fwd_t<3, ZONLY>(p, st);
        case 4: return launch_narrow_fwd_t<4, ZONLY>(p, st);
        case 5: return launch_narrow_fwd_t<5, ZONLY>(p, st);
        case 6: return launch_narrow_fwd_t<6, ZONLY>(p, st);
        case 7: return launch_narrow_fwd_t<7, ZONLY>(p, st);
        default: return launch_narrow_fwd_t<8, ZONLY>(p, st);
    }
}

template <int CIN, int OT>
static int launch_narrow_fwd_mma_t(const NarrowFwd& p, hipStream_t st) {
    const int tiles = p.tile_list ? p.n_tiles : cdiv(p.n, TILE);
    if (tiles == 0) return FGC_OK;
    constexpr int KZ = (FGC_M * CIN + 15) / 16 * 16;
    const size_t smem = ((size_t)NB * (KZ + 4) + (size_t)KZ * (OT * 16 + 4)) * 4 + (size_t)2 * NB * 4;
    hipFuncSetAttribute((const void*)conv_narrow_fwd_mma_kernel<CIN, OT>, hipFuncAttributeMaxDynamicSharedMemorySize,
                        (int)smem);
    FGC_LAUNCH("conv_narrow_kernel<fwd>", st, (conv_narrow_fwd_mma_kernel<CIN, OT>), dim3(cdiv(tiles, NB / TILE)), dim3(NB),
               smem, p);
    FGC_CHECK_LAUNCH("conv_narrow_fwd_mma_kernel");
    return FGC_OK;
}

int launch_narrow_fwd(const fgc_conv_desc* d, const float* ag, float* y, float* y_pool, float* zsave, hipStream_t st,
                      bool out_bf16) {
    NarrowFwd p{d->n,    d->rowptr,    d->col, d->x0,   ag, d->W0, d->b, d->c0, d->cout, d->bias_mask, d->act,
                d->alpha, y, y_pool, zsave, narrow_zld(d->c0), d->tile_list, d->n_tiles, out_bf16 ? 1 : 0};
    // the network's first layer (6 -> 32) and its 3-channel sibling: per-node products on the matrix cores
    const bool mma = !(opt(OPT_NO_NARROW_MMA) == 1) && d->cout == 32 &&
                     ((uintptr_t)zsave % 16) == 0;
    if (mma && d->c0 == 6) return launch_narrow_fwd_mma_t<6, 2>(p, st);
    if (mma && d->c0 == 3) return launch_narrow_fwd_mma_t<3, 2>(p, st);
    return launch_narrow_fwd_z<false>(p, st);
}

int narrow_zld(int cin) { return (FGC_M * narrow_cin_pad(cin) + 3) / 4 * 4; }
size_t narrow_bwd_floats(const fgc_conv_desc* d) {
    const int zld = narrow_zld(d->c0);
    const size_t nblk = cdiv(d->n, NB);
    const int splits = narrow_splits(d);
    return (size_t)d->n * zld + 64 + nblk * NARROW_PART + 64 + (size_t)splits * zld * d->cout + 64 +
           (size_t)zld * d->cout + NARROW_PART + 64 + reduce_tmp_floats((int)nblk, NARROW_PART) +
           reduce_tmp_floats(splits, (size_t)zld * d->cout) + 64;
}

int narrow_splits(const fgc_conv_desc* d) {
    return tn_balanced_splits(768 / cdiv(d->cout, 64), cdiv(d->n, 256), d->n);
}

static bool narrow_bwd_mma(const fgc_conv_desc* d, const fgc_conv_bwd_io* io) {
    return !(opt(OPT_NO_NARROW_MMA) == 1) && d->cout == 32 &&
           ((uintptr_t)io->ds % 16) == 0 && (d->c0 == 6 || d->c0 == 3);
}
// stage 1 (s and the db partials) folded into the stage-2 kernel: a function of descriptor, io and environment alone, so that
// every stage call and the reduction agree on where the db partials are and how many there are
bool narrow_fuses_ds(const fgc_conv_desc* d, const fgc_conv_bwd_io* io) {
    if (opt(OPT_NO_FUSED_DS) == 1) return false;
    if (opt(OPT_NO_NARROW_FUSED_DS) == 1) return false;
    if (!narrow_bwd_mma(d, io) || !io->dy) return false;
    const uintptr_t al = (d->flags & FGC_CONV_BF16) ? 8 : 16;
    const float* yy = io->y ? io->y : io->dy;
    if (((uintptr_t)io->dy | (uintptr_t)yy) % al) return false;
    if (io->pool_dy && (!io->pool_y || !io->y || d->n % 4 != 0 || ((uintptr_t)io->pool_y | (uintptr_t)io->pool_dy) % al)) return false;
    return true;
}
int narrow_db_partials(const fgc_conv_desc* d, const fgc_conv_bwd_io* io, int nb_db) {
    return narrow_fuses_ds(d, io) ? cdiv(d->n, NB) : nb_db;
}

// stage 2 of the first layer's backward: everything except the final sums (db partials: stage 1 left them, or - when
// narrow_fuses_ds - this launch leaves one per workgroup in db_part)
int narrow_bwd_logits(const fgc_conv_desc* d, const fgc_conv_bwd_io* io, float* scratch, float* db_part, hipStream_t st) {
    const int cin = d->c0, zld = narrow_zld(cin);
    float* zbuf = scratch;
    float* part = zbuf + (size_t)d->n * zld + 64;
    NarrowFwd pz{d->n, d->rowptr, d->col, d->x0, io->ag, d->W0, d->b, cin, d->cout, d->bias_mask, d->act, d->alpha,
                 nullptr, nullptr, zbuf, zld, nullptr, 0};
    int rc = io->z_saved ? FGC_OK : launch_narrow_fwd_z<true>(pz, st);   // forward left them (FGC_CONV_SAVE_Z)
    if (rc) return rc;
    NarrowBwd pb{d->n, d->rowptr, d->col, d->x0, io->ag, d->W0, io->ds, cin, d->cout, part};
    const bool fuse = narrow_fuses_ds(d, io);
    if (fuse) {
        pb.dy = io->dy;
        pb.y = io->y ? io->y : io->dy;
        pb.pool_y = io->pool_dy ? io->pool_y : nullptr;
        pb.pool_dy = io->pool_dy;
        pb.ds_out = io->ds;
        pb.db_part = db_part;
        pb.act = d->act;
        pb.bias_mask = d->bias_mask;
        pb.in_bf16 = (d->flags & FGC_CONV_BF16) ? 1 : 0;
        pb.alpha = d->alpha;
    }
    const dim3 grid(cdiv(d->n, NB));
#define FGC_NARROW_BWD(C_) \
    case C_: FGC_LAUNCH("conv_narrow_kernel<bwd>", st, (conv_narrow_bwd_kernel<C_, false>), grid, dim3(NB), 0, pb); break;
    const bool mma = narrow_bwd_mma(d, io);
    if (mma) {
        const int KZ = (FGC_M * cin + 15) / 16 * 16;
        const size_t smem = (size_t)(NB + 32) * (KZ + 4) * 4;
        if (fuse && cin == 6) {
            hipFuncSetAttribute((const void*)conv_narrow_bwd_kernel<6, true, true>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                (int)smem);
            FGC_LAUNCH("conv_narrow_kernel<bwd>", st, (conv_narrow_bwd_kernel<6, true, true>), grid, dim3(NB), smem, pb);
        } else if (fuse) {
            hipFuncSetAttribute((const void*)conv_narrow_bwd_kernel<3, true, true>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                (int)smem);
            FGC_LAUNCH("conv_narrow_kernel<bwd>", st, (conv_narrow_bwd_kernel<3, true, true>), grid, dim3(NB), smem, pb);
        } else if (cin == 6) {
            hipFuncSetAttribute((const void*)conv_narrow_bwd_kernel<6, true>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                (int)smem);
            FGC_LAUNCH("conv_narrow_kernel<bwd>", st, (conv_narrow_bwd_kernel<6, true>), grid, dim3(NB), smem, pb);
        } else {
            hipFuncSetAttribute((const void*)conv_narrow_bwd_kernel<3, true>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                (int)smem);
            FGC_LAUNCH("conv_narrow_kernel<bwd>", st, (conv_narrow_bwd_kernel<3, true>), grid, dim3(NB), smem, pb);
        }
    } else
    switch (cin) {
        FGC_NARROW_BWD(1) FGC_NARROW_BWD(2) FGC_NARROW_BWD(3) FGC_NARROW_BWD(4)
        FGC_NARROW_BWD(5) FGC_NARROW_BWD(6) FGC_NARROW_BWD(7)
        default: FGC_LAUNCH("conv_narrow_kernel<bwd>", st, (conv_narrow_bwd_kernel<8, false>), grid, dim3(NB), 0, pb); break;
    }
#undef FGC_NARROW_BWD
    FGC_CHECK_LAUNCH("conv_narrow_bwd_kernel");
    return FGC_OK;
}

// operands of the first layer's weight-gradient GEMM dW0^T = z^T s (for a caller that launches it with other layers')
void narrow_tn_operands(const fgc_conv_desc* d, const fgc_conv_bwd_io* io, float* scratch, const float** A, int* zld_out,
                        float** slab_out, int* rps_out) {
    const int zld = narrow_zld(d->c0);
    const int nblk = cdiv(d->n, NB);
    const int splits = narrow_splits(d);
    float* zbuf = scratch;
    float* part = zbuf + (size_t)d->n * zld + 64;
    *A = io->z_saved ? io->z_saved : zbuf;
    *zld_out = zld;
    *slab_out = part + (size_t)nblk * NARROW_PART + 64;
    *rps_out = cdiv(cdiv(d->n, splits), 4) * 4;
}

// stage 8: dW0 = sum_i s_i (x) z_i through the streaming GEMM, then every partial in two reduction launches
int narrow_bwd_params(const fgc_conv_desc* d, const fgc_conv_bwd_io* io, float* scratch, const float* db_part,
                      int nb_db, int parts, RedJob* jobs_out, hipStream_t st) {
    const int cin = d->c0, cout = d->cout, zld = narrow_zld(cin), CIN = narrow_cin_pad(cin);
    const int nblk = cdiv(d->n, NB);
    const int splits = narrow_splits(d);
    float* zbuf = scratch;
    float* part = zbuf + (size_t)d->n * zld + 64;
    float* slab = part + (size_t)nblk * NARROW_PART + 64;
    float* rtmp = slab + (size_t)splits * zld * cout + 64;
    const int rps = cdiv(cdiv(d->n, splits), 4) * 4;
    const int ns = cdiv(d->n, rps);
    int rc = 0;
    if ((parts & 1) && !(io->flags & FGC_CONV_DEFER_DW)) {
        rc = launch_gemm_tn_stream("gemm_tn_kernel:dW", io->z_saved ? io->z_saved : zbuf, zld, zld, io->ds, cout, d->n, rps,
                                   ns, slab, st);
        if (rc) return rc;
    }
    // dW0[m][o][c] is the z^T s product [m * cin + c][o] summed over its slabs and written transposed per m; du / dv / dc are
    // three ranges of the logit kernel's per-workgroup partial rows
    RedJob jobs[NARROW_RED_JOBS] = {
        {slab, (size_t)zld * cout, ns, zld * cout, cout, FGC_M * CIN, io->dW0, rtmp},
        {part, (size_t)NARROW_PART, nblk, FGC_M * CIN, FGC_M * CIN, FGC_M * CIN, io->du},
        {part + FGC_M * CIN, (size_t)NARROW_PART, nblk, FGC_M * CIN, FGC_M * CIN, FGC_M * CIN, io->dv},
        {part + 2 * FGC_M * CIN, (size_t)NARROW_PART, nblk, FGC_M, FGC_M, FGC_M, io->dc},
        {db_part, (size_t)cout, nb_db, cout, cout, cout, io->db},
    };
    jobs[0].tr = CIN;
    if (jobs_out)
        for (int q = 0; q < NARROW_RED_JOBS; ++q) jobs_out[q] = jobs[q];
    if (parts & 2) {
        rc = reduce_jobs("reduce:params", jobs, NARROW_RED_JOBS, nullptr, st);
        if (rc) return rc;
    }
    return FGC_OK;
}

}  // namespace fgc
