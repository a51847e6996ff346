// Fused "edge softmax -> neighbour aggregate -> f32 MFMA" core shared by the forward
// graph convolution and by its data-gradient (which is a convolution of the same shape
// over the transposed graph).  gfx950 only: wave64, v_mfma_f32_16x16x4_f32, 160 KB LDS.
//
// Work decomposition (one 256-thread workgroup = 4 waves = one tile of T = 32 nodes):
//   phase S  (once)      thread (node = tid/8, kl = tid%8) computes the M = 9 assignment
//                        softmax of edges kl, kl+8, kl+16 of its node -> LDS qbuf
//   per pass p over KC = 4*LPN gathered channels:
//     phase A            thread (node, cl) streams its node's neighbour rows (float4 per
//                        lane, 128 B contiguous per node for LPN = 8) and accumulates
//                        z[m][4] += q[m] * x_j[4]      (36 fp32 accumulators / lane)
//                        -> LDS ztile[node][m*KC + cl*4 ..]
//     phase G            MFMA: acc[T x npad] += ztile[T x kpass] * Wp[kpass x npad]
//                        A fragments by ds_read_b128 (row stride == 8 mod 64 floats: conflict
//                        free), B fragments by one global dwordx4 per 4 MFMAs from the
//                        k-interleaved packed weights (L2 resident, shared by all tiles)
//   epilogue             accumulators -> LDS (k-split partials summed in fixed order)
//                        -> op-specific epilogue (bias/act/pool, or dx assembly)
#pragma once
#include "fgc_common.h"

namespace fgc {

constexpr int TILE = 32;        // nodes per workgroup
#ifndef FGC_KMAX
#define FGC_KMAX 24
#endif
constexpr int KMAX = FGC_KMAX;  // >= K_faces (23) edge slots per node kept in LDS
constexpr int QLD = 12;         // floats per edge in qbuf: q[0..8], [9] = source row of the neighbour (int bits)
// floats between the q tables of consecutive nodes.  With 16 slots the plain stride (16 * 12 floats = 768 B) is a multiple
// of the 256-byte bank row: the two nodes a ds_read_b128 lane group spans read the same banks (2-way conflict on every q
// read of the aggregation phase, three per edge slot).  Four floats of padding move the second node one 16-byte slot on.
constexpr int qnode_stride(int qslots) { return qslots * QLD + (qslots == 16 ? 4 : 0); }
constexpr int NTHREADS = 256;
constexpr int MAX_NPAD = 128;   // GEMM N limit (cout / cin of the transposed op)

struct CoreParams {
    int n;                   // nodes
    const int* rowptr;       // CSR used for gathering (forward: out-edges, bwd-data: in-edges)
    const int* col;
    const int* eid;          // bwd-data only: forward edge id of each in-edge (else NULL)
    const float* src0;       // gathered rows, source 0: [(n>>shift), c0]
    const float* src1;       // source 1 or NULL
    int c0, c1, shift;       // gathered width cg = c0 + c1
    int cg;                  // gathered channels
    int nout;                // GEMM N (real), npad = roundup16
    int npad;
    int passes, kc, kpass, zstride;
    const float* ag;         // logits table [(n>>shift_ag), 24]
    int ag_shift;            // row = node >> ag_shift
    int ctr_off, nbr_off;    // 0 (a) / 12 (g): which half the centre / the neighbour contributes
    const float* Wp;         // packed B operand [passes*kpass/4][npad] float4
    const int* tile_list;    // tiles to compute (NULL = all cdiv(n, TILE) of them); the grid has one block per entry
    int n_tiles;
};

// ---- workgroup -> tile map ----------------------------------------------------------------
// Workgroups are dealt round-robin over the 8 XCDs (b and b+8 share an XCD and its private 4 MB L2).  The node order
// is spatially coherent, so tiles that are close in index gather overlapping neighbour rows: give every XCD a
// CONTIGUOUS range of tiles, then those rows are fetched into one L2 and hit there for the neighbouring tiles,
// instead of being fetched by all eight L2s from the Infinity Cache.  Bijective for any tile count.  Speed only.
__device__ __forceinline__ int xcd_tile(int b, int ntiles) {
    const int q = ntiles >> 3, r = ntiles & 7;
    const int xcd = b & 7, idx = b >> 3;
    return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
}

// first node of this workgroup's tile
__device__ __forceinline__ int block_tile0(const CoreParams& p) {
    const int t = xcd_tile(blockIdx.x, gridDim.x);
    return (p.tile_list ? p.tile_list[t] : t) * TILE;
}
static inline int core_grid(const CoreParams& p) { return p.tile_list ? p.n_tiles : cdiv(p.n, TILE); }

// ---- LDS carve ---------------------------------------------------------------------------
struct Smem {
    float* ztile;   // [TILE][zstride]
    float* qbuf;    // [TILE][KMAX][QLD]  (slot 9 of each row: neighbour source row, already >> shift)
    int* deg;       // [TILE] degrees, 4 scratch ints, [TILE] first-edge ids (no static __shared__: keeps the
                    // dynamic base 16-B aligned)
    float* extra;   // op specific
};

// qslots: edge slots per node the kernel keeps (KMAX, or 16 where the host guarantees max_deg <= 16: 12 KB less LDS)
__device__ __forceinline__ Smem carve(char* base, int zstride, int qslots = KMAX, int nodes = TILE) {
    Smem s;
    s.ztile = reinterpret_cast<float*>(base);
    size_t off = (size_t)nodes * zstride * 4;
    s.qbuf = reinterpret_cast<float*>(base + off);
    off += (size_t)nodes * qnode_stride(qslots) * 4;
    s.deg = reinterpret_cast<int*>(base + off);
    off += (2 * nodes + 4) * 4;
    s.extra = reinterpret_cast<float*>(base + off);
    return s;
}
static inline size_t smem_core_bytes(int zstride, int qslots = KMAX) {
    return (size_t)TILE * zstride * 4 + (size_t)TILE * qnode_stride(qslots) * 4 + (2 * TILE + 4) * 4;
}

// ---- phase S: per-edge soft assignment ------------------------------------------------------
// q_ikm = softmax_m(ctr[m] + nbr[m])  (model.py:79-94; c is already folded into the a half)
// Edges [kbase, kbase+KMAX) of every node of the tile are processed per call (one call covers
// every node whose degree is <= KMAX = 24, i.e. every reference K-list; longer in-edge lists of
// asymmetric graphs take several chunks).  Returns this thread's node degree.
// NT nodes of the tile (s carved for NT nodes), LPN lanes per node (a power of two): NT * LPN threads
template <bool WITH_DL, int QS = KMAX, int NT = TILE, int LPN = 8>
__device__ __forceinline__ int softmax_phase(const CoreParams& p, const Smem& s, int tile0, int kbase,
                                             const float* dl, float* dgsum /* [9] += sum of dl over my edges */) {
    const int tid = threadIdx.x;
    const int node = tid / LPN, kl = tid % LPN;
    const int i = tile0 + node;
    int d = 0, e0 = 0;
    float ctr[FGC_M];
    // logit table, edge list and per-edge d-logits through buffer descriptors: a gather costs one v_mad_u32_u24 for its
    // 32-bit offset instead of a 64-bit multiply-add chain on the vector ALU (which the fp32 MFMA shares)
    const __amdgpu_buffer_rsrc_t ag_rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.ag), 0, -1, 0x00020000);
    const __amdgpu_buffer_rsrc_t col_rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<int*>(p.col), 0, -1, 0x00020000);
    if (i < p.n) {
        e0 = p.rowptr[i];
        d = p.rowptr[i + 1] - e0;
        const unsigned ao = __umul24((unsigned)(i >> p.ag_shift), FGC_AG_LD * 4u) + (unsigned)p.ctr_off * 4u;
        const f32x4 a0 = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(ag_rs, ao, 0, 0));
        const f32x4 a1 = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(ag_rs, ao + 16u, 0, 0));
        ctr[0] = a0[0]; ctr[1] = a0[1]; ctr[2] = a0[2]; ctr[3] = a0[3];
        ctr[4] = a1[0]; ctr[5] = a1[1]; ctr[6] = a1[2]; ctr[7] = a1[3];
        ctr[8] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(ag_rs, ao + 32u, 0, 0));
    }
    if (kl == 0) {
        s.deg[node] = d;
        s.deg[NT + 4 + node] = e0;
    }
    const int kend = min(d, kbase + KMAX);
    // all neighbour ids first, then all their logit rows: two memory round trips for the (up to 3) edges of this
    // thread instead of two per edge
    constexpr int EPT = (QS + LPN - 1) / LPN;  // edges per thread (QS < KMAX: the caller guarantees degrees <= QS)
    int jj[EPT];
    f32x4 g0[EPT], g1[EPT];
    float g8[EPT];
#pragma unroll
    for (int t = 0; t < EPT; ++t) {
        const int kk = kbase + kl + LPN * t;
        // (unconditional, clamped into the node's list: no exec-masked load.  A node WITHOUT edges reads the entry in front
        //  of its empty list: for the padding nodes at the end of a level e0 == nnz, one past the array)
        const int jv = __builtin_amdgcn_raw_buffer_load_b32(col_rs, (unsigned)(kend > 0 ? e0 + min(kk, kend - 1) : max(e0 - 1, 0)) * 4u, 0, 0);
        jj[t] = kk < kend ? jv : 0;
    }
#pragma unroll
    for (int t = 0; t < EPT; ++t) {
        const unsigned go = __umul24((unsigned)(jj[t] >> p.ag_shift), FGC_AG_LD * 4u) + (unsigned)p.nbr_off * 4u;
        g0[t] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(ag_rs, go, 0, 0));
        g1[t] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(ag_rs, go + 16u, 0, 0));
        g8[t] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(ag_rs, go + 32u, 0, 0));
    }
#pragma unroll
    for (int t = 0; t < EPT; ++t) {
        const int kk = kbase + kl + LPN * t;
        if (kk >= kend) continue;
        const int e = e0 + kk;
        const int k = kk - kbase;
        const int j = jj[t];
        float l[FGC_M];
        l[0] = ctr[0] + g0[t][0]; l[1] = ctr[1] + g0[t][1]; l[2] = ctr[2] + g0[t][2]; l[3] = ctr[3] + g0[t][3];
        l[4] = ctr[4] + g1[t][0]; l[5] = ctr[5] + g1[t][1]; l[6] = ctr[6] + g1[t][2]; l[7] = ctr[7] + g1[t][3];
        l[8] = ctr[8] + g8[t];
        float mx = l[0];
#pragma unroll
        for (int m = 1; m < FGC_M; ++m) mx = fmaxf(mx, l[m]);
        float sum = 0.f;
#pragma unroll
        for (int m = 0; m < FGC_M; ++m) {
            l[m] = __expf(l[m] - mx);   // v_exp_f32 (2 ulp; arguments <= 0): the accurate expf is 5x the instructions
            sum += l[m];
        }
        const float inv = 1.0f / sum;
        float* q = s.qbuf + (size_t)node * qnode_stride(QS) + k * QLD;
        *reinterpret_cast<f32x4*>(q) = f32x4{l[0] * inv, l[1] * inv, l[2] * inv, l[3] * inv};
        *reinterpret_cast<f32x4*>(q + 4) = f32x4{l[4] * inv, l[5] * inv, l[6] * inv, l[7] * inv};
        q[8] = l[8] * inv;
        q[9] = __int_as_float(j >> p.shift);
        if (WITH_DL) {
            const float* dr = dl + (size_t)p.eid[e] * FGC_DL_LD;
            const f32x4 d0 = *reinterpret_cast<const f32x4*>(dr);
            const f32x4 d1 = *reinterpret_cast<const f32x4*>(dr + 4);
            dgsum[0] += d0[0]; dgsum[1] += d0[1]; dgsum[2] += d0[2]; dgsum[3] += d0[3];
            dgsum[4] += d1[0]; dgsum[5] += d1[1]; dgsum[6] += d1[2]; dgsum[7] += d1[3];
            dgsum[8] += dr[8];
        }
    }
    return d;
}

// number of KMAX-edge chunks the tile needs (max over the block); also the barrier that
// publishes qbuf/deg written by softmax_phase.
__device__ __forceinline__ int edge_chunks(const Smem& s, int my_degree) {
    const int any_long = __syncthreads_or(my_degree > KMAX);
    if (!any_long) return 1;
    int* maxd = s.deg + TILE;
    if (threadIdx.x == 0) *maxd = 0;
    __syncthreads();
    atomicMax(maxd, my_degree);
    __syncthreads();
    return (*maxd + KMAX - 1) / KMAX;
}

// ---- gathered row chunk load ----------------------------------------------------------------
// channels [cbase, cbase+4) of the concatenated source row `row`
template <bool VEC4>
__device__ __forceinline__ f32x4 load_chunk(const CoreParams& p, int row, int cbase) {
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
    if (VEC4) {
        if (cbase < p.c0) {
            v = *reinterpret_cast<const f32x4*>(p.src0 + (size_t)row * p.c0 + cbase);
        } else if (cbase < p.cg) {
            v = *reinterpret_cast<const f32x4*>(p.src1 + (size_t)row * p.c1 + (cbase - p.c0));
        }
    } else {
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const int c = cbase + t;
            if (c < p.c0) v[t] = p.src0[(size_t)row * p.c0 + c];
            else if (c < p.cg) v[t] = p.src1[(size_t)row * p.c1 + (c - p.c0)];
        }
    }
    return v;
}

// ---- row batches: all gathers of a node are issued before the first FMA (one memory round trip per pass) ------
constexpr int RB = 16;  // rows held in registers at once (degree 13 on a closed valence-6 mesh; K_faces = 23 max)

template <bool VEC4>
__device__ __forceinline__ void load_rows(const CoreParams& p, const float* qb, int d, int k0, int cbase,
                                          f32x4 (&xv)[RB]) {
#pragma unroll
    for (int t = 0; t < RB; ++t) {
        xv[t] = f32x4{0.f, 0.f, 0.f, 0.f};
        if (k0 + t < d) xv[t] = load_chunk<VEC4>(p, __float_as_int(qb[(k0 + t) * QLD + 9]), cbase);
    }
}

__device__ __forceinline__ void fma_rows(const float* qb, int d, int k0, const f32x4 (&xv)[RB], f32x4 (&z)[FGC_M]) {
#pragma unroll
    for (int t = 0; t < RB; ++t) {
        if (k0 + t < d) {
            const float* q = qb + (k0 + t) * QLD;
            const f32x4 q0 = *reinterpret_cast<const f32x4*>(q);
            const f32x4 q1 = *reinterpret_cast<const f32x4*>(q + 4);
            const float q8 = q[8];
            z[0] += q0[0] * xv[t]; z[1] += q0[1] * xv[t]; z[2] += q0[2] * xv[t]; z[3] += q0[3] * xv[t];
            z[4] += q1[0] * xv[t]; z[5] += q1[1] * xv[t]; z[6] += q1[2] * xv[t]; z[7] += q1[3] * xv[t];
            z[8] += q8 * xv[t];
        }
    }
}

// ---- early gathers: the neighbour rows of the first RB edges are requested at kernel entry (row ids straight from
// the CSR), so that the memory round trip runs under the softmax phase; later passes are requested one pass ahead.
template <int LPN>
__device__ __forceinline__ void early_row_ids(const CoreParams& p, int tile0, int (&rows)[RB], int& d) {
    const int node = threadIdx.x / LPN;
    const int i = tile0 + node;
    d = 0;
    int e0 = 0;
    if (node < TILE && i < p.n) {
        e0 = p.rowptr[i];
        d = p.rowptr[i + 1] - e0;
    }
#pragma unroll
    for (int t = 0; t < RB; ++t) rows[t] = d > 0 ? (p.col[e0 + min(t, d - 1)] >> p.shift) : 0;
}

template <int LPN, bool VEC4>
__device__ __forceinline__ void issue_rows(const CoreParams& p, const int (&rows)[RB], int pass, f32x4 (&xv)[RB]) {
    const int cl = threadIdx.x % LPN;
    const int cbase = pass * KC + cl * 4;
#pragma unroll
    for (int t = 0; t < RB; ++t) xv[t] = load_chunk<VEC4>(p, rows[t], cbase);
}

// ---- phase A: z[m][4] = sum_k q[k][m] * x_j(k)[4] ---------------------------------------------
// thread (node, cl): node = tid / LPN (only the first TILE*LPN threads work), cl = tid % LPN
// accumulates into z (caller zeroes it); covers the edge chunk currently held in qbuf
template <int LPN, bool VEC4>
__device__ __forceinline__ void aggregate_pass(const CoreParams& p, const Smem& s, int pass, int kbase,
                                               f32x4 (&z)[FGC_M]) {
    const int tid = threadIdx.x;
    const int node = tid / LPN, cl = tid % LPN;
    if (node >= TILE) return;
    const int d = min(max(s.deg[node] - kbase, 0), KMAX);
    const int cbase = pass * KC + cl * 4;
    const float* qb = s.qbuf + (size_t)node * KMAX * QLD;
    for (int k0 = 0; k0 < d; k0 += RB) {
        f32x4 xv[RB];
        load_rows<VEC4>(p, qb, d, k0, cbase, xv);
        fma_rows(qb, d, k0, xv, z);
    }
}

template <int LPN>
__device__ __forceinline__ void store_ztile(const CoreParams& p, const Smem& s, const f32x4 (&z)[FGC_M]) {
    const int tid = threadIdx.x;
    const int node = tid / LPN, cl = tid % LPN;
    if (node >= TILE) return;
    float* zr = s.ztile + (size_t)node * ZSTRIDE + cl * 4;
#pragma unroll
    for (int m = 0; m < FGC_M; ++m) *reinterpret_cast<f32x4*>(zr + m * KC) = z[m];
}

// zero the k padding columns [M*kc, kpass) of the z tile (only LPN = 2 has any)
__device__ __forceinline__ void zero_zpad(const CoreParams& p, const Smem& s) {
    const int padw = KPASS - FGC_M * KC;
    if (padw <= 0) return;
    for (int t = threadIdx.x; t < TILE * padw; t += NTHREADS) {
        const int r = t / padw, c = t % padw;
        s.ztile[(size_t)r * ZSTRIDE + FGC_M * KC + c] = 0.f;
    }
}

// ---- phase G: MFMA over one pass ------------------------------------------------------------
// Wave w owns column tiles ct = ct0, ct0+ctstep, ... (< nct) and the k-groups [kg0, kg1).
struct WaveTiling {
    int ct0, ctstep, kparts, kpart;
};
__device__ __forceinline__ WaveTiling wave_tiling(int npad, int w) {
    const int nct = npad >> 4;
    WaveTiling t;
    t.kparts = nct >= 3 ? 1 : (nct == 2 ? 2 : 4);
    const int wpk = 4 / t.kparts;  // waves per k-part
    t.kpart = w / wpk;
    t.ct0 = w % wpk;
    t.ctstep = wpk;
    return t;
}

constexpr int CTW = 2;  // column tiles per wave (npad <= 128)
constexpr int RT = TILE / 16;

__device__ __forceinline__ void gemm_pass(const CoreParams& p, const Smem& s, int pass, const WaveTiling& wt,
                                          f32x4 (&acc)[RT][CTW]) {
    const int lane = threadIdx.x & 63;
    const int lr = lane & 15, lq = lane >> 4;
    const int nct = p.npad >> 4;
    const int kg_total = KPASS >> 4;
    // wave-uniform by construction; readfirstlane makes that provable, so the k loop is a scalar loop (no exec
    // masking) and the compiler can emit counted s_waitcnt vmcnt(N) for the fragment ring instead of vmcnt(0)
    const int kg0 = __builtin_amdgcn_readfirstlane(kg_total * wt.kpart / wt.kparts);
    const int kg1 = __builtin_amdgcn_readfirstlane(kg_total * (wt.kpart + 1) / wt.kparts);
    const f32x4* Wp4 = reinterpret_cast<const f32x4*>(p.Wp);
    const size_t wrow0 = (size_t)pass * (KPASS >> 2);
    bool ctv[CTW];
#pragma unroll
    for (int c = 0; c < CTW; ++c)
        ctv[c] = __builtin_amdgcn_readfirstlane((wt.ct0 + c * wt.ctstep) < nct ? 1 : 0) != 0;  // wave-uniform
    if (!ctv[0]) return;
    // B fragments (packed weights, L2 resident) are fetched two k-groups ahead of the MFMAs that consume them
    // UNCONDITIONAL loads (indices clamped into the packed operand): a load inside an exec-masked branch makes
    // hipcc fall back to s_waitcnt vmcnt(0), which would drain the whole prefetch ring at every k-group
    auto loadb = [&](int g, f32x4 (&b)[CTW]) {
        const int gg = min(g, kg1 - 1);
#pragma unroll
        for (int c = 0; c < CTW; ++c) {
            const int ct = min(wt.ct0 + c * wt.ctstep, nct - 1);
            b[c] = Wp4[(wrow0 + gg * 4 + lq) * p.npad + ct * 16 + lr];
        }
    };
    // Four fragment buffers with FIXED roles (the loop is unrolled by 4, no register copies: a copy of a
    // just-requested fragment would make the compiler wait for it).  Buffer u is refilled for k-group g+4 right
    // after its MFMAs were issued, i.e. three k-groups (>= 768 MFMA cycles) before it is needed again.
    f32x4 b[4][CTW];
#pragma unroll
    for (int u = 0; u < 4; ++u) loadb(kg0 + u, b[u]);
    auto loada = [&](int g, f32x4 (&a)[RT]) {
        const int gg = min(g, kg1 - 1);
#pragma unroll
        for (int r = 0; r < RT; ++r)
            a[r] = *reinterpret_cast<const f32x4*>(s.ztile + (size_t)(r * 16 + lr) * ZSTRIDE + gg * 16 + lq * 4);
    };
    auto mm = [&](const f32x4 (&a)[RT], const f32x4 (&bb)[CTW]) {
#pragma unroll
        for (int t = 0; t < 4; ++t) {
#pragma unroll
            for (int c = 0; c < CTW; ++c) {
                if (!ctv[c]) continue;
#pragma unroll
                for (int r = 0; r < RT; ++r)
                    acc[r][c] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[r][t], bb[c][t], acc[r][c], 0, 0, 0);
            }
        }
    };
    // steady state: whole groups of four, nothing conditional inside (every path issues the same loads, so the
    // compiler's s_waitcnt bookkeeping can count them); the A fragments (LDS) run one k-group ahead in two
    // alternating register sets
    f32x4 a0[RT], a1[RT];
    int g = kg0;
    loada(g, a0);
    for (; g + 4 <= kg1; g += 4) {
        loada(g + 1, a1);
        mm(a0, b[0]);
        loadb(g + 4, b[0]);
        loada(g + 2, a0);
        mm(a1, b[1]);
        loadb(g + 5, b[1]);
        loada(g + 3, a1);
        mm(a0, b[2]);
        loadb(g + 6, b[2]);
        loada(g + 4, a0);
        mm(a1, b[3]);
        loadb(g + 7, b[3]);
    }
    // remainder (< 4 k-groups): their B fragments are already in b[0..2], a0 holds group g
    if (g < kg1) {
        loada(g + 1, a1);
        mm(a0, b[0]);
    }
    if (g + 1 < kg1) {
        loada(g + 2, a0);
        mm(a1, b[1]);
    }
    if (g + 2 < kg1) mm(a0, b[2]);
}

// accumulators -> LDS out tile [kparts][TILE][oldd]; caller must have synchronised so that
// the region (aliasing ztile) is free.
__device__ __forceinline__ void store_acc(float* otile, int oldd, const WaveTiling& wt, int npad,
                                          const f32x4 (&acc)[RT][CTW]) {
    const int lane = threadIdx.x & 63;
    const int lr = lane & 15, lq = lane >> 4;
    const int nct = npad >> 4;
    float* base = otile + (size_t)wt.kpart * TILE * oldd;
#pragma unroll
    for (int c = 0; c < CTW; ++c) {
        const int ct = wt.ct0 + c * wt.ctstep;
        if (ct >= nct) continue;
#pragma unroll
        for (int r = 0; r < RT; ++r) {
#pragma unroll
            for (int t = 0; t < 4; ++t) base[(size_t)(r * 16 + lq * 4 + t) * oldd + ct * 16 + lr] = acc[r][c][t];
        }
    }
}

}  // namespace fgc
