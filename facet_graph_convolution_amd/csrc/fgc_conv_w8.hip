// Eight-wave form of the fused graph-conv core (forward conv and backward-data conv).
//
// Same algorithm, LDS layout and tile (32 nodes) as fgc_conv_core.h, but a workgroup has 512 threads = 8 waves:
// 16 lanes per node (float2 per lane in the gather, 18 accumulators instead of 36), two softmax edges per thread
// instead of three, one column tile per wave in the MFMA phase.  Registers stay under 128 per thread, so with the
// unchanged 75 KB of LDS two workgroups = 16 waves are resident per CU (4 per SIMD instead of 2): twice the waves to
// cover the gather latency and to keep the matrix pipe busy across the phase barriers.
// Used when the gathered width and the output width are multiples of 16 / powers of two (all network layers) and
// no node has more than 24 edges; everything else takes the 4-wave kernels.
#include <stdlib.h>

#include "fgc_conv_w8.h"

namespace fgc {

// developer knock-outs for phase timing (results are wrong with any bit set; never set in the shipped build):
// 1 = no MFMA instructions (operand loads kept), 2 = no aggregation FMAs (gathers kept), 4 = no soft-assignment math,
// 8 = no matrix phase at all (barriers kept), 16 = no row gathers, 32 = no output epilogue, 64 = no logit-row gathers,
// 128 = data kernel: no r stores, 256 = data kernel: no dl gathers, 512 = no packed-weight loads (MFMAs on constants)
#ifndef FGC_KO
#define FGC_KO 0
#endif
constexpr int W8_THREADS = 512;
constexpr int W8_LPN = 16;
typedef float f32x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ f32x2 load_chunk2(const CoreParams& p, int row, int cbase) {
    // channels [cbase, cbase+2) of the concatenated source row (c0, c1 even; pointers 8-byte aligned)
    const float* ptr = cbase < p.c0 ? p.src0 + (size_t)row * p.c0 + cbase
                                    : p.src1 + (size_t)row * p.c1 + (cbase - p.c0);
    f32x2 v = {0.f, 0.f};
    if (cbase < p.cg) v = *reinterpret_cast<const f32x2*>(ptr);
    return v;
}

// N edge slots of one node: row ids (with q[8]) -> N buffer loads -> N x 9 packed FMAs.  Unconditional.
template <int N, bool BF>
__device__ __forceinline__ void edge_batch(__amdgpu_buffer_rsrc_t rsrc, unsigned rowbytes, unsigned laneoff,
                                           const float* qk, f32x2 (&z)[FGC_M]) {
    f32x2 q89[N], xv[N];
    unsigned xw[N];
#pragma unroll
    for (int t = 0; t < N; ++t) q89[t] = *reinterpret_cast<const f32x2*>(qk + t * QLD + 8);   // q[8], row id
#pragma unroll
    for (int t = 0; t < N; ++t) {
        const unsigned off = __umul24((unsigned)__float_as_int(q89[t][1]), rowbytes) + laneoff;
        if constexpr (BF) xw[t] = __builtin_amdgcn_raw_buffer_load_b32(rsrc, off, 0, 0);      // two bf16 channels
        else xv[t] = __builtin_bit_cast(f32x2, __builtin_amdgcn_raw_buffer_load_b64(rsrc, off, 0, 0));
    }
    if constexpr (BF) {
#pragma unroll
        for (int t = 0; t < N; ++t) xv[t] = bf2_to_f2(xw[t]);
    }
#pragma unroll
    for (int t = 0; t < N; ++t) {
        const f32x4 q0 = *reinterpret_cast<const f32x4*>(qk + t * QLD);
        const f32x4 q1 = *reinterpret_cast<const f32x4*>(qk + t * QLD + 4);
        z[0] += q0[0] * xv[t]; z[1] += q0[1] * xv[t]; z[2] += q0[2] * xv[t]; z[3] += q0[3] * xv[t];
        z[4] += q1[0] * xv[t]; z[5] += q1[1] * xv[t]; z[6] += q1[2] * xv[t]; z[7] += q1[3] * xv[t];
        z[8] += q89[t][0] * xv[t];
    }
}

// PIPE form (FAST, 16 slots): the rows of a pass are requested one phase ahead of their FMAs - pass 0 together with the
// logit rows of the soft assignment, pass p + 1 before the matrix phase of pass p - so that a tile's chain of dependent
// memory round trips is rowptr -> col -> (logit rows | rows of pass 0) instead of one more trip per 8-slot batch and
// pass.  Slots [T0, T0 + N) of one node; `raw` holds the rows as loaded (fp32 pair, or one dword of two bf16).
template <int T0, int N, bool BF>
__device__ __forceinline__ void gather_rows(__amdgpu_buffer_rsrc_t rsrc, unsigned rowbytes, unsigned laneoff,
                                            const float* qb, f32x2 (&raw)[16]) {
    unsigned rid[N];
#pragma unroll
    for (int t = 0; t < N; ++t) rid[t] = (unsigned)__float_as_int(qb[(T0 + t) * QLD + 9]);
#pragma unroll
    for (int t = 0; t < N; ++t) {
        const unsigned off = __umul24(rid[t], rowbytes) + laneoff;
        if (FGC_KO & 16) { raw[T0 + t] = f32x2{__uint_as_float(off), 1.f}; continue; }
        if constexpr (BF) raw[T0 + t][0] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsrc, off, 0, 0));
        else raw[T0 + t] = __builtin_bit_cast(f32x2, __builtin_amdgcn_raw_buffer_load_b64(rsrc, off, 0, 0));
    }
}
template <int T0, int N, bool BF>
__device__ __forceinline__ void fma_rows(const float* qb, const f32x2 (&raw)[16], f32x2 (&z)[FGC_M]) {
#pragma unroll
    for (int t = 0; t < N; ++t) {
        const float* qk = qb + (T0 + t) * QLD;
        f32x2 xv;
        if constexpr (BF) xv = bf2_to_f2(__builtin_bit_cast(unsigned, raw[T0 + t][0]));
        else xv = raw[T0 + t];
        if (FGC_KO & 2) { z[t % FGC_M] += xv; continue; }
        const f32x4 q0 = *reinterpret_cast<const f32x4*>(qk);
        const f32x4 q1 = *reinterpret_cast<const f32x4*>(qk + 4);
        const float q8 = qk[8];
        z[0] += q0[0] * xv; z[1] += q0[1] * xv; z[2] += q0[2] * xv; z[3] += q0[3] * xv;
        z[4] += q1[0] * xv; z[5] += q1[1] * xv; z[6] += q1[2] * xv; z[7] += q1[3] * xv;
        z[8] += q8 * xv;
    }
}
// the 16 lanes that write and read a node's q table sit in one wave, whose LDS accesses execute in order: ordering the
// compiler is all that is needed between the soft-assignment and the aggregation phase (no s_barrier)
__device__ __forceinline__ void wave_lds_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// FAST: every pass gathers 32 valid channels from ONE source (cg % 32 == 0 and the concat boundary on a pass boundary:
// all network layers except conv1).  Then the edge loop is branch-free: qbuf is padded with zero-weight edges up to
// the wave's largest degree rounded up to 8, all lanes run the same (scalar) trip count, rows are fetched with
// buffer loads whose 32-bit offset is one v_mad_u32_u24 from the row id, and the row id travels with q[8] in one
// ds_read_b64.  The generic form keeps per-lane degree tests (exec masking) and 64-bit addressing.
// QS: edge slots per node kept in LDS - 16 when the host knows that no node has more edges (each of the 16 softmax lanes
// of a node then owns one slot instead of two, and the tile needs 12 KB less LDS), KMAX otherwise
// BF: bf16 storage (FGC_CONV_BF16).  The gathered rows, y / y_pool (forward), r and dx (data gradient) are bf16; the
// aggregate tile is kept in LDS as bf16 and the tile product runs on v_mfma_f32_16x16x32_bf16 (fp32 accumulators).  The
// soft assignment, the aggregation FMAs and every epilogue stay fp32.  FAST shapes only.
// NT: nodes per workgroup, 32 (eight waves) or 16 (four waves: a half tile each; twice the workgroups, four instead of two
// resident per CU at the same 16 waves, and twice the packed-weight traffic per node).  NT = 16 needs npad <= 64.
// EROW (data gradient of the pair form, fgc_conv_pair.hip): the gathered operand has one row per EDGE of the forward graph
// (the per-pair dt rows), so the row of slot k is the edge id p.eid[e], not the neighbour col[e] >> shift.
// H2 (forward, fp32, half tiles; option W8_HALF2): the aggregate tile is written and consumed in TWO halves of the assignment
// index - assignments 0-4 (160 k), then 5-8 (128 k) - so it needs 10.75 KB instead of 18.9: 23.4 KB per workgroup, a SIXTH
// resident workgroup per CU if the registers stay at 80 (the launch bound asks for six waves per SIMD), for two more
// barriers per pass.  Round-3 candidate "resident workgroups are the currency"; measured in DESIGN.md section 10.
template <bool DATA, bool FAST, int QS, bool BF = false, int NT = 32, bool EROW = false, bool H2 = false>
__global__ __launch_bounds__(NT * 16, H2 ? 6 : 4) void conv_w8_kernel(CoreParams p, FwdEpilogue fe, DataEpilogue de) {
    static_assert(!H2 || (!DATA && !BF && NT == 16), "two-half aggregate tile: the fp32 forward kernel on half tiles");
    constexpr int ZS = H2 ? 168 : ZSTRIDE;             // floats per node of the (half) aggregate tile: == 8 mod 16
    static_assert(!BF || FAST, "the bf16 form exists for the fast shapes only");
    static_assert(!EROW || (DATA && FAST), "rows by edge id: the data-gradient kernel's fast shapes");
    static_assert(NT == 32 || (NT == 16 && FAST && QS == 16), "half tiles: the pipelined 16-slot form only");
    constexpr int TILE = NT, RT = NT / 16, NW = NT / 4, LW = NT == 32 ? 3 : 2, W8_THREADS = NT * 16;   // (shadow the 32-node constants)
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    const Smem s = carve(smem_raw, BF ? ZSTRIDE_BF / 2 : ZS, QS, NT);
    constexpr int SPL = QS / 16 + (QS % 16 ? 1 : 0);   // slots per softmax lane: k = kl + 16 * t
    constexpr bool PIPE = FAST && QS == 16;            // rows requested one phase ahead (gather_rows / fma_rows)
    // DATA: the node's da | dg rows for the epilogue live in the two spare floats of the node's first nine edge slots
    // (qbuf[node][m][10] = da[m], [11] = dg[m]): no LDS of their own, which is what lets a fifth half-tile workgroup fit
    auto dag_slot = [&](int nd, int m) { return s.qbuf + (size_t)nd * qnode_stride(QS) + m * QLD + 10; };
    int tile0;
    if constexpr (NT == 32) {
        tile0 = block_tile0(p);
    } else {   // half tiles: workgroup h of 2 * tiles; the XCD map runs over half tiles
        const int h = xcd_tile(blockIdx.x, gridDim.x);
        tile0 = (p.tile_list ? p.tile_list[h >> 1] : (h >> 1)) * 32 + (h & 1) * 16;
    }
    const int tid = threadIdx.x;
    const int node = tid >> 4, kl = tid & 15;
    const int wave = tid >> 6, lane = tid & 63, lr = lane & 15, lq = lane >> 4;
    // ---------------- MFMA tiling: wave w owns column tile (w % nct) and k-part (w / nct)
    const int nct = p.npad >> 4;              // 1, 2, 4 or 8
    const int nsh = 31 - __builtin_clz(nct);
    const int kparts = NW >> nsh;
    const int ct = __builtin_amdgcn_readfirstlane(wave & (nct - 1));
    const int kpart = __builtin_amdgcn_readfirstlane(wave >> nsh);
    // units of the reduction index per pass: 16-deep k-groups (fp32, four MFMAs deep) or 32-deep k-steps (bf16)
    constexpr int UPP = BF ? KPASS / 32 : KPASS / 16;
    const int u0 = __builtin_amdgcn_readfirstlane((UPP * kpart) >> (LW - nsh));
    const int u1 = __builtin_amdgcn_readfirstlane((UPP * (kpart + 1)) >> (LW - nsh));
    // packed weights through a buffer descriptor: the lane's part of the offset is computed once, the unit's part is
    // scalar (the plain indexed form spent two 64-bit multiplies per fragment load on the vector ALU)
    //   fp32: [unit][4 rows of float4: lq][npad columns]     bf16: [unit][column tile][lane] x 8 bf16
    const __amdgpu_buffer_rsrc_t w_rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.Wp), 0, -1, 0x00020000);
    const unsigned w_lane = BF ? (unsigned)((ct * 64 + lane) * 16) : (unsigned)((lq * p.npad + ct * 16 + lr) * 16);
    const unsigned w_unit = BF ? (unsigned)(nct * 1024) : (unsigned)(p.npad * 64);
    auto loadw = [&](int pass, int u) {
        const int uu = min(u, u1 - 1);
        if (FGC_KO & 512) return u32x4{(unsigned)uu, (unsigned)pass, 0x3f800000u, 0x3f000000u};   // (no packed-weight loads)
        return __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(w_rs, w_lane, (unsigned)(pass * UPP + uu) * w_unit, 0));
    };
    f32x2 xa[16];                                      // PIPE: the rows in flight
    const float* qb = s.qbuf + (size_t)node * qnode_stride(QS);
    // PIPE: request this lane's two channels of the node's neighbour rows for a pass (slots 0-7, and 8-11 / 12-15 when
    // a node of the wave has that many edges; every slot of the table holds a valid row id)
    auto issue = [&](int pass, int dw) {
        constexpr unsigned ESZ = BF ? 2u : 4u;
        const bool first = pass * KC < p.c0;                                   // wave-uniform
        const float* base = first ? p.src0 : p.src1;
        const unsigned rowbytes = (unsigned)(first ? p.c0 : p.c1) * ESZ;
        const unsigned laneoff = (unsigned)(pass * KC + 2 * kl - (first ? 0 : p.c0)) * ESZ;
        const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(base), 0, -1, 0x00020000);
        gather_rows<0, 8, BF>(rsrc, rowbytes, laneoff, qb, xa);
        // (13 is the degree of a regular triangle mesh's facet graph - 12 neighbours and the facet itself: slot 12 on its own
        //  instead of a batch of four saves 3 of 16 row requests and 27 of 144 packed FMAs per lane and pass there)
        if (dw > 8) gather_rows<8, 4, BF>(rsrc, rowbytes, laneoff, qb, xa);
        if (dw > 12) gather_rows<12, 1, BF>(rsrc, rowbytes, laneoff, qb, xa);
        if (dw > 13) gather_rows<13, 1, BF>(rsrc, rowbytes, laneoff, qb, xa);
        if (dw > 14) gather_rows<14, 2, BF>(rsrc, rowbytes, laneoff, qb, xa);
    };

    // ---------------- phase S: per-edge soft assignment (edges kl and kl + 16 of this thread's node)
    int dwave = 0;  // FAST: wave-uniform trip count of the edge loop
    float dgsum[FGC_M];
#pragma unroll
    for (int m = 0; m < FGC_M; ++m) dgsum[m] = 0.f;
    {
        const int i = tile0 + node;
        int d = 0, e0 = 0;
        float ctr[FGC_M];
#pragma unroll
        for (int m = 0; m < FGC_M; ++m) ctr[m] = 0.f;
        // the logit table and the edge list through buffer descriptors: a gather then costs one v_mad_u32_u24 for its
        // 32-bit offset instead of a 64-bit multiply-add chain (the fp32 MFMA and the vector ALU share a SIMD's issue:
        // every vector instruction saved here is matrix time gained)
        const __amdgpu_buffer_rsrc_t ag_rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.ag), 0, -1, 0x00020000);
        const __amdgpu_buffer_rsrc_t col_rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<int*>(p.col), 0, -1, 0x00020000);
        if (i < p.n) {
            e0 = p.rowptr[i];
            d = min(p.rowptr[i + 1] - e0, QS);
            const unsigned ao = __umul24((unsigned)(i >> p.ag_shift), FGC_AG_LD * 4u) + (unsigned)p.ctr_off * 4u;
            const f32x4 a0 = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(ag_rs, ao, 0, 0));
            const f32x4 a1 = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(ag_rs, ao + 16u, 0, 0));
            ctr[0] = a0[0]; ctr[1] = a0[1]; ctr[2] = a0[2]; ctr[3] = a0[3];
            ctr[4] = a1[0]; ctr[5] = a1[1]; ctr[6] = a1[2]; ctr[7] = a1[3];
            ctr[8] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(ag_rs, ao + 32u, 0, 0));
        }
        // DATA: the node's da row (written by the d-logits kernel) is needed only after the dl sums below; asked for
        // here it costs no extra memory round trip
        f32x4 da0 = {0.f, 0.f, 0.f, 0.f}, da1 = {0.f, 0.f, 0.f, 0.f};
        float da8 = 0.f;
        if (DATA && kl == 0 && i < p.n) {
            const float* dr = de.dag + (size_t)i * FGC_AG_LD;
            da0 = *reinterpret_cast<const f32x4*>(dr);
            da1 = *reinterpret_cast<const f32x4*>(dr + 4);
            da8 = dr[8];
        }
        if (kl == 0) {
            s.deg[node] = d;
            // 1 / degree for the epilogue, divided once per node here instead of once per output element there
            s.deg[TILE + 4 + node] = __float_as_int(d > 0 ? 1.0f / (float)d : 0.f);
        }
        if (FAST) {  // largest degree among the 4 nodes of this wave, rounded up to a whole batch of 8 edge slots
            const int dmax = max(max(__builtin_amdgcn_readlane(d, 0), __builtin_amdgcn_readlane(d, 16)),
                                 max(__builtin_amdgcn_readlane(d, 32), __builtin_amdgcn_readlane(d, 48)));
            dwave = dmax;
        }
        const int dfill = (dwave + 7) & ~7;
        int jj[SPL], er[SPL];
        f32x4 g0[SPL], g1[SPL];
        float g8[SPL];
#pragma unroll
        for (int t = 0; t < SPL; ++t) {
            const int k = kl + 16 * t;
            // (unconditional, clamped into the node's list: no exec-masked load.  A node WITHOUT edges reads the entry in
            //  front of its empty list: for the padding nodes at the end of a level e0 == nnz, one past the array)
            const unsigned eo = (unsigned)(d > 0 ? e0 + min(k, d - 1) : max(e0 - 1, 0)) * 4u;
            const int jv = __builtin_amdgcn_raw_buffer_load_b32(col_rs, eo, 0, 0);
            jj[t] = k < d ? jv : 0;
            if constexpr (EROW) {
                const __amdgpu_buffer_rsrc_t eid_rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<int*>(p.eid), 0, -1, 0x00020000);
                const int ev = __builtin_amdgcn_raw_buffer_load_b32(eid_rs, eo, 0, 0);
                er[t] = k < d ? ev : 0;
            } else {
                er[t] = jj[t] >> p.shift;
            }
        }
#pragma unroll
        for (int t = 0; t < SPL; ++t) {
            const unsigned go = __umul24((unsigned)(jj[t] >> p.ag_shift), FGC_AG_LD * 4u) + (unsigned)p.nbr_off * 4u;
            if (FGC_KO & 64) { g0[t] = g1[t] = f32x4{__uint_as_float(go), 0.f, 1.f, 2.f}; g8[t] = 0.f; continue; }
            g0[t] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(ag_rs, go, 0, 0));
            g1[t] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(ag_rs, go + 16u, 0, 0));
            g8[t] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(ag_rs, go + 32u, 0, 0));
        }
        if constexpr (PIPE) {
            // the row ids go to the table first (slots past the degree: row 0 with weight zero), and the rows of pass 0
            // leave right behind the logit rows
            s.qbuf[(size_t)node * qnode_stride(QS) + kl * QLD + 9] = __int_as_float(er[0]);
            wave_lds_sync();
            issue(0, dwave);
        }
#pragma unroll
        for (int t = 0; t < SPL; ++t) {
            const int k = kl + 16 * t;
            if (k >= d) {
                if (FAST && (PIPE || k < dfill)) {  // zero-weight slot pointing at a valid row
                    float* q = s.qbuf + (size_t)node * qnode_stride(QS) + k * QLD;
                    *reinterpret_cast<f32x4*>(q) = f32x4{0.f, 0.f, 0.f, 0.f};
                    *reinterpret_cast<f32x4*>(q + 4) = f32x4{0.f, 0.f, 0.f, 0.f};
                    q[8] = 0.f;
                    if (!PIPE) q[9] = __int_as_float(0);
                }
                continue;
            }
            float l[FGC_M];
            l[0] = ctr[0] + g0[t][0]; l[1] = ctr[1] + g0[t][1]; l[2] = ctr[2] + g0[t][2]; l[3] = ctr[3] + g0[t][3];
            l[4] = ctr[4] + g1[t][0]; l[5] = ctr[5] + g1[t][1]; l[6] = ctr[6] + g1[t][2]; l[7] = ctr[7] + g1[t][3];
            l[8] = ctr[8] + g8[t];
            float mx = l[0];
            if (FGC_KO & 4) {
                float* q = s.qbuf + (size_t)node * qnode_stride(QS) + k * QLD;
                *reinterpret_cast<f32x4*>(q) = f32x4{l[0], l[1], l[2], l[3]};
                *reinterpret_cast<f32x4*>(q + 4) = f32x4{l[4], l[5], l[6], l[7]};
                q[8] = l[8];
                continue;
            }
#pragma unroll
            for (int m = 1; m < FGC_M; ++m) mx = fmaxf(mx, l[m]);
            float sum = 0.f;
            const float nmx = -mx * 1.4426950408889634f;
#pragma unroll
            for (int m = 0; m < FGC_M; ++m) {
                l[m] = __builtin_amdgcn_exp2f(fmaf(l[m], 1.4426950408889634f, nmx));   // exp(l - mx): one fma + v_exp_f32
                sum += l[m];
            }
            const float inv = 1.0f / sum;
            float* q = s.qbuf + (size_t)node * qnode_stride(QS) + k * QLD;
            *reinterpret_cast<f32x4*>(q) = f32x4{l[0] * inv, l[1] * inv, l[2] * inv, l[3] * inv};
            *reinterpret_cast<f32x4*>(q + 4) = f32x4{l[4] * inv, l[5] * inv, l[6] * inv, l[7] * inv};
            q[8] = l[8] * inv;
            if (!PIPE) q[9] = __int_as_float(er[t]);
            if (DATA && !(FGC_KO & 256)) {
                const __amdgpu_buffer_rsrc_t dl_rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(de.dl), 0, -1, 0x00020000);
                const unsigned dof = __umul24((unsigned)(EROW ? er[t] : p.eid[e0 + k]), FGC_DL_LD * 4u);
                const f32x4 d0 = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(dl_rs, dof, 0, 0));
                const f32x4 d1 = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(dl_rs, dof + 16u, 0, 0));
                dgsum[0] += d0[0]; dgsum[1] += d0[1]; dgsum[2] += d0[2]; dgsum[3] += d0[3];
                dgsum[4] += d1[0]; dgsum[5] += d1[1]; dgsum[6] += d1[2]; dgsum[7] += d1[3];
                dgsum[8] += __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(dl_rs, dof + 32u, 0, 0));
            }
        }
        if (DATA) {  // dg_j = sum over in-edges of dl: reduce the 16 softmax lanes of the node
#pragma unroll
            for (int m = 0; m < FGC_M; ++m) {
                float v = dgsum[m];
                FGC_ROW16_SUM(v);
                dgsum[m] = v;
            }
            if (kl == 0) {
                if (i < p.n) {
                    const float da[FGC_M] = {da0[0], da0[1], da0[2], da0[3], da1[0], da1[1], da1[2], da1[3], da8};
#pragma unroll
                    for (int m = 0; m < FGC_M; ++m) *reinterpret_cast<f32x2*>(dag_slot(node, m)) = f32x2{da[m], dgsum[m]};
                    float* o = de.dag + (size_t)i * FGC_AG_LD + 12;
                    *reinterpret_cast<f32x4*>(o) = f32x4{dgsum[0], dgsum[1], dgsum[2], dgsum[3]};
                    *reinterpret_cast<f32x4*>(o + 4) = f32x4{dgsum[4], dgsum[5], dgsum[6], dgsum[7]};
                    *reinterpret_cast<f32x4*>(o + 8) = f32x4{dgsum[8], 0.f, 0.f, 0.f};
                    // da | dg behind the node's r row: [du; dv] = (da | dg)^T x rides in the dW0 GEMM
                    if constexpr (BF) {
                        u32x2* rt = reinterpret_cast<u32x2*>(reinterpret_cast<unsigned short*>(de.r) + (size_t)i * de.rld +
                                                             (FGC_M * p.cg));
                        rt[0] = f4_to_bf4(f32x4{da[0], da[1], da[2], da[3]});
                        rt[1] = f4_to_bf4(f32x4{da[4], da[5], da[6], da[7]});
                        rt[2] = f4_to_bf4(f32x4{da[8], 0.f, 0.f, 0.f});
                        rt[3] = f4_to_bf4(f32x4{dgsum[0], dgsum[1], dgsum[2], dgsum[3]});
                        rt[4] = f4_to_bf4(f32x4{dgsum[4], dgsum[5], dgsum[6], dgsum[7]});
                        rt[5] = f4_to_bf4(f32x4{dgsum[8], 0.f, 0.f, 0.f});
                    } else {
                    float* rt = de.r + (size_t)i * de.rld + (FGC_M * p.cg);
                    *reinterpret_cast<f32x4*>(rt) = f32x4{da[0], da[1], da[2], da[3]};
                    *reinterpret_cast<f32x4*>(rt + 4) = f32x4{da[4], da[5], da[6], da[7]};
                    *reinterpret_cast<f32x4*>(rt + 8) = f32x4{da[8], 0.f, 0.f, 0.f};
                    *reinterpret_cast<f32x4*>(rt + 12) = f32x4{dgsum[0], dgsum[1], dgsum[2], dgsum[3]};
                    *reinterpret_cast<f32x4*>(rt + 16) = f32x4{dgsum[4], dgsum[5], dgsum[6], dgsum[7]};
                    *reinterpret_cast<f32x4*>(rt + 20) = f32x4{dgsum[8], 0.f, 0.f, 0.f};
                    }
                }   // (rows past n are skipped by the epilogue)
            }
        }
    }
    if constexpr (PIPE) wave_lds_sync();      // a node's table is written and read by the 16 lanes of one wave
    else __syncthreads();

    f32x4 acc[RT];
#pragma unroll
    for (int r = 0; r < RT; ++r) acc[r] = f32x4{0.f, 0.f, 0.f, 0.f};
    const bool want_gemm = !DATA || de.dx0 != nullptr;

    const int cl = kl;
    const int d = s.deg[node];
    auto do_pass = [&](int pass) {
        // ---------------- phase A: z[m][2] = sum_k q[k][m] * x_j(k)[2]; every row is requested before the first FMA
        f32x2 z[FGC_M];
#pragma unroll
        for (int m = 0; m < FGC_M; ++m) z[m] = f32x2{0.f, 0.f};
        const int cbase = pass * KC + 2 * cl;
        if constexpr (PIPE) {
            fma_rows<0, 8, BF>(qb, xa, z);
            if (dwave > 8) fma_rows<8, 4, BF>(qb, xa, z);
            if (dwave > 12) fma_rows<12, 1, BF>(qb, xa, z);
            if (dwave > 13) fma_rows<13, 1, BF>(qb, xa, z);
            if (dwave > 14) fma_rows<14, 2, BF>(qb, xa, z);
            // the next pass' rows travel under this pass' matrix phase
            if (pass + 1 < p.passes) issue(pass + 1, dwave);
        } else if (FAST) {
            const bool first = pass * KC < p.c0;                                   // wave-uniform
            const float* base = first ? p.src0 : p.src1;
            constexpr unsigned ESZ = BF ? 2u : 4u;                                  // bytes per stored channel
            const unsigned rowbytes = (unsigned)(first ? p.c0 : p.c1) * ESZ;
            const unsigned laneoff = (unsigned)(first ? cbase : cbase - p.c0) * ESZ;
            const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(base), 0, -1, 0x00020000);
            // whole batches of 8 edge slots, then the remainder rounded up to a pair (its own straight-line code:
            // every row of a batch is requested before the first FMA, and nothing in a batch is conditional)
            int k0 = 0;
            for (; k0 + 8 <= dwave; k0 += 8) edge_batch<8, BF>(rsrc, rowbytes, laneoff, qb + k0 * QLD, z);
            const int rem = dwave - k0;
            if (rem > 4) {
                if (rem > 6) edge_batch<8, BF>(rsrc, rowbytes, laneoff, qb + k0 * QLD, z);
                else edge_batch<6, BF>(rsrc, rowbytes, laneoff, qb + k0 * QLD, z);
            } else if (rem > 0) {
                if (rem > 2) edge_batch<4, BF>(rsrc, rowbytes, laneoff, qb + k0 * QLD, z);
                else edge_batch<2, BF>(rsrc, rowbytes, laneoff, qb + k0 * QLD, z);
            }
        } else
        for (int k0 = 0; k0 < d; k0 += RB) {
            f32x2 xv[RB];
#pragma unroll
            for (int t = 0; t < RB; ++t) {
                xv[t] = f32x2{0.f, 0.f};
                if (k0 + t < d) xv[t] = load_chunk2(p, __float_as_int(qb[(k0 + t) * QLD + 9]), cbase);
            }
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
        }        if (DATA) {  // r[j, m*cout + channel] straight from the accumulators
            const int j = tile0 + node;
            if (!(FGC_KO & 128) && j < p.n && cbase < p.cg) {
                if constexpr (BF) {
                    unsigned* rr = reinterpret_cast<unsigned*>(reinterpret_cast<unsigned short*>(de.r) + (size_t)j * de.rld + cbase);
#pragma unroll
                    for (int m = 0; m < FGC_M; ++m) rr[(m * p.cg) >> 1] = f2_to_bf2(z[m][0], z[m][1]);
                } else {
                float* rr = de.r + (size_t)j * de.rld + cbase;
#pragma unroll
                for (int m = 0; m < FGC_M; ++m) *reinterpret_cast<f32x2*>(rr + m * p.cg) = z[m];
                }
            }
            if (!want_gemm) return;
        }
        // fp32: the k-groups [gl, gh) of this pass against the tile whose first column is k-group gz
        auto gphase32 = [&](int gl, int gh, int gz) {
        auto loadw = [&](int pass_, int u) {
            const int uu = min(u, gh - 1);
            if (FGC_KO & 512) return u32x4{(unsigned)uu, (unsigned)pass_, 0x3f800000u, 0x3f000000u};   // (no packed-weight loads)
            return __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(w_rs, w_lane, (unsigned)(pass_ * UPP + uu) * w_unit, 0));
        };
        // A fragments: one ds_read_b128 per row tile (4 k of a 16-deep group)
        auto loada = [&](int g, f32x4 (&a)[RT]) {
            const int gg = min(g, gh - 1) - gz;
#pragma unroll
            for (int r = 0; r < RT; ++r)
                a[r] = *reinterpret_cast<const f32x4*>(s.ztile + (size_t)(r * 16 + lr) * ZS + gg * 16 + lq * 4);
        };
        auto mm = [&](const f32x4 (&a)[RT], const u32x4& bw) {
            const f32x4 b = __builtin_bit_cast(f32x4, bw);
#pragma unroll
            for (int t = 0; t < 4; ++t)
#pragma unroll
                for (int r = 0; r < RT; ++r) {
                    if (FGC_KO & 1) asm volatile("" ::"v"(a[r][t]), "v"(b[t]));
                    else acc[r] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[r][t], b[t], acc[r], 0, 0, 0);
                }
        };
            int g = gl;
            u32x4 b0 = loadw(pass, g), b1 = loadw(pass, g + 1), b2 = loadw(pass, g + 2), b3 = loadw(pass, g + 3);
            f32x4 a0[RT], a1[RT];
            loada(g, a0);
            for (; g + 4 <= gh; g += 4) {
                loada(g + 1, a1);
                mm(a0, b0);
                b0 = loadw(pass, g + 4);
                loada(g + 2, a0);
                mm(a1, b1);
                b1 = loadw(pass, g + 5);
                loada(g + 3, a1);
                mm(a0, b2);
                b2 = loadw(pass, g + 6);
                loada(g + 4, a0);
                mm(a1, b3);
                b3 = loadw(pass, g + 7);
            }
            if (g < gh) {
                loada(g + 1, a1);
                mm(a0, b0);
            }
            if (g + 1 < gh) {
                loada(g + 2, a0);
                mm(a1, b1);
            }
            if (g + 2 < gh) mm(a0, b2);
        };
        if constexpr (H2) {
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                constexpr int M0[2] = {0, 5}, NM[2] = {5, 4};
                if (pass > 0 || h > 0) __syncthreads();  // the previous half's MFMA reads of the tile are done
                float* zr = s.ztile + (size_t)node * ZS + 2 * cl;
#pragma unroll
                for (int m = 0; m < NM[h]; ++m) *reinterpret_cast<f32x2*>(zr + m * KC) = z[M0[h] + m];
                __syncthreads();
                if (FGC_KO & 8) continue;
                // this half's k-groups (10, then 8), split over the k-parts like a whole pass
                const int gb = M0[h] * 2, ng = NM[h] * 2;
                const int gl = __builtin_amdgcn_readfirstlane(gb + ((ng * kpart) >> (LW - nsh)));
                const int gh = __builtin_amdgcn_readfirstlane(gb + ((ng * (kpart + 1)) >> (LW - nsh)));
                gphase32(gl, gh, gb);
            }
            return;
        }
        if (pass > 0) __syncthreads();  // previous pass' MFMA reads of ztile are done
        if constexpr (BF) {
            unsigned* zr = reinterpret_cast<unsigned*>(s.ztile) + (size_t)node * (ZSTRIDE_BF / 2) + cl;
#pragma unroll
            for (int m = 0; m < FGC_M; ++m) zr[m * (KC / 2)] = f2_to_bf2(z[m][0], z[m][1]);
        } else {
            float* zr = s.ztile + (size_t)node * ZSTRIDE + 2 * cl;
#pragma unroll
            for (int m = 0; m < FGC_M; ++m) *reinterpret_cast<f32x2*>(zr + m * KC) = z[m];
        }
        __syncthreads();
        // ---------------- phase G: acc[32 x 16] += ztile[32 x k-part] * Wp[k-part x 16]
        if (FGC_KO & 8) return;
        const char* zb = reinterpret_cast<const char*>(s.ztile);
        auto mmb = [&](int ks, const u32x4& b) {
#pragma unroll
            for (int r = 0; r < RT; ++r) {
                const u32x4 a = *reinterpret_cast<const u32x4*>(zb + (size_t)(r * 16 + lr) * (ZSTRIDE_BF * 2) + ks * 64 + lq * 16);
                acc[r] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b),
                                                                acc[r], 0, 0, 0);
            }
        };
        if constexpr (BF) {
            // nine 32-deep k-steps per pass, split over the wave's k-part
            int ks = u0;
            u32x4 b0 = loadw(pass, ks), b1 = loadw(pass, ks + 1);
            for (; ks + 2 <= u1; ks += 2) {
                mmb(ks, b0);
                b0 = loadw(pass, ks + 2);
                mmb(ks + 1, b1);
                b1 = loadw(pass, ks + 3);
            }
            if (ks < u1) mmb(ks, b0);
        } else {
            gphase32(u0, u1, 0);
        }
    };
    for (int pass = 0; pass < p.passes; ++pass) do_pass(pass);
    if (!want_gemm) return;
    if (FGC_KO & 32) { if (acc[0][0] == 123.f && acc[RT - 1][3] == 5.f) s.ztile[tid] = acc[0][1]; return; }
    __syncthreads();
    // ---------------- accumulators -> LDS (aliases ztile), k-parts summed in fixed order by the epilogue
    const int oldd = p.npad + 4;
    float* otile = s.ztile;
    {
        float* base = otile + (size_t)kpart * TILE * oldd;
#pragma unroll
        for (int r = 0; r < RT; ++r)
#pragma unroll
            for (int t = 0; t < 4; ++t) base[(size_t)(r * 16 + lq * 4 + t) * oldd + ct * 16 + lr] = acc[r][t];
    }
    __syncthreads();
    if (!DATA) {
        // thread -> (pooled row group, column): W8_THREADS is a multiple of every supported width, so a thread keeps its
        // column over the walk and the bias is read once
        // (FAST: the width is 16, 32, 64 or 128: shifts instead of the ~40-instruction integer divisions)
        const int osh = 31 - __builtin_clz(p.nout);
        const bool pow2 = FAST && (p.nout & (p.nout - 1)) == 0;
        const int o = pow2 ? (tid & (p.nout - 1)) : tid % p.nout, pstep = pow2 ? (W8_THREADS >> osh) : W8_THREADS / p.nout;
        const float bias_o = fe.bias[o];
        // (threads past the last whole row of columns sit out: widths that do not divide W8_THREADS)
        for (int pr = tid < pstep * p.nout ? (pow2 ? tid >> osh : tid / p.nout) : TILE; pr < TILE / 4; pr += pstep) {
            float mx = -INFINITY;
            bool any = false;
            const size_t ybase = (size_t)(tile0 + pr * 4) * p.nout + o;   // (one 64-bit multiply for the four rows)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int row = pr * 4 + q;
                const int i = tile0 + row;
                if (i >= p.n) continue;
                float val = 0.f;
                for (int kp = 0; kp < kparts; ++kp) val += otile[((size_t)kp * TILE + row) * oldd + o];
                const int dd = s.deg[row];
                // deg was clamped to KMAX for the edge loops; the true degree equals it here (host guarantees <= 24)
                val *= __int_as_float(s.deg[TILE + 4 + row]);
                if (!fe.bias_mask || dd > 0) val += bias_o;
                if (fe.act) val = fmaxf(val, 0.f) - fe.alpha * fmaxf(-val, 0.f);
                st_act(fe.y, ybase + (size_t)(q * p.nout), val, BF);
                mx = fmaxf(mx, val);
                any = true;
            }
            if (fe.y_pool && any) st_act(fe.y_pool, (size_t)((tile0 >> 2) + pr) * p.nout + o, mx, BF);
        }
    } else {
        const int group = 1 << de.shiftf;
        const int nsrc = TILE / group;
        // thread -> (source row, input channel): the channel stays put over the walk (W8_THREADS is a multiple of every
        // supported width), so the 18 u / v entries of the logit term are read once per thread, not once per element
        const int csh = 31 - __builtin_clz(de.cin);
        const bool pow2 = FAST && (de.cin & (de.cin - 1)) == 0;
        const int c = pow2 ? (tid & (de.cin - 1)) : tid % de.cin, sstep = pow2 ? (W8_THREADS >> csh) : W8_THREADS / de.cin;
        float uc[FGC_M], vc[FGC_M];
#pragma unroll
        for (int m = 0; m < FGC_M; ++m) {
            uc[m] = de.u[m * de.cin + c];
            vc[m] = de.v[m * de.cin + c];
        }
        for (int sr = tid < sstep * de.cin ? (pow2 ? tid >> csh : tid / de.cin) : nsrc; sr < nsrc; sr += sstep) {
            float val = 0.f;
            bool any = false;
            for (int q = 0; q < group; ++q) {
                const int row = sr * group + q;
                if (tile0 + row >= p.n) continue;
                any = true;
                float g = 0.f;
                for (int kp = 0; kp < kparts; ++kp) g += otile[((size_t)kp * TILE + row) * oldd + c];
#pragma unroll
                for (int m = 0; m < FGC_M; ++m) {
                    const f32x2 dd = *reinterpret_cast<const f32x2*>(dag_slot(row, m));
                    g = fmaf(dd[0], uc[m], g);
                    g = fmaf(dd[1], vc[m], g);
                }
                val += g;
            }
            if (!any) continue;
            const size_t srow = (size_t)((tile0 >> de.shiftf) + sr);
            if (c < de.c0f) {
                const size_t o = srow * de.c0f + c;
                st_act(de.dx0, o, de.acc0 ? ld_act(de.dx0, o, BF) + val : val, BF);
            } else if (de.dx1) {
                const size_t o = srow * de.c1f + (c - de.c0f);
                st_act(de.dx1, o, de.acc1 ? ld_act(de.dx1, o, BF) + val : val, BF);
            }
        }
    }
}

bool w8_supported(const CoreParams& p, int max_deg) {
    if (opt(OPT_NO_W8) == 1) return false;
    const int nct = p.npad >> 4;
    if (!(nct == 1 || nct == 2 || nct == 4 || nct == 8)) return false;
    if (max_deg <= 0 || max_deg > KMAX) return false;
    if (p.kc != 32) return false;
    if ((p.c0 & 1) || (p.c1 & 1)) return false;
    if (((uintptr_t)p.src0 & 7) || (p.src1 && ((uintptr_t)p.src1 & 7))) return false;
    return true;
}

static bool w8_fast(const CoreParams& p) {
    if (opt(OPT_NO_W8FAST) == 1) return false;
    return p.cg % 32 == 0 && (p.c1 == 0 || p.c0 % 32 == 0) && (size_t)p.n * 4 * 128 < 0xFFFFFFFFull;
}

// Half tiles (16 nodes, four waves per workgroup, five workgroups per CU at 31 KB of LDS each): the forward kernel of
// layers up to 64 outputs wide, and the data-gradient kernel of such layers on big levels.  Measured on the 100k-facet
// mesh: level-0 forward 98.8 -> 85.6 us (independently phased small workgroups overlap their gather, aggregate and matrix
// phases better than two big ones, and that outweighs reading the packed weights twice as often), 64-wide level-1 forward
// 62.1 -> 60.5.  The data-gradient kernel lost at four workgroups per CU (126.4 -> 129.6: its da | dg rows cost 1.5 KB
// of LDS, one workgroup fewer than the forward kernel); with those rows moved into spare floats of the edge table it
// runs five as well: level 0 128.6 / 112.5 -> 121.5 / 103.4 us, but the small coarse levels (a few hundred workgroups)
// got slower (level 2: 29.6 -> 32.3), so it switches at FGC_W8_DATA16_MIN_N nodes (default 81920 = four rounds of
// workgroups).  FGC_W8_NT16 = 0: never, 2: the data kernel always (developer switch).
template <bool DATA>
static bool w8_half_tiles(const CoreParams& p) {
    const int mode = (int)opt(OPT_W8_NT16), min_n = (int)opt(OPT_W8_DATA16_MIN_N);
    if ((p.npad >> 4) > 4 || mode < 1) return false;
    return !DATA || mode >= 2 || p.n >= min_n;
}

template <bool DATA, bool FAST, int QS, bool BF = false, bool EROW = false>
static int launch_w8f(const CoreParams& p, const FwdEpilogue& fe, const DataEpilogue& de, size_t smem, hipStream_t st) {
    static bool attr = false;
    if (!attr) {
        hipFuncSetAttribute((const void*)conv_w8_kernel<DATA, FAST, QS, BF, 32, EROW>, hipFuncAttributeMaxDynamicSharedMemorySize,
                            160 * 1024);
        attr = true;
    }
    if constexpr (FAST && QS == 16) {
        if (w8_half_tiles<DATA>(p)) {
            constexpr int NT = 16;
            const size_t zrow = BF ? (size_t)ZSTRIDE_BF * 2 : (size_t)ZSTRIDE * 4;
            size_t smem16 = NT * zrow + (size_t)NT * qnode_stride(16) * 4 + (2 * NT + 4) * 4 + 64;
            if (DATA) smem16 += (size_t)opt(OPT_W8_DATA_SMEM_PAD);   // (developer knob: fewer resident workgroups)
            if constexpr (!DATA && !BF && !EROW) {
                if (opt(OPT_W8_HALF2) == 1) {   // aggregate tile in two halves: 23.4 KB, six workgroups per CU
                    const size_t smem_h2 = smem16 - (size_t)NT * (ZSTRIDE - 168) * 4;
                    FGC_LAUNCH("conv_w8_kernel<fwd>", st, (conv_w8_kernel<DATA, FAST, QS, BF, NT, EROW, true>), dim3(2 * core_grid(p)),
                               dim3(NT * 16), smem_h2, p, fe, de);
                    FGC_CHECK_LAUNCH("conv_w8_kernel (half tiles, two-half aggregate tile)");
                    return FGC_OK;
                }
            }
            FGC_LAUNCH(DATA ? "conv_w8_kernel<data>" : "conv_w8_kernel<fwd>", st, (conv_w8_kernel<DATA, FAST, QS, BF, NT, EROW>),
                       dim3(2 * core_grid(p)), dim3(NT * 16), smem16, p, fe, de);
            FGC_CHECK_LAUNCH("conv_w8_kernel (half tiles)");
            return FGC_OK;
        }
    }
    smem -= (size_t)TILE * (qnode_stride(KMAX) - qnode_stride(QS)) * 4;     // the caller sized the tile for KMAX slots
    if (BF) smem -= (size_t)TILE * (ZSTRIDE * 4 - ZSTRIDE_BF * 2);   // ... and for the fp32 aggregate tile
    FGC_LAUNCH(DATA ? "conv_w8_kernel<data>" : "conv_w8_kernel<fwd>", st, (conv_w8_kernel<DATA, FAST, QS, BF, 32, EROW>),
               dim3(core_grid(p)), dim3(W8_THREADS), smem, p, fe, de);
    FGC_CHECK_LAUNCH("conv_w8_kernel");
    return FGC_OK;
}

// bf16 storage: the fast shapes with 32 .. 128 output columns (the out tile that aliases the bf16 aggregate tile fits)
bool w8_bf16_supported(const CoreParams& p, int max_deg) {
    const int nct = p.npad >> 4;
    return w8_supported(p, max_deg) && w8_fast(p) && (nct == 2 || nct == 4 || nct == 8) && p.nout == p.npad;
}

// (the matrix-pipe form stores r in 16-byte pieces)
static bool bfm_r_ok(const DataEpilogue& de) { return ((uintptr_t)de.r & 15) == 0 && de.rld % 8 == 0; }

template <bool DATA>
static int launch_w8(const CoreParams& p, const FwdEpilogue& fe, const DataEpilogue& de, size_t smem, int max_deg,
                     bool bf16, hipStream_t st) {
    if (bf16) {
        // degrees <= 16: the aggregation on the bf16 matrix pipe (fgc_conv_bfm.hip; option NO_BFM = 1: the vector form)
        if (bfm_supported(p, max_deg, DATA, de.cin) && (!DATA || bfm_r_ok(de)))
            return DATA ? launch_data_bfm(p, de, w8_half_tiles<DATA>(p), false, st) : launch_fwd_bfm(p, fe, w8_half_tiles<DATA>(p), st);
        return max_deg <= 16 ? launch_w8f<DATA, true, 16, true>(p, fe, de, smem, st)
                             : launch_w8f<DATA, true, KMAX, true>(p, fe, de, smem, st);
    }
    if (!w8_fast(p)) return launch_w8f<DATA, false, KMAX>(p, fe, de, smem, st);
    return max_deg <= 16 ? launch_w8f<DATA, true, 16>(p, fe, de, smem, st)
                         : launch_w8f<DATA, true, KMAX>(p, fe, de, smem, st);
}

int launch_fwd_w8(const CoreParams& p, const FwdEpilogue& ep, size_t smem, int max_deg, hipStream_t st, bool bf16) {
    DataEpilogue de{};
    return launch_w8<false>(p, ep, de, smem, max_deg, bf16, st);
}
int launch_data_w8(const CoreParams& p, const DataEpilogue& ep, size_t smem, int max_deg, hipStream_t st, bool bf16) {
    FwdEpilogue fe{};
    return launch_w8<true>(p, fe, ep, smem, max_deg, bf16, st);
}
// the gathered rows are indexed by the edge id p.eid (pair form); fast shapes only
bool w8_erow_supported(const CoreParams& p, int max_deg) { return w8_supported(p, max_deg) && w8_fast(p) && p.eid != nullptr; }
int launch_data_w8_erow(const CoreParams& p, const DataEpilogue& ep, size_t smem, int max_deg, hipStream_t st, bool bf16) {
    FwdEpilogue fe{};
    if (bf16 && bfm_supported(p, max_deg, true, ep.cin) && bfm_r_ok(ep)) return launch_data_bfm(p, ep, w8_half_tiles<true>(p), true, st);
    if (bf16)
        return max_deg <= 16 ? launch_w8f<true, true, 16, true, true>(p, fe, ep, smem, st)
                             : launch_w8f<true, true, KMAX, true, true>(p, fe, ep, smem, st);
    return max_deg <= 16 ? launch_w8f<true, true, 16, false, true>(p, fe, ep, smem, st)
                         : launch_w8f<true, true, KMAX, false, true>(p, fe, ep, smem, st);
}

}  // namespace fgc
