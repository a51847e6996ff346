// bf16 storage: the fused graph-conv core with the AGGREGATION on the bf16 matrix pipe (round 6).
//
// conv_w8_kernel<..., BF = true> (fgc_conv_w8.hip) gathers two bf16 channels per lane and accumulates
// z[m] += q[m] * x_j on the vector ALU: 9 packed FMAs + an unpack per edge slot, lane and pass - 83 % vector-ALU busy with
// the matrix pipe at 3 - 6 % (profiles/r5_pmc_sq_tables_bf16.txt).  Here the per-node product
//     z_i^T [32 channels x 9] = X_i^T [32 channels x 16 edge slots] . q_i [16 edge slots x 9]         (model.py:482-488)
// is two v_mfma_f32_16x16x32_bf16 per node and pass (one per 16 channels):
//   * the neighbour rows of a node never pass through registers: ONE buffer_load_dwordx4 ... lds per node and pass
//     gathers its 16 edge slots x 64 bytes straight into a [slot][32 channels] LDS image (per-lane source address = a row
//     gather; slots past the degree fetch row 0 and carry weight zero);
//   * ds_read_b64_tr_b16 delivers that image transposed - lane (channel, k-group) gets 4 consecutive edge slots of its
//     channel per read - as the A operand; the 32 k slots of the MFMA are [q_hi of slots 0-15 | q_lo of slots 0-15] against
//     the same 16 rows twice, so the soft assignment keeps 16 significand bits (q = hi + lo, two bf16 terms) at no extra
//     matrix instruction: the product is as exact as the fp32 FMAs it replaces up to 2^-17 relative;
//   * the B operand q_i^T [k][m] is written by the soft-assignment lanes as bf16 into a [m][32] table per node (18 two-byte
//     LDS stores per lane and tile) and read back once per tile as one ds_read_b128 per node (kept in 4 registers);
//   * the result tile D[channel][m] leaves as bf16 into the aggregate tile [node][m * 32 + channel] the tile product
//     z . W~ (unchanged: v_mfma_f32_16x16x32_bf16 against the packed weights) reads, and - data gradient - into r.
// Everything around it (soft assignment in fp32 from the logit tables, epilogues, packed-weight layout, XCD tile map) is
// the 16-slot pipelined form of fgc_conv_w8.hip; shapes: FAST (whole 32-channel passes from one source), degrees <= 16.
// Rows in flight live in LDS (1 KB per node, wave-private: no workgroup barrier between a gather and its use), so a lane
// holds no row registers and no z accumulators: 16 + 18 registers less than the vector form (76 in the forward half-tile kernel).
#include <stdlib.h>

#include "fgc_conv_w8.h"

// developer knock-outs for phase timing (tools/build_variant.sh; results are wrong with any bit set, never set in the shipped
// build): 1 = no row gathers, 2 = no aggregation (transposed reads, MFMAs, aggregate-tile stores), 4 = no q^T stores,
// 8 = no tile product.  BFM_SEPARATE_QT = 1: q^T and the row ids in LDS of their own (the first form: 37.6 KB per half tile)
#ifndef BFM_KO
#define BFM_KO 0
#endif
#ifndef BFM_RING
#define BFM_RING 4
#endif
// Nodes per sweep of the aggregation phase.  2 (two sweeps of two nodes per pass): half the fragment registers of a four-node
// sweep - the forward half-tile kernel 89 -> 76 registers, i.e. SIX workgroups per CU by registers as by LDS (five with 4):
// level-0 forward 48.7 -> 45.1 us, step 1.025 -> 1.020 ms in alternating same-box runs; the data-gradient kernels (five by LDS
// either way) unchanged.  Forcing 80 registers on the four-node sweep instead (BFM_LB_FWD16 = 6: 9 spilled) lost: 55 us.
#ifndef BFM_AH
#define BFM_AH 2
#endif
#ifndef BFM_LB_FWD16
#define BFM_LB_FWD16 4
#endif
#ifndef BFM_SEPARATE_QT
#define BFM_SEPARATE_QT 0
#endif

namespace fgc {

namespace {

typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef short s16x8 __attribute__((ext_vector_type(8)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

constexpr int QT_ROW = 64;                 // bytes of one assignment's k slots: 16 x hi | 16 x lo (bf16)
constexpr int QT_NODE = FGC_M * QT_ROW;    // 576 B of q^T per node
constexpr int XS_NODE = 16 * 64;           // staged rows of a node and pass: 16 slots x 32 bf16 channels
constexpr int ZROW = ZSTRIDE_BF * 2;       // bytes per node of the bf16 aggregate tile (608)

// LDS of a workgroup.  q^T of a node lives in the node's row of the aggregate tile (576 of its 608 bytes) and the node's 16
// row ids in the first 64 bytes of its row image: both are written in the soft-assignment phase and are in registers (qf,
// rrow) of the ONE wave that owns the node before that wave writes the node's aggregates / requests its rows - LDS
// operations of a wave execute in order, so no barrier is involved.  26.3 KB per half tile: six workgroups per CU
// (forward; the data gradient's da | dg rows make it 27.4 KB: five).
struct BfmSmem {
    char* ztile;   // [NT][ZROW]           aggregate tile (aliased by the fp32 out tile of the epilogue)
    char* qt;      // [NT][qt_node]        q^T as hi | lo bf16, 9 rows of 64 B
    char* xs;      // [NT][16][64 B]       gathered rows of the current pass (wave-private regions of 4 nodes)
    int* rid;      // [NT][rid_node / 4]   source row of every edge slot
    int* deg;      // [NT] degrees, 4 ints, [NT] 1 / degree
    float* dag;    // [NT][9][2]           data gradient: da | dg of the node for the epilogue
};
constexpr int QT_STRIDE = BFM_SEPARATE_QT ? QT_NODE : ZROW;       // bytes between the q^T tables of two nodes
constexpr int RID_STRIDE = BFM_SEPARATE_QT ? 64 : XS_NODE;        // bytes between the row-id lists of two nodes
__host__ __device__ constexpr size_t bfm_smem_bytes(int nt, bool data) {
    return (size_t)nt * (ZROW + XS_NODE + (BFM_SEPARATE_QT ? QT_NODE + 64 : 0)) + (2 * nt + 4) * 4 +
           (data ? (size_t)nt * FGC_M * 8 : 0) + (BFM_SEPARATE_QT ? 448 : 0);
}
__device__ __forceinline__ BfmSmem bfm_carve(char* base, int nt) {
    BfmSmem s;
    s.ztile = base;
    size_t off = (size_t)nt * ZROW;
    s.qt = BFM_SEPARATE_QT ? base + off : base;
    if (BFM_SEPARATE_QT) off += (size_t)nt * QT_NODE;
    s.xs = base + off;
    off += (size_t)nt * XS_NODE;
    s.rid = reinterpret_cast<int*>(BFM_SEPARATE_QT ? base + off : s.xs);
    if (BFM_SEPARATE_QT) off += (size_t)nt * 64;
    s.deg = reinterpret_cast<int*>(base + off);
    off += (2 * nt + 4) * 4;
    s.dag = reinterpret_cast<float*>(base + off);
    return s;
}

// the lanes of one wave write and read their own nodes' tables: LDS operations of a wave execute in order, the compiler is
// all that has to be told
__device__ __forceinline__ void wave_lds_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}
// workgroup barrier that publishes LDS stores WITHOUT draining the vector-memory counter: the next pass' row gathers
// (LDS-DMA, counted on vmcnt) stay in flight across it.  __syncthreads() would emit s_waitcnt vmcnt(0) here.
__device__ __forceinline__ void lds_barrier() {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
}
__device__ __forceinline__ void wait_vm0() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
__device__ __forceinline__ void wait_lgkm0() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }

typedef __attribute__((address_space(3))) void* lds_ptr_t;
__device__ __forceinline__ lds_ptr_t to_lds(const void* p) {
    return (lds_ptr_t)(size_t)(unsigned)(size_t)p;   // (generic -> LDS: the low 32 bits of a shared-window address are the LDS offset)
}

}  // namespace

// NT nodes per workgroup (16: four waves, 32: eight waves); a wave owns four nodes in the soft-assignment and aggregation
// phases and one column tile x k-part of the tile product.  DATA / EROW as in conv_w8_kernel.
template <bool DATA, int NT, bool EROW>
__global__ __launch_bounds__(NT * 16, (!DATA && NT == 16) ? BFM_LB_FWD16 : 4) void conv_bfm_kernel(CoreParams p, FwdEpilogue fe, DataEpilogue de) {
    constexpr int TILE = NT, RT = NT / 16, NW = NT / 4, LW = NT == 32 ? 3 : 2, THREADS = NT * 16;
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    const BfmSmem s = bfm_carve(smem_raw, NT);
    int tile0;
    if constexpr (NT == 32) {
        tile0 = block_tile0(p);
    } else {
        const int h = xcd_tile(blockIdx.x, gridDim.x);
        tile0 = (p.tile_list ? p.tile_list[h >> 1] : (h >> 1)) * 32 + (h & 1) * 16;
    }
    const int tid = threadIdx.x;
    const int node = tid >> 4, kl = tid & 15;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63, lr = lane & 15, lq = lane >> 4;
    // ---------------- tile product: wave w owns column tile (w % nct) and k-part (w / nct)
    const int nct = p.npad >> 4;              // 2, 4 or 8
    const int nsh = 31 - __builtin_clz(nct);
    const int kparts = NW >> nsh;
    const int ct = __builtin_amdgcn_readfirstlane(wave & (nct - 1));
    const int kpart = __builtin_amdgcn_readfirstlane(wave >> nsh);
    constexpr int UPP = KPASS / 32;           // 32-deep k-steps per pass
    const int u0 = __builtin_amdgcn_readfirstlane((UPP * kpart) >> (LW - nsh));
    const int u1 = __builtin_amdgcn_readfirstlane((UPP * (kpart + 1)) >> (LW - nsh));
    const __amdgpu_buffer_rsrc_t w_rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.Wp), 0, -1, 0x00020000);
    const unsigned w_lane = (unsigned)((ct * 64 + lane) * 16);
    const unsigned w_unit = (unsigned)(nct * 1024);
    auto loadw = [&](int pass, int u) {
        const int uu = min(u, u1 - 1);
        return __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(w_rs, w_lane, (unsigned)(pass * UPP + uu) * w_unit, 0));
    };

    // ---------------- row gathers: one LDS-DMA per node and pass.  Lane l fetches 16 bytes of edge slot l >> 2; the image
    // is [slot][64 B] with the two 32-byte halves of slots 8-15 swapped (source chunk = position ^ 2), which keeps the
    // transposed reads of a 32-lane half - slots 0-3 and 8-11 of the same 16 channels - on different banks.
    unsigned rrow[4];                          // source rows of slot lane >> 2 of this wave's four nodes
    const unsigned dma_chunk = (unsigned)(((lane & 3) ^ ((lane >> 5) << 1)) * 16);
    auto issue = [&](int pass) {
        const bool first = pass * KC < p.c0;                                   // wave-uniform
        const float* base = first ? p.src0 : p.src1;
        const unsigned rowbytes = (unsigned)(first ? p.c0 : p.c1) * 2u;
        const unsigned passoff = (unsigned)(pass * KC - (first ? 0 : p.c0)) * 2u + dma_chunk;
        const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(base), 0, -1, 0x00020000);
        if (BFM_KO & 1) return;
#pragma unroll
        for (int a = 0; a < 4; ++a)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, to_lds(s.xs + (size_t)(wave * 4 + a) * XS_NODE), 16,
                                                     __umul24(rrow[a], rowbytes) + passoff, 0, 0, 0);
    };

    // ---------------- phase S: per-edge soft assignment (edge slot kl of this thread's node), fp32
    float dgsum[FGC_M];
#pragma unroll
    for (int m = 0; m < FGC_M; ++m) dgsum[m] = 0.f;
    {
        const int i = tile0 + node;
        int d = 0, e0 = 0;
        float ctr[FGC_M];
#pragma unroll
        for (int m = 0; m < FGC_M; ++m) ctr[m] = 0.f;
        const __amdgpu_buffer_rsrc_t ag_rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.ag), 0, -1, 0x00020000);
        const __amdgpu_buffer_rsrc_t col_rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<int*>(p.col), 0, -1, 0x00020000);
        if (i < p.n) {
            e0 = p.rowptr[i];
            d = min(p.rowptr[i + 1] - e0, 16);
            const unsigned ao = __umul24((unsigned)(i >> p.ag_shift), FGC_AG_LD * 4u) + (unsigned)p.ctr_off * 4u;
            const f32x4 a0 = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(ag_rs, ao, 0, 0));
            const f32x4 a1 = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(ag_rs, ao + 16u, 0, 0));
            ctr[0] = a0[0]; ctr[1] = a0[1]; ctr[2] = a0[2]; ctr[3] = a0[3];
            ctr[4] = a1[0]; ctr[5] = a1[1]; ctr[6] = a1[2]; ctr[7] = a1[3];
            ctr[8] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(ag_rs, ao + 32u, 0, 0));
        }
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
            s.deg[TILE + 4 + node] = __float_as_int(d > 0 ? 1.0f / (float)d : 0.f);
        }
        const bool valid = kl < d;
        // (unconditional, clamped into the node's list: no exec-masked load.  A node WITHOUT edges reads the entry in front
        //  of its empty list: for the padding nodes at the end of a level e0 == nnz, one past the array)
        const unsigned eo = (unsigned)(d > 0 ? e0 + min(kl, d - 1) : max(e0 - 1, 0)) * 4u;
        const int jv = __builtin_amdgcn_raw_buffer_load_b32(col_rs, eo, 0, 0);
        const int jj = valid ? jv : 0;
        int ev = 0;
        if constexpr (DATA) {
            const __amdgpu_buffer_rsrc_t eid_rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<int*>(p.eid), 0, -1, 0x00020000);
            const int e = __builtin_amdgcn_raw_buffer_load_b32(eid_rs, eo, 0, 0);
            ev = valid ? e : 0;
        }
        const int er = EROW ? ev : (jj >> p.shift);
        const unsigned go = __umul24((unsigned)(jj >> p.ag_shift), FGC_AG_LD * 4u) + (unsigned)p.nbr_off * 4u;
        const f32x4 g0 = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(ag_rs, go, 0, 0));
        const f32x4 g1 = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(ag_rs, go + 16u, 0, 0));
        const float g8 = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(ag_rs, go + 32u, 0, 0));
        f32x4 d0 = {0.f, 0.f, 0.f, 0.f}, d1 = {0.f, 0.f, 0.f, 0.f};
        float d8 = 0.f;
        if constexpr (DATA) {   // per-edge d-logits of the in-edge (row 0 for the empty slots, dropped below)
            const __amdgpu_buffer_rsrc_t dl_rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(de.dl), 0, -1, 0x00020000);
            const unsigned dof = __umul24((unsigned)ev, FGC_DL_LD * 4u);
            d0 = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(dl_rs, dof, 0, 0));
            d1 = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(dl_rs, dof + 16u, 0, 0));
            d8 = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(dl_rs, dof + 32u, 0, 0));
        }
        // the row ids go to LDS first and the rows of pass 0 leave right behind the logit rows
        s.rid[node * (RID_STRIDE / 4) + kl] = er;
        wave_lds_sync();
#pragma unroll
        for (int a = 0; a < 4; ++a) rrow[a] = (unsigned)s.rid[(wave * 4 + a) * (RID_STRIDE / 4) + (lane >> 2)];
        issue(0);

        float l[FGC_M];
        l[0] = ctr[0] + g0[0]; l[1] = ctr[1] + g0[1]; l[2] = ctr[2] + g0[2]; l[3] = ctr[3] + g0[3];
        l[4] = ctr[4] + g1[0]; l[5] = ctr[5] + g1[1]; l[6] = ctr[6] + g1[2]; l[7] = ctr[7] + g1[3];
        l[8] = ctr[8] + g8;
        float mx = l[0];
#pragma unroll
        for (int m = 1; m < FGC_M; ++m) mx = fmaxf(mx, l[m]);
        float sum = 0.f;
        const float nmx = -mx * 1.4426950408889634f;
#pragma unroll
        for (int m = 0; m < FGC_M; ++m) {
            l[m] = __builtin_amdgcn_exp2f(fmaf(l[m], 1.4426950408889634f, nmx));   // exp(l - mx): one fma + v_exp_f32
            sum += l[m];
        }
        const float inv = valid ? 1.0f / sum : 0.f;      // empty slots: weight zero on row 0
        // q = hi + lo, two bf16 terms, transposed into the node's [m][16 hi | 16 lo] table
        if (!(BFM_KO & 4)) {
            unsigned short* qrow = reinterpret_cast<unsigned short*>(s.qt + (size_t)node * QT_STRIDE) + kl;
#pragma unroll
            for (int m = 0; m < FGC_M; m += 2) {
                const float qa = l[m] * inv, qb2 = m + 1 < FGC_M ? l[m + 1] * inv : 0.f;
                const unsigned hp = f2_to_bf2(qa, qb2);
                const f32x2c hb = bf2_to_f2(hp);
                const unsigned lp = f2_to_bf2(qa - hb[0], qb2 - hb[1]);
                qrow[m * (QT_ROW / 2)] = (unsigned short)hp;
                qrow[m * (QT_ROW / 2) + 16] = (unsigned short)lp;
                if (m + 1 < FGC_M) {
                    qrow[(m + 1) * (QT_ROW / 2)] = (unsigned short)(hp >> 16);
                    qrow[(m + 1) * (QT_ROW / 2) + 16] = (unsigned short)(lp >> 16);
                }
            }
        }
        if constexpr (DATA) {  // dg_j = sum over in-edges of dl: reduce the 16 softmax lanes of the node
            dgsum[0] = valid ? d0[0] : 0.f; dgsum[1] = valid ? d0[1] : 0.f; dgsum[2] = valid ? d0[2] : 0.f;
            dgsum[3] = valid ? d0[3] : 0.f; dgsum[4] = valid ? d1[0] : 0.f; dgsum[5] = valid ? d1[1] : 0.f;
            dgsum[6] = valid ? d1[2] : 0.f; dgsum[7] = valid ? d1[3] : 0.f; dgsum[8] = valid ? d8 : 0.f;
#pragma unroll
            for (int m = 0; m < FGC_M; ++m) {
                float v = dgsum[m];
                FGC_ROW16_SUM(v);
                dgsum[m] = v;
            }
            if (kl == 0 && i < p.n) {
                const float da[FGC_M] = {da0[0], da0[1], da0[2], da0[3], da1[0], da1[1], da1[2], da1[3], da8};
#pragma unroll
                for (int m = 0; m < FGC_M; ++m)
                    *reinterpret_cast<f32x2*>(s.dag + ((size_t)node * FGC_M + m) * 2) = f32x2{da[m], dgsum[m]};
                float* o = de.dag + (size_t)i * FGC_AG_LD + 12;
                *reinterpret_cast<f32x4*>(o) = f32x4{dgsum[0], dgsum[1], dgsum[2], dgsum[3]};
                *reinterpret_cast<f32x4*>(o + 4) = f32x4{dgsum[4], dgsum[5], dgsum[6], dgsum[7]};
                *reinterpret_cast<f32x4*>(o + 8) = f32x4{dgsum[8], 0.f, 0.f, 0.f};
                // da | dg behind the node's r row: [du; dv] = (da | dg)^T x rides in the dW0 GEMM
                u32x2* rt = reinterpret_cast<u32x2*>(reinterpret_cast<unsigned short*>(de.r) + (size_t)i * de.rld + (FGC_M * p.cg));
                rt[0] = f4_to_bf4(f32x4{da[0], da[1], da[2], da[3]});
                rt[1] = f4_to_bf4(f32x4{da[4], da[5], da[6], da[7]});
                rt[2] = f4_to_bf4(f32x4{da[8], 0.f, 0.f, 0.f});
                rt[3] = f4_to_bf4(f32x4{dgsum[0], dgsum[1], dgsum[2], dgsum[3]});
                rt[4] = f4_to_bf4(f32x4{dgsum[4], dgsum[5], dgsum[6], dgsum[7]});
                rt[5] = f4_to_bf4(f32x4{dgsum[8], 0.f, 0.f, 0.f});
            }
        }
    }
    wave_lds_sync();
    // B operand of the aggregation: q^T of this wave's four nodes, lane (m = lr, k-group lq) -> 8 consecutive k slots
    // (rows m >= 9 read past the node's table: columns of the product that are never stored)
    u32x4 qf[4];
#pragma unroll
    for (int a = 0; a < 4; ++a)
        qf[a] = *reinterpret_cast<const u32x4*>(s.qt + (size_t)(wave * 4 + a) * QT_STRIDE + lr * QT_ROW + lq * 16);

    f32x4 acc[RT];
#pragma unroll
    for (int r = 0; r < RT; ++r) acc[r] = f32x4{0.f, 0.f, 0.f, 0.f};
    const bool want_gemm = !DATA || de.dx0 != nullptr;

    // transposed reads: lane 4q + pc of a 16-lane group supplies row q, columns 4 pc .. 4 pc + 3 of the group's 4 x 16
    // block; group g reads edge slots 8 (g & 1) + 4 h + q (h = 0, 1: two reads = 8 k slots), 16 channels c2
    const int tq = lr >> 2, tpc = lr & 3;
    const unsigned tr_off = (unsigned)((8 * (lq & 1) + tq) * 64 + ((((lq & 1) << 1) | (tpc >> 1)) * 16) + (tpc & 1) * 8);

    auto do_pass = [&](int pass) {
        // ---------------- phase A: z^T[32 x 9] = X^T[32 x 16] q[16 x 9] per node on the matrix pipe
        wait_vm0();                                         // this pass' rows have landed in LDS
        if (pass > 0) lds_barrier();                        // previous pass' reads of the aggregate tile are done
        // (BFM_AH nodes at a time - 4: all sixteen transposed reads of the wave's four nodes first, then the eight independent
        //  MFMAs back to back, then the conversions and stores; 2: the same in two halves, with half the fragment registers)
        if (!(BFM_KO & 2)) {
#pragma unroll
            for (int hh = 0; hh < 4 / BFM_AH; ++hh) {
                s16x8 xt[BFM_AH][2];
#pragma unroll
                for (int a = 0; a < BFM_AH; ++a) {
                    const char* xb = s.xs + (size_t)(wave * 4 + BFM_AH * hh + a) * XS_NODE;
#pragma unroll
                    for (int c2 = 0; c2 < 2; ++c2) {
                        const char* ta = xb + (tr_off ^ (unsigned)(c2 * 32));
                        const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)to_lds(ta));
                        const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)to_lds(ta + 256));
                        xt[a][c2] = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
                    }
                }
                f32x4 zt[BFM_AH][2];
#pragma unroll
                for (int a = 0; a < BFM_AH; ++a)
#pragma unroll
                    for (int c2 = 0; c2 < 2; ++c2)
                        zt[a][c2] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, xt[a][c2]),
                                                                           __builtin_bit_cast(bf16x8, qf[BFM_AH * hh + a]),
                                                                           f32x4{0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
                // the next pass' rows travel under this pass' matrix phase: once the LAST transposed reads have returned (their
                // MFMAs were issued) the wave's row image is free
                if (hh == 4 / BFM_AH - 1 && pass + 1 < p.passes) {
                    wait_lgkm0();
                    issue(pass + 1);
                }
#pragma unroll
                for (int a = 0; a < BFM_AH; ++a) {
                    const int nd = wave * 4 + BFM_AH * hh + a;
#pragma unroll
                    for (int c2 = 0; c2 < 2; ++c2) {
                        // lane (m = lr, lq): channels 16 c2 + 4 lq .. + 3 of assignment m
                        const u32x2 zp = f4_to_bf4(zt[a][c2]);
                        if (lr < FGC_M) {
                            // (the 16-byte pieces of an assignment's 64 bytes sit at piece ^ ((m >> 1) & 3): assignments m and
                            //  m + 2 are 128 bytes = 32 banks apart, unswizzled the nine lanes of a store group hit two banks
                            //  five-fold - 17 us of a 52 us level-0 launch by knock-out; the tile product reads with the same XOR)
                            *reinterpret_cast<u32x2*>(s.ztile + (size_t)nd * ZROW + lr * (KC * 2) +
                                                      (((c2 * 2 + (lq >> 1)) ^ ((lr >> 1) & 3)) * 16) + (lq & 1) * 8) = zp;
                        }
                    }
                }
            }
        } else if (pass + 1 < p.passes) {
            wait_lgkm0();
            issue(pass + 1);
        }
        lds_barrier();
        if constexpr (DATA) {
            // r[j, m * cg + 32 pass ..] = the node's aggregates of this pass, copied out of the finished tile in 16-byte
            // pieces - four lanes per 64 contiguous bytes, 36 pieces per node - under the other waves' tile product (the
            // accumulator layout would store 8-byte pieces, nine 32-byte runs per instruction: +9 us on the level-0 launch)
            for (int idx = tid; idx < TILE * 36; idx += THREADS) {
                const int nd = (idx * 1821) >> 16, rem = idx - nd * 36;      // idx / 36 for idx < 1152
                const int m = rem >> 2, ch = rem & 3;
                const int j = tile0 + nd;
                const u32x4 v = *reinterpret_cast<const u32x4*>(s.ztile + (size_t)nd * ZROW + m * 64 + ((ch ^ ((m >> 1) & 3)) * 16));
                if (j < p.n)
                    *reinterpret_cast<u32x4*>(reinterpret_cast<unsigned short*>(de.r) + (size_t)j * de.rld + m * p.cg + pass * KC + ch * 8) = v;
            }
            if (!want_gemm) return;
        }
        // ---------------- phase G: acc[NT x 16] += ztile[NT x k-part] * Wp[k-part x 16]
        if (BFM_KO & 8) return;
        auto mmb = [&](int ks, const u32x4& b) {
#pragma unroll
            for (int r = 0; r < RT; ++r) {
                const u32x4 av = *reinterpret_cast<const u32x4*>(s.ztile + (size_t)(r * 16 + lr) * ZROW + ks * 64 +
                                                                 ((lq ^ ((ks >> 1) & 3)) * 16));
                acc[r] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, av), __builtin_bit_cast(bf16x8, b),
                                                                acc[r], 0, 0, 0);
            }
        };
        // the packed-weight fragments run through a ring of BFM_RING registers sets with fixed roles (loads unconditional on
        // clamped indices, loop bounds scalar: counted s_waitcnt vmcnt): on the coarse levels a launch is ONE wave of tiles
        // and its time is a tile's chain of L2 round trips - two fragments in flight left every other k-step waiting
        u32x4 bw[BFM_RING];
        int ks = u0;
#pragma unroll
        for (int t = 0; t < BFM_RING; ++t) bw[t] = loadw(pass, ks + t);
        for (; ks + BFM_RING <= u1; ks += BFM_RING) {
#pragma unroll
            for (int t = 0; t < BFM_RING; ++t) {
                mmb(ks + t, bw[t]);
                bw[t] = loadw(pass, ks + BFM_RING + t);
            }
        }
#pragma unroll
        for (int t = 0; t < BFM_RING - 1; ++t)
            if (ks + t < u1) mmb(ks + t, bw[t]);
    };
    for (int pass = 0; pass < p.passes; ++pass) do_pass(pass);
    if (!want_gemm) return;
    lds_barrier();
    // ---------------- accumulators -> LDS (aliases the aggregate tile), k-parts summed in fixed order by the epilogue
    const int oldd = p.npad + 4;
    float* otile = reinterpret_cast<float*>(s.ztile);
    {
        float* base = otile + (size_t)kpart * TILE * oldd;
#pragma unroll
        for (int r = 0; r < RT; ++r)
#pragma unroll
            for (int t = 0; t < 4; ++t) base[(size_t)(r * 16 + lq * 4 + t) * oldd + ct * 16 + lr] = acc[r][t];
    }
    lds_barrier();
    if (!DATA) {
        // thread -> (pooled row group, column): the width is 32, 64 or 128, THREADS is a multiple of it
        const int osh = 31 - __builtin_clz(p.nout);
        const int o = tid & (p.nout - 1), pstep = THREADS >> osh;
        const float bias_o = fe.bias[o];
        for (int pr = tid >> osh; pr < TILE / 4; pr += pstep) {
            float mx = -INFINITY;
            bool any = false;
            const size_t ybase = (size_t)(tile0 + pr * 4) * p.nout + o;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int row = pr * 4 + q;
                const int i = tile0 + row;
                if (i >= p.n) continue;
                float val = 0.f;
                for (int kp = 0; kp < kparts; ++kp) val += otile[((size_t)kp * TILE + row) * oldd + o];
                const int dd = s.deg[row];
                val *= __int_as_float(s.deg[TILE + 4 + row]);
                if (!fe.bias_mask || dd > 0) val += bias_o;
                if (fe.act) val = fmaxf(val, 0.f) - fe.alpha * fmaxf(-val, 0.f);
                st_act(fe.y, ybase + (size_t)(q * p.nout), val, 1);
                mx = fmaxf(mx, val);
                any = true;
            }
            if (fe.y_pool && any) st_act(fe.y_pool, (size_t)((tile0 >> 2) + pr) * p.nout + o, mx, 1);
        }
    } else {
        const int group = 1 << de.shiftf;
        const int nsrc = TILE / group;
        const int csh = 31 - __builtin_clz(de.cin);
        const int c = tid & (de.cin - 1), sstep = THREADS >> csh;
        float uc[FGC_M], vc[FGC_M];
#pragma unroll
        for (int m = 0; m < FGC_M; ++m) {
            uc[m] = de.u[m * de.cin + c];
            vc[m] = de.v[m * de.cin + c];
        }
        for (int sr = tid >> csh; sr < nsrc; sr += sstep) {
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
                    const f32x2 dd = *reinterpret_cast<const f32x2*>(s.dag + ((size_t)row * FGC_M + m) * 2);
                    g = fmaf(dd[0], uc[m], g);
                    g = fmaf(dd[1], vc[m], g);
                }
                val += g;
            }
            if (!any) continue;
            const size_t srow = (size_t)((tile0 >> de.shiftf) + sr);
            if (c < de.c0f) {
                const size_t o = srow * de.c0f + c;
                st_act(de.dx0, o, de.acc0 ? ld_act(de.dx0, o, 1) + val : val, 1);
            } else if (de.dx1) {
                const size_t o = srow * de.c1f + (c - de.c0f);
                st_act(de.dx1, o, de.acc1 ? ld_act(de.dx1, o, 1) + val : val, 1);
            }
        }
    }
}

// shapes: what w8_bf16_supported accepts, with every degree <= 16, power-of-two epilogue widths, and row offsets that fit
// the 24-bit multiply of the gather
bool bfm_supported(const CoreParams& p, int max_deg, bool data, int cin_fwd) {
    if (opt(OPT_NO_BFM) == 1) return false;
    if (max_deg <= 0 || max_deg > 16) return false;
    const int w = data ? cin_fwd : p.nout;
    if (w < 32 || (w & (w - 1))) return false;                 // epilogue: thread -> column by shifts
    if (p.nout != p.npad) return false;
    if (((uintptr_t)p.src0 & 15) || (p.src1 && ((uintptr_t)p.src1 & 15))) return false;   // 16-byte row pieces
    if ((p.c0 % 8) || (p.c1 % 8)) return false;
    return true;
}

template <bool DATA, bool EROW>
static int launch_bfm_t(const CoreParams& p, const FwdEpilogue& fe, const DataEpilogue& de, bool half, hipStream_t st) {
    static bool attr = false;
    if (!attr) {
        hipFuncSetAttribute((const void*)conv_bfm_kernel<DATA, 32, EROW>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        hipFuncSetAttribute((const void*)conv_bfm_kernel<DATA, 16, EROW>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        attr = true;
    }
    const char* name = DATA ? "conv_w8_kernel<data>" : "conv_w8_kernel<fwd>";   // (same role, same profile key as the vector form)
    if (half) {
        FGC_LAUNCH(name, st, (conv_bfm_kernel<DATA, 16, EROW>), dim3(2 * core_grid(p)), dim3(256), bfm_smem_bytes(16, DATA), p, fe, de);
    } else {
        FGC_LAUNCH(name, st, (conv_bfm_kernel<DATA, 32, EROW>), dim3(core_grid(p)), dim3(512), bfm_smem_bytes(32, DATA), p, fe, de);
    }
    FGC_CHECK_LAUNCH("conv_bfm_kernel");
    return FGC_OK;
}

int launch_fwd_bfm(const CoreParams& p, const FwdEpilogue& ep, bool half, hipStream_t st) {
    DataEpilogue de{};
    return launch_bfm_t<false, false>(p, ep, de, half, st);
}
int launch_data_bfm(const CoreParams& p, const DataEpilogue& ep, bool half, bool erow, hipStream_t st) {
    FwdEpilogue fe{};
    return erow ? launch_bfm_t<true, true>(p, fe, ep, half, st) : launch_bfm_t<true, false>(p, fe, ep, half, st);
}

}  // namespace fgc
