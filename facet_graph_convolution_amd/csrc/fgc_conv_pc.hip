// Producer / consumer form of the fused graph-conv core (forward conv and backward-data conv).
//
// One persistent 512-thread workgroup per CU walks tiles of 32 nodes.  Its 8 waves have fixed roles, two per
// SIMD, so that the SIMD's vector ALU / memory pipe and its matrix pipe work at the same time:
//   waves 0-3  PRODUCERS  per-edge softmax (once per tile) and neighbour aggregation of one 32-channel pass:
//                         gather x_j rows (float4 per lane, 128 B per node), z[m] += q[m]*x_j  -> LDS ztile[s&1]
//   waves 4-7  CONSUMERS  f32 MFMA of the PREVIOUS item (tile, pass) from ztile[(s-1)&1] against the packed
//                         weights (L2), then the op's epilogue after a tile's last pass
// Items are pipelined one deep through the double-buffered z tile; two workgroup barriers per item.
// The soft-assignment buffer is written and read by the same 8 lanes of one wave, so it needs no barrier.
#include <stdlib.h>

#include "fgc_conv_pc.h"

namespace fgc {

constexpr int PC_THREADS = 512;

struct PcSmem {
    float* z0;       // [2][TILE][zstride]; buffer b = z0 + b * TILE * zstride (computed, never stored in an array:
                     // a runtime-indexed pointer array makes hipcc fall back to flat loads + vmcnt(0) waits)
    float* qbuf;     // [TILE][KMAX][QLD]
    float* otile;    // [4][TILE][MAX_NPAD + 4] worst case, sized by host
    float* dagt;     // [2][TILE][24]  (backward-data only)
    int* deg;        // [2][TILE]
};

__device__ __forceinline__ PcSmem pc_carve(char* base, int zstride, int otile_floats) {
    PcSmem s;
    size_t off = 0;
    s.z0 = reinterpret_cast<float*>(base);
    off += (size_t)2 * TILE * zstride * 4;
    s.qbuf = reinterpret_cast<float*>(base + off);
    off += (size_t)TILE * KMAX * QLD * 4;
    s.otile = reinterpret_cast<float*>(base + off);
    off += (size_t)otile_floats * 4;
    s.dagt = reinterpret_cast<float*>(base + off);
    off += (size_t)2 * TILE * 24 * 4;
    s.deg = reinterpret_cast<int*>(base + off);
    return s;
}

static int pc_otile_floats(const ConvGeom& g) {
    const int nct = g.npad / 16;
    const int kparts = nct >= 3 ? 1 : (nct == 2 ? 2 : 4);
    return kparts * TILE * (g.npad + 4);
}

size_t pc_smem_bytes(const ConvGeom& g) {
    return (size_t)2 * TILE * g.zstride * 4 + (size_t)TILE * KMAX * QLD * 4 + (size_t)pc_otile_floats(g) * 4 +
           (size_t)2 * TILE * 24 * 4 + 2 * TILE * 4 + 64;
}

// ---- forward epilogue on the 256 consumer threads -------------------------------------------------
__device__ __forceinline__ void fwd_epilogue(const CoreParams& p, const FwdEpilogue& ep, const float* otile, int oldd,
                                             int kparts, const int* deg, int tile0, int ctid) {
    for (int t = ctid; t < (TILE / 4) * p.nout; t += 256) {
        const int pr = t / p.nout, o = t % p.nout;
        float mx = -INFINITY;
        bool any = false;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int row = pr * 4 + q;
            const int i = tile0 + row;
            if (i >= p.n) continue;
            float val = 0.f;
            for (int kp = 0; kp < kparts; ++kp) val += otile[((size_t)kp * TILE + row) * oldd + o];
            const int d = deg[row];
            const float inv = d > 0 ? 1.0f / (float)d : 0.f;
            val *= inv;
            if (!ep.bias_mask || d > 0) val += ep.bias[o];
            if (ep.act) val = fmaxf(val, 0.f) - ep.alpha * fmaxf(-val, 0.f);
            ep.y[(size_t)i * p.nout + o] = val;
            mx = fmaxf(mx, val);
            any = true;
        }
        if (ep.y_pool && any) ep.y_pool[(size_t)((tile0 >> 2) + pr) * p.nout + o] = mx;
    }
}

// ---- backward-data epilogue ---------------------------------------------------------------------------
__device__ __forceinline__ void data_epilogue(const CoreParams& p, const DataEpilogue& ep, const float* otile, int oldd,
                                              int kparts, const float* dagt, int tile0, int ctid) {
    const int group = 1 << ep.shiftf;
    const int nsrc = TILE / group;
    for (int t = ctid; t < nsrc * ep.cin; t += 256) {
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

template <bool DATA>
__global__ __launch_bounds__(PC_THREADS, 2) void conv_pc_kernel(CoreParams p, FwdEpilogue fe, DataEpilogue de,
                                                               int ntiles, int otile_floats) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    const PcSmem S = pc_carve(smem_raw, p.zstride, otile_floats);
    const int tid = threadIdx.x;
    const bool producer = tid < 256;
    const int ctid = tid - 256;
    const WaveTiling wt = wave_tiling(p.npad, producer ? 0 : (ctid >> 6));
    const int oldd = p.npad + 4;
    const int P = p.passes;
    const bool want_gemm = !DATA || de.dx0 != nullptr;

    // my tiles: a CONTIGUOUS range, so that successive tiles of a workgroup are spatial neighbours (the node order
    // is spatially coherent) and their gathers hit rows this CU / XCD fetched a moment ago
    const int chunk = (ntiles + gridDim.x - 1) / gridDim.x;
    const int tbeg = blockIdx.x * chunk;
    const int my_tiles = max(0, min(ntiles, tbeg + chunk) - tbeg);
    const int nitems = my_tiles * P;

    f32x4 acc[RT][CTW];
#pragma unroll
    for (int r = 0; r < RT; ++r)
#pragma unroll
        for (int c = 0; c < CTW; ++c) acc[r][c] = f32x4{0.f, 0.f, 0.f, 0.f};
    // the matrix waves win issue arbitration against the co-resident gather/VALU wave of their SIMD
    if (__builtin_amdgcn_readfirstlane(tid) >= 256) __builtin_amdgcn_s_setprio(1);
    f32x4 xnext[RB];          // producer: rows of the NEXT pass, in flight while this pass is aggregated
    bool have_next = false;
#pragma unroll
    for (int t = 0; t < RB; ++t) xnext[t] = f32x4{0.f, 0.f, 0.f, 0.f};

    for (int sidx = 0; sidx <= nitems; ++sidx) {
        if (producer) {
            if (sidx < nitems) {
                const int ti = sidx / P, pass = sidx - ti * P;
                const int tile0 = (tbeg + ti) * TILE;
                Smem s;
                s.ztile = S.z0 + (sidx & 1) * (TILE * p.zstride);
                s.qbuf = S.qbuf;
                s.deg = S.deg + (ti & 1) * TILE;
                s.extra = nullptr;
                if (pass == 0) {
                    // per-edge soft assignment of this tile; q rows are produced and consumed by the same 8 lanes
                    if (DATA) {
                        float dgsum[FGC_M];
#pragma unroll
                        for (int m = 0; m < FGC_M; ++m) dgsum[m] = 0.f;
                        softmax_phase<true>(p, s, tile0, 0, de.dl, dgsum);
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
                            float* t = S.dagt + ((ti & 1) * TILE + node) * 24;
                            if (j < p.n) {
                                const float* da = de.dag + (size_t)j * FGC_AG_LD;
#pragma unroll
                                for (int m = 0; m < FGC_M; ++m) {
                                    t[m] = da[m];
                                    t[12 + m] = dgsum[m];
                                }
                                float* o = de.dag + (size_t)j * FGC_AG_LD + 12;
                                *reinterpret_cast<f32x4*>(o) = f32x4{dgsum[0], dgsum[1], dgsum[2], dgsum[3]};
                                *reinterpret_cast<f32x4*>(o + 4) = f32x4{dgsum[4], dgsum[5], dgsum[6], dgsum[7]};
                                *reinterpret_cast<f32x4*>(o + 8) = f32x4{dgsum[8], 0.f, 0.f, 0.f};
                            } else {
#pragma unroll
                                for (int m = 0; m < 24; ++m) t[m] = 0.f;
                            }
                        }
                    } else {
                        softmax_phase<false>(p, s, tile0, 0, nullptr, nullptr);
                    }
                    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                    __builtin_amdgcn_wave_barrier();
                }
                f32x4 z[FGC_M];
#pragma unroll
                for (int m = 0; m < FGC_M; ++m) z[m] = f32x4{0.f, 0.f, 0.f, 0.f};
                {
                    const int node = tid >> 3, cl = tid & 7;
                    const int d = min(s.deg[node], KMAX);
                    const float* qb = s.qbuf + (size_t)node * KMAX * QLD;
                    const int cbase = pass * p.kc + cl * 4;
                    f32x4 xc[RB];
                    if (have_next) {
#pragma unroll
                        for (int t = 0; t < RB; ++t) xc[t] = xnext[t];
                    } else {
                        load_rows<true>(p, qb, d, 0, cbase, xc);
                    }
                    have_next = pass + 1 < P;
                    if (have_next) load_rows<true>(p, qb, d, 0, cbase + p.kc, xnext);
                    fma_rows(qb, d, 0, xc, z);
                    if (d > RB) {  // rare: more than 16 neighbours
                        f32x4 xt[RB];
                        load_rows<true>(p, qb, d, RB, cbase, xt);
                        fma_rows(qb, d, RB, xt, z);
                    }
                }
                if (DATA) {
                    const int node = tid >> 3, cl = tid & 7;
                    const int j = tile0 + node;
                    const int ch0 = pass * p.kc + cl * 4;
                    if (j < p.n && ch0 < p.cg) {
                        float* rr = de.r + (size_t)j * de.rld + ch0;
#pragma unroll
                        for (int m = 0; m < FGC_M; ++m) *reinterpret_cast<f32x4*>(rr + m * p.cg) = z[m];
                    }
                }
                store_ztile<8>(p, s, z);
            }
        } else if (sidx >= 1 && want_gemm) {
            const int it = sidx - 1;
            const int ti = it / P, pass = it - ti * P;
            Smem s;
            s.ztile = S.z0 + (it & 1) * (TILE * p.zstride);
            s.qbuf = nullptr;
            s.deg = nullptr;
            s.extra = nullptr;
            gemm_pass(p, s, pass, wt, acc);
            if (pass == P - 1) {
                store_acc(S.otile, oldd, wt, p.npad, acc);
#pragma unroll
                for (int r = 0; r < RT; ++r)
#pragma unroll
                    for (int c = 0; c < CTW; ++c) acc[r][c] = f32x4{0.f, 0.f, 0.f, 0.f};
            }
        }
        __syncthreads();  // (A) ztile[sidx&1] and otile are complete
        if (!producer && sidx >= 1 && want_gemm) {
            const int it = sidx - 1;
            const int ti = it / P, pass = it - ti * P;
            if (pass == P - 1) {
                const int tile0 = (tbeg + ti) * TILE;
                if (DATA) data_epilogue(p, de, S.otile, oldd, wt.kparts, S.dagt + (ti & 1) * TILE * 24, tile0, ctid);
                else fwd_epilogue(p, fe, S.otile, oldd, wt.kparts, S.deg + (ti & 1) * TILE, tile0, ctid);
            }
        }
        __syncthreads();  // (B) otile may be rewritten, deg/dagt of the older tile may be reused
    }
}

template <bool DATA>
static int launch_pc(const CoreParams& p, const FwdEpilogue& fe, const DataEpilogue& de, const ConvGeom& g,
                     hipStream_t st) {
    const int ntiles = cdiv(p.n, TILE);
    const size_t smem = pc_smem_bytes(g);
    static bool attr_set = false;
    if (!attr_set) {
        hipFuncSetAttribute((const void*)conv_pc_kernel<DATA>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        attr_set = true;
    }
    const int grid = ntiles < 256 ? ntiles : 256;
    FGC_LAUNCH(DATA ? "conv_pc_kernel<data>" : "conv_pc_kernel<fwd>", st, (conv_pc_kernel<DATA>), dim3(grid),
               dim3(PC_THREADS), smem, p, fe, de, ntiles, pc_otile_floats(g));
    FGC_CHECK_LAUNCH("conv_pc_kernel");
    return FGC_OK;
}

int launch_fwd_pc(const CoreParams& p, const FwdEpilogue& ep, const ConvGeom& g, hipStream_t st) {
    DataEpilogue de{};
    return launch_pc<false>(p, ep, de, g, st);
}
int launch_data_pc(const CoreParams& p, const DataEpilogue& ep, const ConvGeom& g, hipStream_t st) {
    FwdEpilogue fe{};
    return launch_pc<true>(p, fe, ep, g, st);
}

}  // namespace fgc
