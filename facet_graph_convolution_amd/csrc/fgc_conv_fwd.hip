// Forward graph convolution (replaces custom_conv2d, /root/reference/Code/model.py:427-504).
#include <stdlib.h>

#include <algorithm>

#include "fgc_conv_w8.h"
#include "fgc_conv_narrow.h"
#include "fgc_conv_pair.h"
#include "fgc_pack.h"

namespace fgc {

// ---------------------------------------------------------------------------------------------
// weight packing: W0[m][o][c] -> k-interleaved B operand of the aggregate-first GEMM
//   row kk = pass*kpass + m*kc + cl  (c = pass*kc + cl), column = o, stored [kk/4][npad][kk%4]
// transposed = 1 packs the data-gradient operand instead: k runs over (pass, m, ol) with
// o = pass*kc + ol and the column is c.
// ---------------------------------------------------------------------------------------------
__global__ void pack_weight_kernel(const float* __restrict__ W0, float* __restrict__ Wp, int cin, int cout,
                                   int kdim, int ncols, int npad, int kc, int kpass, int passes, int transposed) {
    pack_weight_body(W0, Wp, cin, cout, kdim, ncols, npad, kc, kpass, passes, transposed, blockIdx.x, gridDim.x);
}

// ---------------------------------------------------------------------------------------------
// assignment logits on the matrix cores: ag[rows, 24] = x[rows, cin] * [u | v]^T (+ c).  One wave per 16 rows, A
// fragments straight from global memory (one dwordx4 per lane per 16 channels), B = [u|v] staged once per workgroup
// in LDS (column tile 0 = the 9 a-logits, tile 1 = the 9 g-logits).  Streams x exactly once: HBM-bound.
// ---------------------------------------------------------------------------------------------
// BF: x0 / x1 are bf16 tensors (FGC_CONV_BF16), widened on load; the table stays fp32
template <bool VEC4, bool BF = false>
__global__ __launch_bounds__(256) void proj_mfma_kernel(const float* __restrict__ x0, const float* __restrict__ x1,
                                                        int c0, int c1, int rows, const float* __restrict__ u,
                                                        const float* __restrict__ c, const float* __restrict__ v,
                                                        float* __restrict__ ag) {
    __shared__ __attribute__((aligned(16))) float Bs[128 * 32];   // [k][32]
    const int cin = c0 + c1;
    const int kpad = (cin + 15) & ~15;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, lr = lane & 15, lq = lane >> 4;
    for (int t = tid; t < kpad * 32; t += 256) {
        const int k = t >> 5, col = t & 31;
        float val = 0.f;
        if (k < cin) {
            if (col < FGC_M) val = u[col * cin + k];
            else if (col >= 16 && col < 16 + FGC_M) val = v[(col - 16) * cin + k];
        }
        Bs[t] = val;
    }
    __syncthreads();
    const float cbias = lr < FGC_M ? c[lr] : 0.f;
    const int ntile = (rows + 15) >> 4;
    for (int tile = blockIdx.x * 4 + wave; tile < ntile; tile += gridDim.x * 4) {
        const int row = min(tile * 16 + lr, rows - 1);
        f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
        for (int kb = 0; kb < kpad; kb += 16) {
            const int cb = kb + 4 * lq;
            f32x4 a = {0.f, 0.f, 0.f, 0.f};
            if (BF) {
                const unsigned short* h0 = reinterpret_cast<const unsigned short*>(x0);
                const unsigned short* h1 = reinterpret_cast<const unsigned short*>(x1);
                if (cb < c0) a = bf4_to_f4(*reinterpret_cast<const u32x2*>(h0 + (size_t)row * c0 + cb));
                else if (cb < cin) a = bf4_to_f4(*reinterpret_cast<const u32x2*>(h1 + (size_t)row * c1 + (cb - c0)));
            } else if (VEC4) {
                if (cb < c0) a = *reinterpret_cast<const f32x4*>(x0 + (size_t)row * c0 + cb);
                else if (cb < cin) a = *reinterpret_cast<const f32x4*>(x1 + (size_t)row * c1 + (cb - c0));
            } else {
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    const int ch = cb + t;
                    if (ch < c0) a[t] = x0[(size_t)row * c0 + ch];
                    else if (ch < cin) a[t] = x1[(size_t)row * c1 + (ch - c0)];
                }
            }
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[t], Bs[(cb + t) * 32 + lr], acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[t], Bs[(cb + t) * 32 + 16 + lr], acc1, 0, 0, 0);
            }
        }
        // C layout: column = lr (logit index), row = lq*4 + reg
        if (lr < 12) {
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const int r = tile * 16 + lq * 4 + t;
                if (r < rows) {
                    ag[(size_t)r * FGC_AG_LD + lr] = lr < FGC_M ? acc0[t] + cbias : 0.f;
                    ag[(size_t)r * FGC_AG_LD + 12 + lr] = lr < FGC_M ? acc1[t] : 0.f;
                }
            }
        }
    }
}

// Streaming form of proj_mfma_kernel for the network's fp32 shapes (cin = 16 KB with KB = 2, 4 or 8; both sources 16-byte
// aligned, the concat boundary on a multiple of 16 channels - the source of a 16-channel block is then wave-uniform: no
// exec-masked load).  Same products in the same order (bit-identical table).  What differs is how memory is asked for and
// answered: all KB fragments of a wave's NEXT 16-row tile are in flight while it multiplies the current one (the old loop
// waited for every 16-channel block before its eight MFMAs: four to sixteen dependent round trips per tile), the [u | v]
// fragments stay in registers, and the 16 x 24 result tile leaves through a 1.5 KB LDS tile as 96 coalesced 16-byte stores
// (1 536 contiguous bytes) instead of eight scalar stores per lane with a 96-byte stride.
template <int KB>
__global__ __launch_bounds__(256) void proj_stream_kernel(const float* __restrict__ x0, const float* __restrict__ x1, int c0, int c1,
                                                          int rows, const float* __restrict__ u, const float* __restrict__ c,
                                                          const float* __restrict__ v, float* __restrict__ ag) {
    __shared__ __attribute__((aligned(16))) float otile[4][16 * FGC_AG_LD];
    const int cin = c0 + c1;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, lr = lane & 15, lq = lane >> 4;
    // B fragments: element (kb, t) of lane (lr, lq) = [u | v][column lr][channel kb*16 + 4*lq + t] (columns >= 9: zero)
    f32x4 bu[KB], bv[KB];
#pragma unroll
    for (int kb = 0; kb < KB; ++kb) {
        const int ch = kb * 16 + 4 * lq;
        const int col = min(lr, FGC_M - 1);
        const f32x4 uu = *reinterpret_cast<const f32x4*>(u + (size_t)col * cin + ch);
        const f32x4 vv = *reinterpret_cast<const f32x4*>(v + (size_t)col * cin + ch);
        const float keep = lr < FGC_M ? 1.f : 0.f;
        bu[kb] = uu * keep;
        bv[kb] = vv * keep;
    }
    const float cbias = lr < FGC_M ? c[lr] : 0.f;
    const int ntile = (rows + 15) >> 4;
    const int stride = gridDim.x * 4;
    f32x4 an[KB], ac[KB];
    auto request = [&](int tile, f32x4 (&a)[KB]) {         // (clamped: always valid loads)
        const int row = min(min(tile, ntile - 1) * 16 + lr, rows - 1);
#pragma unroll
        for (int kb = 0; kb < KB; ++kb) {
            const int cb = kb * 16;                         // wave-uniform source
            const float* src = cb < c0 ? x0 + (size_t)row * c0 + cb : x1 + (size_t)row * c1 + (cb - c0);
            a[kb] = *reinterpret_cast<const f32x4*>(src + 4 * lq);
        }
    };
    int tile = blockIdx.x * 4 + wave;
    request(tile, an);
    float* ot = otile[wave];
    for (; tile < ntile; tile += stride) {
#pragma unroll
        for (int kb = 0; kb < KB; ++kb) ac[kb] = an[kb];
        request(tile + stride, an);
        f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int kb = 0; kb < KB; ++kb)
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(ac[kb][t], bu[kb][t], acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(ac[kb][t], bv[kb][t], acc1, 0, 0, 0);
            }
        // C layout: column = lr (logit index), row = lq*4 + reg  ->  the tile's 16 table rows in LDS
        if (lr < 12) {
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                ot[(lq * 4 + t) * FGC_AG_LD + lr] = lr < FGC_M ? acc0[t] + cbias : 0.f;
                ot[(lq * 4 + t) * FGC_AG_LD + 12 + lr] = lr < FGC_M ? acc1[t] : 0.f;
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        // 16 rows x 96 bytes = 96 pieces of 16 bytes, contiguous in the table
        const int nvalid = (min(rows - tile * 16, 16) * FGC_AG_LD) >> 2;
        f32x4* out = reinterpret_cast<f32x4*>(ag + (size_t)tile * 16 * FGC_AG_LD);
        const f32x4* in = reinterpret_cast<const f32x4*>(ot);
        if (lane < nvalid) out[lane] = in[lane];
        if (lane + 64 < nvalid) out[lane + 64] = in[lane + 64];
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
    }
}

// the table of a narrow first layer (cin <= 8): a row per thread on the vector ALU, the arithmetic of narrow_logits_row
__global__ __launch_bounds__(256) void proj_narrow_kernel(const float* __restrict__ x, int rows, int cin, const float* __restrict__ u,
                                                          const float* __restrict__ c, const float* __restrict__ v,
                                                          float* __restrict__ ag) {
    narrow_logits_body(x, rows, cin, u, c, v, ag, blockIdx.x, gridDim.x);
}

// ---------------------------------------------------------------------------------------------
// forward kernel
// ---------------------------------------------------------------------------------------------
template <int LPN, bool VEC4>
__global__ __launch_bounds__(NTHREADS) void conv_fwd_kernel(CoreParams p, FwdEpilogue ep) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    const Smem s = carve(smem_raw, ZSTRIDE);
    const int tile0 = block_tile0(p);
    const WaveTiling wt = wave_tiling(p.npad, threadIdx.x >> 6);

    const int dmine = softmax_phase<false>(p, s, tile0, 0, nullptr, nullptr);
    zero_zpad(p, s);
    const int nchunks = edge_chunks(s, dmine);  // block-uniform; contains the barrier

    f32x4 acc[RT][CTW];
#pragma unroll
    for (int r = 0; r < RT; ++r)
#pragma unroll
        for (int c = 0; c < CTW; ++c) acc[r][c] = f32x4{0.f, 0.f, 0.f, 0.f};

    for (int pass = 0; pass < p.passes; ++pass) {
        f32x4 z[FGC_M];
#pragma unroll
        for (int m = 0; m < FGC_M; ++m) z[m] = f32x4{0.f, 0.f, 0.f, 0.f};
        aggregate_pass<LPN, VEC4>(p, s, pass, 0, z);
        for (int ch = 1; ch < nchunks; ++ch) {  // rare: degree > 24
            __syncthreads();
            softmax_phase<false>(p, s, tile0, ch * KMAX, nullptr, nullptr);
            __syncthreads();
            aggregate_pass<LPN, VEC4>(p, s, pass, ch * KMAX, z);
        }
        if (nchunks > 1 && pass + 1 < p.passes) {  // restore chunk 0 for the next pass
            __syncthreads();
            softmax_phase<false>(p, s, tile0, 0, nullptr, nullptr);
            __syncthreads();
        }
        if (pass > 0) __syncthreads();  // previous pass' MFMA reads of ztile are done
        store_ztile<LPN>(p, s, z);
        __syncthreads();
        gemm_pass(p, s, pass, wt, acc);
    }
    __syncthreads();
    // accumulators -> LDS (aliases ztile)
    const int oldd = p.npad + 4;
    float* otile = s.ztile;
    store_acc(otile, oldd, wt, p.npad, acc);
    __syncthreads();

    // epilogue: thread handles (pooled row pr, column o): 4 consecutive nodes
    const int kparts = wt.kparts;
    for (int t = threadIdx.x; t < (TILE / 4) * p.nout; t += NTHREADS) {
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
            const int d = s.deg[row];
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

static size_t packed_floats(const ConvGeom& g) { return (size_t)g.passes * g.kpass * g.npad; }

}  // namespace fgc

using namespace fgc;

extern "C" size_t fgc_conv_workspace_bytes(const fgc_conv_desc* d) {
    FGC_OPT_SCOPE(d);
    if (!d) return 0;
    const ConvGeom g = conv_geom(d->c0 + d->c1, d->cout);
    size_t b = align_up(packed_floats(g) * sizeof(float), 256);
    if ((d->flags & FGC_CONV_SAVE_Z) && narrow_supported(d))
        b = std::max(b, align_up((size_t)d->n * narrow_zld(d->c0) * sizeof(float), 256));
    return b;
}

namespace fgc {

int validate_conv_desc(const fgc_conv_desc* d, const char* who) {
    FGC_CHECK_ARG(d != nullptr, "%s: null descriptor", who);
    FGC_CHECK_ARG(d->n > 0 && d->nnz >= 0, "%s: bad n=%d nnz=%d", who, d->n, d->nnz);
    FGC_CHECK_ARG(d->rowptr && d->col, "%s: null CSR", who);
    FGC_CHECK_ARG(d->x0 && d->c0 > 0, "%s: null x0 / c0=%d", who, d->c0);
    FGC_CHECK_ARG((d->x1 == nullptr) == (d->c1 == 0) && d->c1 >= 0, "%s: x1/c1 mismatch (c1=%d)", who, d->c1);
    FGC_CHECK_ARG(d->shift == 0 || d->shift == 2, "%s: shift must be 0 or 2 (got %d)", who, d->shift);
    FGC_CHECK_ARG(d->shift == 0 || (d->n % 4) == 0, "%s: upsampled input needs n %% 4 == 0 (n=%d)", who, d->n);
    FGC_CHECK_ARG(d->cout > 0 && d->cout <= MAX_NPAD, "%s: cout=%d outside [1,%d]", who, d->cout, MAX_NPAD);
    FGC_CHECK_ARG(d->c0 + d->c1 <= MAX_NPAD, "%s: cin=%d above %d", who, d->c0 + d->c1, MAX_NPAD);
    FGC_CHECK_ARG(d->W0 && d->b && d->u && d->c && d->v, "%s: null parameter pointer", who);
    return FGC_OK;
}

bool conv_vec4_ok(const fgc_conv_desc* d) {
    const bool al = ((uintptr_t)d->x0 % 16 == 0) && (d->x1 == nullptr || (uintptr_t)d->x1 % 16 == 0);
    return al && d->c0 % 4 == 0 && d->c1 % 4 == 0;
}

void fill_core_params(CoreParams& p, const ConvGeom& g, int n, const int* rowptr, const int* col, const int* eid,
                      const float* s0, const float* s1, int c0, int c1, int shift, int nout, const float* ag,
                      int ag_shift, int ctr_off, int nbr_off, const float* Wp) {
    p.n = n;
    p.rowptr = rowptr;
    p.col = col;
    p.eid = eid;
    p.src0 = s0;
    p.src1 = s1;
    p.c0 = c0;
    p.c1 = c1;
    p.shift = shift;
    p.cg = c0 + c1;
    p.nout = nout;
    p.npad = g.npad;
    p.passes = g.passes;
    p.kc = g.kc;
    p.kpass = g.kpass;
    p.zstride = g.zstride;
    p.ag = ag;
    p.ag_shift = ag_shift;
    p.ctr_off = ctr_off;
    p.nbr_off = nbr_off;
    p.Wp = Wp;
    p.tile_list = nullptr;
    p.n_tiles = 0;
}

size_t conv_smem_bytes(const ConvGeom& g, size_t extra) {
    size_t core = smem_core_bytes(g.zstride);
    // the out tile aliases ztile: [kparts<=4][TILE][npad+4]
    const int nct = g.npad / 16;
    const int kparts = nct >= 3 ? 1 : (nct == 2 ? 2 : 4);
    const size_t ot = (size_t)kparts * TILE * (g.npad + 4) * 4;
    const size_t zt = (size_t)TILE * g.zstride * 4;
    if (ot > zt) core += ot - zt;
    return core + extra;
}

}  // namespace fgc

template <int LPN>
static int launch_fwd(const CoreParams& p, const FwdEpilogue& ep, bool vec4, size_t smem, hipStream_t st) {
    const int grid = core_grid(p);
    if (vec4) {
        hipFuncSetAttribute((const void*)conv_fwd_kernel<LPN, true>, hipFuncAttributeMaxDynamicSharedMemorySize,
                            (int)smem);
        FGC_LAUNCH("conv_fwd_kernel<LPN, true>", st, (conv_fwd_kernel<LPN, true>), dim3(grid), dim3(NTHREADS), smem, p, ep);
    } else {
        hipFuncSetAttribute((const void*)conv_fwd_kernel<LPN, false>, hipFuncAttributeMaxDynamicSharedMemorySize,
                            (int)smem);
        FGC_LAUNCH("conv_fwd_kernel<LPN, false>", st, (conv_fwd_kernel<LPN, false>), dim3(grid), dim3(NTHREADS), smem, p, ep);
    }
    FGC_CHECK_LAUNCH("fgc_conv_fwd");
    return FGC_OK;
}

extern "C" int fgc_conv_fwd(const fgc_conv_desc* d, float* ag, float* y, float* y_pool, void* workspace,
                            size_t workspace_bytes, void* stream) {
    FGC_OPT_SCOPE(d);
    int rc = validate_conv_desc(d, "fgc_conv_fwd");
    if (rc) return rc;
    FGC_CHECK_ARG(ag && y, "fgc_conv_fwd: null ag / y");
    FGC_CHECK_ARG(y_pool == nullptr || d->n % 4 == 0, "fgc_conv_fwd: pooled output needs n %% 4 == 0 (n=%d)", d->n);
    const int cin = d->c0 + d->c1;
    const ConvGeom g = conv_geom(cin, d->cout);
    FGC_CHECK_ARG(workspace && workspace_bytes >= fgc_conv_workspace_bytes(d),
                  "fgc_conv_fwd: workspace too small (%zu < %zu)", workspace_bytes, fgc_conv_workspace_bytes(d));
    FGC_CHECK_ARG((uintptr_t)workspace % 16 == 0 && (uintptr_t)ag % 16 == 0, "fgc_conv_fwd: workspace/ag need 16-byte alignment");
    FGC_CHECK_ARG(!(d->flags & FGC_CONV_PACKED) || d->packed_layout == 0 || d->packed_layout == conv_layout_id(d),
                  "fgc_conv_fwd: FGC_CONV_PACKED, but the operands were packed in layout %llu and the options now select %llu "
                  "(an option changed between fgc_conv_pack and this call)", (unsigned long long)d->packed_layout,
                  (unsigned long long)conv_layout_id(d));
    hipStream_t st = (hipStream_t)stream;
    float* Wp = (float*)workspace;
    const int rows = d->src_rows > 0 ? d->src_rows : (d->n >> d->shift);
    const int prow0 = d->proj_rows ? d->proj_row0 : 0;
    const int prows = d->proj_rows ? (d->proj_rows < 0 ? 0 : d->proj_rows) : rows;
    FGC_CHECK_ARG(prow0 >= 0 && prow0 + prows <= rows && (d->proj_rows != 0 || d->proj_row0 == 0),
                  "fgc_conv_fwd: logits rows [%d, %d) outside the %d source rows", prow0, prow0 + prows, rows);
    FGC_CHECK_ARG(d->tile_list == nullptr || (d->n_tiles >= 0 && d->n_tiles <= cdiv(d->n, TILE)),
                  "fgc_conv_fwd: n_tiles=%d outside [0, %d]", d->n_tiles, cdiv(d->n, TILE));

    if (pairs_ok(d)) {   // 4x-upsampled input, pair graph given: the layer on its coarse source rows (fgc_conv_pair.hip)
        FGC_CHECK_ARG(y_pool == nullptr, "fgc_conv_fwd: the pair form has no pooled output");
        return launch_pair_fwd(d, ag, y, workspace, st);
    }
    const bool narrow = narrow_supported(d);   // cin <= 8: vector-ALU kernel, no packed operand (fgc_conv_narrow.hip)
    const bool bf16 = (d->flags & FGC_CONV_BF16) != 0;
    FGC_CHECK_ARG(!bf16 || narrow || (conv_vec4_ok(d) && cin % 32 == 0 && (d->c1 == 0 || d->c0 % 32 == 0) && d->cout % 32 == 0),
                  "fgc_conv_fwd: FGC_CONV_BF16 needs widths that are multiples of 32 and 16-byte aligned tensors (c0=%d c1=%d "
                  "cout=%d)", d->c0, d->c1, d->cout);
    if (!narrow && !(d->flags & FGC_CONV_PACKED) && bf16) {
        PackJobs J;
        J.njobs = 1;
        const size_t tot = (size_t)g.passes * 9 * (g.npad >> 4) * 512;
        J.job[0] = PackJob{d->W0, Wp, 4, cin, d->cout, cin, d->cout, g.npad, g.kc, g.kpass, g.passes, 0, 0};
        J.nblocks = cdiv((int)tot, 1024);
        FGC_LAUNCH("pack_many_kernel", st, pack_many_kernel, dim3(J.nblocks), dim3(256), 0, J);
        FGC_CHECK_LAUNCH("fgc_conv_fwd/pack");
    } else if (!narrow && !(d->flags & FGC_CONV_PACKED)) {
        const size_t tot = packed_floats(g);
        FGC_LAUNCH("pack_weight_kernel", st, pack_weight_kernel, dim3(cdiv((int)tot, 256 * 4)), dim3(256), 0, d->W0, Wp, cin,
                   d->cout, cin, d->cout, g.npad, g.kc, g.kpass, g.passes, 0);
        FGC_CHECK_LAUNCH("fgc_conv_fwd/pack");
    }
    if (prows > 0) {
        const int pg = std::min(cdiv(cdiv(prows, 16), 4), 1024);
        const bool xbf = bf16 && !narrow;        // (a narrow first layer reads its fp32 input)
        const size_t esz = xbf ? 2 : 4;
        const float* px0 = reinterpret_cast<const float*>(reinterpret_cast<const char*>(d->x0) + (size_t)prow0 * d->c0 * esz);
        const float* px1 = d->x1 ? reinterpret_cast<const float*>(reinterpret_cast<const char*>(d->x1) + (size_t)prow0 * d->c1 * esz)
                                 : nullptr;
        float* pag = ag + (size_t)prow0 * FGC_AG_LD;
        if (narrow)
            FGC_LAUNCH("proj_mfma_kernel", st, proj_narrow_kernel, dim3(cdiv(prows, 256)), dim3(256), 0, px0, prows, d->c0, d->u, d->c,
                       d->v, pag);
        else if (xbf)
            FGC_LAUNCH("proj_mfma_kernel", st, (proj_mfma_kernel<true, true>), dim3(pg), dim3(256), 0, px0, px1, d->c0, d->c1,
                       prows, d->u, d->c, d->v, pag);
        else if (conv_vec4_ok(d) && (cin == 32 || cin == 64 || cin == 128) && d->c0 % 16 == 0 && (uintptr_t)pag % 16 == 0 &&
                 ((uintptr_t)d->u | (uintptr_t)d->v) % 16 == 0 && opt(OPT_NO_PROJ_STREAM) != 1) {
            if (cin == 32)
                FGC_LAUNCH("proj_mfma_kernel", st, (proj_stream_kernel<2>), dim3(pg), dim3(256), 0, px0, px1, d->c0, d->c1, prows, d->u,
                           d->c, d->v, pag);
            else if (cin == 64)
                FGC_LAUNCH("proj_mfma_kernel", st, (proj_stream_kernel<4>), dim3(pg), dim3(256), 0, px0, px1, d->c0, d->c1, prows, d->u,
                           d->c, d->v, pag);
            else
                FGC_LAUNCH("proj_mfma_kernel", st, (proj_stream_kernel<8>), dim3(pg), dim3(256), 0, px0, px1, d->c0, d->c1, prows, d->u,
                           d->c, d->v, pag);
        } else if (conv_vec4_ok(d))
            FGC_LAUNCH("proj_mfma_kernel", st, (proj_mfma_kernel<true>), dim3(pg), dim3(256), 0, px0, px1, d->c0, d->c1,
                       prows, d->u, d->c, d->v, pag);
        else
            FGC_LAUNCH("proj_mfma_kernel", st, (proj_mfma_kernel<false>), dim3(pg), dim3(256), 0, px0, px1, d->c0, d->c1,
                       prows, d->u, d->c, d->v, pag);
        FGC_CHECK_LAUNCH("fgc_conv_fwd/proj");
    }
    if (d->tile_list && d->n_tiles == 0) return FGC_OK;
    if (narrow)
        return launch_narrow_fwd(d, ag, y, y_pool, (d->flags & FGC_CONV_SAVE_Z) ? (float*)workspace : nullptr, st, bf16);

    CoreParams p;
    fill_core_params(p, g, d->n, d->rowptr, d->col, nullptr, d->x0, d->x1, d->c0, d->c1, d->shift, d->cout, ag,
                     d->shift, 0, 12, Wp);
    p.tile_list = d->tile_list;
    p.n_tiles = d->n_tiles;
    FwdEpilogue ep{d->b, d->bias_mask, d->act, d->alpha, y, y_pool};
    const size_t smem = conv_smem_bytes(g, 0);
    const bool vec4 = conv_vec4_ok(d);
    if (bf16) {
        FGC_CHECK_ARG(w8_bf16_supported(p, d->max_deg), "fgc_conv_fwd: FGC_CONV_BF16: unsupported shape (cin=%d cout=%d "
                      "max_deg=%d)", cin, d->cout, d->max_deg);
        return launch_fwd_w8(p, ep, smem, d->max_deg, st, true);
    }
    if (g.lpn == 8 && w8_supported(p, d->max_deg)) return launch_fwd_w8(p, ep, smem, d->max_deg, st);
    return launch_fwd<8>(p, ep, vec4, smem, st);
}
