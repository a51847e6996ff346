// Forward graph convolution (replaces custom_conv2d, /root/reference/Code/model.py:427-504).
#include <stdlib.h>

#include "fgc_conv_pc.h"

namespace fgc {

// ---------------------------------------------------------------------------------------------
// weight packing: W0[m][o][c] -> k-interleaved B operand of the aggregate-first GEMM
//   row kk = pass*kpass + m*kc + cl  (c = pass*kc + cl), column = o, stored [kk/4][npad][kk%4]
// transposed = 1 packs the data-gradient operand instead: k runs over (pass, m, ol) with
// o = pass*kc + ol and the column is c.
// ---------------------------------------------------------------------------------------------
__global__ void pack_weight_kernel(const float* __restrict__ W0, float* __restrict__ Wp, int cin, int cout,
                                   int kdim, int ncols, int npad, int kc, int kpass, int passes, int transposed) {
    const size_t total = (size_t)passes * kpass * npad;
    for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
         idx += (size_t)gridDim.x * blockDim.x) {
        const int e = idx & 3;
        const size_t v4 = idx >> 2;
        const int colp = v4 % npad;
        const int kk = (int)(v4 / npad) * 4 + e;
        const int pass = kk / kpass, kin = kk % kpass;
        const int m = kin / kc, cl = kin % kc;
        const int kch = pass * kc + cl;
        float val = 0.f;
        if (m < FGC_M && kch < kdim && colp < ncols) {
            val = transposed ? W0[((size_t)m * cout + kch) * cin + colp] : W0[((size_t)m * cout + colp) * cin + kch];
        }
        Wp[idx] = val;
    }
}

// ---------------------------------------------------------------------------------------------
// assignment logits: ag[r][m] = u[m].x_r + c[m], ag[r][12+m] = v[m].x_r   (model.py:79-80,94)
// one lane per source row; x staged through LDS in 32-channel chunks.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void proj_kernel(const float* __restrict__ x0, const float* __restrict__ x1,
                                                   int c0, int c1, int rows, const float* __restrict__ u,
                                                   const float* __restrict__ c, const float* __restrict__ v,
                                                   float* __restrict__ ag) {
    __shared__ float xs[256][33];
    __shared__ float us[FGC_M][32];
    __shared__ float vs[FGC_M][32];
    const int cin = c0 + c1;
    const int r0 = blockIdx.x * 256;
    const int tid = threadIdx.x;
    float a[FGC_M], g[FGC_M];
#pragma unroll
    for (int m = 0; m < FGC_M; ++m) {
        a[m] = 0.f;
        g[m] = 0.f;
    }
    for (int cb = 0; cb < cin; cb += 32) {
        const int cw = min(32, cin - cb);
        __syncthreads();
        for (int t = tid; t < 256 * 32; t += 256) {
            const int rr = t >> 5, cc = t & 31;
            const int r = r0 + rr, ch = cb + cc;
            float val = 0.f;
            if (r < rows && cc < cw) val = ch < c0 ? x0[(size_t)r * c0 + ch] : x1[(size_t)r * c1 + (ch - c0)];
            xs[rr][cc] = val;
        }
        for (int t = tid; t < FGC_M * 32; t += 256) {
            const int m = t >> 5, cc = t & 31;
            us[m][cc] = cc < cw ? u[m * cin + cb + cc] : 0.f;
            vs[m][cc] = cc < cw ? v[m * cin + cb + cc] : 0.f;
        }
        __syncthreads();
        for (int cc = 0; cc < cw; ++cc) {
            const float xv = xs[tid][cc];
#pragma unroll
            for (int m = 0; m < FGC_M; ++m) {
                a[m] = fmaf(us[m][cc], xv, a[m]);
                g[m] = fmaf(vs[m][cc], xv, g[m]);
            }
        }
    }
    const int r = r0 + tid;
    if (r < rows) {
        float* o = ag + (size_t)r * FGC_AG_LD;
#pragma unroll
        for (int m = 0; m < FGC_M; ++m) {
            o[m] = a[m] + c[m];
            o[12 + m] = g[m];
        }
        o[9] = o[10] = o[11] = 0.f;
        o[21] = o[22] = o[23] = 0.f;
    }
}

// ---------------------------------------------------------------------------------------------
// forward kernel
// ---------------------------------------------------------------------------------------------
template <int LPN, bool VEC4>
__global__ __launch_bounds__(NTHREADS) void conv_fwd_kernel(CoreParams p, FwdEpilogue ep) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    const Smem s = carve(smem_raw, p.zstride);
    const int tile0 = blockIdx.x * TILE;
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
    if (!d) return 0;
    const ConvGeom g = conv_geom(d->c0 + d->c1, d->cout);
    return align_up(packed_floats(g) * sizeof(float), 256);
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
    const int grid = cdiv(p.n, TILE);
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
    int rc = validate_conv_desc(d, "fgc_conv_fwd");
    if (rc) return rc;
    FGC_CHECK_ARG(ag && y, "fgc_conv_fwd: null ag / y");
    FGC_CHECK_ARG(y_pool == nullptr || d->n % 4 == 0, "fgc_conv_fwd: pooled output needs n %% 4 == 0 (n=%d)", d->n);
    const int cin = d->c0 + d->c1;
    const ConvGeom g = conv_geom(cin, d->cout);
    FGC_CHECK_ARG(workspace && workspace_bytes >= fgc_conv_workspace_bytes(d),
                  "fgc_conv_fwd: workspace too small (%zu < %zu)", workspace_bytes, fgc_conv_workspace_bytes(d));
    FGC_CHECK_ARG((uintptr_t)workspace % 16 == 0 && (uintptr_t)ag % 16 == 0, "fgc_conv_fwd: workspace/ag need 16-byte alignment");
    hipStream_t st = (hipStream_t)stream;
    float* Wp = (float*)workspace;

    const size_t tot = packed_floats(g);
    FGC_LAUNCH("pack_weight_kernel", st, pack_weight_kernel, dim3(cdiv((int)tot, 256 * 4)), dim3(256), 0, d->W0, Wp, cin, d->cout,
                       cin, d->cout, g.npad, g.kc, g.kpass, g.passes, 0);
    FGC_CHECK_LAUNCH("fgc_conv_fwd/pack");
    const int rows = d->src_rows > 0 ? d->src_rows : (d->n >> d->shift);
    FGC_LAUNCH("proj_kernel", st, proj_kernel, dim3(cdiv(rows, 256)), dim3(256), 0, d->x0, d->x1, d->c0, d->c1, rows, d->u,
                       d->c, d->v, ag);
    FGC_CHECK_LAUNCH("fgc_conv_fwd/proj");

    CoreParams p;
    fill_core_params(p, g, d->n, d->rowptr, d->col, nullptr, d->x0, d->x1, d->c0, d->c1, d->shift, d->cout, ag,
                     d->shift, 0, 12, Wp);
    FwdEpilogue ep{d->b, d->bias_mask, d->act, d->alpha, y, y_pool};
    const size_t smem = conv_smem_bytes(g, 0);
    const bool vec4 = conv_vec4_ok(d);
    if (g.lpn == 8 && vec4 && d->max_deg > 0 && d->max_deg <= KMAX && (getenv("FGC_PC") && getenv("FGC_PC")[0] == '1'))
        return launch_fwd_pc(p, ep, g, st);
    switch (g.lpn) {
        case 2: return launch_fwd<2>(p, ep, vec4, smem, st);
        case 4: return launch_fwd<4>(p, ep, vec4, smem, st);
        default: return launch_fwd<8>(p, ep, vec4, smem, st);
    }
}
