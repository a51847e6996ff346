// custom_lin (model.py:763-769) on its own: y = x W + b and its gradients through libfgc, for callers of the operator API
// that compose custom_lin -> lrelu -> custom_lin themselves (model.py:937-941).  The network path never comes here: its two
// linear layers live in fgc_mlp_fwd / fgc_mlp_bwd with the 1024-wide hidden layer kept on chip.
//
// One strided fp32 GEMM kernel on v_mfma_f32_16x16x4_f32 (exact fp32 products, fp32 sums) serves the three products:
//   forward   y  [n, cout]     = x  [n, cin]   . W [cin, cout] + b
//   backward  dx [n, cin]      = dy [n, cout]  . W^T
//             [dW; db] [cin + 1, cout] = [x | 1]^T . dy        (K = n: split over workgroups, fixed-order slab sum)
// Workgroup = 64 x 64 tile of the product, four waves of 32 x 32, 16-deep steps staged through LDS with generic strides
// (each operand is read once per tile row / column of workgroups; at these shapes the products are bound by HBM and the
// fp32 matrix pipe like every dense product of the path).
#include <algorithm>

#include "fgc_common.h"
#include "fgc_reduce.h"

namespace fgc {

constexpr int LIN_BM = 64, LIN_BN = 64, LIN_BK = 16;

struct LinOperand {
    const float* p;
    long rs, cs;     // element (i, j) at p[i * rs + j * cs]
    int rows, cols;  // extent; reads outside are 0 (ones_row: row == rows reads 1)
    int ones_row;
};

__device__ __forceinline__ float lin_at(const LinOperand& o, int i, int j) {
    if (i < o.rows && j < o.cols) return o.p[(long)i * o.rs + (long)j * o.cs];
    return (o.ones_row && i == o.rows && j < o.cols) ? 1.f : 0.f;
}

// C[M, N] (+ bias[N]) = A[M, K] . B[K, N] over k in [k0, k1) of this workgroup's split; split s writes slab s
__global__ __launch_bounds__(256) void lin_gemm_kernel(LinOperand A, LinOperand B, int M, int N, int K, int ksplit,
                                                       const float* __restrict__ bias, float* __restrict__ C, long ldc,
                                                       long slab_stride) {
    __shared__ float As[LIN_BK][LIN_BM + 4];   // k-major: the MFMA A fragment reads 16 consecutive rows of one k
    __shared__ float Bs[LIN_BK][LIN_BN + 4];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, lr = lane & 15, lq = lane >> 4;
    const int m0 = blockIdx.x * LIN_BM, n0 = blockIdx.y * LIN_BN;   // (row tiles on x: up to 2^31 of them)
    const int split = blockIdx.z;
    const int kper = ((K + ksplit - 1) / ksplit + LIN_BK - 1) / LIN_BK * LIN_BK;
    const int k0 = split * kper, k1 = min(K, k0 + kper);
    const int wm = (wave >> 1) * 32, wn = (wave & 1) * 32;
    f32x4 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    for (int kb = k0; kb < k1; kb += LIN_BK) {
        // 64 x 16 elements of each operand, four per thread; the faster-varying thread index follows the operand's
        // unit-stride direction where there is one
        float av[4], bv[4];
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const int e = tid + t * 256;
            int am, ak, bk, bn;
            if (A.cs == 1) { ak = e & 15; am = e >> 4; } else { am = e & 63; ak = e >> 6; }
            if (B.cs == 1) { bn = e & 63; bk = e >> 6; } else { bk = e & 15; bn = e >> 4; }
            av[t] = (kb + ak < k1) ? lin_at(A, m0 + am, kb + ak) : 0.f;
            bv[t] = (kb + bk < k1) ? lin_at(B, kb + bk, n0 + bn) : 0.f;
        }
        __syncthreads();
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const int e = tid + t * 256;
            int am, ak, bk, bn;
            if (A.cs == 1) { ak = e & 15; am = e >> 4; } else { am = e & 63; ak = e >> 6; }
            if (B.cs == 1) { bn = e & 63; bk = e >> 6; } else { bk = e & 15; bn = e >> 4; }
            As[ak][am] = av[t];
            Bs[bk][bn] = bv[t];
        }
        __syncthreads();
#pragma unroll
        for (int kk = 0; kk < LIN_BK; kk += 4) {
            float a[2], b[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) a[i] = As[kk + lq][wm + i * 16 + lr];
#pragma unroll
            for (int j = 0; j < 2; ++j) b[j] = Bs[kk + lq][wn + j * 16 + lr];
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i], b[j], acc[i][j], 0, 0, 0);
        }
    }
    // C layout of the 16 x 16 MFMA: column = lr, row = 4 * lq + register
    float* out = C + (long)split * slab_stride;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const int row = m0 + wm + i * 16 + lq * 4 + t, col = n0 + wn + j * 16 + lr;
                if (row < M && col < N) out[(long)row * ldc + col] = acc[i][j][t] + (bias ? bias[col] : 0.f);
            }
}

static int lin_splits(int n) {
    // [dW; db]: K = rows of x.  One split per 2048 rows, at most 256 slabs
    return std::max(1, std::min(256, cdiv(n, 2048)));
}

}  // namespace fgc

using namespace fgc;

extern "C" size_t fgc_lin_bwd_workspace_bytes(int32_t n, int32_t cin, int32_t cout) {
    const int ns = lin_splits(n);
    const size_t count = (size_t)(cin + 1) * cout;
    return align_up(((size_t)ns * count + reduce_tmp_floats(ns, count) + 64) * 4, 256);
}

extern "C" int fgc_lin_fwd(const float* x, int32_t n, int32_t cin, int32_t cout, const float* W, const float* b, float* y,
                           void* stream) {
    FGC_CHECK_ARG(x && W && y && n > 0 && cin > 0 && cout > 0, "fgc_lin_fwd: bad arguments (n=%d cin=%d cout=%d)", n, cin, cout);
    hipStream_t st = (hipStream_t)stream;
    const LinOperand A{x, cin, 1, n, cin, 0}, B{W, cout, 1, cin, cout, 0};
    FGC_LAUNCH("lin_gemm_kernel:fwd", st, lin_gemm_kernel, dim3(cdiv(n, LIN_BM), cdiv(cout, LIN_BN), 1), dim3(256), 0, A, B, n,
               cout, cin, 1, b, y, (long)cout, 0L);
    FGC_CHECK_LAUNCH("fgc_lin_fwd");
    return FGC_OK;
}

extern "C" int fgc_lin_bwd(const float* x, const float* dy, int32_t n, int32_t cin, int32_t cout, const float* W, float* dx,
                           float* dW, float* db, void* workspace, size_t workspace_bytes, void* stream) {
    FGC_CHECK_ARG(x && dy && W && dW && db && n > 0 && cin > 0 && cout > 0, "fgc_lin_bwd: bad arguments (n=%d cin=%d cout=%d)",
                  n, cin, cout);
    FGC_CHECK_ARG(workspace && workspace_bytes >= fgc_lin_bwd_workspace_bytes(n, cin, cout) && (uintptr_t)workspace % 16 == 0,
                  "fgc_lin_bwd: workspace too small or misaligned (%zu < %zu)", workspace_bytes,
                  fgc_lin_bwd_workspace_bytes(n, cin, cout));
    hipStream_t st = (hipStream_t)stream;
    if (dx) {   // dx = dy . W^T: B(k = output channel, j = input channel) = W[j, k]
        const LinOperand A{dy, cout, 1, n, cout, 0}, B{W, 1, cout, cout, cin, 0};
        FGC_LAUNCH("lin_gemm_kernel:dx", st, lin_gemm_kernel, dim3(cdiv(n, LIN_BM), cdiv(cin, LIN_BN), 1), dim3(256), 0, A, B, n,
                   cin, cout, 1, (const float*)nullptr, dx, (long)cin, 0L);
    }
    // [dW; db] = [x | 1]^T . dy: A(i = input channel or the ones row, k = node) = x[k, i]
    const int ns = lin_splits(n);
    const size_t count = (size_t)(cin + 1) * cout;
    float* slab = (float*)workspace;
    float* tmp = slab + (size_t)ns * count;
    const LinOperand A{x, 1, cin, cin, n, 1}, B{dy, cout, 1, n, cout, 0};
    FGC_LAUNCH("lin_gemm_kernel:dW", st, lin_gemm_kernel, dim3(cdiv(cin + 1, LIN_BM), cdiv(cout, LIN_BN), ns), dim3(256), 0, A, B,
               cin + 1, cout, n, ns, (const float*)nullptr, slab, (long)cout, (long)count);
    FGC_CHECK_LAUNCH("fgc_lin_bwd");
    const RedJob jobs[2] = {
        {slab, count, ns, cin * cout, cout, cout, dW, tmp},
        {slab + (size_t)cin * cout, count, ns, cout, cout, cout, db},
    };
    return reduce_jobs("reduce:lin", jobs, 2, nullptr, st);
}
