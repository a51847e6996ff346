// Operand packing of the conv weights, as device functions so that one launch can pack for several layers.
#pragma once
#include "fgc_common.h"

namespace fgc {

// W0[m][o][c] -> k-interleaved B operand of the aggregate-first GEMM
//   row kk = pass*kpass + m*kc + cl  (c = pass*kc + cl), column = o, stored [kk/4][npad][kk%4]
// transposed = 1 packs the data-gradient operand instead: k runs over (pass, m, ol) with o = pass*kc + ol and the
// column is c.  Workgroup `bid` of `nb` (256 threads each).
__device__ __forceinline__ void pack_weight_body(const float* __restrict__ W0, float* __restrict__ Wp, int cin, int cout,
                                                 int kdim, int ncols, int npad, int kc, int kpass, int passes,
                                                 int transposed, int bid, int nb) {
    const size_t total = (size_t)passes * kpass * npad;
    for (size_t idx = (size_t)bid * blockDim.x + threadIdx.x; idx < total; idx += (size_t)nb * blockDim.x) {
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

// d-logits operand: Wq[pass][o/4][kk][o%4] = W0[m][o][pass*kc+cl], kk = m*kc+cl  (K = cout, N = kpass)
__device__ __forceinline__ void pack_logit_weight_body(const float* __restrict__ W0, float* __restrict__ Wq, int cin,
                                                       int cout, int opad, int kc, int kpass, int passes, int bid,
                                                       int nb) {
    const size_t total = (size_t)passes * opad * kpass;
    for (size_t idx = (size_t)bid * blockDim.x + threadIdx.x; idx < total; idx += (size_t)nb * blockDim.x) {
        const int e = idx & 3;
        const size_t v4 = idx >> 2;
        const int kk = v4 % kpass;
        const size_t rest = v4 / kpass;
        const int o4 = rest % (opad >> 2);
        const int pass = (int)(rest / (opad >> 2));
        const int o = o4 * 4 + e;
        const int m = kk / kc, cl = kk % kc;
        const int c = pass * kc + cl;
        Wq[idx] = (m < FGC_M && o < cout && c < cin) ? W0[((size_t)m * cout + o) * cin + c] : 0.f;
    }
}

}  // namespace fgc
