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

// ---- bf16 operands (FGC_CONV_BF16) for v_mfma_f32_16x16x32_bf16: a lane's B fragment is 8 consecutive k of ONE column,
// so the operands are stored fragment by fragment, [pass][k-step][column tile][lane][8]: one 16-byte load per MFMA.
//   lane l of a fragment: k = k-step*32 + 8*(l>>4) + j (j = 0..7), column = column tile*16 + (l&15)
// Aggregate-first GEMM (transposed = 0) / data-gradient GEMM (transposed = 1), k = m*32 + cl as in pack_weight_body.
__device__ __forceinline__ void pack_weight_bf16_body(const float* __restrict__ W0, unsigned short* __restrict__ Wp,
                                                      int cin, int cout, int kdim, int ncols, int npad, int passes,
                                                      int transposed, int bid, int nb) {
    const int nct = npad >> 4;
    const size_t total = (size_t)passes * 9 * nct * 512;
    for (size_t idx = (size_t)bid * blockDim.x + threadIdx.x; idx < total; idx += (size_t)nb * blockDim.x) {
        const int j = idx & 7, lane = (idx >> 3) & 63;
        const size_t rest = idx >> 9;
        const int ct = rest % nct, ks = (rest / nct) % 9, pass = (int)(rest / ((size_t)nct * 9));
        const int k = ks * 32 + 8 * (lane >> 4) + j;
        const int m = k >> 5, cl = k & 31;
        const int kch = pass * 32 + cl, colp = ct * 16 + (lane & 15);
        float val = 0.f;
        if (kch < kdim && colp < ncols)
            val = transposed ? W0[((size_t)m * cout + kch) * cin + colp] : W0[((size_t)m * cout + colp) * cin + kch];
        Wp[idx] = f_to_bf(val);
    }
}
// d-logits operand: dz[node][kk = m*32 + cl] = sum_o s[node][o] W0[m][o][pass*32 + cl]; k = o (cout / 32 k-steps), 18
// column tiles of kk
__device__ __forceinline__ void pack_logit_weight_bf16_body(const float* __restrict__ W0, unsigned short* __restrict__ Wq,
                                                            int cin, int cout, int passes, int bid, int nb) {
    const int kso = (cout + 31) >> 5;
    const size_t total = (size_t)passes * kso * 18 * 512;
    for (size_t idx = (size_t)bid * blockDim.x + threadIdx.x; idx < total; idx += (size_t)nb * blockDim.x) {
        const int j = idx & 7, lane = (idx >> 3) & 63;
        const size_t rest = idx >> 9;
        const int ct = rest % 18, ks = (rest / 18) % kso, pass = (int)(rest / ((size_t)18 * kso));
        const int o = ks * 32 + 8 * (lane >> 4) + j;
        const int kk = ct * 16 + (lane & 15);
        const int m = kk >> 5, cl = kk & 31;
        const int c = pass * 32 + cl;
        Wq[idx] = f_to_bf((o < cout && c < cin) ? W0[((size_t)m * cout + o) * cin + c] : 0.f);
    }
}

// pair form, bf16 storage: a plain bf16 copy of W0 [9 cout, cin] (the transform kernel reads it in its native layout)
__device__ __forceinline__ void pack_plain_bf16_body(const float* __restrict__ W0, unsigned short* __restrict__ Wb, int count,
                                                     int bid, int nb) {
    for (int i = bid * blockDim.x + threadIdx.x; i < count; i += nb * blockDim.x) Wb[i] = f_to_bf(W0[i]);
}

// every packed operand of several layers in one launch (fgc_conv_pack)
struct PackJob {
    const float* W0;
    float* dst;
    int kind;   // 0: forward operand, 1: data-gradient operand (transposed), 2: d-logits operand; +4: the bf16 forms;
                // 7: plain bf16 copy of W0 (pair form; kdim = element count)
    int cin, cout, kdim, ncols, npad, kc, kpass, passes, opad;
    int block0;
};
constexpr int PACK_MAX_JOBS = 24;
struct PackJobs {
    PackJob job[PACK_MAX_JOBS];
    int njobs, nblocks;
};
__global__ void pack_many_kernel(PackJobs J);

}  // namespace fgc
