// Device helpers of the fp32-on-the-bf16-matrix-pipe products (three-term operand splits): shared by the per-facet MLP
// (fgc_mlp_bf16.hip, where the scheme is described) and the dz GEMM of the d-logits kernel (fgc_conv_bwd.hip).
#pragma once
#include "fgc_common.h"

namespace fgc {

// v = p0 + p1 + p2, each a bf16 (round to nearest even): p0 = bf16(v), p1 = bf16(v - p0), p2 = bf16(v - p0 - p1)
__device__ __forceinline__ void split3(const f32x4& v, u32x2& p0, u32x2& p1, u32x2& p2) {
    p0 = f4_to_bf4(v);
    const f32x4 r1 = v - bf4_to_f4(p0);
    p1 = f4_to_bf4(r1);
    const f32x4 r2 = r1 - bf4_to_f4(p1);
    p2 = f4_to_bf4(r2);
}


// split eight fp32 values (two f32x4: fragment elements 0-3 and 4-7) into the three planes of one A / B fragment
__device__ __forceinline__ void split3_frag(const f32x4& lo, const f32x4& hi, u32x4 (&p)[3]) {
    u32x2 l[3], h[3];
    split3(lo, l[0], l[1], l[2]);
    split3(hi, h[0], h[1], h[2]);
#pragma unroll
    for (int q = 0; q < 3; ++q) p[q] = u32x4{l[q][0], l[q][1], h[q][0], h[q][1]};
}


// six MFMAs of one split product, smallest terms first
__device__ __forceinline__ f32x4 mfma_split(const u32x4 (&a)[3], const u32x4 (&b)[3], f32x4 acc) {
#define FGC_M16(A_, B_) acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, A_), __builtin_bit_cast(bf16x8, B_), acc, 0, 0, 0)
    FGC_M16(a[0], b[2]);
    FGC_M16(a[2], b[0]);
    FGC_M16(a[1], b[1]);
    FGC_M16(a[0], b[1]);
    FGC_M16(a[1], b[0]);
    FGC_M16(a[0], b[0]);
#undef FGC_M16
    return acc;
}

// The same product into TWO accumulators: the five small terms into `lo`, a0 b0 into `hi`.  For a sum that runs over many
// calls (dx over the 1024 hidden columns): added to one accumulator that already holds the large partial sum, every small
// term is rounded to that sum's last place by the matrix pipe's adder - measured 6e-7 of max |dx| against 1.4e-7 for the
// fp32 MFMA kernel; kept among themselves they keep their bits until the one addition at the end.
__device__ __forceinline__ void mfma_split2(const u32x4 (&a)[3], const u32x4 (&b)[3], f32x4& hi, f32x4& lo) {
#define FGC_M16(A_, B_, ACC_) ACC_ = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, A_), __builtin_bit_cast(bf16x8, B_), ACC_, 0, 0, 0)
    FGC_M16(a[0], b[2], lo);
    FGC_M16(a[2], b[0], lo);
    FGC_M16(a[1], b[1], lo);
    FGC_M16(a[0], b[1], lo);
    FGC_M16(a[1], b[0], lo);
    FGC_M16(a[0], b[0], hi);
#undef FGC_M16
}


}  // namespace fgc
