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

// ---- operands of the per-facet MLP (fgc_mlp.hip, fgc_mlp_bf16.hip) and the rotation of the input rows: device bodies here so
// ---- that the step's one housekeeping launch (fgc_conv_pack with an fgc_pack_extra) can run them beside the conv packs

// W1 [cin, hidden] -> Wp1[k/4][hidden][k%4], k padded to kpad with zeros
__device__ __forceinline__ void mlp_pack_body(const float* __restrict__ W1, float* __restrict__ Wp, int cin, int kpad, int hidden,
                                              int bid, int nb) {
    const size_t total = (size_t)kpad * hidden;
    for (size_t idx = (size_t)bid * blockDim.x + threadIdx.x; idx < total; idx += (size_t)nb * blockDim.x) {
        const int e = idx & 3;
        const size_t v4 = idx >> 2;
        const int col = v4 % hidden;
        const int k = (int)(v4 / hidden) * 4 + e;
        Wp[idx] = k < cin ? W1[(size_t)k * hidden + col] : 0.f;
    }
}

// W1 [cin, hidden] fp32 -> B fragments [k-step][column tile][lane][8] bf16 (k = input channel, 32 per step)
__device__ __forceinline__ void mlp_pack_bf16_body(const float* __restrict__ W1, unsigned short* __restrict__ Wp, int cin, int hidden,
                                                   int bid, int nb) {
    const int nct = hidden >> 4;
    const size_t total = (size_t)(cin >> 5) * nct * 512;
    for (size_t idx = (size_t)bid * blockDim.x + threadIdx.x; idx < total; idx += (size_t)nb * blockDim.x) {
        const int j = idx & 7, lane = (idx >> 3) & 63;
        const size_t rest = idx >> 9;
        const int ct = rest % nct, ks = (int)(rest / nct);
        const int c = ks * 32 + 8 * (lane >> 4) + j;
        Wp[idx] = f_to_bf(W1[(size_t)c * hidden + ct * 16 + (lane & 15)]);
    }
}

// the same fragments as three planes hi / mid / lo of a three-term split (v = bf16(v) + bf16(v - v0) + bf16(v - v0 - v1))
__device__ __forceinline__ void mlp_pack_split_body(const float* __restrict__ W1, unsigned short* __restrict__ Wp, int cin, int hidden,
                                                    int bid, int nb) {
    const int nct = hidden >> 4;
    const size_t total = (size_t)(cin >> 5) * nct * 512;
    for (size_t idx = (size_t)bid * blockDim.x + threadIdx.x; idx < total; idx += (size_t)nb * blockDim.x) {
        const int j = idx & 7, lane = (idx >> 3) & 63;
        const size_t rest = idx >> 9;
        const int ct = rest % nct, ks = (int)(rest / nct);
        const int c = ks * 32 + 8 * (lane >> 4) + j;
        const float w = W1[(size_t)c * hidden + ct * 16 + (lane & 15)];
        const unsigned short w0 = f_to_bf(w);
        const float r1 = w - bf_to_f(w0);
        const unsigned short w1 = f_to_bf(r1);
        const unsigned short w2 = f_to_bf(r1 - bf_to_f(w1));
        Wp[idx] = w0;
        Wp[total + idx] = w1;
        Wp[2 * total + idx] = w2;
    }
}

// The 1024 -> 3 layer of the bf16 backward pass on the matrix pipe.  g[row][col] = sum_o dy[row][o] W2[col][o] has K = 3: on the
// vector ALU it cost three FMAs per hidden element, a third of the kernel's vector work.  One bf16 MFMA has 32 k slots:
// slots 0-2 carry dy_hi x W2_hi, 3-5 dy_lo x W2_hi, 8-10 dy_hi x W2_lo (v_hi = bf16(v), v_lo = bf16(v - v_hi): 16
// significand bits per operand; g is rounded to bf16 right afterwards for the products that consume it).
//   A (per 16 rows):  lane (lr = row, lq):  lq 0: {hi0 hi1 hi2 lo0 lo1 lo2 0 0}   lq 1: {hi0 hi1 hi2 0 ...}   else 0
//   B (per 16 hidden columns, packed once per launch): lq 0: {Whi0-2 Whi0-2 0 0}   lq 1: {Wlo0-2 0 ...}   else 0
__device__ __forceinline__ void mlp_pack_w2_bf16_body(const float* __restrict__ W2, u32x4* __restrict__ W2p, int hidden, int cout,
                                                      int bid) {
    const int idx = bid * blockDim.x + threadIdx.x;
    if (idx >= (hidden >> 4) * 32) return;       // lanes 0..31 of a fragment (lq 0 and 1); lanes 32..63 are zero, not stored
    const int lane = idx & 31, ct = idx >> 5, lr = lane & 15, lq = lane >> 4;
    unsigned short hi[3], lo[3];
#pragma unroll
    for (int o = 0; o < 3; ++o) {
        const float w = o < cout ? W2[(size_t)(ct * 16 + lr) * cout + o] : 0.f;
        hi[o] = f_to_bf(w);
        lo[o] = f_to_bf(w - bf_to_f(hi[o]));
    }
    W2p[idx] = lq == 0 ? u32x4{hi[0] | ((unsigned)hi[1] << 16), hi[2] | ((unsigned)hi[0] << 16), hi[1] | ((unsigned)hi[2] << 16), 0u}
                       : u32x4{lo[0] | ((unsigned)lo[1] << 16), (unsigned)lo[2], 0u, 0u};
}
// W1 [cin, hidden] fp32 -> B fragments of dx += dh W1^T in the k order of the dx kernel's transposed dh (pair pp of column
// tiles, input-channel tile m, lane (lr = channel, lq)): element j = c2*4 + t is W1[m*16 + lr][pp*32 + c2*16 + 4*lq + t]
__device__ __forceinline__ void mlp_pack_w1dx_bf16_body(const float* __restrict__ W1, unsigned short* __restrict__ Wd, int cin,
                                                        int hidden, int bid, int nb) {
    const size_t total = (size_t)cin * hidden;
    const int mt = cin >> 4;
    for (size_t idx = (size_t)bid * blockDim.x + threadIdx.x; idx < total; idx += (size_t)nb * blockDim.x) {
        const int j = idx & 7, lane = (idx >> 3) & 63;
        const size_t rest = idx >> 9;
        const int m = rest % mt, pp = (int)(rest / mt);
        const int ch = m * 16 + (lane & 15), col = pp * 32 + (j >> 2) * 16 + 4 * (lane >> 4) + (j & 3);
        Wd[idx] = f_to_bf(W1[(size_t)ch * hidden + col]);
    }
}

// ---- the fp32 network's MLP BACKWARD on split operands (fgc_mlp_bf16.hip: mlp_bwd_dx_split_kernel / mlp_bwd_w_split_kernel) ----
// three-term split of one fp32 value, v = p[0] + p[1] + p[2] (each a bf16, round to nearest even)
__device__ __forceinline__ void split3_scalar(float v, unsigned short (&p)[3]) {
    p[0] = f_to_bf(v);
    const float r1 = v - bf_to_f(p[0]);
    p[1] = f_to_bf(r1);
    p[2] = f_to_bf(r1 - bf_to_f(p[1]));
}
// d-logits operand for the dz GEMM on split operands (conv_bwd_logits_deep_kernel<.., SPLIT>): the B fragments of
// pack_logit_weight_bf16_body in three planes, [pass][k-step][column tile][plane][lane][8]:
//   lane l: o = k-step*32 + 8*(l>>4) + j (j = 0..7), column kk = column tile*16 + (l&15) = m*32 + cl, channel pass*32 + cl
__device__ __forceinline__ void pack_logit_weight_split_body(const float* __restrict__ W0, unsigned short* __restrict__ Wq, int cin,
                                                             int cout, int passes, int bid, int nb) {
    const int kso = (cout + 31) >> 5;
    const size_t total = (size_t)passes * kso * 18 * 3 * 512;
    for (size_t idx = (size_t)bid * blockDim.x + threadIdx.x; idx < total; idx += (size_t)nb * blockDim.x) {
        const int j = idx & 7, lane = (idx >> 3) & 63;
        const size_t rest = idx >> 9;
        const int plane = rest % 3, ct = (rest / 3) % 18, ks = (rest / 54) % kso, pass = (int)(rest / ((size_t)54 * kso));
        const int o = ks * 32 + 8 * (lane >> 4) + j;
        const int kk = ct * 16 + (lane & 15);
        const int m = kk >> 5, cl = kk & 31;
        const int c = pass * 32 + cl;
        unsigned short p[3];
        split3_scalar((o < cout && c < cin) ? W0[((size_t)m * cout + o) * cin + c] : 0.f, p);
        Wq[idx] = p[plane];
    }
}
// The K = 3 product g[row][col] = sum_o dy[row][o] W2[col][o] with fp32-equivalent accuracy in ONE bf16 MFMA: both operands
// split into three terms, the six products dy_p W2_q that matter (p + q <= 2) times three outputs fill 18 of the 32 k slots.
//   slot s = pair * 3 + o,  pair -> (dy plane, W2 plane) = (0,0) (0,1) (1,0) (1,1) (0,2) (2,0);  lane (lr, lq) holds slots
//   8 lq + j, j = 0..7, of row / column lr (slots >= 18: zero).  `side` 0: the dy operand, 1: the W2 operand.
__device__ __forceinline__ u32x4 split_k3_frag(const float (&v)[3], int lq, int side) {
    unsigned short p[3][3];   // [o][plane]
#pragma unroll
    for (int o = 0; o < 3; ++o) split3_scalar(v[o], p[o]);
    constexpr int PD[6] = {0, 0, 1, 1, 0, 2}, PW[6] = {0, 1, 0, 1, 2, 0};
    unsigned w[9];            // slots 2 i, 2 i + 1
#pragma unroll
    for (int i = 0; i < 9; ++i) {
        const int s0 = 2 * i, s1 = 2 * i + 1;
        w[i] = (unsigned)p[s0 % 3][side ? PW[s0 / 3] : PD[s0 / 3]] | ((unsigned)p[s1 % 3][side ? PW[s1 / 3] : PD[s1 / 3]] << 16);
    }
    // lane group lq takes dwords 4 lq .. 4 lq + 3 (bit masks, not branches: the compiler turns a chain of selects on lq into
    // divergent control flow)
    const unsigned m0 = lq == 0 ? ~0u : 0u, m1 = lq == 1 ? ~0u : 0u, m2 = lq == 2 ? ~0u : 0u;
    return u32x4{(w[0] & m0) | (w[4] & m1) | (w[8] & m2), (w[1] & m0) | (w[5] & m1), (w[2] & m0) | (w[6] & m1),
                 (w[3] & m0) | (w[7] & m1)};
}
// W2 [hidden, cout] -> the W2 operand of that product, one 16-byte fragment per (column tile, lane): [hidden / 16][64]
__device__ __forceinline__ void mlp_pack_w2_split_body(const float* __restrict__ W2, u32x4* __restrict__ W2s, int hidden, int cout,
                                                       int bid) {
    const int idx = bid * blockDim.x + threadIdx.x;
    if (idx >= (hidden >> 4) * 64) return;
    const int lane = idx & 63, ct = idx >> 6, lr = lane & 15, lq = lane >> 4;
    float w[3];
#pragma unroll
    for (int o = 0; o < 3; ++o) w[o] = o < cout ? W2[(size_t)(ct * 16 + lr) * cout + o] : 0.f;
    W2s[idx] = split_k3_frag(w, lq, 1);
}
// W1 in the k order of the dx kernel's transposed dh (mlp_pack_w1dx_bf16_body), as three planes of a three-term split
__device__ __forceinline__ void mlp_pack_w1dx_split_body(const float* __restrict__ W1, unsigned short* __restrict__ Wd, int cin,
                                                         int hidden, int bid, int nb) {
    const size_t total = (size_t)cin * hidden;
    const int mt = cin >> 4;
    for (size_t idx = (size_t)bid * blockDim.x + threadIdx.x; idx < total; idx += (size_t)nb * blockDim.x) {
        const int j = idx & 7, lane = (idx >> 3) & 63;
        const size_t rest = idx >> 9;
        const int m = rest % mt, pp = (int)(rest / mt);
        const int ch = m * 16 + (lane & 15), col = pp * 32 + (j >> 2) * 16 + 4 * (lane >> 4) + (j & 3);
        unsigned short p[3];
        split3_scalar(W1[(size_t)ch * hidden + col], p);
        Wd[idx] = p[0];
        Wd[total + idx] = p[1];
        Wd[2 * total + idx] = p[2];
    }
}

// one 3-vector times R^T, with the association spelled out (wherever a row is rotated it gets the same bits)
__device__ __forceinline__ void rot3(const float (&r)[9], float a, float b, float c, float& o0, float& o1, float& o2) {
    o0 = fmaf(r[2], c, fmaf(r[1], b, r[0] * a));
    o1 = fmaf(r[5], c, fmaf(r[4], b, r[3] * a));
    o2 = fmaf(r[8], c, fmaf(r[7], b, r[6] * a));
}

// assignment logits of ONE row of a narrow input (cin <= 8, the first layer): ag[0..8] = u x + c, ag[12..20] = v x, pads 0.
// One fixed order of fused multiply-adds wherever this table is computed (a launch of its own, or a job of the step's
// housekeeping launch): the same bits.
__device__ __forceinline__ void narrow_logits_row(const float* xr, int cin, const float* __restrict__ u, const float* __restrict__ c,
                                                  const float* __restrict__ v, float* __restrict__ agr) {
    // the 24 floats of the row in registers (fully unrolled: k < cin <= 8 by predicate, the same FMA chain per logit as
    // ever), then six 16-byte stores: a row per thread with 24 scalar stores made every store instruction of a wave touch
    // 64 different rows, four bytes each
    float o[FGC_AG_LD];
#pragma unroll
    for (int m = 0; m < FGC_M; ++m) {
        float a = c[m], g = 0.f;
#pragma unroll
        for (int k = 0; k < 8; ++k)
            if (k < cin) {
                a = fmaf(u[m * cin + k], xr[k], a);
                g = fmaf(v[m * cin + k], xr[k], g);
            }
        o[m] = a;
        o[12 + m] = g;
    }
#pragma unroll
    for (int m = FGC_M; m < 12; ++m) {
        o[m] = 0.f;
        o[12 + m] = 0.f;
    }
    if ((reinterpret_cast<uintptr_t>(agr) & 15) == 0) {
#pragma unroll
        for (int q = 0; q < FGC_AG_LD / 4; ++q)
            reinterpret_cast<f32x4*>(agr)[q] = f32x4{o[4 * q], o[4 * q + 1], o[4 * q + 2], o[4 * q + 3]};
    } else {
#pragma unroll
        for (int q = 0; q < FGC_AG_LD; ++q) agr[q] = o[q];
    }
}
__device__ __forceinline__ void narrow_logits_body(const float* __restrict__ x, int rows, int cin, const float* __restrict__ u,
                                                   const float* __restrict__ c, const float* __restrict__ v, float* __restrict__ ag,
                                                   int bid, int nb) {
    for (int r = bid * blockDim.x + threadIdx.x; r < rows; r += nb * blockDim.x) {
        float xr[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) xr[k] = k < cin ? x[(size_t)r * cin + k] : 0.f;
        narrow_logits_row(xr, cin, u, c, v, ag + (size_t)r * FGC_AG_LD);
    }
}
// rotation of the rows of a 3- or 6-channel input AND the first layer's logit table of the rotated rows, one row per thread
__device__ __forceinline__ void rotate_logits_body(const float* __restrict__ x, float* __restrict__ y, int rows, int vecs,
                                                   const float* __restrict__ Rd, const float* __restrict__ u,
                                                   const float* __restrict__ c, const float* __restrict__ v, float* __restrict__ ag,
                                                   int bid, int nb) {
    float r9[9];
#pragma unroll
    for (int i = 0; i < 9; ++i) r9[i] = Rd[i];
    const int cin = 3 * vecs;
    for (int r = bid * blockDim.x + threadIdx.x; r < rows; r += nb * blockDim.x) {
        float xr[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int q = 0; q < 2; ++q) {          // (vecs <= 2: checked by the host; static indices keep xr in registers)
            if (q < vecs) {
                const size_t i = (size_t)r * vecs + q;
                const float a = x[3 * i], b = x[3 * i + 1], cc = x[3 * i + 2];
                rot3(r9, a, b, cc, xr[3 * q], xr[3 * q + 1], xr[3 * q + 2]);
                y[3 * i] = xr[3 * q];
                y[3 * i + 1] = xr[3 * q + 1];
                y[3 * i + 2] = xr[3 * q + 2];
            }
        }
        narrow_logits_row(xr, cin, u, c, v, ag + (size_t)r * FGC_AG_LD);
    }
}

// rotation augmentation (train.py:439-451): every 3-vector of x times R^T
__device__ __forceinline__ void rotate_rows_body(const float* __restrict__ x, float* __restrict__ y, int64_t nvec,
                                                 const float* __restrict__ Rd, int bid, int nb) {
    float r[9];
#pragma unroll
    for (int i = 0; i < 9; ++i) r[i] = Rd[i];
    for (int64_t i = (int64_t)bid * blockDim.x + threadIdx.x; i < nvec; i += (int64_t)nb * blockDim.x) {
        const float a = x[3 * i], b = x[3 * i + 1], c = x[3 * i + 2];
        rot3(r, a, b, c, y[3 * i], y[3 * i + 1], y[3 * i + 2]);
    }
}

// every packed operand of several layers in one launch (fgc_conv_pack)
struct PackJob {
    const float* W0;
    float* dst;
    int kind;   // 0: forward operand, 1: data-gradient operand (transposed), 2: d-logits operand; +4: the bf16 forms;
                // 7: plain bf16 copy of W0 (pair form; kdim = element count);
                // 8: rotate_rows_body (W0 = x, dst = y, aux = R, kdim = 3-vectors);
                // 14: rotate_logits_body (W0 = x, dst = y, aux = R, kdim = rows, cin = 3-vectors per row, lg_* = the table);
                // MLP operands, W0 = W1 [cin, ncols]: 9 mlp_pack_body (kdim = kpad), 10 mlp_pack_split_body,
                // 11 mlp_pack_bf16_body, 12 mlp_pack_w1dx_bf16_body; 13 mlp_pack_w2_bf16_body (W0 = W2 [ncols, cout]);
                // 15 mlp_pack_w1dx_split_body, 16 mlp_pack_w2_split_body (W0 = W2 [ncols, cout])
                // 17 pack_logit_weight_split_body
    int cin, cout, kdim, ncols, npad, kc, kpass, passes, opad;
    int block0;
    const float* aux;
    const float *lg_u, *lg_c, *lg_v;     // kind 14: the first layer's assignment parameters and its logit table
    float* lg_ag;
};
constexpr int PACK_MAX_JOBS = 28;
struct PackJobs {
    PackJob job[PACK_MAX_JOBS];
    int njobs, nblocks;
};
__global__ void pack_many_kernel(PackJobs J);

// host side: the jobs that leave an MLP's operands in the workspaces of fgc_mlp_fwd / fgc_mlp_bwd (fgc_mlp.hip) or of their
// bf16 forms (fgc_mlp_bf16.hip) exactly as those entry points lay them out themselves; `totals[i]` = elements of job i (a
// workgroup takes 1024).  Return the number of jobs written (<= 4), or -1 for a shape the entry points refuse.
int mlp_pack_jobs_f32(const fgc_pack_extra* e, PackJob* jobs, size_t* totals);
int mlp_pack_jobs_bf16(const fgc_pack_extra* e, PackJob* jobs, size_t* totals);

}  // namespace fgc
