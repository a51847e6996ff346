// Per-facet MLP cin -> hidden -> cout (replaces lrelu(custom_lin(x,1024)) -> custom_lin(.,3),
// /root/reference/Code/model.py:763-769,937-941).  The [n, hidden] activation never leaves the CU:
// each wave produces 16-column slabs of it with f32 MFMA, applies bias + leaky ReLU in the
// accumulator layout and folds them straight into the tiny second layer on the VALU.
#include <algorithm>

#include "fgc_reduce.h"
#include "fgc_mlp_split.h"
#include "fgc_pack.h"

namespace fgc {

#ifndef FGC_MLP_T
#define FGC_MLP_T 64
#endif
constexpr int MLP_T = FGC_MLP_T;   // rows per tile (forward kernel)
constexpr int MLP_RT = MLP_T / 16;
constexpr int MLP_COUT_MAX = 4;
constexpr int MLP_THREADS = 256;
// LDS row strides (floats).  ds_read_b32 / ds_write_b32 bank = dword % 32 within a 32-lane half, ds_read_b128 bank =
// dword % 64 within a 16-lane group: a stride == 4 mod 8 keeps both the row-per-lane b128 reads (16 rows -> 16
// distinct 16-byte slots) and the 4-rows-apart b32 accesses of lanes l, l+16 (4 * stride == 16 mod 32) conflict free.
constexpr int MLP_XPAD = 4;    // x tile rows: kpad + 4
constexpr int MLP_DHS = 20;    // transposed dh rows


// W1 [cin, hidden] -> Wp1[k/4][hidden][k%4], k padded to a multiple of 16 with zeros
__global__ void mlp_pack_kernel(const float* __restrict__ W1, float* __restrict__ Wp, int cin, int kpad, int hidden) {
    mlp_pack_body(W1, Wp, cin, kpad, hidden, blockIdx.x, gridDim.x);
}

template <int T = MLP_T>
__device__ __forceinline__ void load_x_tile(const float* __restrict__ x, int n, int cin, int kpad, int xs, int row0,
                                            float* xt) {
    // xt [T][xs]; zero padded rows / channels
    for (int t = threadIdx.x; t < T * kpad; t += MLP_THREADS) {
        const int r = t / kpad, c = t % kpad;
        const int row = row0 + r;
        xt[r * xs + c] = (row < n && c < cin) ? x[(size_t)row * cin + c] : 0.f;
    }
}

// hidden pre-activation slab for column tile ct: h[rt] (C layout: col = lane&15, row = rt*16 + (lane>>4)*4 + reg)
// KG > 0: the number of 16-wide k groups is known at compile time (the loop unrolls and all fragment loads are issued
// before the first MFMA); KG == 0: runtime kg
// init: the bias of the slab's column (lane lr), which rides in the accumulator: one vector add per element less next to the
// fp32 MFMAs, which share the SIMD with the vector ALU
template <int RT, int KG = 0>
__device__ __forceinline__ void hidden_slab(const float* xt, int xs, int kg, const f32x4* __restrict__ Wp4, int hidden,
                                            int ct, f32x4 (&h)[RT], float init = 0.f) {
    const int lane = threadIdx.x & 63, lr = lane & 15, lq = lane >> 4;
#pragma unroll
    for (int r = 0; r < RT; ++r) h[r] = f32x4{init, init, init, init};
    if constexpr (KG > 0) kg = KG;
#pragma unroll KG > 0 ? KG : 1
    for (int g = 0; g < kg; ++g) {
        const f32x4 b = Wp4[(size_t)(g * 4 + lq) * hidden + ct * 16 + lr];
        f32x4 a[RT];
#pragma unroll
        for (int r = 0; r < RT; ++r) a[r] = *reinterpret_cast<const f32x4*>(xt + (r * 16 + lr) * xs + g * 16 + lq * 4);
        // k outer, row tile inner: consecutive MFMAs write different accumulators (a dependent 16x16x4 f32 MFMA
        // needs 40 cycles, an independent one issues every 32)
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int r = 0; r < RT; ++r) h[r] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[r][t], b[t], h[r], 0, 0, 0);
    }
}

// KG = kpad / 16 when the next column tile's weight fragments are prefetched (narrow inputs), 0 = generic
// CO = number of output columns carried in registers (cout <= CO <= CO)
template <int KG, int CO>
__global__ __launch_bounds__(MLP_THREADS, CO <= 3 ? 4 : 3) void mlp_fwd_kernel(const float* __restrict__ x, int n, int cin, int kpad,
                                                              int hidden, int cout, const float* __restrict__ Wp,
                                                              const float* __restrict__ b1,
                                                              const float* __restrict__ W2,
                                                              const float* __restrict__ b2, float alpha,
                                                              float* __restrict__ y, float* __restrict__ abs_partial) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    const int xs = kpad + MLP_XPAD;
    float* xt = reinterpret_cast<float*>(smem_raw);
    float* ypart = xt + MLP_T * xs;  // [4 waves][MLP_T][4]
    float* red = ypart + 4 * MLP_T * 4;  // [4]
    const int row0 = blockIdx.x * MLP_T;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, lr = lane & 15, lq = lane >> 4;
    load_x_tile(x, n, cin, kpad, xs, row0, xt);
    __syncthreads();

    const int kg = kpad >> 4;
    const int nct = hidden >> 4;
    const f32x4* Wp4 = reinterpret_cast<const f32x4*>(Wp);
    float yp[MLP_RT][4][CO];
#pragma unroll
    for (int r = 0; r < MLP_RT; ++r)
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int o = 0; o < CO; ++o) yp[r][t][o] = 0.f;

    constexpr int NB = KG > 0 ? KG : 1;
    f32x4 bnext[NB];
    float bbn = 0.f, w2n[CO];
    auto fetch_weights = [&](int ct) {   // W1 fragments, b1 and W2 rows of column tile ct (clamped: always a valid load)
        ct = min(ct, nct - 1);
#pragma unroll
        for (int g = 0; g < NB; ++g) bnext[g] = Wp4[(size_t)(g * 4 + lq) * hidden + ct * 16 + lr];
        bbn = b1[ct * 16 + lr];
#pragma unroll
        for (int o = 0; o < CO; ++o) w2n[o] = W2[(size_t)(ct * 16 + lr) * cout + min(o, cout - 1)];
    };
    if constexpr (KG > 0) fetch_weights(wave);
    for (int ct = wave; ct < nct; ct += 4) {
        f32x4 h[MLP_RT];
        float bb, w2[CO];
        if constexpr (KG > 0) {
            // this tile's weights arrived while the previous one was computed; ask for the next tile's before the MFMAs
            f32x4 bcur[NB];
#pragma unroll
            for (int g = 0; g < NB; ++g) bcur[g] = bnext[g];
            bb = bbn;
#pragma unroll
            for (int o = 0; o < CO; ++o) w2[o] = o < cout ? w2n[o] : 0.f;
            fetch_weights(ct + 4);
#pragma unroll
            for (int r = 0; r < MLP_RT; ++r) h[r] = f32x4{bb, bb, bb, bb};   // the bias rides in the accumulator
#pragma unroll
            for (int g = 0; g < NB; ++g) {
                f32x4 a[MLP_RT];
#pragma unroll
                for (int r = 0; r < MLP_RT; ++r)
                    a[r] = *reinterpret_cast<const f32x4*>(xt + (r * 16 + lr) * xs + g * 16 + lq * 4);
#pragma unroll
                for (int t = 0; t < 4; ++t)
#pragma unroll
                    for (int r = 0; r < MLP_RT; ++r)
                        h[r] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[r][t], bcur[g][t], h[r], 0, 0, 0);
            }
        } else {
            const int col = ct * 16 + lr;
            bb = b1[col];
            hidden_slab(xt, xs, kg, Wp4, hidden, ct, h, bb);
#pragma unroll
            for (int o = 0; o < CO; ++o) w2[o] = o < cout ? W2[(size_t)col * cout + o] : 0.f;
        }
#pragma unroll
        for (int r = 0; r < MLP_RT; ++r)
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                float v = h[r][t];
                v = lrelu01(v, alpha);    // leaky ReLU for 0 <= alpha <= 1 (checked by the host): bit-identical to
                                            // relu(v) - alpha relu(-v), one instruction less
#pragma unroll
                for (int o = 0; o < CO; ++o) yp[r][t][o] = fmaf(v, w2[o], yp[r][t][o]);
            }
    }
    // reduce over the 16 column lanes
#pragma unroll
    for (int r = 0; r < MLP_RT; ++r)
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int o = 0; o < CO; ++o) {
                float v = yp[r][t][o];
                FGC_ROW16_SUM(v);      // DPP adds; lane lr == 0 (the one that is stored) sums in the xor-butterfly order
                yp[r][t][o] = v;
            }
    if (lr == 0) {
#pragma unroll
        for (int r = 0; r < MLP_RT; ++r)
#pragma unroll
            for (int t = 0; t < 4; ++t)
#pragma unroll
                for (int o = 0; o < CO; ++o)
                    ypart[(wave * MLP_T + r * 16 + lq * 4 + t) * 4 + o] = yp[r][t][o];
    }
    __syncthreads();
    float asum = 0.f;
    for (int t = threadIdx.x; t < MLP_T * cout; t += MLP_THREADS) {
        const int r = t / cout, o = t % cout;
        const int row = row0 + r;
        if (row < n) {
            float v = b2[o];
            v += ypart[(0 * MLP_T + r) * 4 + o];
            v += ypart[(1 * MLP_T + r) * 4 + o];
            v += ypart[(2 * MLP_T + r) * 4 + o];
            v += ypart[(3 * MLP_T + r) * 4 + o];
            y[(size_t)row * cout + o] = v;
            asum += fabsf(v);
        }
    }
    if (abs_partial) {
        // deterministic block reduction: wave shuffle then 4 partials in fixed order
        for (int off = 32; off > 0; off >>= 1) asum += __shfl_xor(asum, off);
        if (lane == 0) red[wave] = asum;
        __syncthreads();
        if (threadIdx.x == 0) abs_partial[blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
    }
}

// ---------------------------------------------------------------------------------------------
// backward: persistent workgroups; hidden is recomputed, never stored.
//   dhid = (dy W2^T) * lrelu'(h);  dW2 = hact^T dy;  db1 = sum dhid;  dW1 = x^T dhid;  dx = dhid W1^T
// Each workgroup owns the hidden-column range [hc0, hc0+hcw) (blockIdx.y) and walks row tiles
// blockIdx.x, +gridDim.x, ...; parameter-gradient partials stay in registers over the walk and are
// written once per workgroup as slabs (reduced in fixed order by reduce_jobs).
// ---------------------------------------------------------------------------------------------
// MLP_BWD_MT = cin tiles of 16, MLP_BWD_CTW = column tiles per wave per workgroup (hcw = 4 waves * CTW * 16 columns).
// <2,4> serves the 32-wide head of the network; <4,2> and <8,1> the 64/128-wide multi-scale heads (model.py:894-899,
// 915-920), trading column width for the wider dx / dW1 accumulators.
#ifndef FGC_MLP_BWD_T
#define FGC_MLP_BWD_T 32
#endif
#ifndef FGC_MLP_BWD_KG
#define FGC_MLP_BWD_KG(mt) 0
#endif
#ifndef FGC_MLP_BWD_WAVES
#define FGC_MLP_BWD_WAVES 2
#endif
constexpr int BWD_T = FGC_MLP_BWD_T;   // rows per tile of the backward kernel
constexpr int BWD_RT = BWD_T / 16;

template <int MLP_BWD_MT, int MLP_BWD_CTW, int CO>
__global__ __launch_bounds__(MLP_THREADS, FGC_MLP_BWD_WAVES) void mlp_bwd_kernel(
    const float* __restrict__ x, const float* __restrict__ dy, int n, int cin, int kpad, int hidden, int cout,
    const float* __restrict__ Wp, const float* __restrict__ W1, const float* __restrict__ b1,
    const float* __restrict__ W2, float alpha, float* __restrict__ dx_slab /* [gridDim.y][n][cin] */,
    float* __restrict__ dW1_slab /* [gridDim.x][cin][hidden] */, float* __restrict__ db1_slab /* [gridDim.x][hidden] */,
    float* __restrict__ dW2_slab /* [gridDim.x][hidden][4] */, float* __restrict__ db2_slab /* [gridDim.x][4] */) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    constexpr int KP = MLP_BWD_MT * 16;                             // == kpad
    constexpr int HCW = 4 * MLP_BWD_CTW * 16;                       // hidden columns of this workgroup
    constexpr int WLD = HCW + 1;                                    // f32x4 per k-group row of the weight slice
    constexpr int XPT = BWD_T * KP / MLP_THREADS;                   // x elements per thread per tile
    constexpr int xs = KP + MLP_XPAD;      // kpad == KP: strides are compile-time so LDS offsets fold into the instructions
    float* xt = reinterpret_cast<float*>(smem_raw);                // [BWD_T][xs]
    float* dyt = xt + BWD_T * xs;                                   // [BWD_T][4]
    float* dht = dyt + BWD_T * 4;                                   // [4 waves][BWD_T][MLP_DHS]
    float* dxp = dht + 4 * BWD_T * MLP_DHS;                              // [4 waves][BWD_T][kpad+4]  ([1][..] when MT > 2)
    // [KP/4][WLD][4]; every region before it is a multiple of four floats, so it is 16-byte aligned as it stands (an
    // integer round trip to align the pointer hides from the compiler that this is LDS: every access to the weight slice
    // became a FLAT load that waits on both the memory and the LDS counter)
    static_assert((BWD_T * xs) % 4 == 0 && (4 * BWD_T * MLP_DHS) % 4 == 0 && (BWD_T * (KP + MLP_XPAD)) % 4 == 0, "LDS carve alignment");
    float* Ws = dxp + (MLP_BWD_MT <= 2 ? 4 : 1) * BWD_T * (KP + MLP_XPAD);
    float* b1s = Ws + (KP / 4) * WLD * 4;                           // [HCW]
    float* W2s = b1s + HCW;                                         // [HCW][4]
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, lr = lane & 15, lq = lane >> 4;
    constexpr int kg = MLP_BWD_MT;
    const f32x4* Wp4 = reinterpret_cast<const f32x4*>(Wp);
    const int hc0 = blockIdx.y * HCW;
    const int ntiles = (n + BWD_T - 1) / BWD_T;

    // this workgroup's weight slice stays in LDS for the whole walk: W1 columns [hc0, hc0+HCW) in the packed
    // [k/4][col][k%4] layout (the +1 row pad spreads the transposed reads of the dx product over the banks), b1, W2
    f32x4* Ws4 = reinterpret_cast<f32x4*>(Ws);
    for (int t = threadIdx.x; t < (KP / 4) * HCW; t += MLP_THREADS) {
        const int k4 = t / HCW, jl = t % HCW;
        Ws4[k4 * WLD + jl] = Wp4[(size_t)k4 * hidden + hc0 + jl];
    }
    for (int t = threadIdx.x; t < HCW; t += MLP_THREADS) {
        b1s[t] = b1[hc0 + t];
#pragma unroll
        for (int o = 0; o < 4; ++o) W2s[t * 4 + o] = o < cout ? W2[(size_t)(hc0 + t) * cout + o] : 0.f;
    }

    f32x4 dW1acc[MLP_BWD_CTW][MLP_BWD_MT];
    float dW2acc[MLP_BWD_CTW][CO];
    float db1acc[MLP_BWD_CTW];
#pragma unroll
    for (int c = 0; c < MLP_BWD_CTW; ++c) {
        db1acc[c] = 0.f;
#pragma unroll
        for (int o = 0; o < CO; ++o) dW2acc[c][o] = 0.f;
#pragma unroll
        for (int m = 0; m < MLP_BWD_MT; ++m) dW1acc[c][m] = f32x4{0.f, 0.f, 0.f, 0.f};
    }

    // the next tile's rows travel through registers while the current tile is being worked on: addresses are clamped
    // so the loads are unconditional (no exec-masked load, no early wait); out-of-range elements are zeroed on store
    float xpre[XPT], dypre, db2acc = 0.f;
    auto fetch_tile = [&](int tl) {
        const int r0 = tl * BWD_T;
#pragma unroll
        for (int i = 0; i < XPT; ++i) {
            const int t = threadIdx.x + i * MLP_THREADS;
            const int r = t / KP, c = t % KP;
            xpre[i] = x[(size_t)min(r0 + r, n - 1) * cin + min(c, cin - 1)];
        }
        const int t = threadIdx.x & (BWD_T * 4 - 1);
        dypre = dy[(size_t)min(r0 + (t >> 2), n - 1) * cout + min(t & 3, cout - 1)];
    };
    auto store_tile = [&](int tl) {
        const int r0 = tl * BWD_T;
#pragma unroll
        for (int i = 0; i < XPT; ++i) {
            const int t = threadIdx.x + i * MLP_THREADS;
            const int r = t / KP, c = t % KP;
            xt[r * xs + c] = (r0 + r < n && c < cin) ? xpre[i] : 0.f;
        }
        if (threadIdx.x < BWD_T * 4) {
            const int t = threadIdx.x;
            const float v = (r0 + (t >> 2) < n && (t & 3) < cout) ? dypre : 0.f;
            dyt[t] = v;
            db2acc += v;     // db2 = column sums of dy: every workgroup row-walk sees each dy row once per hidden slice
        }
    };
    constexpr bool PREFETCH = XPT <= 8;     // the 128-wide head has no registers to spare for it
    if (PREFETCH && (int)blockIdx.x < ntiles) fetch_tile(blockIdx.x);

    for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const int row0 = tile * BWD_T;
        __syncthreads();
        if (!PREFETCH) fetch_tile(tile);
        store_tile(tile);
        if (PREFETCH && tile + (int)gridDim.x < ntiles) fetch_tile(tile + gridDim.x);
        __syncthreads();
        f32x4 dxacc[BWD_RT][MLP_BWD_MT];
#pragma unroll
        for (int r = 0; r < BWD_RT; ++r)
#pragma unroll
            for (int m = 0; m < MLP_BWD_MT; ++m) dxacc[r][m] = f32x4{0.f, 0.f, 0.f, 0.f};

#pragma unroll
        for (int c = 0; c < MLP_BWD_CTW; ++c) {
            // one column tile at a time: without this fence the scheduler interleaves the unrolled iterations to
            // overlap their loads and the live fragments of all of them push the kernel past 256 registers
            // (384 = one wave per SIMD, every memory wait exposed)
            __builtin_amdgcn_sched_barrier(0);
            const int ctl = wave * MLP_BWD_CTW + c;          // column tile within the workgroup's slice
            f32x4 h[BWD_RT];
            const float bb = b1s[ctl * 16 + lr];
            hidden_slab<BWD_RT, FGC_MLP_BWD_KG(MLP_BWD_MT)>(xt, xs, kg, Ws4, WLD, ctl, h, bb);
            const f32x4 w2 = *reinterpret_cast<const f32x4*>(W2s + (ctl * 16 + lr) * 4);
            f32x4 dh[BWD_RT];
#pragma unroll
            for (int r = 0; r < BWD_RT; ++r)
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    const int rr = r * 16 + lq * 4 + t;
                    const f32x4 dyr = *reinterpret_cast<const f32x4*>(dyt + rr * 4);
                    const float pre = h[r][t];
                    const float hact = lrelu01(pre, alpha);   // (0 <= alpha <= 1, checked by the host)
                    // d lrelu: relu'(pre) + alpha*relu'(-pre), both 0 at pre == 0 (TF relu gradient)
                    const float slope = lrelu01_slope(pre, alpha);
                    float g = 0.f;
#pragma unroll
                    for (int o = 0; o < CO; ++o) {
                        g = fmaf(dyr[o], w2[o], g);
                        dW2acc[c][o] = fmaf(hact, dyr[o], dW2acc[c][o]);
                    }
                    g *= slope;
                    dh[r][t] = g;
                    db1acc[c] += g;
                }
            // dW1[cin tile m][col] += sum_rows x[row][cm] * dh[row][col]  (K = rows; dh regs are the B fragment)
#pragma unroll
            for (int r = 0; r < BWD_RT; ++r)
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    const int rr = r * 16 + lq * 4 + t;
#pragma unroll
                    for (int m = 0; m < MLP_BWD_MT; ++m) {
                        const float a = xt[rr * xs + m * 16 + lr];
                        dW1acc[c][m] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, dh[r][t], dW1acc[c][m], 0, 0, 0);
                    }
                }
            // dx += dh[T x 16] * W1^T[16 x cin]: transpose dh through LDS into the A-fragment layout
            float* dhw = dht + wave * BWD_T * MLP_DHS;
#pragma unroll
            for (int r = 0; r < BWD_RT; ++r)
#pragma unroll
                for (int t = 0; t < 4; ++t) dhw[(r * 16 + lq * 4 + t) * MLP_DHS + lr] = dh[r][t];
            // same wave wrote and reads: LDS ops of one wave are ordered; the compiler inserts the lgkmcnt wait
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
#pragma unroll
            for (int m = 0; m < MLP_BWD_MT; ++m) {
                // b[t] = W1[m*16+lr][slice col ctl*16 + lq*4 + t] out of the packed slice (zero beyond cin)
                const int cc = m * 16 + lr;
                const float* wrow = Ws + ((cc >> 2) * WLD + ctl * 16 + lq * 4) * 4 + (cc & 3);
                f32x4 b;
#pragma unroll
                for (int t = 0; t < 4; ++t) b[t] = wrow[t * 4];
                f32x4 a[BWD_RT];
#pragma unroll
                for (int r = 0; r < BWD_RT; ++r) a[r] = *reinterpret_cast<const f32x4*>(dhw + (r * 16 + lr) * MLP_DHS + lq * 4);
#pragma unroll
                for (int t = 0; t < 4; ++t)
#pragma unroll
                    for (int r = 0; r < BWD_RT; ++r)
                        dxacc[r][m] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[r][t], b[t], dxacc[r][m], 0, 0, 0);
            }
        }
        // reduce dx over the 4 waves (fixed order) and write this hidden-range's slab
        constexpr int dxs = KP + MLP_XPAD;
        if constexpr (MLP_BWD_MT <= 2) {
            float* dxw = dxp + wave * BWD_T * dxs;
#pragma unroll
            for (int r = 0; r < BWD_RT; ++r)
#pragma unroll
                for (int m = 0; m < MLP_BWD_MT; ++m)
#pragma unroll
                    for (int t = 0; t < 4; ++t) dxw[(r * 16 + lq * 4 + t) * dxs + m * 16 + lr] = dxacc[r][m][t];
            __syncthreads();
            if (cin == KP) {
                // full-width rows: four floats per thread, one 16-byte store, no division by a runtime cin
#pragma unroll
                for (int i = 0; i < BWD_T * KP / 4 / MLP_THREADS; ++i) {
                    const int t = threadIdx.x + i * MLP_THREADS;
                    const int r = t / (KP / 4), c = (t % (KP / 4)) * 4;
                    const f32x4 p0 = *reinterpret_cast<const f32x4*>(dxp + (0 * BWD_T + r) * dxs + c);
                    const f32x4 p1 = *reinterpret_cast<const f32x4*>(dxp + (1 * BWD_T + r) * dxs + c);
                    const f32x4 p2 = *reinterpret_cast<const f32x4*>(dxp + (2 * BWD_T + r) * dxs + c);
                    const f32x4 p3 = *reinterpret_cast<const f32x4*>(dxp + (3 * BWD_T + r) * dxs + c);
                    const f32x4 v = (p0 + p1) + (p2 + p3);
                    if (row0 + r < n)
                        *reinterpret_cast<f32x4*>(dx_slab + ((size_t)blockIdx.y * n + row0 + r) * KP + c) = v;
                }
            } else {
                for (int t = threadIdx.x; t < BWD_T * cin; t += MLP_THREADS) {
                    const int r = t / cin, c = t % cin;
                    if (row0 + r < n) {
                        const float v = (dxp[(0 * BWD_T + r) * dxs + c] + dxp[(1 * BWD_T + r) * dxs + c]) +
                                        (dxp[(2 * BWD_T + r) * dxs + c] + dxp[(3 * BWD_T + r) * dxs + c]);
                        dx_slab[((size_t)blockIdx.y * n + row0 + r) * cin + c] = v;
                    }
                }
            }
        } else {
            // wide inputs: one shared tile, the waves add into it one after the other (wave 0, 1, 2, 3)
            for (int w = 0; w < 4; ++w) {
                if (wave == w) {
#pragma unroll
                    for (int r = 0; r < BWD_RT; ++r)
#pragma unroll
                        for (int m = 0; m < MLP_BWD_MT; ++m)
#pragma unroll
                            for (int t = 0; t < 4; ++t) {
                                float* p = dxp + (r * 16 + lq * 4 + t) * dxs + m * 16 + lr;
                                *p = (w == 0) ? dxacc[r][m][t] : *p + dxacc[r][m][t];
                            }
                }
                __syncthreads();
            }
            for (int t = threadIdx.x; t < BWD_T * cin; t += MLP_THREADS) {
                const int r = t / cin, c = t % cin;
                if (row0 + r < n) dx_slab[((size_t)blockIdx.y * n + row0 + r) * cin + c] = dxp[r * dxs + c];
            }
        }
    }
    // db2 partial of this row walk (hidden slice 0 only; rows of the tile in a fixed order)
    if (blockIdx.y == 0) {
        __syncthreads();
        if (threadIdx.x < BWD_T * 4) dyt[threadIdx.x] = db2acc;
        __syncthreads();
        if (threadIdx.x < 4) {
            float v = 0.f;
            for (int r = 0; r < BWD_T; ++r) v += dyt[r * 4 + threadIdx.x];
            db2_slab[blockIdx.x * 4 + threadIdx.x] = v;
        }
    }
    // parameter-gradient slabs of this workgroup
#pragma unroll
    for (int c = 0; c < MLP_BWD_CTW; ++c) {
        const int ct = (hc0 >> 4) + wave * MLP_BWD_CTW + c;
        const int col = ct * 16 + lr;
        // dW1acc C layout: column = lane&15 (hidden col), row = lq*4 + t (cin index within tile m)
#pragma unroll
        for (int m = 0; m < MLP_BWD_MT; ++m)
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const int cc = m * 16 + lq * 4 + t;
                if (cc < cin) dW1_slab[((size_t)blockIdx.x * cin + cc) * hidden + col] = dW1acc[c][m][t];
            }
        // dW2 / db1: rows were split over the 4 lane quarters
        float v = db1acc[c];
        v += __shfl_xor(v, 16);
        v += __shfl_xor(v, 32);
        if (lq == 0) db1_slab[(size_t)blockIdx.x * hidden + col] = v;
#pragma unroll
        for (int o = 0; o < CO; ++o) {
            float w = dW2acc[c][o];
            w += __shfl_xor(w, 16);
            w += __shfl_xor(w, 32);
            if (lq == 0) dW2_slab[((size_t)blockIdx.x * hidden + col) * 4 + o] = w;
        }
    }
}

}  // namespace fgc

using namespace fgc;

static int mlp_kpad(int cin) { return (cin + 15) / 16 * 16; }
static int mlp_bwd_ctw(int cin) { return cin <= 32 ? 4 : (cin <= 64 ? 2 : 1); }   // see mlp_bwd_kernel
// the backward kernel's k extent is its template width (16 * MT = 128 / CTW): narrower inputs are zero padded up to it
static int mlp_bwd_kpad(int cin) { return 128 / mlp_bwd_ctw(cin); }
static int mlp_bwd_gx(int n) {
    const int ntiles = cdiv(n, BWD_T);
    return ntiles < 128 ? ntiles : 128;
}

namespace fgc {
// fgc_conv_pack's view of the two workspaces: the forward operand at the start of mlp_fwd_ws (split planes where the shape
// takes the split path, else the k-interleaved fp32 operand), the backward operand at the start of mlp_bwd_ws
int mlp_pack_jobs_f32(const fgc_pack_extra* e, PackJob* jobs, size_t* totals) {
    const int cin = e->mlp_cin, hidden = e->mlp_hidden, cout = e->mlp_cout;
    if (cin <= 0 || cin > 128 || hidden <= 0 || hidden % 64 != 0 || cout <= 0 || cout > MLP_COUT_MAX) return -1;
    int nj = 0;
    if (e->mlp_fwd_ws) {
        if ((uintptr_t)e->mlp_fwd_ws % 16 != 0) return -1;
        if (mlp_fwd_split_ok(nullptr, cin, hidden, cout)) {
            jobs[nj] = PackJob{e->mlp_W1, (float*)e->mlp_fwd_ws, 10, cin, cout, 0, hidden, 0, 0, 0, 0, 0, 0};
            totals[nj++] = (size_t)cin * hidden;
        } else {
            const int kpad = mlp_kpad(cin);
            jobs[nj] = PackJob{e->mlp_W1, (float*)e->mlp_fwd_ws, 9, cin, cout, kpad, hidden, 0, 0, 0, 0, 0, 0};
            totals[nj++] = (size_t)kpad * hidden;
        }
    }
    if (e->mlp_bwd_ws) {
        if (hidden % 256 != 0) return -1;
        if (mlp_bwd_split_ok(nullptr, nullptr, cin, hidden, cout)) {
            const int k = mlp_bwd_split_pack_jobs(e, jobs + nj, totals + nj);
            return k < 0 ? -1 : nj + k;
        }
        const int kpad = mlp_bwd_kpad(cin);
        jobs[nj] = PackJob{e->mlp_W1, (float*)e->mlp_bwd_ws, 9, cin, cout, kpad, hidden, 0, 0, 0, 0, 0, 0};
        totals[nj++] = (size_t)kpad * hidden;
    }
    return nj;
}
}  // namespace fgc

namespace fgc {
int mlp_layout_id(int cin, int hidden, int cout, bool bf16) {
    if (bf16) return 1 | 8;     // (the bf16-storage kernels have one operand form)
    return 1 | (mlp_fwd_split_ok(nullptr, cin, hidden, cout) ? 2 : 0) | (mlp_bwd_split_ok(nullptr, nullptr, cin, hidden, cout) ? 4 : 0);
}
}  // namespace fgc

extern "C" int32_t fgc_mlp_layout_id(int32_t cin, int32_t hidden, int32_t cout, int32_t bf16) {
    return mlp_layout_id(cin, hidden, cout, bf16 != 0);
}

// an FGC_MLP_PACKED call whose flags carry the layout the workspace was packed in (FGC_MLP_LAYOUT): refuse if the options moved
#define FGC_MLP_CHECK_LAYOUT(who, bf16)                                                                                        \
    FGC_CHECK_ARG(!(flags & FGC_MLP_PACKED) || (flags >> 8) == 0 || (flags >> 8) == mlp_layout_id(cin, hidden, cout, bf16),        \
                  who ": FGC_MLP_PACKED, but the operands were packed in layout %d and the options now select %d (an option "  \
                  "changed between fgc_conv_pack and this call)", flags >> 8, mlp_layout_id(cin, hidden, cout, bf16))

extern "C" int32_t fgc_mlp_num_partials(int32_t n) { return cdiv(n, MLP_T); }

extern "C" size_t fgc_mlp_workspace_bytes(int32_t cin, int32_t hidden, int32_t cout) {
    // packed W1 + backward slabs (sized for the worst case n-independent parts; dx slabs are sized by the caller's n
    // through fgc_mlp_bwd_workspace_bytes below)
    (void)cout;
    return std::max(align_up((size_t)mlp_kpad(cin) * hidden * sizeof(float), 256), mlp_split_pack_bytes(cin, hidden));
}

extern "C" size_t fgc_mlp_bwd_workspace_bytes(int32_t n, int32_t cin, int32_t hidden, int32_t cout) {
    (void)cout;
    const size_t gx = mlp_bwd_gx(n), gy = hidden / (64 * mlp_bwd_ctw(cin));
    size_t b = align_up((size_t)mlp_bwd_kpad(cin) * hidden * 4, 256);
    b += align_up(gy * (size_t)n * cin * 4, 256);          // dx slabs
    b += align_up(gx * (size_t)cin * hidden * 4, 256);     // dW1 slabs
    b += align_up(gx * (size_t)hidden * 4, 256);           // db1 slabs
    b += align_up(gx * (size_t)hidden * 4 * 4, 256);       // dW2 slabs
    b += align_up((size_t)1024 * 4 * 4, 256);              // db2 partials
    b += align_up((reduce_tmp_floats(1024, 4) + reduce_tmp_floats((int)gx, (size_t)cin * hidden) +
                   reduce_tmp_floats((int)gx, (size_t)hidden * 5)) * 4 + 256, 256);
    if (mlp_bwd_split_ok(nullptr, nullptr, cin, hidden, cout)) b = std::max(b, mlp_bwd_split_workspace_bytes(n, cin, hidden));
    return b;
}

extern "C" int fgc_mlp_fwd(const float* x, int32_t n, int32_t cin, int32_t hidden, int32_t cout, const float* W1,
                           const float* b1, const float* W2, const float* b2, float alpha, float* y,
                           float* abs_partial, int32_t flags, void* workspace, size_t workspace_bytes, void* stream) {
    FGC_CHECK_ARG(x && W1 && b1 && W2 && b2 && y, "fgc_mlp_fwd: null pointer");
    const bool packed = (flags & FGC_MLP_PACKED) != 0;
    FGC_CHECK_ARG(n > 0 && cin > 0 && cin <= 128, "fgc_mlp_fwd: n=%d cin=%d (cin must be in [1,128])", n, cin);
    FGC_CHECK_ARG(hidden > 0 && hidden % 64 == 0, "fgc_mlp_fwd: hidden=%d must be a multiple of 64", hidden);
    FGC_CHECK_ARG(cout > 0 && cout <= MLP_COUT_MAX, "fgc_mlp_fwd: cout=%d outside [1,%d]", cout, MLP_COUT_MAX);
    FGC_CHECK_ARG(workspace && workspace_bytes >= fgc_mlp_workspace_bytes(cin, hidden, cout),
                  "fgc_mlp_fwd: workspace too small");
    FGC_CHECK_ARG(alpha >= 0.f && alpha <= 1.f, "fgc_mlp_fwd: alpha=%g outside [0,1] (the reference uses 0.1, model.py:846)", alpha);
    FGC_MLP_CHECK_LAYOUT("fgc_mlp_fwd", false);
    hipStream_t st = (hipStream_t)stream;
    // the two 1024-wide products on the bf16 matrix pipe with three-term operand splits (fgc_mlp_bf16.hip) where the shape
    // allows: fp32-equivalent results, the matrix time a sixth of the fp32 MFMA's
    if (mlp_fwd_split_ok(x, cin, hidden, cout) && (uintptr_t)workspace % 16 == 0)
        return launch_mlp_fwd_split(x, n, cin, hidden, cout, W1, b1, W2, b2, alpha, y, abs_partial, workspace, packed, st);
    // (the operands fgc_conv_pack leaves are the split planes whenever the SHAPE takes the split path)
    FGC_CHECK_ARG(!packed || !mlp_fwd_split_ok(nullptr, cin, hidden, cout),
                  "fgc_mlp_fwd: FGC_MLP_PACKED needs 16-byte aligned x and workspace for this shape");
    const int kpad = mlp_kpad(cin);
    float* Wp = (float*)workspace;
    if (!packed) {
        FGC_LAUNCH("mlp_pack_kernel", st, mlp_pack_kernel, dim3(cdiv(kpad * hidden, 1024)), dim3(256), 0, W1, Wp, cin, kpad, hidden);
        FGC_CHECK_LAUNCH("fgc_mlp_fwd/pack");
    }
    const size_t smem = (size_t)(MLP_T * (kpad + MLP_XPAD) + 4 * MLP_T * 4 + 4) * 4;
#define FGC_MLP_FWD_LAUNCH(KG)                                                                                     \
    do {                                                                                                           \
        if (cout <= 3)                                                                                             \
            FGC_LAUNCH("mlp_fwd_kernel", st, (mlp_fwd_kernel<KG, 3>), dim3(cdiv(n, MLP_T)), dim3(MLP_THREADS), smem, x, n, \
                       cin, kpad, hidden, cout, Wp, b1, W2, b2, alpha, y, abs_partial);                            \
        else                                                                                                       \
            FGC_LAUNCH("mlp_fwd_kernel", st, (mlp_fwd_kernel<KG, 4>), dim3(cdiv(n, MLP_T)), dim3(MLP_THREADS), smem, x, n, \
                       cin, kpad, hidden, cout, Wp, b1, W2, b2, alpha, y, abs_partial);                            \
    } while (0)
    if (kpad == 16) FGC_MLP_FWD_LAUNCH(1);
    else if (kpad == 32) FGC_MLP_FWD_LAUNCH(2);
    else FGC_MLP_FWD_LAUNCH(0);
#undef FGC_MLP_FWD_LAUNCH
    FGC_CHECK_LAUNCH("fgc_mlp_fwd");
    return FGC_OK;
}

extern "C" int fgc_mlp_bwd(const float* x, const float* dy, int32_t n, int32_t cin, int32_t hidden, int32_t cout,
                           const float* W1, const float* b1, const float* W2, float alpha, float* dx, float* dW1,
                           float* db1, float* dW2, float* db2, int32_t flags, void* workspace, size_t workspace_bytes,
                           void* stream) {
    FGC_CHECK_ARG(x && dy && W1 && b1 && W2 && dx && dW1 && db1 && dW2 && db2, "fgc_mlp_bwd: null pointer");
    FGC_CHECK_ARG(n > 0 && cin > 0 && cin <= 128, "fgc_mlp_bwd: n=%d cin=%d (cin must be in [1,128])", n, cin);
    FGC_CHECK_ARG(hidden > 0 && hidden % 256 == 0, "fgc_mlp_bwd: hidden=%d must be a multiple of 256", hidden);
    FGC_CHECK_ARG(cout > 0 && cout <= MLP_COUT_MAX, "fgc_mlp_bwd: cout=%d outside [1,%d]", cout, MLP_COUT_MAX);
    FGC_CHECK_ARG(alpha >= 0.f && alpha <= 1.f, "fgc_mlp_bwd: alpha=%g outside [0,1] (the reference uses 0.1, model.py:846)", alpha);
    FGC_MLP_CHECK_LAYOUT("fgc_mlp_bwd", false);
    FGC_CHECK_ARG(workspace && workspace_bytes >= fgc_mlp_bwd_workspace_bytes(n, cin, hidden, cout),
                  "fgc_mlp_bwd: workspace too small (%zu < %zu)", workspace_bytes,
                  fgc_mlp_bwd_workspace_bytes(n, cin, hidden, cout));
    hipStream_t st = (hipStream_t)stream;
    // every 1024-wide product on the bf16 matrix pipe with three-term operand splits (fgc_mlp_bf16.hip) where the shape allows
    // (the dx kernel reads b1 in 16-byte pieces as well)
    if (mlp_bwd_split_ok(x, dx, cin, hidden, cout) && (uintptr_t)workspace % 16 == 0 && (uintptr_t)b1 % 16 == 0)
        return launch_mlp_bwd_split(x, dy, n, cin, hidden, cout, W1, b1, W2, alpha, dx, dW1, db1, dW2, db2, workspace,
                                    (flags & FGC_MLP_PACKED) != 0, st);
    // (the operands fgc_conv_pack leaves are the split planes whenever the SHAPE takes the split path)
    FGC_CHECK_ARG(!(flags & FGC_MLP_PACKED) || !mlp_bwd_split_ok(nullptr, nullptr, cin, hidden, cout),
                  "fgc_mlp_bwd: FGC_MLP_PACKED needs 16-byte aligned x, dx, b1 and workspace for this shape");
    const int kpad = mlp_bwd_kpad(cin);
    const int ctw = mlp_bwd_ctw(cin);
    const int gx = mlp_bwd_gx(n), gy = hidden / (64 * ctw);
    char* w = (char*)workspace;
    float* Wp = (float*)w;
    w += align_up((size_t)kpad * hidden * 4, 256);
    float* dx_slab = (float*)w;
    w += align_up((size_t)gy * n * cin * 4, 256);
    float* dW1_slab = (float*)w;
    w += align_up((size_t)gx * cin * hidden * 4, 256);
    float* db1_slab = (float*)w;
    w += align_up((size_t)gx * hidden * 4, 256);
    float* dW2_slab = (float*)w;
    w += align_up((size_t)gx * hidden * 4 * 4, 256);
    float* db2_part = (float*)w;
    w += align_up((size_t)1024 * 4 * 4, 256);
    float* rtmp = (float*)w;

    if (!(flags & FGC_MLP_PACKED)) {
        FGC_LAUNCH("mlp_pack_kernel", st, mlp_pack_kernel, dim3(cdiv(kpad * hidden, 1024)), dim3(256), 0, W1, Wp, cin, kpad, hidden);
        FGC_CHECK_LAUNCH("fgc_mlp_bwd/pack");
    }
    const int hcw = 64 * ctw;   // + the workgroup's weight slice [kpad/4][hcw+1][4], b1 [hcw], W2 [hcw][4]
    const size_t smem = (size_t)(BWD_T * (kpad + MLP_XPAD) + BWD_T * 4 + 4 * BWD_T * MLP_DHS + (ctw == 4 ? 4 : 1) * BWD_T * (kpad + MLP_XPAD) +
                                 4 + (kpad / 4) * (hcw + 1) * 4 + hcw * 5) * 4;
#define FGC_MLP_BWD_LAUNCH(MT, CTW)                                                                                       \
    do {                                                                                                                  \
        if (cout <= 3) {                                                                                                  \
            hipFuncSetAttribute((const void*)mlp_bwd_kernel<MT, CTW, 3>, hipFuncAttributeMaxDynamicSharedMemorySize,      \
                                (int)smem);                                                                               \
            FGC_LAUNCH("mlp_bwd_kernel", st, (mlp_bwd_kernel<MT, CTW, 3>), dim3(gx, gy), dim3(MLP_THREADS), smem, x, dy,  \
                       n, cin, kpad, hidden, cout, Wp, W1, b1, W2, alpha, dx_slab, dW1_slab, db1_slab, dW2_slab, db2_part); \
        } else {                                                                                                          \
            hipFuncSetAttribute((const void*)mlp_bwd_kernel<MT, CTW, 4>, hipFuncAttributeMaxDynamicSharedMemorySize,      \
                                (int)smem);                                                                               \
            FGC_LAUNCH("mlp_bwd_kernel", st, (mlp_bwd_kernel<MT, CTW, 4>), dim3(gx, gy), dim3(MLP_THREADS), smem, x, dy,  \
                       n, cin, kpad, hidden, cout, Wp, W1, b1, W2, alpha, dx_slab, dW1_slab, db1_slab, dW2_slab, db2_part); \
        }                                                                                                                 \
    } while (0)
    if (ctw == 4) FGC_MLP_BWD_LAUNCH(2, 4);
    else if (ctw == 2) FGC_MLP_BWD_LAUNCH(4, 2);
    else FGC_MLP_BWD_LAUNCH(8, 1);
#undef FGC_MLP_BWD_LAUNCH
    FGC_CHECK_LAUNCH("fgc_mlp_bwd");
    // fixed-order reductions, all five in two launches
    const RedJob jobs[5] = {
        {dx_slab, (size_t)n * cin, gy, n * cin, cin, cin, dx},
        {dW1_slab, (size_t)cin * hidden, gx, cin * hidden, hidden, hidden, dW1},
        {db1_slab, (size_t)hidden, gx, hidden, hidden, hidden, db1},
        {dW2_slab, (size_t)hidden * 4, gx, hidden * 4, 4, cout, dW2},
        {db2_part, (size_t)4, gx, 4, 4, cout, db2},
    };
    const int rc = reduce_jobs("reduce:mlp", jobs, 5, rtmp, st);
    if (rc) return rc;
    FGC_CHECK_LAUNCH("fgc_mlp_bwd/reduce");
    return FGC_OK;
}
