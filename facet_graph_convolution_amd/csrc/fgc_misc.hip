// Element-wise and reduction ops of the denoising path.  All are HBM-bound streaming kernels:
// float4 accesses where the layout allows, grid capped at 2048 blocks with grid-stride loops,
// reductions in two deterministic stages (no float atomics, bitwise reproducible).
#include <math.h>

#include "fgc_common.h"
#include "fgc_pack.h"

namespace fgc {

constexpr int EW_THREADS = 256;
static inline int ew_grid(int64_t count) {
    int64_t b = (count + EW_THREADS - 1) / EW_THREADS;
    return (int)(b < 1 ? 1 : (b > 2048 ? 2048 : b));
}

__device__ __forceinline__ float lrelu_f(float v, float alpha) { return fmaxf(v, 0.f) - alpha * fmaxf(-v, 0.f); }
// derivative expressed through the OUTPUT y (alpha > 0 keeps the sign): 1, alpha, or 0 at exactly 0
// (tf.nn.relu's gradient is 0 at 0 for both relu terms of model.py:830)
__device__ __forceinline__ float lrelu_slope(float y, float alpha) { return y > 0.f ? 1.f : (y < 0.f ? alpha : 0.f); }

__global__ void lrelu_fwd_kernel(const float* __restrict__ x, float* __restrict__ y, int64_t count, float alpha) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < count; i += (int64_t)gridDim.x * blockDim.x)
        y[i] = lrelu_f(x[i], alpha);
}
__global__ void lrelu_bwd_kernel(const float* __restrict__ y, const float* __restrict__ dy, float* __restrict__ dx,
                                 int64_t count, float alpha) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < count; i += (int64_t)gridDim.x * blockDim.x)
        dx[i] = dy[i] * lrelu_slope(y[i], alpha);
}

__global__ void pool4_fwd_kernel(const float* __restrict__ x, float* __restrict__ y, int64_t count, int c) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < count; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t r = i / c;
        const int col = (int)(i % c);
        const float* p = x + (r * 4) * c + col;
        y[i] = fmaxf(fmaxf(p[0], p[c]), fmaxf(p[2 * c], p[3 * c]));
    }
}
// tf.reduce_max gradient: dy split evenly over the entries equal to the max
// bf16: all four tensors are bf16 (FGC_CONV_BF16 storage; rounding is monotone, so the stored maximum still equals the
// stored entries it came from)
__global__ void pool4_bwd_kernel(const float* __restrict__ x, const float* __restrict__ y,
                                 const float* __restrict__ dy, float* __restrict__ dx, int64_t count, int c,
                                 int accumulate, int bf16) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < count; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t r = i / c;
        const int col = (int)(i % c);
        const float m = ld_act(y, i, bf16);
        const size_t b = (size_t)(r * 4) * c + col;
        float e[4];
        float ne = 0.f;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            e[k] = ld_act(x, b + (size_t)k * c, bf16) == m ? 1.f : 0.f;
            ne += e[k];
        }
        const float g = ld_act(dy, i, bf16) / ne;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const size_t o = b + (size_t)k * c;
            st_act(dx, o, accumulate ? ld_act(dx, o, bf16) + e[k] * g : e[k] * g, bf16);
        }
    }
}
__global__ void upsample4_fwd_kernel(const float* __restrict__ x, float* __restrict__ y, int64_t count_out, int c) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < count_out;
         i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t r = i / c;
        y[i] = x[(r >> 2) * c + i % c];
    }
}
__global__ void upsample4_bwd_kernel(const float* __restrict__ dy, float* __restrict__ dx, int64_t count_in, int c,
                                     int accumulate) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < count_in;
         i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t r = i / c;
        const int col = (int)(i % c);
        const float* p = dy + (r * 4) * c + col;
        const float v = (p[0] + p[c]) + (p[2 * c] + p[3 * c]);
        dx[i] = accumulate ? dx[i] + v : v;
    }
}

// the same four ops for any group size 2^steps (custom_binary_tree_pooling / custom_upsampling with steps != 2,
// model.py:779-788,817-825): the operator API only - the network's 4:1 forms are fused into the conv kernels
__global__ void pool_fwd_kernel(const float* __restrict__ x, float* __restrict__ y, int64_t count, int c, int group) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < count; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t r = i / c;
        const float* p = x + (r * group) * c + (int)(i % c);
        // tf.reduce_max propagates NaN (fmaxf would drop it): a NaN entry makes the group's maximum NaN
        float m = p[0];
        for (int k = 1; k < group; ++k) {
            const float v = p[(int64_t)k * c];
            m = (v > m || v != v) ? v : m;
        }
        y[i] = m;
    }
}
__global__ void pool_bwd_kernel(const float* __restrict__ x, const float* __restrict__ y, const float* __restrict__ dy,
                                float* __restrict__ dx, int64_t count, int c, int group, int accumulate) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < count; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t r = i / c;
        const int64_t b = (r * group) * c + (int)(i % c);
        const float m = y[i];
        float ne = 0.f;
        for (int k = 0; k < group; ++k) ne += x[b + (int64_t)k * c] == m ? 1.f : 0.f;
        // tf.reduce_max: the gradient is split evenly over the entries equal to the maximum (none equals a NaN maximum:
        // no entry gets a gradient, like tf's equal() mask - and no 0 / 0)
        const float g = ne > 0.f ? dy[i] / ne : 0.f;
        for (int k = 0; k < group; ++k) {
            const int64_t o = b + (int64_t)k * c;
            const float v = x[o] == m ? g : 0.f;
            dx[o] = accumulate ? dx[o] + v : v;
        }
    }
}
__global__ void upsample_fwd_kernel(const float* __restrict__ x, float* __restrict__ y, int64_t count_out, int c, int group) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < count_out; i += (int64_t)gridDim.x * blockDim.x)
        y[i] = x[((i / c) / group) * c + i % c];
}
__global__ void upsample_bwd_kernel(const float* __restrict__ dy, float* __restrict__ dx, int64_t count_in, int c, int group,
                                    int accumulate) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < count_in; i += (int64_t)gridDim.x * blockDim.x) {
        const float* p = dy + ((i / c) * group) * c + (int)(i % c);
        float v = 0.f;
        for (int k = 0; k < group; ++k) v += p[(int64_t)k * c];
        dx[i] = accumulate ? dx[i] + v : v;
    }
}

// ---- block reduction helper (deterministic) ---------------------------------------------------
__device__ __forceinline__ float block_sum(float v, float* red /* >= 4 floats LDS */) {
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    __syncthreads();
    if (lane == 0) red[wave] = v;
    __syncthreads();
    float t = 0.f;
    const int nw = blockDim.x >> 6;
    for (int w = 0; w < nw; ++w) t += red[w];
    return t;
}

// ---- normalizeTensor (utils.py:1700-1715) ---------------------------------------------------
constexpr int NORM_ROWS_PER_BLOCK = 1024;

__global__ __launch_bounds__(256) void abs_partial_kernel(const float* __restrict__ x, int64_t count,
                                                          float* __restrict__ part) {
    __shared__ float red[4];
    const int64_t i0 = (int64_t)blockIdx.x * NORM_ROWS_PER_BLOCK * 3;
    const int64_t i1 = min(count, i0 + (int64_t)NORM_ROWS_PER_BLOCK * 3);
    float v = 0.f;
    for (int64_t i = i0 + threadIdx.x; i < i1; i += 256) v += fabsf(x[i]);
    v = block_sum(v, red);
    if (threadIdx.x == 0) part[blockIdx.x] = v;
}
// FUSED: every workgroup sums the partials itself (same fixed order everywhere, so the same value) instead of waiting
// for a one-workgroup launch to do it; workgroup 0 leaves the mean in scratch[0] for the backward pass
template <bool FUSED>
__global__ __launch_bounds__(256) void normalize_fwd_kernel(const float* __restrict__ x, int n,
                                                            float* __restrict__ scratch, float* __restrict__ y,
                                                            const float* __restrict__ part, int nparts, float count) {
    __shared__ float red[4];
    const float eps = 1e-5f;
    float s;
    if (FUSED) {
        float v = 0.f;
        for (int i = threadIdx.x; i < nparts; i += 256) v += part[i];
        v = block_sum(v, red);
        s = v / count + 1e-5f;
        if (blockIdx.x == 0 && threadIdx.x == 0) scratch[0] = s;
    } else {
        s = scratch[0];
    }
    for (int r = blockIdx.x * blockDim.x + threadIdx.x; r < n; r += gridDim.x * blockDim.x) {
        const float a = x[3 * r] / s, b = x[3 * r + 1] / s, c = x[3 * r + 2] / s;
        const float norm = sqrtf(eps + (a * a + b * b + c * c));
        const float inv = norm > eps ? 1.0f / (norm + eps) : 0.f;
        y[3 * r] = a * inv;
        y[3 * r + 1] = b * inv;
        y[3 * r + 2] = c * inv;
    }
}
// backward.  xs = x/s, out = xs*inv(norm).  d xs = dy*inv - xs * (dy.xs) * inv^2 / norm   (norm > eps)
// d s = -sum(d xs . x)/s^2 ;  d x = d xs / s + sign(x) * d s / (3n)
__global__ __launch_bounds__(256) void normalize_bwd_stage1(const float* __restrict__ x, const float* __restrict__ dy,
                                                            int n, const float* __restrict__ scratch,
                                                            float* __restrict__ dxs /* reuse dx buffer */,
                                                            float* __restrict__ part) {
    __shared__ float red[4];
    const float eps = 1e-5f;
    const float s = scratch[0];
    const int r0 = blockIdx.x * NORM_ROWS_PER_BLOCK;
    const int r1 = min(n, r0 + NORM_ROWS_PER_BLOCK);
    float acc = 0.f;
    for (int r = r0 + threadIdx.x; r < r1; r += 256) {
        const float x0 = x[3 * r], x1 = x[3 * r + 1], x2 = x[3 * r + 2];
        const float a = x0 / s, b = x1 / s, c = x2 / s;
        const float norm = sqrtf(eps + (a * a + b * b + c * c));
        float g0 = 0.f, g1 = 0.f, g2 = 0.f;
        if (norm > eps) {
            const float inv = 1.0f / (norm + eps);
            const float d0 = dy[3 * r], d1 = dy[3 * r + 1], d2 = dy[3 * r + 2];
            const float dot = d0 * a + d1 * b + d2 * c;
            const float k = dot * inv * inv / norm;
            g0 = d0 * inv - a * k;
            g1 = d1 * inv - b * k;
            g2 = d2 * inv - c * k;
        }
        dxs[3 * r] = g0;
        dxs[3 * r + 1] = g1;
        dxs[3 * r + 2] = g2;
        acc += g0 * x0 + g1 * x1 + g2 * x2;
    }
    acc = block_sum(acc, red);
    if (threadIdx.x == 0) part[blockIdx.x] = acc;
}
template <bool FUSED>
__global__ __launch_bounds__(256) void normalize_bwd_stage2(const float* __restrict__ x, int64_t count,
                                                            float* __restrict__ scratch, float inv_count,
                                                            float* __restrict__ dx, const float* __restrict__ part,
                                                            int nparts) {
    __shared__ float red[4];
    const float s = scratch[0];
    float ds;
    if (FUSED) {   // as normalize_bwd_finish, in every workgroup
        float v = 0.f;
        for (int i = threadIdx.x; i < nparts; i += 256) v += part[i];
        v = block_sum(v, red);
        ds = -v / (s * s);
        if (blockIdx.x == 0 && threadIdx.x == 0) scratch[1] = ds;
        ds *= inv_count;
    } else {
        ds = scratch[1] * inv_count;
    }
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < count; i += (int64_t)gridDim.x * blockDim.x) {
        const float xv = x[i];
        const float sg = xv > 0.f ? 1.f : (xv < 0.f ? -1.f : 0.f);
        dx[i] = dx[i] / s + sg * ds;
    }
}

// ---- angular loss on sampled rows (train.py:509-517, 1272-1294) ---------------------------------
// one workgroup of 1024 threads; the sample gathers (index -> two rows) are dependent round trips, so each thread issues
// all of its (up to four per sweep) before using any
__global__ __launch_bounds__(1024) void angular_loss_fwd_kernel(const float* __restrict__ fn,
                                                                const float* __restrict__ gt,
                                                                const int* __restrict__ idx, int ns,
                                                                float* __restrict__ out) {
    __shared__ float red[16];
    const float close = 0.9999999f;
    float lsum = 0.f, rsum = 0.f;
    for (int s0 = 0; s0 < ns; s0 += 4096) {
        int r[4];
        float g[4][3], f[4][3];
#pragma unroll
        for (int k = 0; k < 4; ++k) r[k] = idx[min(s0 + (int)threadIdx.x + k * 1024, ns - 1)];
#pragma unroll
        for (int k = 0; k < 4; ++k)
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                g[k][c] = gt[3 * r[k] + c];
                f[k][c] = fn[3 * r[k] + c];
            }
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const bool fake = (fabsf(g[k][0]) + fabsf(g[k][1]) + fabsf(g[k][2])) <= 10e-4f;
            if (s0 + (int)threadIdx.x + k * 1024 < ns && !fake) {
                const float dt = f[k][0] * g[k][0] + f[k][1] * g[k][1] + f[k][2] * g[k][2];
                lsum += 180.f * acosf(fminf(fmaxf(dt, -close), close)) / 3.14159265358979323846f;
                rsum += 1.f;
            }
        }
    }
    lsum = block_sum(lsum, red);
    rsum = block_sum(rsum, red);
    if (threadIdx.x == 0) {
        out[0] = lsum / rsum;
        out[1] = rsum;
    }
}
// d loss / d fn[r] = -(180/pi) / sqrt(1-c^2) * gt[r] / nreal   when |dot| < close (clip gradient is 0 outside)
__global__ void angular_loss_bwd_kernel(const float* __restrict__ fn, const float* __restrict__ gt,
                                        const int* __restrict__ idx, int ns, const float* __restrict__ loss_out,
                                        float dloss, float* __restrict__ dfn) {
    const float close = 0.9999999f;
    const float nreal = loss_out[1];
    for (int s = blockIdx.x * blockDim.x + threadIdx.x; s < ns; s += gridDim.x * blockDim.x) {
        const int r = idx[s];
        const float g0 = gt[3 * r], g1 = gt[3 * r + 1], g2 = gt[3 * r + 2];
        const bool fake = (fabsf(g0) + fabsf(g1) + fabsf(g2)) <= 10e-4f;
        if (fake) continue;
        const float dt = fn[3 * r] * g0 + fn[3 * r + 1] * g1 + fn[3 * r + 2] * g2;
        // tf.minimum/maximum pass the gradient to the un-clipped input only while it is strictly inside
        // (ties: tf gives the gradient to the first argument when equal -> still n_dt for maximum(n_dt,-c)
        //  at equality; measure-zero, ignored)
        if (dt > close || dt < -close) continue;
        const float k = -(180.f / 3.14159265358979323846f) / sqrtf(1.f - dt * dt) * dloss / nreal;
        // duplicates in sample_ind add identical values: order-independent, so atomics stay deterministic
        atomicAdd(&dfn[3 * r], k * g0);
        atomicAdd(&dfn[3 * r + 1], k * g1);
        atomicAdd(&dfn[3 * r + 2], k * g2);
    }
}

// ---- the loss end of a training step in two launches (fgc_loss_step) --------------------------------
// normalize_fwd, the rotation of the ground truth, the sampled angular loss, its gradient and both stages of
// normalize_bwd were seven launches of 5 ... 13 us each, every one waiting for the one before.  Only the SAMPLED rows
// carry a loss, normalizeTensor's backward is linear in the incoming gradient, and the number of real samples enters
// every gradient as one global factor, so:
//   A (one workgroup per 256 samples, no communication between them): s = mean|y| + eps from the MLP's partials (every
//     workgroup sums them itself, in the same order); per sample: normalised row, rotated ground truth, angle, and the
//     gradient of the angle taken back through the row's normalisation - WITHOUT the 1 / (real samples) factor -
//     scattered into `gacc` (rows of d xs; duplicates of a sample add identical values, so the order of the atomic
//     adds does not matter); per workgroup one partial {sum of angles, real samples, sum(d xs . y)}
//   B (all rows): every workgroup sums A's partials (fixed order) -> loss, real samples, ds; dy = gacc / (real s) +
//     sign(y) ds / (3n), n_conv = the normalised rows; gacc is zero again afterwards
// (One workgroup for all samples was the first version: its 12k scattered float atomics from ONE compute unit took 55 us.)
constexpr int LOSS_SAMPLES_PER_BLOCK = 256;
__device__ __forceinline__ void norm_row(const float* y3, float s, float (&a)[3], float& norm, float& inv) {
    const float eps = 1e-5f;
    a[0] = y3[0] / s;
    a[1] = y3[1] / s;
    a[2] = y3[2] / s;
    norm = sqrtf(eps + (a[0] * a[0] + a[1] * a[1] + a[2] * a[2]));
    inv = norm > eps ? 1.0f / (norm + eps) : 0.f;
}
// SHARD (fgc_loss_shard_*): the rows of y are split over ranks.  The sum of |y| over ALL ranks is in scratch[0] (the caller
// all-reduced it), `count` counts the elements of the whole tensor, idx holds this rank's samples as local row ids (ns may
// be 0: the launch still has the step's full number of workgroups, the empty ones leave zero partials), and the partials go
// to scratch[4 + 3 b] - the caller all-reduces that table before the all-rows kernel.
template <bool SHARD>
__global__ __launch_bounds__(LOSS_SAMPLES_PER_BLOCK) void loss_step_samples_kernel(
    const float* __restrict__ y, float count, const float* __restrict__ part, int nparts, const float* __restrict__ gt,
    const float* __restrict__ Rd, const int* __restrict__ idx, int ns, float* __restrict__ gacc,
    float* __restrict__ scratch /* [2 + 3 * blocks]; SHARD: [4 + 3 * blocks] */) {
    __shared__ float red[4];
    const float close = 0.9999999f, eps = 1e-5f;
    float s;
    if constexpr (SHARD) {
        s = scratch[0] / count + 1e-5f;
    } else {
        float v = 0.f;
        for (int i = threadIdx.x; i < nparts; i += LOSS_SAMPLES_PER_BLOCK) v += part[i];
        s = block_sum(v, red) / count + 1e-5f;
    }
    const int sidx = blockIdx.x * LOSS_SAMPLES_PER_BLOCK + threadIdx.x;
    float lsum = 0.f, rsum = 0.f, acc = 0.f;
    if (sidx < ns) {
        const int r = idx[sidx];
        float g0[3], yr[3], g[3], a[3], norm, inv;
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            g0[c] = gt[3 * (size_t)r + c];
            yr[c] = y[3 * (size_t)r + c];
        }
        if (Rd) {
            g[0] = Rd[0] * g0[0] + Rd[1] * g0[1] + Rd[2] * g0[2];
            g[1] = Rd[3] * g0[0] + Rd[4] * g0[1] + Rd[5] * g0[2];
            g[2] = Rd[6] * g0[0] + Rd[7] * g0[1] + Rd[8] * g0[2];
        } else {
            g[0] = g0[0]; g[1] = g0[1]; g[2] = g0[2];
        }
        norm_row(yr, s, a, norm, inv);
        const float dt = (a[0] * inv) * g[0] + (a[1] * inv) * g[1] + (a[2] * inv) * g[2];
        if ((fabsf(g[0]) + fabsf(g[1]) + fabsf(g[2])) > 10e-4f) {     // a real row (train.py:1283-1292)
            lsum = 180.f * acosf(fminf(fmaxf(dt, -close), close)) / 3.14159265358979323846f;
            rsum = 1.f;
            // (tf.minimum / maximum pass the gradient only while the cosine is strictly inside the clip)
            if (!(dt > close || dt < -close) && norm > eps) {
                const float kf = -(180.f / 3.14159265358979323846f) / sqrtf(1.f - dt * dt);
                const float d[3] = {kf * g[0], kf * g[1], kf * g[2]};
                const float dot = d[0] * a[0] + d[1] * a[1] + d[2] * a[2];
                const float kk = dot * inv * inv / norm;
#pragma unroll
                for (int c = 0; c < 3; ++c) {
                    const float gx = d[c] * inv - a[c] * kk;
                    acc += gx * yr[c];
                    atomicAdd(&gacc[3 * (size_t)r + c], gx);
                }
            }
        }
    }
    lsum = block_sum(lsum, red);
    rsum = block_sum(rsum, red);
    acc = block_sum(acc, red);
    if (threadIdx.x == 0) {
        if (!SHARD && blockIdx.x == 0) scratch[0] = s;
        float* o = scratch + (SHARD ? 4 : 2) + 3 * blockIdx.x;
        o[0] = lsum;
        o[1] = rsum;
        o[2] = acc;
    }
}
__global__ __launch_bounds__(256) void abs_sum_kernel(const float* __restrict__ part, int nparts, float* __restrict__ out) {
    __shared__ float red[4];
    float v = 0.f;
    for (int i = threadIdx.x; i < nparts; i += 256) v += part[i];
    v = block_sum(v, red);
    if (threadIdx.x == 0) out[0] = v;
}
// SHARD: scratch[0] = the all-reduced sum of |y|, scratch[4 ...] = the all-reduced partial table; inv_count = 1 / elements
// of the WHOLE tensor
template <bool SHARD>
__global__ __launch_bounds__(256) void loss_step_rows_kernel(const float* __restrict__ y, int n, float* __restrict__ scratch,
                                                             int nblk, float inv_count, float count,
                                                             float* __restrict__ gacc, float* __restrict__ dy,
                                                             float* __restrict__ nconv, float* __restrict__ loss_out) {
    __shared__ float red[4];
    constexpr int P0 = SHARD ? 4 : 2;
    float l = 0.f, rr = 0.f, ac = 0.f;
    for (int i = threadIdx.x; i < nblk; i += 256) {
        l += scratch[P0 + 3 * i];
        rr += scratch[P0 + 1 + 3 * i];
        ac += scratch[P0 + 2 + 3 * i];
    }
    l = block_sum(l, red);
    rr = block_sum(rr, red);
    ac = block_sum(ac, red);
    const float s = SHARD ? scratch[0] / count + 1e-5f : scratch[0];     // (the same expression as in the samples kernel)
    const float inr = 1.0f / rr;                      // (no real sample: inf / nan, as the reference's 0 / 0)
    const float ds_all = -(ac * inr) / (s * s);
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        loss_out[0] = l * inr;
        loss_out[1] = rr;
        scratch[1] = ds_all;
    }
    const float ds = ds_all * inv_count, gs = inr / s;
    for (int r = blockIdx.x * blockDim.x + threadIdx.x; r < n; r += gridDim.x * blockDim.x) {
        const float yr[3] = {y[3 * r], y[3 * r + 1], y[3 * r + 2]};
        float a[3], norm, inv;
        norm_row(yr, s, a, norm, inv);
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const float g = gacc[3 * r + c];
            gacc[3 * r + c] = 0.f;
            const float sg = yr[c] > 0.f ? 1.f : (yr[c] < 0.f ? -1.f : 0.f);
            dy[3 * r + c] = g * gs + sg * ds;
            if (nconv) nconv[3 * r + c] = a[c] * inv;
        }
    }
}

// ---- rotation augmentation (train.py:439-451) ---------------------------------------------------
__global__ void rotate_rows_kernel(const float* __restrict__ x, float* __restrict__ y, int64_t nvec,
                                   const float* __restrict__ Rd) {
    rotate_rows_body(x, y, nvec, Rd, blockIdx.x, gridDim.x);
}

// ---- TF1 Adam ------------------------------------------------------------------------------
__global__ void adam_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                            float* __restrict__ v, int64_t count, float lr_t, float b1, float b2, float eps) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < count; i += (int64_t)gridDim.x * blockDim.x) {
        const float gi = g[i];
        const float mi = b1 * m[i] + (1.f - b1) * gi;
        const float vi = b2 * v[i] + (1.f - b2) * gi * gi;
        m[i] = mi;
        v[i] = vi;
        p[i] -= lr_t * mi / (sqrtf(vi) + eps);
    }
}

// ---- inference epilogue (train.py:115-121,136) -------------------------------------------------
__global__ void infer_epilogue_kernel(const float* __restrict__ nc, const int* __restrict__ perm, int nf,
                                      float* __restrict__ out) {
    for (int f = blockIdx.x * blockDim.x + threadIdx.x; f < nf; f += gridDim.x * blockDim.x) {
        const int r = perm[f];
        float a = nc[3 * r], b = nc[3 * r + 1], c = nc[3 * r + 2];
#pragma unroll
        for (int it = 0; it < 2; ++it) {  // utils.normalize = normalizeOnce twice (utils.py:26-35)
            const float k = 1.0f / (sqrtf(a * a + b * b + c * c) + 0.00000001f);
            a *= k;
            b *= k;
            c *= k;
        }
        out[3 * f] = a;
        out[3 * f + 1] = b;
        out[3 * f + 2] = c;
    }
}

// ---- halo pack / unpack -------------------------------------------------------------------------
__global__ void gather_rows_kernel(const float* __restrict__ src, const int* __restrict__ idx, int64_t count, int c,
                                   float* __restrict__ dst) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < count; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t r = i / c;
        dst[i] = src[(int64_t)idx[r] * c + i % c];
    }
}
// several row copies in one launch (fgc_copy_rows_jobs): workgroup -> job through the table's block prefix
struct RowJobs {
    fgc_row_job job[FGC_ROW_JOBS_MAX];
    int block0[FGC_ROW_JOBS_MAX + 1];
    int njobs;
};
constexpr int ROWJOB_ELEMS = 2048;      // dwords per workgroup
// A facet-sharded step runs 23 of these per rank (14 packs, 9 unpacks: DESIGN.md section 7), each moving 0.1 - 1 MB: what they
// cost is their dependent loads (job table -> row index -> row) and their instructions per element, not their bytes.  Rows
// whose width is a multiple of four dwords (every activation and s row; the 12-dword d-logit rows too) between 16-byte
// aligned buffers move as 16-byte pieces with 32-bit index arithmetic: one division per piece instead of a 64-bit division
// per dword (round 6: 9.2 -> see DESIGN.md section 7 us per launch under the event timer).
__global__ __launch_bounds__(EW_THREADS) void copy_rows_jobs_kernel(RowJobs J) {
    int j = 0;
#pragma unroll 1
    for (int t = 1; t < J.njobs; ++t)
        if ((int)blockIdx.x >= J.block0[t]) j = t;
    const fgc_row_job job = J.job[j];
    const int64_t total = (int64_t)job.rows * job.width;
    const int64_t e0 = (int64_t)(blockIdx.x - J.block0[j]) * ROWJOB_ELEMS;
    const bool vec = (job.width & 3) == 0 && ((((uintptr_t)job.src) | ((uintptr_t)job.dst)) & 15) == 0 && total < (1ll << 31);
    if (vec) {   // (uniform per workgroup)
        const unsigned w4 = (unsigned)job.width >> 2;                     // 16-byte pieces per row
        const unsigned p0 = (unsigned)(e0 >> 2), p1 = (unsigned)(min(e0 + ROWJOB_ELEMS, total) >> 2);
        const f32x4* src4 = reinterpret_cast<const f32x4*>(job.src);
        f32x4* dst4 = reinterpret_cast<f32x4*>(job.dst);
        // both pieces of a thread are requested before either is stored
        unsigned pc[ROWJOB_ELEMS / 4 / EW_THREADS];
        f32x4 v[ROWJOB_ELEMS / 4 / EW_THREADS];
#pragma unroll
        for (int t = 0; t < ROWJOB_ELEMS / 4 / EW_THREADS; ++t) {
            pc[t] = p0 + threadIdx.x + t * EW_THREADS;
            const unsigned pp = min(pc[t], p1 - 1);
            const unsigned r = pp / w4, c = pp - r * w4;
            const unsigned sr = job.idx ? (unsigned)job.idx[r] : r;
            v[t] = src4[(size_t)sr * w4 + c];
        }
#pragma unroll
        for (int t = 0; t < ROWJOB_ELEMS / 4 / EW_THREADS; ++t)
            if (pc[t] < p1) dst4[pc[t]] = v[t];
        return;
    }
    for (int64_t e = e0 + threadIdx.x; e < min(e0 + ROWJOB_ELEMS, total); e += EW_THREADS) {
        const int64_t r = e / job.width;
        const int c = (int)(e - r * job.width);
        const int64_t sr = job.idx ? (int64_t)job.idx[r] : r;
        job.dst[e] = job.src[sr * job.width + c];
    }
}
__global__ void scatter_add_rows_kernel(const float* __restrict__ src, const int* __restrict__ idx, int64_t count,
                                        int c, float* __restrict__ dst) {
    // idx must be duplicate-free (each halo row has one owner): plain read-modify-write, deterministic
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < count; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t r = i / c;
        dst[(int64_t)idx[r] * c + i % c] += src[i];
    }
}

}  // namespace fgc

using namespace fgc;
#define ST ((hipStream_t)stream)

extern "C" int fgc_lrelu_fwd(const float* x, float* y, int64_t count, float alpha, void* stream) {
    FGC_CHECK_ARG(x && y && count >= 0, "fgc_lrelu_fwd: bad arguments");
    if (!count) return FGC_OK;
    FGC_LAUNCH("lrelu_fwd_kernel", ST, lrelu_fwd_kernel, dim3(ew_grid(count)), dim3(EW_THREADS), 0, x, y, count, alpha);
    FGC_CHECK_LAUNCH("fgc_lrelu_fwd");
    return FGC_OK;
}
extern "C" int fgc_lrelu_bwd(const float* y, const float* dy, float* dx, int64_t count, float alpha, void* stream) {
    FGC_CHECK_ARG(y && dy && dx && count >= 0, "fgc_lrelu_bwd: bad arguments");
    if (!count) return FGC_OK;
    FGC_LAUNCH("lrelu_bwd_kernel", ST, lrelu_bwd_kernel, dim3(ew_grid(count)), dim3(EW_THREADS), 0, y, dy, dx, count, alpha);
    FGC_CHECK_LAUNCH("fgc_lrelu_bwd");
    return FGC_OK;
}
extern "C" int fgc_pool4_fwd(const float* x, float* y, int32_t n_out, int32_t c, void* stream) {
    FGC_CHECK_ARG(x && y && n_out > 0 && c > 0, "fgc_pool4_fwd: bad arguments");
    const int64_t cnt = (int64_t)n_out * c;
    FGC_LAUNCH("pool4_fwd_kernel", ST, pool4_fwd_kernel, dim3(ew_grid(cnt)), dim3(EW_THREADS), 0, x, y, cnt, c);
    FGC_CHECK_LAUNCH("fgc_pool4_fwd");
    return FGC_OK;
}
extern "C" int fgc_pool4_bwd(const float* x, const float* y, const float* dy, float* dx, int32_t n_out, int32_t c,
                             int32_t accumulate, void* stream) {
    FGC_CHECK_ARG(x && y && dy && dx && n_out > 0 && c > 0, "fgc_pool4_bwd: bad arguments");
    const int64_t cnt = (int64_t)n_out * c;
    FGC_LAUNCH("pool4_bwd_kernel", ST, pool4_bwd_kernel, dim3(ew_grid(cnt)), dim3(EW_THREADS), 0, x, y, dy, dx, cnt, c, accumulate, 0);
    FGC_CHECK_LAUNCH("fgc_pool4_bwd");
    return FGC_OK;
}
extern "C" int fgc_pool4_bwd_bf16(const void* x, const void* y, const void* dy, void* dx, int32_t n_out, int32_t c,
                                  int32_t accumulate, void* stream) {
    FGC_CHECK_ARG(x && y && dy && dx && n_out > 0 && c > 0, "fgc_pool4_bwd_bf16: bad arguments");
    const int64_t cnt = (int64_t)n_out * c;
    FGC_LAUNCH("pool4_bwd_kernel", ST, pool4_bwd_kernel, dim3(ew_grid(cnt)), dim3(EW_THREADS), 0, (const float*)x, (const float*)y,
               (const float*)dy, (float*)dx, cnt, c, accumulate, 1);
    FGC_CHECK_LAUNCH("fgc_pool4_bwd_bf16");
    return FGC_OK;
}
extern "C" int fgc_upsample4_fwd(const float* x, float* y, int32_t n_in, int32_t c, void* stream) {
    FGC_CHECK_ARG(x && y && n_in > 0 && c > 0, "fgc_upsample4_fwd: bad arguments");
    const int64_t cnt = (int64_t)n_in * 4 * c;
    FGC_LAUNCH("upsample4_fwd_kernel", ST, upsample4_fwd_kernel, dim3(ew_grid(cnt)), dim3(EW_THREADS), 0, x, y, cnt, c);
    FGC_CHECK_LAUNCH("fgc_upsample4_fwd");
    return FGC_OK;
}
extern "C" int fgc_upsample4_bwd(const float* dy, float* dx, int32_t n_in, int32_t c, int32_t accumulate,
                                 void* stream) {
    FGC_CHECK_ARG(dy && dx && n_in > 0 && c > 0, "fgc_upsample4_bwd: bad arguments");
    const int64_t cnt = (int64_t)n_in * c;
    FGC_LAUNCH("upsample4_bwd_kernel", ST, upsample4_bwd_kernel, dim3(ew_grid(cnt)), dim3(EW_THREADS), 0, dy, dx, cnt, c, accumulate);
    FGC_CHECK_LAUNCH("fgc_upsample4_bwd");
    return FGC_OK;
}

extern "C" int fgc_pool_fwd(const float* x, float* y, int32_t n_out, int32_t c, int32_t group, void* stream) {
    FGC_CHECK_ARG(x && y && n_out > 0 && c > 0 && group >= 1, "fgc_pool_fwd: bad arguments");
    const int64_t cnt = (int64_t)n_out * c;
    FGC_LAUNCH("pool_fwd_kernel", ST, pool_fwd_kernel, dim3(ew_grid(cnt)), dim3(EW_THREADS), 0, x, y, cnt, c, group);
    FGC_CHECK_LAUNCH("fgc_pool_fwd");
    return FGC_OK;
}
extern "C" int fgc_pool_bwd(const float* x, const float* y, const float* dy, float* dx, int32_t n_out, int32_t c,
                            int32_t group, int32_t accumulate, void* stream) {
    FGC_CHECK_ARG(x && y && dy && dx && n_out > 0 && c > 0 && group >= 1, "fgc_pool_bwd: bad arguments");
    const int64_t cnt = (int64_t)n_out * c;
    FGC_LAUNCH("pool_bwd_kernel", ST, pool_bwd_kernel, dim3(ew_grid(cnt)), dim3(EW_THREADS), 0, x, y, dy, dx, cnt, c, group,
               accumulate);
    FGC_CHECK_LAUNCH("fgc_pool_bwd");
    return FGC_OK;
}
extern "C" int fgc_upsample_fwd(const float* x, float* y, int32_t n_in, int32_t c, int32_t group, void* stream) {
    FGC_CHECK_ARG(x && y && n_in > 0 && c > 0 && group >= 1, "fgc_upsample_fwd: bad arguments");
    const int64_t cnt = (int64_t)n_in * group * c;
    FGC_LAUNCH("upsample_fwd_kernel", ST, upsample_fwd_kernel, dim3(ew_grid(cnt)), dim3(EW_THREADS), 0, x, y, cnt, c, group);
    FGC_CHECK_LAUNCH("fgc_upsample_fwd");
    return FGC_OK;
}
extern "C" int fgc_upsample_bwd(const float* dy, float* dx, int32_t n_in, int32_t c, int32_t group, int32_t accumulate,
                                void* stream) {
    FGC_CHECK_ARG(dy && dx && n_in > 0 && c > 0 && group >= 1, "fgc_upsample_bwd: bad arguments");
    const int64_t cnt = (int64_t)n_in * c;
    FGC_LAUNCH("upsample_bwd_kernel", ST, upsample_bwd_kernel, dim3(ew_grid(cnt)), dim3(EW_THREADS), 0, dy, dx, cnt, c, group,
               accumulate);
    FGC_CHECK_LAUNCH("fgc_upsample_bwd");
    return FGC_OK;
}

extern "C" int32_t fgc_norm_num_partials(int32_t n) { return cdiv(n, NORM_ROWS_PER_BLOCK); }

extern "C" int fgc_normalize_fwd(const float* x, int32_t n, const float* abs_partial, int32_t num_partials, float* y,
                                 float* scratch, void* stream) {
    FGC_CHECK_ARG(x && y && scratch && n > 0, "fgc_normalize_fwd: bad arguments");
    const float inv_count = 3.0f * (float)n;  // element count (the kernel divides)
    const float* part = abs_partial;
    int np = num_partials;
    if (!(abs_partial && num_partials > 0)) {
        np = fgc_norm_num_partials(n);
        FGC_LAUNCH("abs_partial_kernel", ST, abs_partial_kernel, dim3(np), dim3(256), 0, x, (int64_t)n * 3, scratch + 2);
        part = scratch + 2;
    }
    FGC_LAUNCH("normalize_fwd_kernel", ST, normalize_fwd_kernel<true>, dim3(ew_grid(n)), dim3(EW_THREADS), 0, x, n, scratch, y,
               part, np, inv_count);
    FGC_CHECK_LAUNCH("fgc_normalize_fwd");
    return FGC_OK;
}
extern "C" int fgc_normalize_bwd(const float* x, const float* dy, int32_t n, float* dx, float* scratch,
                                 void* stream) {
    FGC_CHECK_ARG(x && dy && dx && scratch && n > 0, "fgc_normalize_bwd: bad arguments");
    // scratch[0] (= mean|x| + eps) is the value left by fgc_normalize_fwd on the same x
    const int np = fgc_norm_num_partials(n);
    FGC_LAUNCH("normalize_bwd_stage1", ST, normalize_bwd_stage1, dim3(np), dim3(256), 0, x, dy, n, scratch, dx, scratch + 2);
    FGC_LAUNCH("normalize_bwd_stage2", ST, normalize_bwd_stage2<true>, dim3(ew_grid((int64_t)n * 3)), dim3(EW_THREADS), 0, x,
               (int64_t)n * 3, scratch, 1.0f / (3.0f * (float)n), dx, scratch + 2, np);
    FGC_CHECK_LAUNCH("fgc_normalize_bwd");
    return FGC_OK;
}

extern "C" int fgc_normalize_apply(const float* x, int32_t n, const float* scratch, float* y, void* stream) {
    FGC_CHECK_ARG(x && y && scratch && n > 0, "fgc_normalize_apply: bad arguments");
    FGC_LAUNCH("normalize_fwd_kernel", ST, normalize_fwd_kernel<false>, dim3(ew_grid(n)), dim3(EW_THREADS), 0, x, n,
               const_cast<float*>(scratch), y, nullptr, 0, 1.f);
    FGC_CHECK_LAUNCH("fgc_normalize_apply");
    return FGC_OK;
}
extern "C" int fgc_normalize_bwd_partial(const float* x, const float* dy, int32_t n, const float* scratch, float* dx,
                                         float* partial, void* stream) {
    FGC_CHECK_ARG(x && dy && dx && scratch && partial && n > 0, "fgc_normalize_bwd_partial: bad arguments");
    FGC_LAUNCH("normalize_bwd_stage1", ST, normalize_bwd_stage1, dim3(fgc_norm_num_partials(n)), dim3(256), 0, x, dy, n,
               scratch, dx, partial);
    FGC_CHECK_LAUNCH("fgc_normalize_bwd_partial");
    return FGC_OK;
}
extern "C" int fgc_normalize_bwd_apply(const float* x, int32_t n, float total_count, const float* scratch, float* dx,
                                       void* stream) {
    FGC_CHECK_ARG(x && dx && scratch && n > 0 && total_count > 0, "fgc_normalize_bwd_apply: bad arguments");
    FGC_LAUNCH("normalize_bwd_stage2", ST, normalize_bwd_stage2<false>, dim3(ew_grid((int64_t)n * 3)), dim3(EW_THREADS), 0, x,
               (int64_t)n * 3, const_cast<float*>(scratch), 1.0f / total_count, dx, nullptr, 0);
    FGC_CHECK_LAUNCH("fgc_normalize_bwd_apply");
    return FGC_OK;
}

extern "C" int fgc_angular_loss_fwd(const float* fn, const float* gt, const int32_t* sample_ind, int32_t ns,
                                    float* loss_out, void* stream) {
    FGC_CHECK_ARG(fn && gt && sample_ind && loss_out && ns > 0, "fgc_angular_loss_fwd: bad arguments");
    FGC_LAUNCH("angular_loss_fwd_kernel", ST, angular_loss_fwd_kernel, dim3(1), dim3(1024), 0, fn, gt, sample_ind, ns, loss_out);
    FGC_CHECK_LAUNCH("fgc_angular_loss_fwd");
    return FGC_OK;
}
extern "C" int fgc_angular_loss_bwd(const float* fn, const float* gt, const int32_t* sample_ind, int32_t ns, int32_t n,
                                    const float* loss_out, float dloss, float* dfn, void* stream) {
    FGC_CHECK_ARG(fn && gt && sample_ind && loss_out && dfn && ns > 0 && n > 0, "fgc_angular_loss_bwd: bad arguments");
    if (hipMemsetAsync(dfn, 0, (size_t)n * 3 * sizeof(float), ST) != hipSuccess) {
        fgc::set_error("fgc_angular_loss_bwd: memset failed");
        return FGC_EHIP;
    }
    FGC_LAUNCH("angular_loss_bwd_kernel", ST, angular_loss_bwd_kernel, dim3(cdiv(ns, 256)), dim3(256), 0, fn, gt, sample_ind, ns, loss_out,
                       dloss, dfn);
    FGC_CHECK_LAUNCH("fgc_angular_loss_bwd");
    return FGC_OK;
}

extern "C" int32_t fgc_loss_step_scratch_floats(int32_t ns) { return 2 + 3 * cdiv(ns, LOSS_SAMPLES_PER_BLOCK); }

extern "C" int fgc_loss_step(const float* y, int32_t n, const float* abs_partial, int32_t num_partials, const float* gt,
                             const float* R, const int32_t* sample_ind, int32_t ns, float* gacc, float* n_conv, float* dy,
                             float* loss_out, float* scratch, void* stream) {
    FGC_CHECK_ARG(y && abs_partial && num_partials > 0 && gt && sample_ind && gacc && dy && loss_out && scratch && n > 0 &&
                  ns > 0, "fgc_loss_step: bad arguments");
    const int nblk = cdiv(ns, LOSS_SAMPLES_PER_BLOCK);
    FGC_LAUNCH("loss_step_samples_kernel", ST, loss_step_samples_kernel<false>, dim3(nblk), dim3(LOSS_SAMPLES_PER_BLOCK), 0, y,
               3.0f * (float)n, abs_partial, num_partials, gt, R, sample_ind, ns, gacc, scratch);
    FGC_LAUNCH("loss_step_rows_kernel", ST, loss_step_rows_kernel<false>, dim3(ew_grid(n)), dim3(EW_THREADS), 0, y, n, scratch, nblk,
               1.0f / (3.0f * (float)n), 3.0f * (float)n, gacc, dy, n_conv, loss_out);
    FGC_CHECK_LAUNCH("fgc_loss_step");
    return FGC_OK;
}

extern "C" int32_t fgc_loss_shard_floats(int32_t ns_total) { return 4 + 3 * cdiv(ns_total, LOSS_SAMPLES_PER_BLOCK); }

extern "C" int fgc_loss_shard_abs_sum(const float* abs_partial, int32_t num_partials, float* sums, void* stream) {
    FGC_CHECK_ARG(abs_partial && num_partials > 0 && sums, "fgc_loss_shard_abs_sum: bad arguments");
    FGC_LAUNCH("abs_sum_kernel", ST, abs_sum_kernel, dim3(1), dim3(256), 0, abs_partial, num_partials, sums);
    FGC_CHECK_LAUNCH("fgc_loss_shard_abs_sum");
    return FGC_OK;
}

extern "C" int fgc_loss_shard_samples(const float* y, float total_count, const float* gt, const float* R,
                                      const int32_t* sample_local, int32_t ns_local, int32_t ns_total, float* gacc, float* sums,
                                      void* stream) {
    FGC_CHECK_ARG(y && gt && gacc && sums && total_count > 0 && ns_total > 0 && ns_local >= 0 && ns_local <= ns_total &&
                  (sample_local || !ns_local), "fgc_loss_shard_samples: bad arguments (ns_local=%d ns_total=%d)", ns_local, ns_total);
    // (the whole step's number of workgroups on every rank: the partial table has the same shape everywhere)
    FGC_LAUNCH("loss_step_samples_kernel", ST, loss_step_samples_kernel<true>, dim3(cdiv(ns_total, LOSS_SAMPLES_PER_BLOCK)),
               dim3(LOSS_SAMPLES_PER_BLOCK), 0, y, total_count, (const float*)nullptr, 0, gt, R, sample_local ? sample_local : (const int32_t*)sums,
               ns_local, gacc, sums);
    FGC_CHECK_LAUNCH("fgc_loss_shard_samples");
    return FGC_OK;
}

extern "C" int fgc_loss_shard_rows(const float* y, int32_t n_local, float total_count, int32_t ns_total, float* sums, float* gacc,
                                   float* n_conv, float* dy, float* loss_out, void* stream) {
    FGC_CHECK_ARG(y && sums && gacc && dy && loss_out && n_local > 0 && total_count > 0 && ns_total > 0,
                  "fgc_loss_shard_rows: bad arguments");
    FGC_LAUNCH("loss_step_rows_kernel", ST, loss_step_rows_kernel<true>, dim3(ew_grid(n_local)), dim3(EW_THREADS), 0, y, n_local, sums,
               cdiv(ns_total, LOSS_SAMPLES_PER_BLOCK), 1.0f / total_count, total_count, gacc, dy, n_conv, loss_out);
    FGC_CHECK_LAUNCH("fgc_loss_shard_rows");
    return FGC_OK;
}

extern "C" int fgc_rotate_rows(const float* x, float* y, int32_t n, int32_t vecs, const float* R, void* stream) {
    FGC_CHECK_ARG(x && y && R && n > 0 && vecs > 0, "fgc_rotate_rows: bad arguments");
    const int64_t nvec = (int64_t)n * vecs;
    FGC_LAUNCH("rotate_rows_kernel", ST, rotate_rows_kernel, dim3(ew_grid(nvec)), dim3(EW_THREADS), 0, x, y, nvec, R);
    FGC_CHECK_LAUNCH("fgc_rotate_rows");
    return FGC_OK;
}

extern "C" int fgc_adam_step(float* p, const float* g, float* m, float* v, int64_t count, int32_t t, float lr,
                             float b1, float b2, float eps, void* stream) {
    FGC_CHECK_ARG(p && g && m && v && count > 0 && t >= 1, "fgc_adam_step: bad arguments");
    const double lr_t = (double)lr * sqrt(1.0 - pow((double)b2, t)) / (1.0 - pow((double)b1, t));
    FGC_LAUNCH("adam_kernel", ST, adam_kernel, dim3(ew_grid(count)), dim3(EW_THREADS), 0, p, g, m, v, count, (float)lr_t, b1, b2,
                       eps);
    FGC_CHECK_LAUNCH("fgc_adam_step");
    return FGC_OK;
}

extern "C" int fgc_infer_epilogue(const float* n_conv, const int32_t* perm, int32_t num_faces, float* out,
                                  void* stream) {
    FGC_CHECK_ARG(n_conv && perm && out && num_faces > 0, "fgc_infer_epilogue: bad arguments");
    FGC_LAUNCH("infer_epilogue_kernel", ST, infer_epilogue_kernel, dim3(ew_grid(num_faces)), dim3(EW_THREADS), 0, n_conv, perm, num_faces,
                       out);
    FGC_CHECK_LAUNCH("fgc_infer_epilogue");
    return FGC_OK;
}

extern "C" int fgc_gather_rows(const float* src, const int32_t* idx, int32_t count, int32_t c, float* dst,
                               void* stream) {
    FGC_CHECK_ARG(src && idx && dst && count >= 0 && c > 0, "fgc_gather_rows: bad arguments");
    if (!count) return FGC_OK;
    const int64_t cnt = (int64_t)count * c;
    FGC_LAUNCH("gather_rows_kernel", ST, gather_rows_kernel, dim3(ew_grid(cnt)), dim3(EW_THREADS), 0, src, idx, cnt, c, dst);
    FGC_CHECK_LAUNCH("fgc_gather_rows");
    return FGC_OK;
}
extern "C" int fgc_copy_rows_jobs(const fgc_row_job* jobs, int32_t njobs, void* stream) {
    FGC_CHECK_ARG(njobs >= 0 && njobs <= FGC_ROW_JOBS_MAX && (jobs || !njobs), "fgc_copy_rows_jobs: 0 <= njobs <= %d",
                  FGC_ROW_JOBS_MAX);
    RowJobs J;
    J.njobs = 0;
    int blocks = 0;
    for (int j = 0; j < njobs; ++j) {
        const fgc_row_job& q = jobs[j];
        FGC_CHECK_ARG(q.rows >= 0 && q.width > 0, "fgc_copy_rows_jobs: job %d: rows %d, width %d", j, q.rows, q.width);
        if (!q.rows) continue;
        FGC_CHECK_ARG(q.src && q.dst, "fgc_copy_rows_jobs: job %d: null pointer", j);
        J.job[J.njobs] = q;
        J.block0[J.njobs] = blocks;
        blocks += (int)(((int64_t)q.rows * q.width + ROWJOB_ELEMS - 1) / ROWJOB_ELEMS);
        ++J.njobs;
    }
    J.block0[J.njobs] = blocks;
    if (!blocks) return FGC_OK;
    FGC_LAUNCH("copy_rows_jobs_kernel", ST, copy_rows_jobs_kernel, dim3(blocks), dim3(EW_THREADS), 0, J);
    FGC_CHECK_LAUNCH("fgc_copy_rows_jobs");
    return FGC_OK;
}
extern "C" int fgc_scatter_add_rows(const float* src, const int32_t* idx, int32_t count, int32_t c, float* dst,
                                    void* stream) {
    FGC_CHECK_ARG(src && idx && dst && count >= 0 && c > 0, "fgc_scatter_add_rows: bad arguments");
    if (!count) return FGC_OK;
    const int64_t cnt = (int64_t)count * c;
    FGC_LAUNCH("scatter_add_rows_kernel", ST, scatter_add_rows_kernel, dim3(ew_grid(cnt)), dim3(EW_THREADS), 0, src, idx, cnt, c, dst);
    FGC_CHECK_LAUNCH("fgc_scatter_add_rows");
    return FGC_OK;
}
