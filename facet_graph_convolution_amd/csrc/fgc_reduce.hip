// Fixed-order slab reduction: workgroup = 64 columns x 4 slab lanes; each lane sums every 4th slab of its group
// of up to 64, the 4 lane partials are combined in a fixed order; groups are reduced recursively.
#include "fgc_reduce.h"

namespace fgc {

__global__ __launch_bounds__(256) void reduce_group_kernel(const float* __restrict__ slab, int nslabs, size_t count,
                                                           int in_ld, int out_ld, float* __restrict__ out,
                                                           int final_stage) {
    __shared__ float part[4][64];
    const int col = threadIdx.x & 63, sl = threadIdx.x >> 6;
    const size_t j = (size_t)blockIdx.x * 64 + col;
    const int s0 = blockIdx.y * RED_GROUP;
    const int s1 = min(nslabs, s0 + RED_GROUP);
    float acc = 0.f;
    if (j < count) {
        int s = s0 + sl;
        for (; s + 12 < s1; s += 16) {  // 4 independent loads in flight
            const float a = slab[(size_t)s * count + j], b = slab[(size_t)(s + 4) * count + j],
                        c = slab[(size_t)(s + 8) * count + j], d = slab[(size_t)(s + 12) * count + j];
            acc += a;
            acc += b;
            acc += c;
            acc += d;
        }
        for (; s < s1; s += 4) acc += slab[(size_t)s * count + j];
    }
    part[sl][col] = acc;
    __syncthreads();
    if (sl == 0 && j < count) {
        const float v = (part[0][col] + part[1][col]) + (part[2][col] + part[3][col]);
        if (final_stage) {
            const int c = (int)(j % in_ld);
            if (c < out_ld) out[(j / in_ld) * out_ld + c] = v;
        } else {
            out[(size_t)blockIdx.y * count + j] = v;
        }
    }
}

int reduce_slabs(const char* what, const float* slab, int nslabs, size_t count, int in_ld, int out_ld, float* out,
                 float* tmp, hipStream_t st) {
    const float* src = slab;
    int n = nslabs;
    float* t = tmp;
    while (true) {
        const int groups = (n + RED_GROUP - 1) / RED_GROUP;
        const bool fin = groups == 1;
        float* dst = fin ? out : t;
        FGC_LAUNCH(what, st, reduce_group_kernel, dim3((unsigned)((count + 63) / 64), groups), dim3(256), 0, src, n,
                   count, in_ld, out_ld, dst, fin ? 1 : 0);
        if (fin) break;
        src = t;
        t += (size_t)groups * count;
        n = groups;
    }
    FGC_CHECK_LAUNCH("reduce_slabs");
    return FGC_OK;
}

}  // namespace fgc
