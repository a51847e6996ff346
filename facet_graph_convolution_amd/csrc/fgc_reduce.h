// Fixed-order (bitwise reproducible) reduction of partial-result slabs.
#pragma once
#include "fgc_common.h"

namespace fgc {
constexpr int RED_GROUP = 64;  // slabs summed by one workgroup
// floats of scratch reduce_slabs needs
static inline size_t reduce_tmp_floats(int nslabs, size_t count) {
    size_t tot = 0;
    int g = nslabs;
    while (g > RED_GROUP) {
        g = (g + RED_GROUP - 1) / RED_GROUP;
        tot += (size_t)g * count;
    }
    return tot;
}
// out[(j / in_ld) * out_ld + j % in_ld] = sum_s slab[s * count + j]   for j % in_ld < out_ld
int reduce_slabs(const char* what, const float* slab, int nslabs, size_t count, int in_ld, int out_ld, float* out,
                 float* tmp, hipStream_t st);
}  // namespace fgc
