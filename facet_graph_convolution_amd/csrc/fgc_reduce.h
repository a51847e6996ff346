// Fixed-order (bitwise reproducible) reduction of partial-result slabs.
#pragma once
#include "fgc_common.h"

namespace fgc {
constexpr int RED_GROUP = 64;  // slabs summed by one workgroup
constexpr int RED_MAX_JOBS = 6;
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

// Several independent reductions in (at most) two launches: stage 1 sums groups of RED_GROUP slabs of every job, stage
// 2 sums the group results (jobs with a single group finish in stage 1).  Same summation order as reduce_slabs, so the
// results are bit-identical to one reduce_slabs call per job as long as nslabs <= RED_GROUP^2 (longer lists are summed
// in proportionally larger groups: still a fixed order, a different one).
struct RedJob {
    const float* slab;   // element j of slab s at slab[s * stride + j]
    size_t stride;
    int nslabs;
    int count;           // elements per slab that are reduced (j < count)
    int in_ld, out_ld;   // as in reduce_slabs
    float* out;
};
// scratch: sum over jobs of reduce_tmp_floats(nslabs, count) (an upper bound: at most RED_GROUP * count per job)
int reduce_jobs(const char* what, const RedJob* jobs, int njobs, float* tmp, hipStream_t st);
}  // namespace fgc
