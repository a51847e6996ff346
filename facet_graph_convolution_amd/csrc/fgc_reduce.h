// Fixed-order (bitwise reproducible) reduction of partial-result slabs.
#pragma once
#include "fgc_common.h"

namespace fgc {
constexpr int RED_GROUP = 64;  // slabs summed by one workgroup
constexpr int RED_MAX_JOBS = 48;  // one launch pair covers the parameter gradients of a whole network (5 per layer)
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
// results are bit-identical to one reduce_slabs call per job for nslabs <= RED_GROUP (lists of up to 2 * RED_GROUP slabs are
// summed by ONE workgroup per column block in a single stage, longer than RED_GROUP^2 in proportionally larger groups:
// still a fixed order, a different one).
struct RedJob {
    const float* slab;   // element j of slab s at slab[s * stride + j]
    size_t stride;
    int nslabs;
    int count;           // elements per slab that are reduced (j < count)
    int in_ld, out_ld;   // as in reduce_slabs
    float* out;
    float* tmp = nullptr;   // when set: this job's (and the following jobs') stage-1 results go here instead of
                            // continuing in the scratch of the job before
};
// scratch: sum over jobs of reduce_tmp_floats(nslabs, count) (an upper bound: at most RED_GROUP * count per job);
// tmp may be NULL when the first job names its own
int reduce_jobs(const char* what, const RedJob* jobs, int njobs, float* tmp, hipStream_t st);
}  // namespace fgc
