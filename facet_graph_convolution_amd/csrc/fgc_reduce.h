// Fixed-order (bitwise reproducible) reduction of partial-result slabs.
#pragma once
#include "fgc_common.h"

namespace fgc {
constexpr int RED_GROUP = 64;  // slabs summed by one workgroup
constexpr int RED_MAX_JOBS = 48;  // one launch pair covers the parameter gradients of a whole network (5 per layer)
// floats of scratch a job of reduce_jobs with this many slabs may need
static inline size_t reduce_tmp_floats(int nslabs, size_t count) {
    size_t tot = 0;
    int g = nslabs;
    while (g > RED_GROUP) {
        g = (g + RED_GROUP - 1) / RED_GROUP;
        tot += (size_t)g * count;
    }
    return tot;
}
// Several independent reductions in (at most) two launches:
//   out[(j / in_ld) * out_ld + j % in_ld] = sum_s slab[s * stride + j]   for j % in_ld < out_ld   (RedJob::tr: see there)
// Stage 1 sums groups of RED_GROUP slabs of every job, stage 2 sums the group results (jobs with a single group finish in
// stage 1; lists of up to 2 * RED_GROUP slabs are summed by ONE workgroup per column block in a single stage, longer than
// RED_GROUP^2 in proportionally larger groups).  One entry point, one fixed order per (nslabs, count): the per-layer and the
// whole-network calls of a parameter gradient agree bit for bit.
struct RedJob {
    const float* slab;   // element j of slab s at slab[s * stride + j]
    size_t stride;
    int nslabs;
    int count;           // elements per slab that are reduced (j < count)
    int in_ld, out_ld;   // row lengths of the slab and of `out` (see above)
    float* out;
    float* tmp = nullptr;   // when set: this job's (and the following jobs') stage-1 results go here instead of
                            // continuing in the scratch of the job before
    int tr = 0;             // > 0: the slab is [rows][in_ld] with row = m * tr + c; the result goes out TRANSPOSED per m as
                            // out[(m * in_ld + col) * tr + c] for rows < out_ld (the first layer's dW0[m][o][c] out of
                            // its z^T s product [m * cin + c][o]); out_ld then counts rows, not columns
};
// scratch: sum over jobs of reduce_tmp_floats(nslabs, count) (an upper bound: at most RED_GROUP * count per job);
// tmp may be NULL when the first job names its own
int reduce_jobs(const char* what, const RedJob* jobs, int njobs, float* tmp, hipStream_t st);
}  // namespace fgc
