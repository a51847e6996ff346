// Fixed-order slab reduction: workgroup = 64 columns x 4 slab lanes; each lane sums every 4th slab of its group
// of up to 64, the 4 lane partials are combined in a fixed order; groups are reduced recursively.
#include <algorithm>

#include "fgc_reduce.h"

namespace fgc {

// The job table travels as a kernel argument and every workgroup scans it for its job: a short table (MAXJ = 6) for the
// few-jobs-many-workgroups calls, the long one only for the whole-network call.
template <int MAXJ>
struct RedJobsT {
    RedJob job[MAXJ];
    float* stage_out[MAXJ];   // where this stage writes (tmp, or the job's out when `fin`)
    int fin[MAXJ];
    int block0[MAXJ + 1];     // first workgroup of each job
    int xblocks[MAXJ];        // workgroups along the element axis
    int group[MAXJ];          // slabs per workgroup (RED_GROUP unless the job has more than RED_GROUP^2 slabs)
    int vec[MAXJ];            // 1: the workgroup covers 64 float4 columns instead of 64 floats (same order per element)
    int njobs;
};
constexpr int RED_FEW_JOBS = 6;
static_assert(sizeof(RedJobsT<RED_MAX_JOBS>) <= 4096, "the job table travels as a kernel argument");

template <int MAXJ>
__global__ __launch_bounds__(256) void reduce_jobs_kernel(RedJobsT<MAXJ> J) {
    __shared__ float part[4][64];
    int q = 0;
#pragma unroll
    for (int t = 1; t < MAXJ; ++t)
        if (t < J.njobs && (int)blockIdx.x >= J.block0[t]) q = t;
    const RedJob& job = J.job[q];
    const int b = blockIdx.x - J.block0[q];
    if (J.group[q] < 0) {
        // a few long slabs (the MLP's four dx slabs): four elements per thread, 1024 per workgroup, the same summation
        // order as the general form below (slab s belongs to lane s % 4; (l0 + l1) + (l2 + l3))
        const size_t j4 = ((size_t)b * 256 + threadIdx.x) * 4;
        if (j4 < (size_t)job.count) {
            f32x4 acc[4];
#pragma unroll
            for (int l = 0; l < 4; ++l) {
                acc[l] = f32x4{0.f, 0.f, 0.f, 0.f};
                if (l < job.nslabs) acc[l] += *reinterpret_cast<const f32x4*>(job.slab + (size_t)l * job.stride + j4);
            }
            *reinterpret_cast<f32x4*>(job.out + j4) = (acc[0] + acc[1]) + (acc[2] + acc[3]);
        }
        return;
    }
    const int bx = b % J.xblocks[q], by = b / J.xblocks[q];
    const int col = threadIdx.x & 63, sl = threadIdx.x >> 6;
    const int s0 = by * J.group[q];
    const int s1 = min(job.nslabs, s0 + J.group[q]);
    const float* slab = job.slab;
    const size_t stride = job.stride;
    if (J.vec[q]) {
        __shared__ __attribute__((aligned(16))) float part4[4][64][4];
        const int j = (bx * 64 + col) * 4;
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
        if (j < job.count) {
            int s = s0 + sl;
            for (; s + 12 < s1; s += 16) {  // 4 independent loads in flight
                const f32x4 a = *reinterpret_cast<const f32x4*>(slab + (size_t)s * stride + j),
                            bb = *reinterpret_cast<const f32x4*>(slab + (size_t)(s + 4) * stride + j),
                            c = *reinterpret_cast<const f32x4*>(slab + (size_t)(s + 8) * stride + j),
                            d = *reinterpret_cast<const f32x4*>(slab + (size_t)(s + 12) * stride + j);
                acc += a;
                acc += bb;
                acc += c;
                acc += d;
            }
            for (; s < s1; s += 4) acc += *reinterpret_cast<const f32x4*>(slab + (size_t)s * stride + j);
        }
        *reinterpret_cast<f32x4*>(part4[sl][col]) = acc;
        __syncthreads();
        if (sl == 0 && j < job.count) {
            const f32x4 v = (*reinterpret_cast<const f32x4*>(part4[0][col]) + *reinterpret_cast<const f32x4*>(part4[1][col])) +
                            (*reinterpret_cast<const f32x4*>(part4[2][col]) + *reinterpret_cast<const f32x4*>(part4[3][col]));
            float* out = J.stage_out[q];
            // in_ld == out_ld for vectorised jobs: element j of the slab is element j of the result in both stages
            *reinterpret_cast<f32x4*>(out + (J.fin[q] ? (size_t)j : (size_t)by * job.count + j)) = v;
        }
        return;
    }
    const int j = bx * 64 + col;
    float acc = 0.f;
    if (j < job.count) {
        int s = s0 + sl;
        for (; s + 12 < s1; s += 16) {  // 4 independent loads in flight
            const float a = slab[(size_t)s * stride + j], bb = slab[(size_t)(s + 4) * stride + j],
                        c = slab[(size_t)(s + 8) * stride + j], d = slab[(size_t)(s + 12) * stride + j];
            acc += a;
            acc += bb;
            acc += c;
            acc += d;
        }
        for (; s < s1; s += 4) acc += slab[(size_t)s * stride + j];
    }
    part[sl][col] = acc;
    __syncthreads();
    if (sl == 0 && j < job.count) {
        const float v = (part[0][col] + part[1][col]) + (part[2][col] + part[3][col]);
        float* out = J.stage_out[q];
        if (J.fin[q] && job.tr > 0) {
            const int row = j / job.in_ld, col = j % job.in_ld;
            if (row < job.out_ld) out[((size_t)(row / job.tr) * job.in_ld + col) * job.tr + row % job.tr] = v;
        } else if (J.fin[q]) {
            const int c = j % job.in_ld;
            if (c < job.out_ld) out[(size_t)(j / job.in_ld) * job.out_ld + c] = v;
        } else {
            out[(size_t)by * job.count + j] = v;
        }
    }
}

template <int MAXJ>
static int reduce_jobs_impl(const char* what, const RedJob* jobs, int njobs, float* tmp, hipStream_t st) {
    RedJobsT<MAXJ> A, B;
    A.njobs = njobs;
    B.njobs = 0;
    int nb = 0, nb2 = 0;
    float* t = tmp;
    for (int q = 0; q < njobs; ++q) {
        const RedJob& j = jobs[q];
        if (j.tmp) t = j.tmp;
        // two stages always suffice: very long slab lists get proportionally larger groups
        int group = std::max(RED_GROUP, (j.nslabs + RED_GROUP - 1) / RED_GROUP);
        // up to two groups' worth of slabs (the MLP's 128 parameter slabs): one workgroup walks them all - a second launch
        // costs more than the longer walk
        if (j.nslabs <= 2 * RED_GROUP) group = std::max(group, j.nslabs);
        const int groups = (j.nslabs + group - 1) / group;
        if (j.nslabs <= 0 || j.count <= 0) {
            set_error("reduce_jobs: job %d has %d slabs of %d elements", q, j.nslabs, j.count);
            return FGC_EINVAL;
        }
        if (groups > 1 && !t) {
            set_error("reduce_jobs: job %d needs scratch and none was given", q);
            return FGC_EINVAL;
        }
        A.job[q] = j;
        const bool wide = j.tr == 0 && j.nslabs <= 4 && j.count >= (1 << 16) && j.count % 4 == 0 && j.in_ld == j.out_ld &&
                          j.stride % 4 == 0 && ((uintptr_t)j.slab % 16) == 0 && ((uintptr_t)j.out % 16) == 0;
        if (wide) {
            A.xblocks[q] = (j.count / 4 + 255) / 256;
            A.block0[q] = nb;
            nb += A.xblocks[q];
            A.group[q] = -1;
            A.vec[q] = 0;
            A.fin[q] = 1;
            A.stage_out[q] = j.out;
            continue;
        }
        // four columns per thread where the layout allows it: a quarter of the workgroups, 16-byte accesses
        const bool vec = j.tr == 0 && j.count >= 4096 && j.count % 4 == 0 && j.in_ld == j.out_ld && j.stride % 4 == 0 &&
                         ((uintptr_t)j.slab % 16) == 0 && ((uintptr_t)j.out % 16) == 0 &&
                         (groups == 1 || ((uintptr_t)t % 16) == 0);
        A.vec[q] = vec ? 1 : 0;
        A.xblocks[q] = vec ? (j.count / 4 + 63) / 64 : (j.count + 63) / 64;
        A.block0[q] = nb;
        nb += A.xblocks[q] * groups;
        A.group[q] = group;
        A.fin[q] = groups == 1;
        A.stage_out[q] = groups == 1 ? j.out : t;
        if (groups > 1) {
            const int k = B.njobs++;
            B.job[k] = j;
            B.job[k].slab = t;
            B.job[k].stride = (size_t)j.count;
            B.job[k].nslabs = groups;
            B.xblocks[k] = A.xblocks[q];
            B.vec[k] = A.vec[q];
            B.block0[k] = nb2;
            nb2 += B.xblocks[k];
            B.group[k] = RED_GROUP;
            B.fin[k] = 1;
            B.stage_out[k] = j.out;
            t += (size_t)groups * j.count;
        }
    }
    A.block0[njobs] = nb;
    FGC_LAUNCH(what, st, reduce_jobs_kernel<MAXJ>, dim3(nb), dim3(256), 0, A);
    if (B.njobs) {
        B.block0[B.njobs] = nb2;
        FGC_LAUNCH(what, st, reduce_jobs_kernel<MAXJ>, dim3(nb2), dim3(256), 0, B);
    }
    FGC_CHECK_LAUNCH("reduce_jobs");
    return FGC_OK;
}

int reduce_jobs(const char* what, const RedJob* jobs, int njobs, float* tmp, hipStream_t st) {
    if (njobs <= 0) return FGC_OK;
    if (njobs > RED_MAX_JOBS) {
        set_error("reduce_jobs: %d jobs (max %d)", njobs, RED_MAX_JOBS);
        return FGC_EINVAL;
    }
    return njobs <= RED_FEW_JOBS ? reduce_jobs_impl<RED_FEW_JOBS>(what, jobs, njobs, tmp, st)
                                 : reduce_jobs_impl<RED_MAX_JOBS>(what, jobs, njobs, tmp, st);
}

}  // namespace fgc
