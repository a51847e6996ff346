// Vertex update from denoised face normals (replaces update_position2, train.py:1467-1557).
//
// One thread per vertex and iteration: walk the vertex' edge slots, fetch the edge record [v1, v2, f1, f2] (one
// 16-byte load), the other endpoint's position and the two face normals, accumulate
//     n_f1 (n_f1 . d) + n_f2 (n_f2 . d),   d = x_other - x_i
// in slot order, x_i += lambda * sum.  Jacobi: every vertex reads the previous iterate, so the iterations ping-pong
// between two buffers and are enqueued back to back on the caller's stream.  Pure gather/stream work: the per-vertex
// footprint is 80 B of slots + 16 B per edge + 12 B per endpoint / normal, HBM- (in practice L2-) bound.
#include "fgc_common.h"

namespace fgc {

__global__ __launch_bounds__(256) void vertex_update_kernel(const float* __restrict__ x, float* __restrict__ xo, int nv,
                                                            const float* __restrict__ nrm, int nf,
                                                            const int4* __restrict__ emap, int ne,
                                                            const int* __restrict__ vemap, int max_edges,
                                                            float lambda) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nv) return;
    const float xi0 = x[3 * (size_t)i], xi1 = x[3 * (size_t)i + 1], xi2 = x[3 * (size_t)i + 2];
    float a0 = 0.f, a1 = 0.f, a2 = 0.f;
    const int* slots = vemap + (size_t)i * max_edges;
    for (int s = 0; s < max_edges; ++s) {
        const int e = slots[s];
        if (e < 0 || e >= ne) continue;           // unused slot: the reference adds exact zeros (train.py:1479,1539)
        const int4 em = emap[e];
        const int j = em.x == i ? em.y : em.x;    // the endpoint that is i itself contributes zero (train.py:1523-1526)
        if (j < 0 || j >= nv) continue;
        const float d0 = x[3 * (size_t)j] - xi0, d1 = x[3 * (size_t)j + 1] - xi1, d2 = x[3 * (size_t)j + 2] - xi2;
        float u0 = 0.f, u1 = 0.f, u2 = 0.f;
        if (em.z >= 0 && em.z < nf) {
            const float n0 = nrm[3 * (size_t)em.z], n1 = nrm[3 * (size_t)em.z + 1], n2 = nrm[3 * (size_t)em.z + 2];
            const float dp = (d0 * n0 + d1 * n1) + d2 * n2;
            u0 = n0 * dp;
            u1 = n1 * dp;
            u2 = n2 * dp;
        }
        if (em.w >= 0 && em.w < nf) {              // boundary edges have f2 = -1: zero normal in the reference
            const float n0 = nrm[3 * (size_t)em.w], n1 = nrm[3 * (size_t)em.w + 1], n2 = nrm[3 * (size_t)em.w + 2];
            const float dp = (d0 * n0 + d1 * n1) + d2 * n2;
            u0 += n0 * dp;
            u1 += n1 * dp;
            u2 += n2 * dp;
        }
        a0 += u0;
        a1 += u1;
        a2 += u2;
    }
    xo[3 * (size_t)i] = xi0 + lambda * a0;
    xo[3 * (size_t)i + 1] = xi1 + lambda * a1;
    xo[3 * (size_t)i + 2] = xi2 + lambda * a2;
}

}  // namespace fgc

using namespace fgc;

extern "C" int fgc_vertex_update(const float* x, float* x_out, float* tmp, int32_t nv, const float* normals, int32_t nf,
                                 const int32_t* e_map, int32_t ne, const int32_t* v_e_map, int32_t max_edges,
                                 int32_t iters, float lambda, void* stream) {
    FGC_CHECK_ARG(x && x_out && tmp && normals && e_map && v_e_map, "fgc_vertex_update: null pointer");
    FGC_CHECK_ARG(nv > 0 && nf > 0 && ne >= 0 && max_edges > 0 && iters >= 0,
                  "fgc_vertex_update: nv=%d nf=%d ne=%d max_edges=%d iters=%d", nv, nf, ne, max_edges, iters);
    FGC_CHECK_ARG(x != x_out && x != tmp && x_out != tmp, "fgc_vertex_update: x, x_out and tmp must be distinct");
    FGC_CHECK_ARG((uintptr_t)e_map % 16 == 0, "fgc_vertex_update: e_map needs 16-byte alignment");
    hipStream_t st = (hipStream_t)stream;
    if (iters == 0) {
        if (hipMemcpyAsync(x_out, x, (size_t)nv * 12, hipMemcpyDeviceToDevice, st) != hipSuccess) {
            fgc::set_error("fgc_vertex_update: copy failed");
            return FGC_EHIP;
        }
        return FGC_OK;
    }
    const float* src = x;
    for (int it = 0; it < iters; ++it) {
        float* dst = ((iters - 1 - it) & 1) ? tmp : x_out;   // the last iteration lands in x_out
        FGC_LAUNCH("vertex_update_kernel", st, vertex_update_kernel, dim3(cdiv(nv, 256)), dim3(256), 0, src, dst, nv,
                   normals, nf, reinterpret_cast<const int4*>(e_map), ne, v_e_map, max_edges, lambda);
        src = dst;
    }
    FGC_CHECK_LAUNCH("fgc_vertex_update");
    return FGC_OK;
}
