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

// ---------------------------------------------------------------------------------------------
// multi-scale vertex update (update_position_MS, train.py:1668-1798)
// ---------------------------------------------------------------------------------------------
// node centres of the finest level: barycentre of the face's vertices; -1 corners read a zero vertex (train.py:1779-1787)
__global__ __launch_bounds__(256) void face_centers_kernel(const float* __restrict__ x, int nv,
                                                           const int* __restrict__ faces, int n0,
                                                           float* __restrict__ fpos) {
    const int f = blockIdx.x * blockDim.x + threadIdx.x;
    if (f >= n0) return;
    float s0 = 0.f, s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int t = 0; t < 3; ++t) {
        const int v = faces[3 * (size_t)f + t];
        if (v >= 0 && v < nv) {
            s0 += x[3 * (size_t)v];
            s1 += x[3 * (size_t)v + 1];
            s2 += x[3 * (size_t)v + 2];
        }
    }
    fpos[3 * (size_t)f] = s0 / 3.0f;
    fpos[3 * (size_t)f + 1] = s1 / 3.0f;
    fpos[3 * (size_t)f + 2] = s2 / 3.0f;
}

// model.py:792-814 with steps = 2: thread = (output row, channel); the zero test is on whole rows, so a thread reads
// all c channels of its four input rows for the flags (c is 3 here)
__global__ __launch_bounds__(256) void pool4_avg_iz_kernel(const float* __restrict__ x, int nout, int c,
                                                           float* __restrict__ y) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= nout * c) return;
    const int r = idx / c, ch = idx % c;
    const float* base = x + (size_t)r * 4 * c;
    bool z[4];
    float v[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        bool all0 = true;
        for (int k = 0; k < c; ++k) all0 = all0 && (base[q * c + k] == 0.f);
        z[q] = all0;
        v[q] = base[q * c + ch];
    }
    // first round: pairs (0,1) and (2,3); a zero row takes its partner's values
    const float a0 = z[0] ? v[1] : v[0], a1 = z[1] ? v[0] : v[1];
    const float b0 = z[2] ? v[3] : v[2], b1 = z[3] ? v[2] : v[3];
    const float m0 = (a0 + a1) / 2.0f, m1 = (b0 + b1) / 2.0f;
    // the mean of a pair is a zero ROW only if both inputs were zero rows... or cancel exactly in every channel: the
    // reference tests the pooled values themselves, so do the same test on them
    bool zm0 = true, zm1 = true;
    for (int k = 0; k < c; ++k) {
        const float p0 = z[0] ? base[c + k] : base[k], p1 = z[1] ? base[k] : base[c + k];
        const float q0 = z[2] ? base[3 * c + k] : base[2 * c + k], q1 = z[3] ? base[2 * c + k] : base[3 * c + k];
        zm0 = zm0 && ((p0 + p1) / 2.0f == 0.f);
        zm1 = zm1 && ((q0 + q1) / 2.0f == 0.f);
    }
    const float c0 = zm0 ? m1 : m0, c1 = zm1 ? m0 : m1;
    y[idx] = (c0 + c1) / 2.0f;
}

// one Jacobi iteration at one scale: x_v += (1/#faces(v)) sum_k n (n . (c - x_v)) over the vertex' face slots
__global__ __launch_bounds__(256) void vertex_update_ms_kernel(const float* __restrict__ x, float* __restrict__ xo,
                                                               int nv, const int* __restrict__ vfaces, int k_v,
                                                               int shift, const float* __restrict__ nrm,
                                                               const float* __restrict__ fpos, int nnodes) {
    const int v = blockIdx.x * blockDim.x + threadIdx.x;
    if (v >= nv) return;
    const float x0 = x[3 * (size_t)v], x1 = x[3 * (size_t)v + 1], x2 = x[3 * (size_t)v + 2];
    float a0 = 0.f, a1 = 0.f, a2 = 0.f;
    int numf = 0;
    for (int k = 0; k < k_v; ++k) {
        const int f = vfaces[(size_t)v * k_v + k];
        if (f < 0) continue;                      // -1 stays -1 under the floor division, i.e. the zero-normal fake node
        ++numf;
        const int node = f >> shift;
        if (node >= nnodes) continue;
        const float n0 = nrm[3 * (size_t)node], n1 = nrm[3 * (size_t)node + 1], n2 = nrm[3 * (size_t)node + 2];
        const float e0 = fpos[3 * (size_t)node] - x0, e1 = fpos[3 * (size_t)node + 1] - x1,
                    e2 = fpos[3 * (size_t)node + 2] - x2;
        const float w = (n0 * e0 + n1 * e1) + n2 * e2;
        a0 += w * n0;
        a1 += w * n1;
        a2 += w * n2;
    }
    // a vertex without faces keeps its position (the reference computes 0 * (1/0) = NaN there)
    const float lm = numf > 0 ? 1.0f / (float)numf : 0.f;
    xo[3 * (size_t)v] = x0 + lm * a0;
    xo[3 * (size_t)v + 1] = x1 + lm * a1;
    xo[3 * (size_t)v + 2] = x2 + lm * a2;
}

__global__ void sub3_kernel(const float* a, const float* b, int n, float* o) {   // o may alias b
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) o[i] = a[i] - b[i];
}

}  // namespace fgc

using namespace fgc;

extern "C" int fgc_face_centers(const float* x, int32_t nv, const int32_t* faces, int32_t n0, float* fpos, void* stream) {
    FGC_CHECK_ARG(x && faces && fpos && nv > 0 && n0 > 0, "fgc_face_centers: bad arguments");
    FGC_LAUNCH("face_centers_kernel", (hipStream_t)stream, face_centers_kernel, dim3(cdiv(n0, 256)), dim3(256), 0, x, nv,
               faces, n0, fpos);
    FGC_CHECK_LAUNCH("fgc_face_centers");
    return FGC_OK;
}

extern "C" int fgc_pool4_avg_iz(const float* x, int32_t n, int32_t c, float* y, void* stream) {
    FGC_CHECK_ARG(x && y && n > 0 && n % 4 == 0 && c > 0, "fgc_pool4_avg_iz: n=%d (multiple of 4) c=%d", n, c);
    const int total = (n / 4) * c;
    FGC_LAUNCH("pool4_avg_iz_kernel", (hipStream_t)stream, pool4_avg_iz_kernel, dim3(cdiv(total, 256)), dim3(256), 0, x,
               n / 4, c, y);
    FGC_CHECK_LAUNCH("fgc_pool4_avg_iz");
    return FGC_OK;
}

extern "C" int fgc_vertex_update_ms(const float* x, float* x_out, int32_t nv, const int32_t* faces, int32_t n0,
                                    const int32_t* v_faces, int32_t k_v, const float* normals0, const float* normals1,
                                    const float* normals2, const int32_t* iters, float* dx_out, float* scratch,
                                    size_t scratch_floats, void* stream) {
    FGC_CHECK_ARG(x && x_out && faces && v_faces && normals0 && normals1 && normals2 && iters && scratch,
                  "fgc_vertex_update_ms: null pointer");
    FGC_CHECK_ARG(nv > 0 && n0 > 0 && n0 % 16 == 0 && k_v > 0, "fgc_vertex_update_ms: nv=%d n0=%d (multiple of 16) k_v=%d",
                  nv, n0, k_v);
    FGC_CHECK_ARG(x != x_out, "fgc_vertex_update_ms: x and x_out must be distinct");
    const int n1 = n0 / 4, n2 = n0 / 16;
    const size_t need = 3 * ((size_t)nv * 2 + n0 + n1 + n2);
    FGC_CHECK_ARG(scratch_floats >= need, "fgc_vertex_update_ms: scratch too small (%zu < %zu floats)", scratch_floats, need);
    hipStream_t st = (hipStream_t)stream;
    float* xa = scratch;
    float* xb = xa + 3 * (size_t)nv;
    float* fp0 = xb + 3 * (size_t)nv;
    float* fp1 = fp0 + 3 * (size_t)n0;
    float* fp2 = fp1 + 3 * (size_t)n1;
    const float* nrm[3] = {normals0, normals1, normals2};
    float* fps[3] = {fp0, fp1, fp2};
    const int nn[3] = {n0, n1, n2};
    if (hipMemcpyAsync(xa, x, (size_t)nv * 12, hipMemcpyDeviceToDevice, st) != hipSuccess) {
        fgc::set_error("fgc_vertex_update_ms: copy failed");
        return FGC_EHIP;
    }
    float* cur = xa;
    float* nxt = xb;
    for (int stage = 0; stage < 3; ++stage) {
        const int scale = 2 - stage;                 // coarse to fine (train.py:1687)
        FGC_CHECK_ARG(iters[stage] >= 0, "fgc_vertex_update_ms: negative iteration count");
        float* start = nullptr;
        if (dx_out) {   // remember where this stage started: dx = x_end - x_start (train.py:1763)
            start = dx_out + (size_t)stage * nv * 3;
            if (hipMemcpyAsync(start, cur, (size_t)nv * 12, hipMemcpyDeviceToDevice, st) != hipSuccess) {
                fgc::set_error("fgc_vertex_update_ms: copy failed");
                return FGC_EHIP;
            }
        }
        for (int it = 0; it < iters[stage]; ++it) {
            FGC_LAUNCH("face_centers_kernel", st, face_centers_kernel, dim3(cdiv(n0, 256)), dim3(256), 0, cur, nv, faces, n0, fp0);
            if (scale >= 1)
                FGC_LAUNCH("pool4_avg_iz_kernel", st, pool4_avg_iz_kernel, dim3(cdiv(n1 * 3, 256)), dim3(256), 0, fp0, n1, 3, fp1);
            if (scale >= 2)
                FGC_LAUNCH("pool4_avg_iz_kernel", st, pool4_avg_iz_kernel, dim3(cdiv(n2 * 3, 256)), dim3(256), 0, fp1, n2, 3, fp2);
            FGC_LAUNCH("vertex_update_ms_kernel", st, vertex_update_ms_kernel, dim3(cdiv(nv, 256)), dim3(256), 0, cur, nxt, nv,
                       v_faces, k_v, 2 * scale, nrm[scale], fps[scale], nn[scale]);
            float* t = cur;
            cur = nxt;
            nxt = t;
        }
        if (dx_out)
            FGC_LAUNCH("sub3_kernel", st, sub3_kernel, dim3(cdiv(nv * 3, 256)), dim3(256), 0, cur, start, nv * 3, start);
    }
    if (hipMemcpyAsync(x_out, cur, (size_t)nv * 12, hipMemcpyDeviceToDevice, st) != hipSuccess) {
        fgc::set_error("fgc_vertex_update_ms: copy failed");
        return FGC_EHIP;
    }
    FGC_CHECK_LAUNCH("fgc_vertex_update_ms");
    return FGC_OK;
}

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
