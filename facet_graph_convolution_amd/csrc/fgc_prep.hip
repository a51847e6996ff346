// Host-side mesh preprocessing (CPU, native): faces -> facet adjacency K-list -> weighted graph ->
// 4 levels of graclus-style pairing -> binary-tree node ordering -> the three K-lists the network
// consumes.  Replaces the pure-Python loops of the reference:
//   getFacesLargeAdj        utils.py:243-295         (bit-exact)
//   computeFacesNormals     utils.py:63-68, 26-35    (fp32, same operation order)
//   getTrianglesBarycenter  utils.py:1264-1294
//   listToSparseWNormals    utils.py:1753-1796
//   coarsen / metis / metis_one_level / compute_perm / perm_adjacency   lib/coarsening.py:5-296
//   sparseToList            utils.py:1799-1827       (bit-exact given the cluster assignments)
// The pairing itself is a random procedure in the reference (numpy's global Mersenne Twister and an
// unstable argsort decide it); here it is deterministic given `seed`, or replays
// recorded cluster assignments (`parents`) bit-exactly.
#include <math.h>
#include <string.h>

#include <algorithm>
#include <numeric>
#include <random>
#include <vector>

#include "fgc_common.h"

namespace {

struct Coo {
    int n = 0;
    std::vector<int> row, col;
    std::vector<float> val;
};

// row-major canonical form: sorted by (row, col), duplicates summed (what scipy's csr conversion yields)
static void canonicalize(Coo& g) {
    const size_t nnz = g.row.size();
    std::vector<size_t> order(nnz);
    std::iota(order.begin(), order.end(), (size_t)0);
    std::stable_sort(order.begin(), order.end(), [&](size_t a, size_t b) {
        if (g.row[a] != g.row[b]) return g.row[a] < g.row[b];
        return g.col[a] < g.col[b];
    });
    Coo o;
    o.n = g.n;
    for (size_t t = 0; t < nnz; ++t) {
        const size_t e = order[t];
        if (!o.row.empty() && o.row.back() == g.row[e] && o.col.back() == g.col[e]) {
            o.val.back() += g.val[e];
        } else {
            o.row.push_back(g.row[e]);
            o.col.push_back(g.col[e]);
            o.val.push_back(g.val[e]);
        }
    }
    g = std::move(o);
}

// lib/coarsening.py:135-192.  rr must be sorted; float32 arithmetic as numpy (NEP 50) evaluates it.
static double one_level(const int* rr, const int* cc, const float* vv, size_t nnz, const int64_t* rid,
                        const float* weights, int N, int* cluster_id) {
    std::vector<char> marked(N, 0);
    std::vector<int> rowstart(N, 0), rowlength(N, 0);
    if (nnz) {
        int oldval = rr[0];
        for (size_t ii = 0; ii < nnz; ++ii) {
            if (rr[ii] > oldval) {
                oldval = rr[ii];
                rowstart[rr[ii]] = (int)ii;
            }
            rowlength[rr[ii]]++;
        }
    }
    for (int i = 0; i < N; ++i) cluster_id[i] = 0;
    float total = 0.0f;  // python float += np.float32 -> float32 accumulation under NEP 50
    int clustercount = 0;
    for (int ii = 0; ii < N; ++ii) {
        const int tid = (int)rid[ii];
        if (marked[tid]) continue;
        float wmax = 0.0f;
        const int rs = rowstart[tid];
        marked[tid] = 1;
        int best = -1;
        for (int jj = 0; jj < rowlength[tid]; ++jj) {
            const int nid = cc[rs + jj];
            float tval;
            if (marked[nid]) {
                tval = 0.0f;
            } else {
                const float a = 1.0f / weights[tid];
                const float b = 1.0f / weights[nid];
                tval = vv[rs + jj] * (a + b);
            }
            if (tval > wmax) {
                wmax = tval;
                best = nid;
            }
        }
        cluster_id[tid] = clustercount;
        if (best > -1) {
            cluster_id[best] = clustercount;
            marked[best] = 1;
        }
        total += wmax;
        clustercount++;
    }
    return (double)total;
}

struct Hierarchy {
    int levels = 0;
    int n0 = 0;                                  // real nodes of level 0
    std::vector<Coo> graphs;                     // levels+1 graphs, UNPERMUTED, canonical
    std::vector<std::vector<int>> parents;       // levels arrays
    std::vector<std::vector<int>> perms;         // levels+1 orderings (new -> old), fake ids >= size
};

static void column_sums(const Coo& g, bool minus_diag, std::vector<float>& deg) {
    deg.assign(g.n, 0.0f);
    for (size_t e = 0; e < g.row.size(); ++e) {
        if (minus_diag && g.row[e] == g.col[e]) continue;
        deg[g.col[e]] += g.val[e];
    }
}

// lib/coarsening.py:194-241 (linear time instead of the O(N^2) np.where scan)
static void compute_perm(const std::vector<std::vector<int>>& parents, std::vector<std::vector<int>>& out) {
    std::vector<std::vector<int>> indices;
    if (parents.empty()) {
        out.clear();
        return;
    }
    const std::vector<int>& last = parents.back();
    const int M_last = *std::max_element(last.begin(), last.end()) + 1;
    std::vector<int> top(M_last);
    std::iota(top.begin(), top.end(), 0);
    indices.push_back(top);
    for (int l = (int)parents.size() - 1; l >= 0; --l) {
        const std::vector<int>& parent = parents[l];
        int pool = (int)parent.size();
        // children of every parent id, ascending
        const std::vector<int>& prev = indices.back();
        int maxid = 0;
        for (int v : prev) maxid = std::max(maxid, v);
        std::vector<int> c0(maxid + 1, -1), c1(maxid + 1, -1);
        for (int i = 0; i < (int)parent.size(); ++i) {
            const int pz = parent[i];
            if (pz > maxid) continue;
            if (c0[pz] < 0) c0[pz] = i;
            else c1[pz] = i;  // the reference asserts <= 2 children
        }
        std::vector<int> layer;
        layer.reserve(prev.size() * 2);
        for (int i : prev) {
            int a = c0[i], b = c1[i];
            if (a >= 0 && b >= 0) {
                layer.push_back(a);
                layer.push_back(b);
            } else if (a >= 0) {
                layer.push_back(a);
                layer.push_back(pool++);
            } else {
                layer.push_back(pool);
                layer.push_back(pool + 1);
                pool += 2;
            }
        }
        indices.push_back(layer);
    }
    out.assign(indices.rbegin(), indices.rend());
}

static void build_level0(const int32_t* adj, int n, int K, const double* pos, const float* nrm, Coo& g) {
    // utils.py:1753-1796
    g.n = n;
    const double sigma = 0.001;
    const double sig_den = 1.0 / (2 * sigma * sigma);
    for (int i = 0; i < n; ++i) {
        for (int k = 1; k < K; ++k) {
            const int j = adj[(size_t)i * K + k] - 1;
            if (j < 0) break;
            float dp = 0.f;  // np.sum over float32 products
            for (int t = 0; t < 3; ++t) dp += nrm[3 * i + t] * nrm[3 * j + t];
            double d2 = 0.0;
            for (int t = 0; t < 3; ++t) {
                const double df = pos[3 * j + t] - pos[3 * i + t];
                d2 += df * df;
            }
            const double nr = sqrt(d2);
            const double w = std::max((double)dp * exp(-(nr * nr) * sig_den), 0.001);
            g.row.push_back(i);
            g.col.push_back(j);
            g.val.push_back((float)w);
        }
    }
}

static void coarsen_graph(const Coo& g, const std::vector<int>& cid, Coo& out) {
    out = Coo();
    out.n = *std::max_element(cid.begin(), cid.end()) + 1;
    out.row.resize(g.row.size());
    out.col.resize(g.row.size());
    out.val = g.val;
    for (size_t e = 0; e < g.row.size(); ++e) {
        out.row[e] = cid[g.row[e]];
        out.col[e] = cid[g.col[e]];
    }
    canonicalize(out);
}

}  // namespace

struct fgc_hierarchy {
    Hierarchy h;
};

extern "C" int fgc_face_features(const float* V, int32_t nv, const uint32_t* F, int32_t nf, float* normals,
                                 double* centres) {
    FGC_CHECK_ARG(V && F && normals && centres && nv > 0 && nf > 0, "fgc_face_features: bad arguments");
    // bounding-box diagonal (utils.py:1271-1280); positions are divided by it without centring
    float mn[3] = {V[0], V[1], V[2]}, mx[3] = {V[0], V[1], V[2]};
    for (int i = 0; i < nv; ++i)
        for (int t = 0; t < 3; ++t) {
            mn[t] = std::min(mn[t], V[3 * i + t]);
            mx[t] = std::max(mx[t], V[3 * i + t]);
        }
    double diag2 = 0.0;
    for (int t = 0; t < 3; ++t) diag2 += pow((double)(mx[t] - mn[t]), 2);
    const float diag = (float)sqrt(diag2);  // numpy: float32 array / python float -> float32 division
    for (int f = 0; f < nf; ++f) {
        const uint32_t a = F[3 * f], b = F[3 * f + 1], c = F[3 * f + 2];
        FGC_CHECK_ARG((int)a < nv && (int)b < nv && (int)c < nv, "fgc_face_features: face %d references vertex >= %d",
                      f, nv);
        const float* p0 = V + 3 * a;
        const float* p1 = V + 3 * b;
        const float* p2 = V + 3 * c;
        float e1[3], e2[3];
        for (int t = 0; t < 3; ++t) {
            e1[t] = p1[t] - p0[t];
            e2[t] = p2[t] - p0[t];
        }
        float n[3];
        // np.cross on float32: separate multiply and subtract roundings (no fma)
        {
            volatile float m0 = e1[1] * e2[2], m1 = e1[2] * e2[1];
            n[0] = m0 - m1;
            volatile float m2 = e1[2] * e2[0], m3 = e1[0] * e2[2];
            n[1] = m2 - m3;
            volatile float m4 = e1[0] * e2[1], m5 = e1[1] * e2[0];
            n[2] = m4 - m5;
        }
        for (int it = 0; it < 2; ++it) {  // utils.normalize = normalizeOnce twice, eps 1e-8 added to the norm
            volatile float s0 = n[0] * n[0], s1 = n[1] * n[1], s2 = n[2] * n[2];
            volatile float ss = s0 + s1;
            ss = ss + s2;
            const float norm = sqrtf(ss) + 0.00000001f;
            const float inv = 1.0f / norm;
            for (int t = 0; t < 3; ++t) n[t] = n[t] * inv;
        }
        for (int t = 0; t < 3; ++t) normals[3 * f + t] = n[t];
        for (int t = 0; t < 3; ++t) {
            const float q0 = p0[t] / diag, q1 = p1[t] / diag, q2 = p2[t] / diag;
            volatile float s = q0 + q1;
            s = s + q2;
            centres[3 * f + t] = (double)(s / 3.0f);
        }
    }
    return FGC_OK;
}

// utils.py:243-295 without its hidden limits (vertex ids < 0.6F+200, valence <= 69): per vertex in ascending id,
// every pair (f1 < f2 in face order) of incident faces is appended to both rows unless the row is full.
extern "C" int fgc_faces_large_adj(const uint32_t* F, int32_t nf, int32_t nv, int32_t K, int32_t* adj,
                                   int64_t* unregistered) {
    FGC_CHECK_ARG(F && adj && nf > 0 && nv > 0 && K > 1 && K < 128, "fgc_faces_large_adj: bad arguments");
    std::vector<int> cnt(nv + 1, 0);
    for (int f = 0; f < nf; ++f)
        for (int t = 0; t < 3; ++t) {
            FGC_CHECK_ARG((int)F[3 * f + t] < nv, "fgc_faces_large_adj: face %d references vertex >= %d", f, nv);
            cnt[F[3 * f + t] + 1]++;
        }
    for (int v = 0; v < nv; ++v) cnt[v + 1] += cnt[v];
    std::vector<int> vf(cnt[nv]);
    std::vector<int> fill(cnt.begin(), cnt.end() - 1);
    for (int f = 0; f < nf; ++f)
        for (int t = 0; t < 3; ++t) vf[fill[F[3 * f + t]]++] = f;
    memset(adj, 0, sizeof(int32_t) * (size_t)nf * K);
    std::vector<int> find(nf, 1);
    for (int f = 0; f < nf; ++f) adj[(size_t)f * K] = f + 1;
    int64_t unreg = 0;
    for (int v = 0; v < nv; ++v) {
        const int b = cnt[v], e = cnt[v + 1];
        for (int a1 = b; a1 < e; ++a1)
            for (int a2 = a1 + 1; a2 < e; ++a2) {
                const int f1 = vf[a1], f2 = vf[a2];
                if (find[f1] == K) unreg++;
                else adj[(size_t)f1 * K + find[f1]++] = f2 + 1;
                if (find[f2] == K) unreg++;
                else adj[(size_t)f2 * K + find[f2]++] = f1 + 1;
            }
    }
    if (unregistered) *unregistered = unreg;
    return FGC_OK;
}

// utils.py:1508-1696.  Two FIFO queues (nodes to expand, already-covered "border" nodes), nodes renumbered in order of
// discovery; growth stops at nodes_num nodes (or min_patch_size through the border queue when the region is exhausted).
extern "C" int fgc_graph_patch(const int32_t* adj, int32_t n, int32_t K, int32_t nodes_num, int32_t seed,
                               const int8_t* mask, int32_t min_patch_size, int32_t* out, int32_t* old_of,
                               int32_t* patch_n, int32_t* next_seed) {
    FGC_CHECK_ARG(adj && mask && out && old_of && patch_n && next_seed, "fgc_graph_patch: null pointer");
    FGC_CHECK_ARG(n > 0 && K > 1 && nodes_num > 0 && seed >= 0 && seed < n, "fgc_graph_patch: n=%d K=%d nodes_num=%d seed=%d",
                  n, K, nodes_num, seed);
    FGC_CHECK_ARG(min_patch_size <= nodes_num, "fgc_graph_patch: min_patch_size=%d above nodes_num=%d", min_patch_size,
                  nodes_num);
    const int cap = nodes_num + K;
    for (size_t t = 0; t < (size_t)cap * K; ++t) out[t] = 0;     // one-indexed output: 0 = empty slot
    std::vector<int> new_of(n, -1);
    std::vector<int> q, border;
    size_t qh = 0, bh = 0;
    int count = 0;
    auto add = [&](int v) {
        new_of[v] = count;
        old_of[count] = v;
        ++count;
    };
    auto nb_of = [&](int cur, int s) { return adj[(size_t)cur * K + s] - 1; };
    // a node may be discovered while count < nodes_num only at loop entry, so at most K - 1 more fit: cap rows suffice
    auto expand = [&](int cur, bool masked_to_border) {
        const int r = new_of[cur];
        out[(size_t)r * K] = r + 1;
        for (int s = 1; s < K; ++s) {
            const int nb = nb_of(cur, s);
            if (nb == -1) break;
            if (new_of[nb] == -1) {
                add(nb);
                if (masked_to_border && mask[nb] == 1) border.push_back(nb);
                else q.push_back(nb);
            }
            out[(size_t)r * K + s] = new_of[nb] + 1;
        }
    };
    add(seed);
    q.push_back(seed);
    while (count < nodes_num && qh < q.size()) expand(q[qh++], true);
    int nxt = -1;
    if (count < min_patch_size) {
        while (count < min_patch_size && bh < border.size()) expand(border[bh++], false);
        while (count < min_patch_size && qh < q.size()) expand(q[qh++], false);
    }
    FGC_CHECK_ARG(count <= cap, "fgc_graph_patch: patch overflow (%d > %d)", count, cap);
    auto finish = [&](std::vector<int>& qq, size_t& head) {
        while (head < qq.size()) {
            const int cur = qq[head++];
            const int r = new_of[cur];
            out[(size_t)r * K] = r + 1;
            int c = 1;
            for (int s = 1; s < K; ++s) {
                const int nb = nb_of(cur, s);
                if (nb == -1) break;
                if (new_of[nb] == -1) {
                    if (mask[nb] == 0) nxt = nb;
                    continue;
                }
                out[(size_t)r * K + c++] = new_of[nb] + 1;
            }
        }
    };
    finish(q, qh);
    finish(border, bh);
    *patch_n = count;
    *next_seed = nxt;
    return FGC_OK;
}

// utils.py:1298-1410: breadth-first mesh patch (faces, their vertices and the K-list among the patch's faces)
extern "C" int fgc_mesh_patch(const float* V, int32_t nv, const int32_t* F, int32_t nf, const int32_t* adj, int32_t K,
                              int32_t face_num, int32_t seed, float* v_out, int32_t v_cap, int32_t* f_out,
                              int32_t* adj_out, int32_t* v_old, int32_t* f_old, int32_t* n_v, int32_t* n_f) {
    FGC_CHECK_ARG(V && F && adj && v_out && f_out && adj_out && v_old && f_old && n_v && n_f, "fgc_mesh_patch: null pointer");
    FGC_CHECK_ARG(nv > 0 && nf > 0 && K > 1 && face_num > 0 && seed >= 0 && seed < nf && v_cap > 0,
                  "fgc_mesh_patch: nv=%d nf=%d K=%d face_num=%d seed=%d v_cap=%d", nv, nf, K, face_num, seed, v_cap);
    const int f_cap = face_num + K;
    for (size_t t = 0; t < (size_t)f_cap * K; ++t) adj_out[t] = 0;   // one-indexed K-list: 0 = empty slot
    std::vector<int> v_new(nv, -1), f_new(nf, -1);
    std::vector<int> q;
    size_t qh = 0;
    int vc = 0, fc = 0;
    bool overflow = false;
    auto add_vertex = [&](int v) {
        if (v_new[v] != -1) return;
        if (vc >= v_cap) {
            overflow = true;
            return;
        }
        v_new[v] = vc;
        v_old[vc] = v;
        for (int t = 0; t < 3; ++t) v_out[3 * (size_t)vc + t] = V[3 * (size_t)v + t];
        ++vc;
    };
    auto add_face = [&](int f) {
        for (int t = 0; t < 3; ++t) add_vertex(F[3 * (size_t)f + t]);
        if (overflow) return;
        for (int t = 0; t < 3; ++t) f_out[3 * (size_t)fc + t] = v_new[F[3 * (size_t)f + t]];
        f_new[f] = fc;
        f_old[fc] = f;
        ++fc;
    };
    add_face(seed);
    q.push_back(seed);
    // a face is only added while fc < face_num at loop entry, then at most K - 1 more: face_num + K rows suffice
    while (fc < face_num && qh < q.size() && !overflow) {
        const int cur = q[qh++];
        const int r = f_new[cur];
        adj_out[(size_t)r * K] = r + 1;
        for (int s = 1; s < K; ++s) {
            const int nb = adj[(size_t)cur * K + s] - 1;
            if (nb == -1) break;
            if (f_new[nb] == -1) {
                add_face(nb);
                if (overflow) break;
                q.push_back(nb);
            }
            adj_out[(size_t)r * K + s] = f_new[nb] + 1;
        }
    }
    FGC_CHECK_ARG(!overflow, "fgc_mesh_patch: more than %d vertices in a patch of %d faces (the reference raises IndexError)",
                  v_cap, face_num);
    // faces still queued when growth stopped: their rows list the neighbours that made it into the patch, compacted
    while (qh < q.size()) {
        const int cur = q[qh++];
        const int r = f_new[cur];
        adj_out[(size_t)r * K] = r + 1;
        int c = 1;
        for (int s = 1; s < K; ++s) {
            const int nb = adj[(size_t)cur * K + s] - 1;
            if (nb == -1) break;
            if (f_new[nb] == -1) continue;
            adj_out[(size_t)r * K + c++] = f_new[nb] + 1;
        }
    }
    *n_v = vc;
    *n_f = fc;
    return FGC_OK;
}

// utils.py:370-395
extern "C" int fgc_vertices_faces(const int32_t* F, int32_t nf, int32_t nv, int32_t k_v, int32_t* out) {
    FGC_CHECK_ARG(F && out && nf > 0 && nv > 0 && k_v > 0, "fgc_vertices_faces: bad arguments");
    for (size_t t = 0; t < (size_t)nv * k_v; ++t) out[t] = -1;
    std::vector<int> cnt(nv, 0);
    for (int f = 0; f < nf; ++f) {
        if (F[3 * (size_t)f] == -1) continue;
        for (int t = 0; t < 3; ++t) {
            const int v = F[3 * (size_t)f + t];
            FGC_CHECK_ARG(v >= 0 && v < nv, "fgc_vertices_faces: face %d references vertex %d", f, v);
            FGC_CHECK_ARG(cnt[v] < k_v, "fgc_vertices_faces: vertex %d is in more than %d faces", v, k_v);
            out[(size_t)v * k_v + cnt[v]++] = f;
        }
    }
    return FGC_OK;
}

// utils.py:91-183.  Faces in order; (v1,v2) and (v1,v3) are searched among v1's edges, (v2,v3) among v2's; a found
// edge takes this face as its second face, a missing one is created in the order 12, 13, 23.
extern "C" int fgc_edge_map(const uint32_t* F, int32_t nf, int32_t nv, int32_t max_edges, int32_t* e_map,
                            int32_t* n_edges, int32_t* v_e_map) {
    FGC_CHECK_ARG(F && e_map && n_edges && v_e_map && nf > 0 && nv > 0 && max_edges > 0, "fgc_edge_map: bad arguments");
    for (size_t t = 0; t < (size_t)nf * 12; ++t) e_map[t] = -1;
    for (size_t t = 0; t < (size_t)nv * max_edges; ++t) v_e_map[t] = -1;
    std::vector<int> cnt(nv, 0);
    int eind = 0;
    auto has = [&](int ce, int v) { return e_map[4 * (size_t)ce] == v || e_map[4 * (size_t)ce + 1] == v; };
    for (int f = 0; f < nf; ++f) {
        const int v[3] = {(int)F[3 * f], (int)F[3 * f + 1], (int)F[3 * f + 2]};
        for (int t = 0; t < 3; ++t)
            FGC_CHECK_ARG(v[t] >= 0 && v[t] < nv, "fgc_edge_map: face %d references vertex %d >= %d", f, v[t], nv);
        bool e12 = false, e13 = false, e23 = false;
        for (int ne = 0; ne < cnt[v[0]]; ++ne) {
            const int ce = v_e_map[(size_t)v[0] * max_edges + ne];
            if (has(ce, v[1])) { e12 = true; e_map[4 * (size_t)ce + 3] = f; }
            if (has(ce, v[2])) { e13 = true; e_map[4 * (size_t)ce + 3] = f; }
        }
        for (int ne = 0; ne < cnt[v[1]]; ++ne) {
            const int ce = v_e_map[(size_t)v[1] * max_edges + ne];
            if (has(ce, v[2])) { e23 = true; e_map[4 * (size_t)ce + 3] = f; }
        }
        const int pairs[3][2] = {{v[0], v[1]}, {v[0], v[2]}, {v[1], v[2]}};
        const bool found[3] = {e12, e13, e23};
        for (int k = 0; k < 3; ++k) {
            if (found[k]) continue;
            const int a = pairs[k][0], b = pairs[k][1];
            FGC_CHECK_ARG(cnt[a] < max_edges && cnt[b] < max_edges,
                          "fgc_edge_map: vertex %d has more than %d edges (utils.py:103 sizes the table)",
                          cnt[a] < max_edges ? b : a, max_edges);
            e_map[4 * (size_t)eind] = a;
            e_map[4 * (size_t)eind + 1] = b;
            e_map[4 * (size_t)eind + 2] = f;
            v_e_map[(size_t)a * max_edges + cnt[a]++] = eind;
            v_e_map[(size_t)b * max_edges + cnt[b]++] = eind;
            ++eind;
        }
    }
    *n_edges = eind;
    return FGC_OK;
}

extern "C" int fgc_metis_one_level(const int32_t* rr, const int32_t* cc, const float* vv, int64_t nnz,
                                   const int64_t* rid, const float* weights, int32_t N, int32_t* cluster_id,
                                   double* total_assoc) {
    FGC_CHECK_ARG(rr && cc && vv && rid && weights && cluster_id && nnz >= 0 && N > 0, "fgc_metis_one_level: bad arguments");
    const double t = one_level(rr, cc, vv, (size_t)nnz, rid, weights, N, cluster_id);
    if (total_assoc) *total_assoc = t;
    return FGC_OK;
}

extern "C" int fgc_hierarchy_build(const int32_t* adj, int32_t n, int32_t K, const double* pos, const float* normals,
                                   int32_t levels, uint64_t seed, const int32_t* const* parents_in,
                                   const int32_t* parents_len, fgc_hierarchy** out) {
    FGC_CHECK_ARG(adj && pos && normals && out && n > 0 && K > 1 && levels >= 1 && levels <= 8,
                  "fgc_hierarchy_build: bad arguments");
    fgc_hierarchy* H = new fgc_hierarchy();
    Hierarchy& h = H->h;
    h.levels = levels;
    h.n0 = n;
    h.graphs.resize(levels + 1);
    build_level0(adj, n, K, pos, normals, h.graphs[0]);
    canonicalize(h.graphs[0]);
    std::mt19937_64 rng(seed);
    std::vector<int64_t> rid;
    for (int l = 0; l < levels; ++l) {
        const Coo& g = h.graphs[l];
        std::vector<int> cid(g.n, 0);
        if (parents_in) {
            if (parents_len[l] != g.n) {
                fgc::set_error("fgc_hierarchy_build: recorded parents[%d] has %d entries, graph has %d nodes", l,
                               parents_len[l], g.n);
                delete H;
                return FGC_EINVAL;
            }
            cid.assign(parents_in[l], parents_in[l] + g.n);
        } else {
            // lib/coarsening.py:52-96: graclus weights = weighted degree; best of 3 visiting orders
            std::vector<float> deg;
            column_sums(g, l == 0, deg);
            if (l == 0) {
                rid.resize(g.n);
                std::iota(rid.begin(), rid.end(), (int64_t)0);
                std::shuffle(rid.begin(), rid.end(), rng);
            }
            // every node needs a row for the scan; isolated nodes become singletons (validated, not inferred
            // from the last row as coarsening.py:138 does)
            double best = 0.0;
            bool have = false;
            std::vector<int> cur(g.n);
            for (int trial = 0; trial < 3; ++trial) {
                const double assoc = one_level(g.row.data(), g.col.data(), g.val.data(), g.row.size(), rid.data(),
                                               deg.data(), g.n, cur.data());
                if (!have || assoc > best) {  // coarsening.py:92-94 (strictly better replaces)
                    cid = cur;
                    best = assoc;
                    have = true;
                }
                rid.resize(g.n);
                std::iota(rid.begin(), rid.end(), (int64_t)0);
                std::shuffle(rid.begin(), rid.end(), rng);
            }
        }
        h.parents.push_back(cid);
        coarsen_graph(g, cid, h.graphs[l + 1]);
        if (!parents_in) {
            // next level's first visiting order: ascending weighted degree (coarsening.py:128-129), stable
            std::vector<float> ss;
            column_sums(h.graphs[l + 1], false, ss);
            rid.resize(ss.size());
            std::iota(rid.begin(), rid.end(), (int64_t)0);
            std::stable_sort(rid.begin(), rid.end(), [&](int64_t a, int64_t b) { return ss[a] < ss[b]; });
        }
    }
    if (!parents_in) {
        // Spatial order of the coarsest clusters (Morton code of their centroid).  The reference leaves that order
        // to the pairing ("Order of last layer is random", coarsening.py:200); every finer level inherits it through
        // the binary tree, so consecutive 16-node blocks become spatial neighbours: the gather of a tile then hits
        // rows its neighbours just fetched (L1/L2), and a contiguous split of the node range is a compact patch
        // (what facet sharding across GPUs needs).  Results are identical up to this relabelling.
        const int L = levels;
        std::vector<int> top(n);
        for (int i = 0; i < n; ++i) {
            int c = i;
            for (int l = 0; l < L; ++l) c = h.parents[l][c];
            top[i] = c;
        }
        const int nc = h.graphs[L].n;
        std::vector<double> cen(3 * (size_t)nc, 0.0);
        std::vector<int> cnt(nc, 0);
        double mn[3] = {1e300, 1e300, 1e300}, mx[3] = {-1e300, -1e300, -1e300};
        for (int i = 0; i < n; ++i) {
            for (int t = 0; t < 3; ++t) {
                cen[3 * (size_t)top[i] + t] += pos[3 * (size_t)i + t];
                mn[t] = std::min(mn[t], pos[3 * (size_t)i + t]);
                mx[t] = std::max(mx[t], pos[3 * (size_t)i + t]);
            }
            cnt[top[i]]++;
        }
        std::vector<uint64_t> code(nc);
        for (int c = 0; c < nc; ++c) {
            uint64_t m = 0;
            uint32_t q[3];
            for (int t = 0; t < 3; ++t) {
                const double v = cnt[c] ? cen[3 * (size_t)c + t] / cnt[c] : 0.0;
                const double u = mx[t] > mn[t] ? (v - mn[t]) / (mx[t] - mn[t]) : 0.0;
                q[t] = (uint32_t)std::min(1023.0, std::max(0.0, u * 1023.0));
            }
            for (int b = 9; b >= 0; --b)
                for (int t = 0; t < 3; ++t) m = (m << 1) | ((q[t] >> b) & 1u);
            code[c] = m;
        }
        std::vector<int> order(nc);
        std::iota(order.begin(), order.end(), 0);
        std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return code[a] < code[b]; });
        std::vector<int> rank(nc);
        for (int r = 0; r < nc; ++r) rank[order[r]] = r;
        for (auto& v : h.parents[L - 1]) v = rank[v];
        coarsen_graph(h.graphs[L - 1], h.parents[L - 1], h.graphs[L]);
    }
    compute_perm(h.parents, h.perms);
    *out = H;
    return FGC_OK;
}

extern "C" void fgc_hierarchy_free(fgc_hierarchy* H) { delete H; }

extern "C" int32_t fgc_hierarchy_size(const fgc_hierarchy* H, int32_t level) {
    if (!H || level < 0 || level > H->h.levels) return -1;
    return (int32_t)H->h.perms[level].size();
}
extern "C" int32_t fgc_hierarchy_real_size(const fgc_hierarchy* H, int32_t level) {
    if (!H || level < 0 || level > H->h.levels) return -1;
    return (int32_t)H->h.graphs[level].n;
}

extern "C" int fgc_hierarchy_new_to_old(const fgc_hierarchy* H, int32_t level, int32_t* out) {
    FGC_CHECK_ARG(H && out && level >= 0 && level <= H->h.levels, "fgc_hierarchy_new_to_old: bad arguments");
    const std::vector<int>& p = H->h.perms[level];
    std::copy(p.begin(), p.end(), out);
    return FGC_OK;
}

extern "C" int fgc_hierarchy_parents(const fgc_hierarchy* H, int32_t level, int32_t* out) {
    FGC_CHECK_ARG(H && out && level >= 0 && level < H->h.levels, "fgc_hierarchy_parents: bad arguments");
    const std::vector<int>& p = H->h.parents[level];
    std::copy(p.begin(), p.end(), out);
    return FGC_OK;
}

// coarsen() post-processing (coarsening.py:13-26: drop self loops, permute, csr) + sparseToList (utils.py:1799-1827)
extern "C" int fgc_hierarchy_klist(const fgc_hierarchy* H, int32_t level, int32_t K, int32_t* adj, int32_t* saturated) {
    FGC_CHECK_ARG(H && adj && level >= 0 && level <= H->h.levels && K > 1, "fgc_hierarchy_klist: bad arguments");
    const Hierarchy& h = H->h;
    const Coo& g = h.graphs[level];
    const std::vector<int>& idx = h.perms[level];  // new -> old
    const int Mnew = (int)idx.size();
    std::vector<int> inv(Mnew, -1);                // old -> new (np.argsort(indices))
    for (int i = 0; i < Mnew; ++i) inv[idx[i]] = i;
    // CSR row starts of the canonical graph
    std::vector<int> rs(g.n + 1, 0);
    for (size_t e = 0; e < g.row.size(); ++e) rs[g.row[e] + 1]++;
    for (int i = 0; i < g.n; ++i) rs[i + 1] += rs[i];
    memset(adj, 0, sizeof(int32_t) * (size_t)Mnew * K);
    int sat = 0;
    std::vector<int> nb;
    for (int i = 0; i < Mnew; ++i) {
        int32_t* row = adj + (size_t)i * K;
        row[0] = i + 1;
        const int old = idx[i];
        if (old >= g.n) continue;  // fake node: isolated
        nb.clear();
        for (int e = rs[old]; e < rs[old + 1]; ++e) {
            if (g.col[e] == old) continue;        // setdiag(0) + eliminate_zeros
            if (g.val[e] == 0.0f) continue;
            nb.push_back(inv[g.col[e]]);
        }
        std::sort(nb.begin(), nb.end());
        int cur = 1;
        for (int j : nb) {
            if (cur == K) {
                sat = 1;
            } else {
                row[cur++] = j + 1;
            }
        }
    }
    if (saturated) *saturated = sat;
    return FGC_OK;
}
