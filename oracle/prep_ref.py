"""ORACLE (test infrastructure only): CPU restatement of the reference's mesh preprocessing.

Plain numpy / Python loops following the reference line by line; meant for meshes of a few thousand faces.
Pinned against tests/golden/prep_*.npz (outputs of the reference source itself) by tests/test_oracle_prep.py.

ref: utils.py:26-35,63-68,243-295,1264-1294,1753-1835; lib/coarsening.py:135-241,269-296; dataClasses.py:172-233
"""
import math

import numpy as np
import scipy.sparse


def normalize_once(a):
    """ref: utils.py:26-31"""
    norms = np.sqrt((a * a).sum(1))[:, np.newaxis] + 0.00000001
    return a * (1 / norms)


def face_normals(verts, faces):
    """ref: utils.py:63-68 (normalize = normalizeOnce twice, utils.py:33-35)"""
    T = verts[faces]
    N = np.cross(T[::, 1] - T[::, 0], T[::, 2] - T[::, 0])
    return normalize_once(normalize_once(N))


def barycentres(vl, fl):
    """ref: utils.py:1264-1294 (normalize=True): divide by the bbox diagonal, no centring"""
    mn, mx = vl.min(0), vl.max(0)
    diag = math.sqrt(sum(math.pow(float(mx[t] - mn[t]), 2) for t in range(3)))
    vl = vl / diag
    out = np.empty([fl.shape[0], 3])
    for f in range(fl.shape[0]):
        out[f, :] = (vl[fl[f, 0], :] + vl[fl[f, 1], :] + vl[fl[f, 2], :]) / 3
    return out


def faces_large_adj(faces, K):
    """ref: utils.py:243-295.  Per vertex (ascending id) every pair of incident faces is appended to both rows."""
    fnum = faces.shape[0]
    fadj = np.zeros([fnum, K], dtype=np.int32)
    find = np.ones([fnum], dtype=np.int64)
    incident = {}
    for f in range(fnum):
        for t in range(3):
            incident.setdefault(int(faces[f, t]), []).append(f)
    fadj[:, 0] = np.arange(fnum) + 1
    for v in sorted(incident):
        fl = incident[v]
        for a in range(len(fl)):
            for b in range(a + 1, len(fl)):
                f1, f2 = fl[a], fl[b]
                if find[f1] < K:
                    fadj[f1, find[f1]] = f2 + 1
                    find[f1] += 1
                if find[f2] < K:
                    fadj[f2, find[f2]] = f1 + 1
                    find[f2] += 1
    return fadj


def metis_one_level(rr, cc, vv, rid, weights):
    """ref: lib/coarsening.py:135-192 (float32 arithmetic as numpy >= 2 evaluates it)"""
    nnz = rr.shape[0]
    N = rr[nnz - 1] + 1
    marked = np.zeros(N, bool)
    rowstart = np.zeros(N, np.int32)
    rowlength = np.zeros(N, np.int32)
    cluster_id = np.zeros(N, np.int32)
    oldval = rr[0]
    for ii in range(nnz):
        if rr[ii] > oldval:
            oldval = rr[ii]
            rowstart[rr[ii]] = ii
        rowlength[rr[ii]] += 1
    clustercount = 0
    total = np.float32(0.0)
    for ii in range(N):
        tid = rid[ii]
        if marked[tid]:
            continue
        wmax = np.float32(0.0)
        rs = rowstart[tid]
        marked[tid] = True
        best = -1
        for jj in range(rowlength[tid]):
            nid = cc[rs + jj]
            if marked[nid]:
                tval = np.float32(0.0)
            else:
                tval = np.float32(vv[rs + jj]) * (np.float32(1.0) / np.float32(weights[tid]) +
                                                   np.float32(1.0) / np.float32(weights[nid]))
            if tval > wmax:
                wmax = tval
                best = nid
        cluster_id[tid] = clustercount
        if best > -1:
            cluster_id[best] = clustercount
            marked[best] = True
        total = np.float32(total + wmax)
        clustercount += 1
    return cluster_id, float(total)


def compute_perm(parents):
    """ref: lib/coarsening.py:194-241"""
    indices = []
    if len(parents) > 0:
        indices.append(list(range(max(parents[-1]) + 1)))
    for parent in parents[::-1]:
        pool = len(parent)
        layer = []
        for i in indices[-1]:
            node = list(np.where(parent == i)[0])
            assert 0 <= len(node) <= 2
            if len(node) == 1:
                node.append(pool)
                pool += 1
            elif len(node) == 0:
                node += [pool, pool + 1]
                pool += 2
            layer.extend(node)
        indices.append(layer)
    return indices[::-1]


def coarsened_klists(adj, parents, K=23, keep=(0, 2, 4)):
    """Structure of graph levels given the cluster assignments, then perm_adjacency + sparseToList.
    ref: lib/coarsening.py:5-31,100-112,269-296; utils.py:1799-1827.  Returns (klists, newToOld)."""
    n = adj.shape[0]
    rows, cols = [], []
    for i in range(n):
        for k in range(1, adj.shape[1]):
            j = adj[i, k] - 1
            if j < 0:
                break
            rows.append(i)
            cols.append(j)
    graphs = [scipy.sparse.csr_matrix((np.ones(len(rows), np.float32), (rows, cols)), shape=(n, n))]
    for cid in parents:
        g = graphs[-1].tocoo()
        nn = int(cid.max()) + 1
        graphs.append(scipy.sparse.csr_matrix((g.data, (cid[g.row], cid[g.col])), shape=(nn, nn)))
    perms = compute_perm(parents)
    klists = []
    for lvl in keep:
        A = graphs[lvl].tocoo()
        A.setdiag(0)
        idx = perms[lvl]
        M, Mnew = A.shape[0], len(idx)
        if Mnew > M:
            A = scipy.sparse.vstack([A, scipy.sparse.coo_matrix((Mnew - M, M), dtype=np.float32)])
            A = scipy.sparse.hstack([A, scipy.sparse.coo_matrix((Mnew, Mnew - M), dtype=np.float32)]).tocoo()
        perm = np.argsort(idx)
        A = scipy.sparse.coo_matrix((A.data, (perm[A.row], perm[A.col])), shape=(Mnew, Mnew)).tocsr()
        A.eliminate_zeros()
        out = np.zeros((Mnew, K), dtype=np.int32)
        out[:, 0] = np.arange(Mnew) + 1
        cur = np.ones(Mnew, dtype=np.int64)
        cx = A.tocoo()
        for i, j in zip(cx.row, cx.col):
            if i != j and cur[i] < K:
                out[i, cur[i]] = j + 1
                cur[i] += 1
        klists.append(out)
    return klists, np.asarray(perms[0])


def inv_perm(perm):
    """ref: utils.py:1830-1835"""
    inverse = [0] * max(len(perm), int(np.amax(perm)) + 1)
    for i, p in enumerate(perm):
        inverse[p] = i
    return np.array(inverse)


def edge_map(faces, max_edges=20):
    """ref: utils.py:91-183 (getEdgeMap).  Returns (e_map [E,4] = [v1, v2, f1, f2 or -1], v_e_map [V, max_edges]
    edge ids per vertex, -1 padded).  Faces are visited in order; per face the edges (v1,v2), (v1,v3) are looked up
    among v1's edges and (v2,v3) among v2's; a found edge gets the face as its second face (overwriting a previous
    one on non-manifold input, as the reference does), a missing edge is created in the order 12, 13, 23."""
    faces = np.asarray(faces).astype(np.int64)
    fnum = faces.shape[0]
    e_map = -np.ones((fnum * 3, 4), dtype=np.int32)
    vnum = int(faces.max()) + 1
    v_e_map = -np.ones((vnum, max_edges), dtype=np.int32)
    cnt = np.zeros(vnum, dtype=np.int64)
    eind = 0
    for f in range(fnum):
        v1, v2, v3 = (int(t) for t in faces[f])
        found = {"12": False, "13": False, "23": False}
        for ne in range(cnt[v1]):
            ce = v_e_map[v1, ne]
            if e_map[ce, 0] == v2 or e_map[ce, 1] == v2:
                found["12"] = True
                e_map[ce, 3] = f
            if e_map[ce, 0] == v3 or e_map[ce, 1] == v3:
                found["13"] = True
                e_map[ce, 3] = f
        for ne in range(cnt[v2]):
            ce = v_e_map[v2, ne]
            if e_map[ce, 0] == v3 or e_map[ce, 1] == v3:
                found["23"] = True
                e_map[ce, 3] = f
        for key, (a, b) in (("12", (v1, v2)), ("13", (v1, v3)), ("23", (v2, v3))):
            if found[key]:
                continue
            e_map[eind, 0], e_map[eind, 1], e_map[eind, 2] = a, b, f
            for v in (a, b):
                if cnt[v] >= max_edges:
                    raise IndexError("vertex %d has more than %d edges (utils.py:103 sizes the table)" % (v, max_edges))
                v_e_map[v, cnt[v]] = eind
                cnt[v] += 1
            eind += 1
    return e_map[:eind], v_e_map


def graph_patch_wmask(adj, nodes_num, seed, mask, min_patch_size):
    """ref: utils.py:1508-1696 (getGraphPatch_wMask).  Breadth-first patch of a one-indexed K-list `adj` grown from
    node `seed` until it holds `nodes_num` nodes.  `mask[j] == 1` marks nodes already covered by earlier patches:
    they are added to the patch (context) but their own neighbours are only expanded while the patch is smaller than
    `min_patch_size`.  Nodes are renumbered in order of discovery.  Rows of nodes that were expanded keep their
    neighbour slots (slot 0 = self); rows of nodes still queued when growth stops are compacted to the neighbours that
    made it into the patch.  Returns (patch K-list one-indexed [n, K], old index of every patch node [n], next seed:
    an uncovered node seen just outside the patch, or -1)."""
    from collections import deque
    adj = np.asarray(adj).astype(np.int64) - 1
    N, K = adj.shape
    out = -np.ones((nodes_num + K, K), dtype=np.int64)
    new_of = -np.ones(N, dtype=np.int64)
    old_of = -np.ones(N, dtype=np.int64)
    count = [0]

    def add(n):
        new_of[n] = count[0]
        old_of[count[0]] = n
        count[0] += 1

    q, border = deque([seed]), deque()
    add(seed)

    def expand(cur, masked_to_border):
        r = new_of[cur]
        out[r, 0] = r
        for s in range(1, K):
            nb = adj[cur, s]
            if nb == -1:
                break
            if new_of[nb] == -1:
                add(nb)
                (border if (masked_to_border and mask[nb] == 1) else q).append(nb)
            out[r, s] = new_of[nb]

    while count[0] < nodes_num and q:
        expand(q.popleft(), True)
    next_seed = -1
    if count[0] < min_patch_size:
        while count[0] < min_patch_size and border:
            expand(border.popleft(), False)
        while count[0] < min_patch_size and q:
            expand(q.popleft(), False)
    for queue_ in (q, border):
        while queue_:
            cur = queue_.popleft()
            r = new_of[cur]
            out[r, 0] = r
            c = 1
            for s in range(1, K):
                nb = adj[cur, s]
                if nb == -1:
                    break
                if new_of[nb] == -1:
                    if mask[nb] == 0:
                        next_seed = nb
                    continue
                out[r, c] = new_of[nb]
                c += 1
    n = count[0]
    return (out[:n] + 1), old_of[:n].copy(), int(next_seed)


def vertices_faces(faces, k_v, vnum=0):
    """ref: utils.py:370-395 (getVerticesFaces): for every vertex the faces (row indices of `faces`) that contain it,
    in face order, -1 padded to k_v; rows whose first vertex is -1 (fake faces) are skipped."""
    faces = np.asarray(faces).astype(np.int64)
    if vnum == 0:
        vnum = int(faces.max()) + 1
    v_f = -np.ones((vnum, k_v), dtype=np.int32)
    cnt = np.zeros(vnum, dtype=np.int64)
    for f in range(faces.shape[0]):
        if faces[f, 0] == -1:
            continue
        for t in range(3):
            v = faces[f, t]
            if cnt[v] >= k_v:
                raise IndexError("vertex %d is in more than %d faces" % (v, k_v))
            v_f[v, cnt[v]] = f
            cnt[v] += 1
    return v_f


def normalize_point_sets(vl1, vl2):
    """ref: utils.py:2077-2104: both point sets divided by the bounding-box diagonal of their union (not centred)."""
    lo = np.minimum(vl1.min(0), vl2.min(0))
    hi = np.maximum(vl1.max(0), vl2.max(0))
    diag = math.sqrt(float(((hi - lo) ** 2).sum()))
    return vl1 / diag, vl2 / diag



def mesh_patch(vIn, fIn, fAdjIn, faceNum, seed):
    """getMeshPatch (utils.py:1298-1410) restated: breadth-first growth from face `seed`; returns
    (vOut, fOut, fAdjOut one-indexed, vOldInd, fOldInd)."""
    import collections
    vIn, fIn, fAdjIn = np.asarray(vIn), np.asarray(fIn), np.asarray(fAdjIn)
    K = fAdjIn.shape[1]
    v_new, f_new = {}, {}
    v_old, f_old, f_out = [], [], []
    adj_out = {}

    def add_face(f):
        for v in fIn[f]:
            v = int(v)
            if v not in v_new:
                v_new[v] = len(v_old)
                v_old.append(v)
        f_new[f] = len(f_old)
        f_old.append(f)
        f_out.append([v_new[int(v)] for v in fIn[f]])

    q = collections.deque([int(seed)])
    add_face(int(seed))
    while len(f_old) < faceNum and q:                      # utils.py:1349-1376
        cur = q.popleft()
        row = [0] * K
        row[0] = f_new[cur] + 1
        for s in range(1, K):
            nb = int(fAdjIn[cur, s]) - 1
            if nb == -1:
                break
            if nb not in f_new:
                add_face(nb)
                q.append(nb)
            row[s] = f_new[nb] + 1
        adj_out[f_new[cur]] = row
    while q:                                               # utils.py:1381-1402: compacted rows of the faces still queued
        cur = q.popleft()
        row = [0] * K
        row[0] = f_new[cur] + 1
        c = 1
        for s in range(1, K):
            nb = int(fAdjIn[cur, s]) - 1
            if nb == -1:
                break
            if nb not in f_new:
                continue
            row[c] = f_new[nb] + 1
            c += 1
        adj_out[f_new[cur]] = row
    nf = len(f_old)
    return (vIn[v_old].astype(np.float32), np.asarray(f_out, dtype=np.int64),
            np.asarray([adj_out.get(r, [0] * K) for r in range(nf)], dtype=np.int64), np.asarray(v_old, dtype=np.int64),
            np.asarray(f_old, dtype=np.int64))


def pair_graph_ref(rowptr, col):
    """numpy restatement of the parent-compressed graph (libfgc fgc_pair_graph) of a level whose convolution reads a
    4x-upsampled coarse tensor (custom_upsampling model.py:817-825 feeding custom_conv2d model.py:905,926): neighbour j
    contributes coarse row j >> 2 and the soft assignment (model.py:74-95) of edge (i, j) depends on (i >> 2, j >> 2)
    only, so per block of four siblings the edges collapse to distinct parents with one multiplicity per child.
    Returns (prow [n/4+1], pcol, pmul uint32 = mult of child 0 | child 1 << 8 | ...), pairs of a block in ascending P."""
    rowptr = np.asarray(rowptr, dtype=np.int64)
    col = np.asarray(col, dtype=np.int64)
    n = len(rowptr) - 1
    nc = n // 4
    src = np.repeat(np.arange(n), np.diff(rowptr))
    key = (src >> 2) * max(nc, 1) + (col >> 2)
    uk, inv = np.unique(key, return_inverse=True)
    mul = np.zeros((len(uk), 4), dtype=np.uint32)
    np.add.at(mul, (inv, src & 3), 1)
    prow = np.zeros(nc + 1, dtype=np.int32)
    np.cumsum(np.bincount(uk // max(nc, 1), minlength=nc), out=prow[1:])
    pmul = mul[:, 0] | (mul[:, 1] << 8) | (mul[:, 2] << 16) | (mul[:, 3] << 24)
    return prow, (uk % max(nc, 1)).astype(np.int32), pmul.astype(np.uint32)
