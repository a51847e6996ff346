"""Host-side helpers that keep the names of the reference's ``utils.py`` where a counterpart exists."""
import numpy as np


def rand_rotation_matrix(deflection=1.0, randnums=None):
    """Uniform random rotation (Arvo's method), same parametrisation as the reference
    (utils.py:2034-2074): theta/phi/z from three uniforms, M = (V V^T - I) Rz(theta)."""
    if randnums is None:
        randnums = np.random.uniform(size=(3,))
    t, p, z = randnums
    t = t * 2.0 * deflection * np.pi
    p = p * 2.0 * np.pi
    z = z * 2.0 * deflection
    r = np.sqrt(z)
    V = np.array([np.sin(p) * r, np.cos(p) * r, np.sqrt(2.0 - z)])
    st, ct = np.sin(t), np.cos(t)
    Rz = np.array(((ct, st, 0.0), (-st, ct, 0.0), (0.0, 0.0, 1.0)))
    return (np.outer(V, V) - np.eye(3)).dot(Rz)


def normalizeOnce(a):
    """utils.py:26-31: row-wise a / (|a| + 1e-8)."""
    a = np.asarray(a)
    flat = a.reshape(-1, a.shape[-1])
    norms = np.sqrt((flat * flat).sum(1))[:, None] + 0.00000001
    return (flat * (1 / norms)).reshape(a.shape)


def normalize(a):
    """utils.py:33-35 (applied twice, as the reference does)."""
    return normalizeOnce(normalizeOnce(a))


def inv_perm(perm):
    """utils.py:1830-1835."""
    perm = np.asarray(perm)
    inv = np.zeros(max(len(perm), int(perm.max()) + 1), dtype=np.int64)
    inv[perm] = np.arange(len(perm))
    return inv


# ------------------------------------------------------------------------------------------------
# native preprocessing (libfgc host routines), same names / argument meaning as the reference utils.py
# ------------------------------------------------------------------------------------------------
def _faces_u32(faces):
    f = np.ascontiguousarray(np.asarray(faces).astype(np.uint32))
    if f.ndim != 2 or f.shape[1] != 3:
        raise ValueError("faces must be [F,3] (triangular faces only)")
    return f


def face_features(verts, faces):
    """(normals float32 [F,3], barycentres/bbox-diagonal float64 [F,3]) in one native pass."""
    from . import _lib
    V = np.ascontiguousarray(np.asarray(verts, dtype=np.float32))
    F = _faces_u32(faces)
    normals = np.empty((F.shape[0], 3), dtype=np.float32)
    centres = np.empty((F.shape[0], 3), dtype=np.float64)
    _lib.check(_lib.lib().fgc_face_features(V.ctypes.data, V.shape[0], F.ctypes.data, F.shape[0],
                                            normals.ctypes.data, centres.ctypes.data), "fgc_face_features")
    return normals, centres


def computeFacesNormals(verts, faces):
    """utils.py:63-68."""
    return face_features(verts, faces)[0]


def getTrianglesBarycenter(vl, fl, normalize=True):
    """utils.py:1264-1294 (normalize=True only: positions divided by the bbox diagonal, not centred)."""
    if not normalize:
        raise NotImplementedError("only the default normalize=True path is on the denoising path")
    return face_features(vl, fl)[1]


def getFacesLargeAdj(faces, K):
    """utils.py:243-295: vertex-sharing facet adjacency as a K-list (one-indexed, slot 0 = self), bit-exact."""
    from . import _lib
    import ctypes as C
    F = _faces_u32(faces)
    nv = int(F.max()) + 1
    adj = np.empty((F.shape[0], K), dtype=np.int32)
    unreg = C.c_int64(0)
    _lib.check(_lib.lib().fgc_faces_large_adj(F.ctypes.data, F.shape[0], nv, K, adj.ctypes.data, C.byref(unreg)),
               "fgc_faces_large_adj")
    if unreg.value > 0:
        print("unregistered connections (faces): " + str(unreg.value / 2))
    return adj


def getVerticesFaces(faces, k_v, vnum=0):
    """utils.py:370-395: faces incident to every vertex (row indices of `faces`, -1 padded to k_v; rows starting with
    -1 are fake faces and skipped), natively and bit-exact."""
    from . import _lib
    F = np.ascontiguousarray(np.asarray(faces).reshape(-1, 3), dtype=np.int32)
    if vnum == 0:
        vnum = int(F.max()) + 1
    out = np.empty((int(vnum), int(k_v)), dtype=np.int32)
    _lib.check(_lib.lib().fgc_vertices_faces(F.ctypes.data, F.shape[0], int(vnum), int(k_v), out.ctypes.data),
               "fgc_vertices_faces")
    return out


def normalizePointSets(vl1, vl2):
    """utils.py:2077-2104: both point sets divided by the bounding-box diagonal of their union (not centred)."""
    vl1, vl2 = np.asarray(vl1), np.asarray(vl2)
    lo = np.minimum(vl1.min(0), vl2.min(0))
    hi = np.maximum(vl1.max(0), vl2.max(0))
    diag = float(np.sqrt(((hi - lo).astype(np.float64) ** 2).sum()))
    return vl1 / diag, vl2 / diag


def getEdgeMap(faces, maxEdges=50):
    """utils.py:91-183: (e_map [E,4] = [v1, v2, f1, f2 or -1], v_e_map [V, maxEdges] edge ids per vertex, -1 padded),
    bit-exact (same visiting order), natively."""
    from . import _lib
    import ctypes as C
    F = _faces_u32(faces)
    nv = int(F.max()) + 1
    e_map = np.empty((F.shape[0] * 3, 4), dtype=np.int32)
    v_e_map = np.empty((nv, maxEdges), dtype=np.int32)
    ne = C.c_int32(0)
    _lib.check(_lib.lib().fgc_edge_map(F.ctypes.data, F.shape[0], nv, int(maxEdges), e_map.ctypes.data, C.byref(ne),
                                       v_e_map.ctypes.data), "fgc_edge_map")
    return e_map[:ne.value].copy(), v_e_map


def getGraphPatch_wMask(fAdjIn, nodesNum, seed, mask, minPatchSize):
    """utils.py:1508-1696: (patch K-list one-indexed [n, K], old index of every patch node [n], nextSeed), natively and
    bit-exact (same breadth-first queue discipline)."""
    from . import _lib
    import ctypes as C
    adj = np.ascontiguousarray(np.asarray(fAdjIn), dtype=np.int32)
    n, K = adj.shape
    m = np.ascontiguousarray(np.asarray(mask) == 1, dtype=np.int8)
    out = np.empty((int(nodesNum) + K, K), dtype=np.int32)
    old = np.empty(int(nodesNum) + K, dtype=np.int32)
    cnt, nxt = C.c_int32(0), C.c_int32(-1)
    _lib.check(_lib.lib().fgc_graph_patch(adj.ctypes.data, n, K, int(nodesNum), int(seed), m.ctypes.data,
                                          int(minPatchSize), out.ctypes.data, old.ctypes.data, C.byref(cnt),
                                          C.byref(nxt)), "fgc_graph_patch")
    return out[:cnt.value].copy(), old[:cnt.value].copy(), int(nxt.value)


def getMeshPatch(vIn, fIn, fAdjIn, faceNum, seed):
    """utils.py:1298-1410: (vOut [nv',3] float32, fOut [nf',3] in patch vertex ids, fAdjOut [nf',K] one-indexed, vOldInd,
    fOldInd), natively and bit-exact (same breadth-first queue discipline; the reference's vertex buffer of
    int(0.6 * faceNum) + K rows is kept, overflowing it raises like the reference's IndexError)."""
    from . import _lib
    import ctypes as C
    V = np.ascontiguousarray(np.asarray(vIn), dtype=np.float32)
    F = np.ascontiguousarray(np.asarray(fIn), dtype=np.int32)
    adj = np.ascontiguousarray(np.asarray(fAdjIn), dtype=np.int32)
    K = adj.shape[1]
    faceNum = int(faceNum)
    v_cap = int(faceNum * 0.6) + K
    v_out = np.empty((v_cap, 3), dtype=np.float32)
    f_out = np.empty((faceNum + K, 3), dtype=np.int32)
    adj_out = np.empty((faceNum + K, K), dtype=np.int32)
    v_old = np.empty(v_cap, dtype=np.int32)
    f_old = np.empty(faceNum + K, dtype=np.int32)
    n_v, n_f = C.c_int32(0), C.c_int32(0)
    rc = _lib.lib().fgc_mesh_patch(V.ctypes.data, V.shape[0], F.ctypes.data, F.shape[0], adj.ctypes.data, K, faceNum,
                                   int(seed), v_out.ctypes.data, v_cap, f_out.ctypes.data, adj_out.ctypes.data,
                                   v_old.ctypes.data, f_old.ctypes.data, C.byref(n_v), C.byref(n_f))
    if rc:
        raise IndexError(_lib.lib().fgc_last_error().decode())
    nv, nf = n_v.value, n_f.value
    return (v_out[:nv].copy(), f_out[:nf].astype(np.int64), adj_out[:nf].astype(np.int64), v_old[:nv].astype(np.int64),
            f_old[:nf].astype(np.int64))


def getBoundingBox(points):
    """utils.py:2130-2137: [[xmin, xmax], [ymin, ymax], [zmin, zmax]]."""
    points = np.asarray(points)
    return np.stack([points.min(axis=0), points.max(axis=0)], axis=1)


def takePointSetSlice(points, boundBox):
    """utils.py:2109-2125: the points inside the (closed) box."""
    points = np.asarray(points)
    boundBox = np.asarray(boundBox)
    return points[np.all((points >= boundBox[:, 0]) & (points <= boundBox[:, 1]), axis=1)]


def coarsen_klists(adj, pos, normals, levels=4, K=23, seed=0, parents=None, keep=(0, 2, 4)):
    """listToSparseWNormals + coarsen + sparseToList (utils.py:1753-1827, lib/coarsening.py:5-31).

    Returns (klists for the graph levels in `keep`, newToOld, parents, has_saturated).  With `parents`
    (recorded cluster assignments of a reference run) the result replays that run bit-exactly."""
    from . import _lib
    import ctypes as C
    L = _lib.lib()
    adj = np.ascontiguousarray(np.asarray(adj, dtype=np.int32))
    pos = np.ascontiguousarray(np.asarray(pos, dtype=np.float64))
    nrm = np.ascontiguousarray(np.asarray(normals, dtype=np.float32))
    n, Kin = adj.shape
    h = C.c_void_p(0)
    if parents is not None:
        arrs = [np.ascontiguousarray(np.asarray(p, dtype=np.int32)) for p in parents]
        if len(arrs) != levels:
            raise ValueError("need %d recorded parent arrays" % levels)
        ptrs = (C.c_void_p * levels)(*[a.ctypes.data for a in arrs])
        lens = np.asarray([len(a) for a in arrs], dtype=np.int32)
        pp, pl = C.cast(ptrs, C.c_void_p), lens.ctypes.data
    else:
        pp, pl = None, None
    _lib.check(L.fgc_hierarchy_build(adj.ctypes.data, n, Kin, pos.ctypes.data, nrm.ctypes.data, levels,
                                     C.c_uint64(seed), pp, pl, C.byref(h)), "fgc_hierarchy_build")
    try:
        klists, sat_any = [], False
        for lvl in keep:
            m = L.fgc_hierarchy_size(h, lvl)
            out = np.empty((m, K), dtype=np.int32)
            sat = C.c_int32(0)
            _lib.check(L.fgc_hierarchy_klist(h, lvl, K, out.ctypes.data, C.byref(sat)), "fgc_hierarchy_klist")
            sat_any = sat_any or bool(sat.value)
            klists.append(out)
        n0 = L.fgc_hierarchy_size(h, 0)
        new_to_old = np.empty(n0, dtype=np.int32)
        _lib.check(L.fgc_hierarchy_new_to_old(h, 0, new_to_old.ctypes.data), "fgc_hierarchy_new_to_old")
        par = []
        for lvl in range(levels):
            p = np.empty(L.fgc_hierarchy_real_size(h, lvl), dtype=np.int32)
            _lib.check(L.fgc_hierarchy_parents(h, lvl, p.ctypes.data), "fgc_hierarchy_parents")
            par.append(p)
    finally:
        L.fgc_hierarchy_free(h)
    return klists, new_to_old, par, sat_any


def metis_one_level(rr, cc, vv, rid, weights):
    """lib/coarsening.py:135-192 (native, bit-exact given its arguments). Returns (cluster_id, totalAssoc)."""
    from . import _lib
    import ctypes as C
    rr = np.ascontiguousarray(rr, dtype=np.int32)
    cc = np.ascontiguousarray(cc, dtype=np.int32)
    vv = np.ascontiguousarray(vv, dtype=np.float32)
    rid = np.ascontiguousarray(rid, dtype=np.int64)
    weights = np.ascontiguousarray(weights, dtype=np.float32)
    N = int(rr[-1]) + 1
    cid = np.empty(N, dtype=np.int32)
    assoc = C.c_double(0)
    _lib.check(_lib.lib().fgc_metis_one_level(rr.ctypes.data, cc.ctypes.data, vv.ctypes.data, len(rr),
                                              rid.ctypes.data, weights.ctypes.data, N, cid.ctypes.data,
                                              C.byref(assoc)), "fgc_metis_one_level")
    return cid, assoc.value


# ---------------------------------------------------------------------------------------------------
# OBJ input / output (utils.py:476-640, 659-697): what infer.py reads and writes
# ---------------------------------------------------------------------------------------------------
def load_mesh(path, filename, K=0, bGetAdj=False):
    """utils.py:476-640 for bGetAdj=False (the only mode the denoising path uses, dataClasses.py:513): reads 'v' and
    'f' records of a Wavefront OBJ (vertex/texture/normal triplets accepted, polygons fan-triangulated around their
    first vertex, everything else ignored).  Returns (vertices float32 [V,3], adj, free_ind, faces [F,3] zero-indexed
    uint16 when V < 65536 else uint32, per-vertex normals) with adj / free_ind empty as in the reference."""
    import os
    if bGetAdj:
        raise NotImplementedError("vertex adjacency (bGetAdj=True) is not on the facet denoising path")
    verts, faces = [], []
    with open(os.path.join(path, filename), "r") as fh:
        for line in fh:
            if line.startswith("#"):
                continue
            tok = line.split()
            if not tok:
                continue
            if tok[0] == "v":
                verts.append((float(tok[1]), float(tok[2]), float(tok[3])))
            elif tok[0] == "f":
                idx = [int(t.split("/")[0]) - 1 for t in tok[1:]]
                for k in range(1, len(idx) - 1):
                    faces.append((idx[0], idx[k], idx[k + 1]))
    V = np.asarray(verts, dtype=np.float32).reshape(-1, 3)
    F = np.asarray(faces, dtype=np.int64).reshape(-1, 3).astype(np.uint16 if V.shape[0] < 65536 else np.uint32)
    return V, [], [], F, computeNormals(V, F)


def computeNormals(verts, faces):
    """utils.py:44-60: per-vertex normals from the unit face normals.  `normals[faces[:, i]] += Nn` is a buffered
    fancy-index add: for each corner i a vertex keeps the contribution of the LAST face listing it there, not the sum
    over faces.  Kept as the reference has it (the denoising path never reads this output, dataClasses.py:513)."""
    verts = np.asarray(verts, dtype=np.float32)
    F = np.asarray(faces).astype(np.int64)
    T = verts[F]
    Nn = normalize(np.cross(T[:, 1] - T[:, 0], T[:, 2] - T[:, 0]))
    out = np.zeros(verts.shape, dtype=np.float32)
    for k in range(3):
        out[F[:, k]] += Nn
    return normalize(out)


def write_mesh(vl, fl, strFileName):
    """utils.py:659-697: 'v x y z' (6 decimals, extra per-vertex columns such as colours written as they are), then
    'f a b c' one-indexed; a face (0,0,..) ends the list and a face (-1,-1,..) is skipped, as in the reference."""
    vl = np.asarray(vl)
    fl = np.asarray(fl).astype(np.int64) + 1
    with open(strFileName, "w") as fh:
        for row in vl:
            fh.write("v " + " ".join("%.6f" % x for x in row) + " \n")
        for row in fl:
            if row[0] == 1 and row[1] == 1:
                break
            if row[0] == 0 and row[1] == 0:
                continue
            fh.write("f " + " ".join(str(int(t)) for t in row) + " \n")
