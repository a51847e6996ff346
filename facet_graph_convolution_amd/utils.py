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
