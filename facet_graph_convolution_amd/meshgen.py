"""Synthetic closed triangle meshes used as benchmark / parity inputs.

The reference ships no meshes (its data are external downloads, README.md:43-49),
so every workload in BASELINE.json is synthetic:

* ``icosphere(3)``        -> 642 vertices, 1 280 faces  (config C1)
* ``torus(250, 200)``     -> 50 000 vertices, 100 000 faces (config C2, the bench workload)
* ``torus(250, 100)``     -> 50 000 faces (C3), ``torus(1000, 500)`` -> 1M (C4), ``torus(500, 500)`` -> 500k (C5)

All vertex valences are 6 (5 at the 12 icosahedron corners), which satisfies the
hidden limits of the reference adjacency builder (utils.py:249-250).
"""
import numpy as np


def icosphere(subdivisions=3):
    """Unit icosphere by midpoint subdivision. Returns (V float32 [Vn,3], F uint32 [Fn,3])."""
    t = (1.0 + 5.0 ** 0.5) / 2.0
    verts = [(-1, t, 0), (1, t, 0), (-1, -t, 0), (1, -t, 0),
             (0, -1, t), (0, 1, t), (0, -1, -t), (0, 1, -t),
             (t, 0, -1), (t, 0, 1), (-t, 0, -1), (-t, 0, 1)]
    verts = [np.asarray(v, dtype=np.float64) / np.linalg.norm(v) for v in verts]
    faces = [(0, 11, 5), (0, 5, 1), (0, 1, 7), (0, 7, 10), (0, 10, 11),
             (1, 5, 9), (5, 11, 4), (11, 10, 2), (10, 7, 6), (7, 1, 8),
             (3, 9, 4), (3, 4, 2), (3, 2, 6), (3, 6, 8), (3, 8, 9),
             (4, 9, 5), (2, 4, 11), (6, 2, 10), (8, 6, 7), (9, 8, 1)]
    for _ in range(subdivisions):
        cache = {}

        def mid(a, b):
            key = (a, b) if a < b else (b, a)
            if key not in cache:
                m = verts[a] + verts[b]
                verts.append(m / np.linalg.norm(m))
                cache[key] = len(verts) - 1
            return cache[key]

        nf = []
        for a, b, c in faces:
            ab, bc, ca = mid(a, b), mid(b, c), mid(c, a)
            nf += [(a, ab, ca), (b, bc, ab), (c, ca, bc), (ab, bc, ca)]
        faces = nf
    return np.asarray(verts, dtype=np.float32), np.asarray(faces, dtype=np.uint32)


def torus(nu, nv, R=1.0, r=0.4):
    """nu x nv quad grid on a torus, each quad split in two -> 2*nu*nv faces, nu*nv vertices."""
    i, j = np.meshgrid(np.arange(nu), np.arange(nv), indexing="ij")
    a = 2.0 * np.pi * i / nu
    b = 2.0 * np.pi * j / nv
    V = np.stack([(R + r * np.cos(b)) * np.cos(a), (R + r * np.cos(b)) * np.sin(a), r * np.sin(b)], axis=-1)
    V = V.reshape(-1, 3).astype(np.float32)

    def vid(ii, jj):
        return (ii % nu) * nv + (jj % nv)

    v00, v10, v11, v01 = vid(i, j), vid(i + 1, j), vid(i + 1, j + 1), vid(i, j + 1)
    F = np.stack([np.stack([v00, v10, v11], -1), np.stack([v00, v11, v01], -1)], axis=2)
    return V, F.reshape(-1, 3).astype(np.uint32)


def add_noise(V, F, sigma_rel=0.2, seed=1):
    """Gaussian displacement along a random direction, sigma = sigma_rel * mean edge length."""
    rs = np.random.RandomState(seed)
    Vd = V.astype(np.float64)
    e = np.concatenate([Vd[F[:, 0]] - Vd[F[:, 1]], Vd[F[:, 1]] - Vd[F[:, 2]], Vd[F[:, 2]] - Vd[F[:, 0]]])
    mean_edge = np.sqrt((e * e).sum(1)).mean()
    d = rs.normal(size=Vd.shape)
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    mag = rs.normal(scale=sigma_rel * mean_edge, size=(Vd.shape[0], 1))
    return (Vd + d * mag).astype(np.float32)


def flip_edges(F, nflips, seed=0):
    """Irregular connectivity for tests: `nflips` random edge flips on a closed manifold triangle mesh (the edge
    shared by triangles (a,b,c) and (b,a,d) becomes (c,d); skipped when (c,d) already exists or a vertex would drop
    below valence 3).  Vertex valences spread to about 3..10, facet degrees beyond 16."""
    F = np.array(F, dtype=np.int64)
    rs = np.random.RandomState(seed)
    edge_faces = {}
    for f, (a, b, c) in enumerate(F):
        for u, v in ((a, b), (b, c), (c, a)):
            edge_faces.setdefault((min(u, v), max(u, v)), []).append(f)
    val = np.bincount(F.ravel())
    done = 0
    keys = list(edge_faces.keys())
    for _ in range(nflips * 20):
        if done >= nflips:
            break
        e = keys[rs.randint(len(keys))]
        fs = edge_faces.get(e)
        if fs is None or len(fs) != 2:
            continue
        f0, f1 = fs
        a, b = e
        c = [v for v in F[f0] if v != a and v != b]
        d = [v for v in F[f1] if v != a and v != b]
        if len(c) != 1 or len(d) != 1:
            continue
        c, d = c[0], d[0]
        if c == d or (min(c, d), max(c, d)) in edge_faces or val[a] <= 3 or val[b] <= 3:
            continue
        # orientation of f0: does it run a -> b -> c ?
        t = list(F[f0])
        ia = t.index(a)
        forward = t[(ia + 1) % 3] == b
        new0, new1 = ((c, a, d), (d, b, c)) if forward else ((c, d, a), (d, c, b))
        for f in (f0, f1):
            x, y, z = F[f]
            for u, v in ((x, y), (y, z), (z, x)):
                lst = edge_faces[(min(u, v), max(u, v))]
                lst.remove(f)
        del edge_faces[e]
        F[f0], F[f1] = new0, new1
        for f in (f0, f1):
            x, y, z = F[f]
            for u, v in ((x, y), (y, z), (z, x)):
                edge_faces.setdefault((min(u, v), max(u, v)), []).append(f)
        keys.append((min(c, d), max(c, d)))
        val[a] -= 1
        val[b] -= 1
        val[c] += 1
        val[d] += 1
        done += 1
    return F.astype(np.int32)
