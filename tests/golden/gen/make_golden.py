#!/usr/bin/env python3
"""Generate the golden fixtures under tests/golden/ by EXECUTING THE REFERENCE SOURCE.

Run in the build container only (the reference tree does not exist on the GPU box):

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/gen/make_golden.py

How the reference is run:
  * ``lib/coarsening.py`` imports unmodified (numpy/scipy only).
  * ``utils.py``, ``dataClasses.py``, ``model.py``, ``train.py`` ``import tensorflow`` at the top;
    TensorFlow is not installed, so ``tests/golden/gen/tf_shim`` (an eager torch-CPU mapping
    of the ~55 tf symbols, our own test tooling) is put first on ``sys.path``.  The
    reference files themselves are read where they lie under /root/reference; no line
    of them is copied or edited.
  * ``time.clock`` (removed in Python 3.8, used at dataClasses.py:39) is aliased.

What the fixtures pin: the reference's *algorithm as written* with torch-CPU kernels
standing in for TF kernels.  Agreement with real TensorFlow binaries is unpinned
(expected at fp32 summation-order level).

Only data is written: inputs, expected outputs, recorded random state of the
coarsening (per-level ``parents``; one ``metis_one_level`` call's arguments).
Weights are NOT stored: they are regenerated from ``RandomState(0)`` in variable
creation order (see ``param_values``; mirrored by the product's ``init_params``).
"""
import os
import sys
import time
import types

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.abspath(os.path.join(HERE, "..", "..", ".."))
OUT = os.path.abspath(os.path.join(HERE, ".."))
REF = "/root/reference/Code"

sys.dont_write_bytecode = True
sys.path.insert(0, os.path.join(HERE, "tf_shim"))
sys.path.insert(0, REF)
sys.path.insert(0, REPO)
if not hasattr(time, "clock"):
    time.clock = time.perf_counter

import warnings  # noqa: E402

warnings.filterwarnings("ignore")

import numpy as np  # noqa: E402
import torch  # noqa: E402

torch.set_num_threads(8)

import tensorflow as tf  # noqa: E402  (the shim)
import lib.coarsening as coarsening  # noqa: E402  (reference, unmodified)
import utils as ref_utils  # noqa: E402
import dataClasses as ref_data  # noqa: E402
import model as ref_model  # noqa: E402
import train as ref_train  # noqa: E402

from facet_graph_convolution_amd.meshgen import icosphere, torus, add_noise  # noqa: E402

F64 = os.environ.get("TF_SHIM_DTYPE", "float32") == "float64"
FDT = torch.float64 if F64 else torch.float32


# ----------------------------------------------------------------------------------------
# helpers
# ----------------------------------------------------------------------------------------
def param_values(shapes_names, seed=0):
    """Weights in variable-creation order from RandomState(seed); std per the reference
    initialisers (model.py:16-44): 0.05 for 'weight'/'assignment', 0.01 for 'bias'."""
    rs = np.random.RandomState(seed)
    vals = []
    for name, shape in shapes_names:
        std = 0.01 if name == "bias" else 0.05
        vals.append(rs.normal(0.0, std, size=shape).astype(np.float32))
    return vals


def run_with_params(fn, seed=0):
    """Run fn() twice: once to discover variable shapes, once with seeded values.
    Returns (result, [(name, leaf tensor)])."""
    tf.VARIABLES.clear()
    tf.VARIABLE_FEED = None
    fn()
    shapes = [(n, tuple(v.shape)) for n, v in tf.VARIABLES]
    vals = param_values(shapes, seed)
    tf.VARIABLES.clear()
    tf.VARIABLE_FEED = iter(vals)
    res = fn()
    tf.VARIABLE_FEED = None
    variables = list(tf.VARIABLES)
    return res, variables


def save(name, **arrays):
    path = os.path.join(OUT, name)
    np.savez_compressed(path, **arrays)
    print("wrote %-32s %8.1f KB" % (name, os.path.getsize(path) / 1024.0))


class Recorder:
    """Wraps the reference coarsening entry points to record their random state."""

    def __init__(self):
        self.one_level_calls = []
        self.parents = None
        self._orig_one = coarsening.metis_one_level
        self._orig_metis = coarsening.metis

    def __enter__(self):
        rec = self

        def one_level(rr, cc, vv, rid, weights):
            cid, assoc = rec._orig_one(rr, cc, vv, rid, weights)
            rec.one_level_calls.append(dict(rr=np.array(rr), cc=np.array(cc), vv=np.array(vv),
                                            rid=np.array(rid), weights=np.array(weights),
                                            cluster_id=np.array(cid), assoc=float(assoc)))
            return cid, assoc

        def metis(W, levels, rid=None):
            graphs, parents = rec._orig_metis(W, levels, rid)
            rec.parents = [np.array(p) for p in parents]
            return graphs, parents

        coarsening.metis_one_level = one_level
        coarsening.metis = metis
        return self

    def __exit__(self, *a):
        coarsening.metis_one_level = self._orig_one
        coarsening.metis = self._orig_metis


# ----------------------------------------------------------------------------------------
# Part A: preprocessing (reference utils.py / dataClasses.py / lib/coarsening.py)
# ----------------------------------------------------------------------------------------
def preprocess_fixture(tag, Vclean, F, seed):
    V = add_noise(Vclean, F, 0.2, seed=1)
    normals = ref_utils.computeFacesNormals(V, F)
    centres = ref_utils.getTrianglesBarycenter(V, F)
    fadj = ref_utils.getFacesLargeAdj(F, 23)
    coo = ref_utils.listToSparseWNormals(fadj, centres, normals)

    np.random.seed(seed)
    ds = ref_data.PreprocessedData(10 ** 9, 2, 3)
    with Recorder() as rec:
        ds.addMesh_TimeEfficient(V, F, GTV=Vclean)
    assert len(ds.in_list) == 1
    x = ds.in_list[0]
    adjs = ds.adj_list[0]
    gt = ds.gt_list[0]
    perm = ds.permutations[0]
    nlev = len(rec.parents)
    first = rec.one_level_calls[0]
    arrays = dict(
        V=V, Vclean=Vclean, F=F,
        normals=normals, centres=centres, fadj=fadj,
        coo_row=coo.row, coo_col=coo.col, coo_val=coo.data,
        x=x, adj0=adjs[0], adj1=adjs[1], adj2=adjs[2], gt=gt,
        num_faces=np.int64(ds.num_faces[0]), permutations=np.asarray(perm),
        n_parent_levels=np.int64(nlev),
        ol_rr=first["rr"], ol_cc=first["cc"], ol_vv=first["vv"], ol_rid=first["rid"],
        ol_weights=first["weights"], ol_cluster_id=first["cluster_id"], ol_assoc=np.float64(first["assoc"]),
    )
    for i, p in enumerate(rec.parents):
        arrays["parents%d" % i] = p
    save("prep_%s.npz" % tag, **arrays)
    return x, adjs, gt, ds


# ----------------------------------------------------------------------------------------
# Part B: model (reference model.py / utils.normalizeTensor / train.faceNormalsLoss)
# ----------------------------------------------------------------------------------------
def conv_case(tag, x_np, adj_np, cout, M, seed, biasMask=True):
    """custom_conv2d forward + gradients for an upstream dy drawn from RandomState(seed+100)."""
    # feeds are cast to the float32 placeholder dtype first (train.py:409), also in the float64 run
    x_np = np.asarray(x_np).astype(np.float32)
    x = torch.tensor(x_np, dtype=FDT).requires_grad_(True)
    adj = torch.tensor(adj_np, dtype=torch.int32)

    def fn():
        return ref_model.custom_conv2d(x, adj, cout, M, biasMask=biasMask)

    (y, _), variables = run_with_params(fn, seed)
    # creation order W0, b, u, c, v (model.py:430-433,447)
    dy = np.random.RandomState(seed + 100).normal(size=tuple(y.shape)).astype(np.float32)
    (y * torch.tensor(dy, dtype=FDT)).sum().backward()
    out = dict(x=x_np.astype(np.float32), adj=adj_np.astype(np.int32), cout=np.int64(cout), M=np.int64(M),
               seed=np.int64(seed), y=y.detach().numpy(), dy=dy, dx=x.grad.numpy().astype(np.float32))
    for (name, v), key in zip(variables, ["W0", "b", "u", "c", "v"]):
        out["d" + key] = v.grad.numpy().astype(np.float32)
    save("conv_%s%s.npz" % (tag, "_f64" if F64 else ""), **out)


def net_case(tag, x_np, adjs_np, gt_np, seed, multi_scale):
    # feeds are cast to the float32 placeholder dtype first (train.py:409-427), also in the float64 run
    x_in = torch.tensor(np.asarray(x_np).astype(np.float32), dtype=FDT)
    adjs = [torch.tensor(a, dtype=torch.int32) for a in adjs_np]
    gt = torch.tensor(np.asarray(gt_np).astype(np.float32), dtype=FDT)
    n0 = x_np.shape[1]
    sample_ind = np.random.RandomState(2).randint(n0, size=4000)
    R = ref_utils.rand_rotation_matrix(randnums=np.random.RandomState(3).uniform(size=3))
    rot = torch.tensor(np.tile(R.reshape(1, 1, 3, 3), (1, n0, 1, 1)).astype(np.float32), dtype=FDT)

    keep = {}

    def fn():
        # the three statements below restate train.py:439-451 (rotation of GT and of both
        # 3-vectors of the input) with the same tf ops; everything else is reference code.
        tfn_rot = tf.reshape(tf.matmul(rot, tf.reshape(gt, [1, -1, 3, 1])), [1, -1, 3])
        fn_rot = tf.reshape(x_in, [1, -1, 2, 3])
        fn_rot = tf.transpose(fn_rot, [0, 1, 3, 2])
        fn_rot = tf.matmul(rot, fn_rot)
        fn_rot = tf.reshape(tf.transpose(fn_rot, [0, 1, 3, 2]), [1, -1, 6])
        out = ref_model.get_model_reg_multi_scale(fn_rot, adjs, 1.0, multiScale=multi_scale)
        keep["raw"] = out
        y0 = out[0] if multi_scale else out
        n_conv = ref_utils.normalizeTensor(y0)
        keep["n_conv"] = n_conv
        keep["fn_rot"] = fn_rot
        keep["tfn_rot"] = tfn_rot
        # train.py:509-517
        samp_n = tf.transpose(tf.gather(tf.transpose(n_conv, [1, 0, 2]), torch.tensor(sample_ind)), [1, 0, 2])
        samp_gt = tf.transpose(tf.gather(tf.transpose(tfn_rot, [1, 0, 2]), torch.tensor(sample_ind)), [1, 0, 2])
        return ref_train.faceNormalsLoss(samp_n, samp_gt)

    loss, variables = run_with_params(fn, seed)
    loss.backward()
    out = dict(seed=np.int64(seed), sample_ind=sample_ind.astype(np.int64), R=R,
               fn_rot=keep["fn_rot"].detach().numpy(), tfn_rot=keep["tfn_rot"].detach().numpy(),
               n_conv=keep["n_conv"].detach().numpy(), loss=np.float64(loss.item()),
               n_vars=np.int64(len(variables)))
    if multi_scale:
        for i, t in enumerate(keep["raw"]):
            out["y%d" % i] = t.detach().numpy()
    else:
        out["y0"] = keep["raw"].detach().numpy()
    for i, (name, v) in enumerate(variables):
        g = v.grad.numpy() if v.grad is not None else np.zeros(tuple(v.shape), np.float32)
        out["g%02d" % i] = g.astype(np.float32)  # float64 run: true gradient rounded once
        out["name%02d" % i] = np.array(name)
    # inference epilogue (train.py:115-121,136): un-permute, drop fake rows, normalize twice
    save("net_%s%s%s.npz" % (tag, "_ms" if multi_scale else "", "_f64" if F64 else ""), **out)


def net_case_patch_size(tag, subdiv, seed):
    """One training step of the reference network at the reference's OWN patch size (settings.py:20: 20 000 faces): icosphere
    subdivision 5 = 20 480 faces through the reference preprocessing (its own coarsening draw, np.random.seed(seed)) and the
    reference model / loss, forward + backward.  Self-contained fixture: the inputs as the float32 / int placeholders see
    them (train.py:409-427), the three K-lists (int16 / int32), loss, normalised output, the 44 gradients."""
    Vc, F = icosphere(subdiv)
    V = add_noise(Vc, F, 0.2, seed=1)
    np.random.seed(seed)
    ds = ref_data.PreprocessedData(10 ** 9, 2, 3)
    t0 = time.time()
    ds.addMesh_TimeEfficient(V, F, GTV=Vc)
    print("reference preprocessing of %d faces: %.1f s" % (F.shape[0], time.time() - t0))
    assert len(ds.in_list) == 1
    x_np, adjs_np, gt_np = ds.in_list[0], ds.adj_list[0], ds.gt_list[0]
    x32 = np.asarray(x_np).astype(np.float32)
    gt32 = np.asarray(gt_np).astype(np.float32)
    x_in = torch.tensor(x32, dtype=FDT)
    adjs = [torch.tensor(np.asarray(a), dtype=torch.int32) for a in adjs_np]
    gt = torch.tensor(gt32, dtype=FDT)
    n0 = x32.shape[1]
    sample_ind = np.random.RandomState(2).randint(n0, size=4000)
    R = ref_utils.rand_rotation_matrix(randnums=np.random.RandomState(3).uniform(size=3))
    rot = torch.tensor(np.tile(R.reshape(1, 1, 3, 3), (1, n0, 1, 1)).astype(np.float32), dtype=FDT)
    keep = {}

    def fn():
        # (train.py:439-451, 509-517 restated with the same tf ops as in net_case; the network and the loss are reference code)
        tfn_rot = tf.reshape(tf.matmul(rot, tf.reshape(gt, [1, -1, 3, 1])), [1, -1, 3])
        fn_rot = tf.reshape(x_in, [1, -1, 2, 3])
        fn_rot = tf.transpose(fn_rot, [0, 1, 3, 2])
        fn_rot = tf.matmul(rot, fn_rot)
        fn_rot = tf.reshape(tf.transpose(fn_rot, [0, 1, 3, 2]), [1, -1, 6])
        y0 = ref_model.get_model_reg_multi_scale(fn_rot, adjs, 1.0, multiScale=False)
        keep["y0"] = y0
        n_conv = ref_utils.normalizeTensor(y0)
        keep["n_conv"] = n_conv
        samp_n = tf.transpose(tf.gather(tf.transpose(n_conv, [1, 0, 2]), torch.tensor(sample_ind)), [1, 0, 2])
        samp_gt = tf.transpose(tf.gather(tf.transpose(tfn_rot, [1, 0, 2]), torch.tensor(sample_ind)), [1, 0, 2])
        return ref_train.faceNormalsLoss(samp_n, samp_gt)

    t0 = time.time()
    loss, variables = run_with_params(fn, seed)
    loss.backward()
    print("reference forward + backward at N0 = %d: %.1f s" % (n0, time.time() - t0))
    out = dict(seed=np.int64(seed), num_faces=np.int64(ds.num_faces[0]), sample_ind=sample_ind.astype(np.int32), R=R,
               x=x32, gt=gt32, loss=np.float64(loss.item()), n_vars=np.int64(len(variables)),
               y0=keep["y0"].detach().numpy().astype(np.float32), n_conv=keep["n_conv"].detach().numpy().astype(np.float32))
    for l, a in enumerate(adjs_np):
        a = np.asarray(a)
        assert a.min() >= 0
        out["adj%d" % l] = a.astype(np.int16 if a.max() < 2 ** 15 else np.int32)
    for i, (name, v) in enumerate(variables):
        out["g%02d" % i] = v.grad.numpy().astype(np.float32)
    save("net_%s%s.npz" % (tag, "_f64" if F64 else ""), **out)


def infer_case(tag, x_np, adjs_np, ds, seed):
    """Forward without rotation + inference epilogue of inferNetOld (train.py:72-75,115-121,136)."""
    x_in = torch.tensor(np.asarray(x_np).astype(np.float32), dtype=FDT)
    adjs = [torch.tensor(a, dtype=torch.int32) for a in adjs_np]

    def fn():
        y = ref_model.get_model_reg_multi_scale(x_in, adjs, 1.0, multiScale=False)
        return ref_utils.normalizeTensor(y)

    n_conv, _ = run_with_params(fn, seed)
    outN = n_conv.detach().numpy().squeeze()
    outN = outN[ds.permutations[0]]
    outN = outN[0:ds.num_faces[0]]
    pred = ref_utils.normalize(outN)
    save("infer_%s%s.npz" % (tag, "_f64" if F64 else ""), n_conv=n_conv.detach().numpy(), predicted_normals=pred)


def vertex_case(tag, V_noisy, F, normals):
    """getEdgeMap (utils.py:91-183) + update_position2 (train.py:1467-1557), as called by inferNetOld
    (train.py:129-139: 60 iterations, MAX_EDGES = 20 slots per vertex).  Also an open mesh variant (boundary edges)."""
    e_map, v_e_map = ref_utils.getEdgeMap(F, maxEdges=20)
    x = torch.tensor(V_noisy.astype(np.float32), dtype=FDT)[None]
    fn = torch.tensor(normals.astype(np.float32), dtype=FDT)[None]
    em = torch.tensor(e_map[None].astype(np.int32))
    vem = torch.tensor(v_e_map[None].astype(np.int32))
    out = {}
    for it in (1, 60):
        out["x_%d" % it] = ref_train.update_position2(x, fn, em, vem, iter_num=it, max_edges=20).detach().numpy()[0]
    save("vertex_%s%s.npz" % (tag, "_f64" if F64 else ""), verts=V_noisy.astype(np.float32), faces=F.astype(np.int32),
         normals=normals.astype(np.float32), edge_map=e_map, v_e_map=v_e_map, **out)


def multiscale_vertex_case(tag, V_noisy, Vclean, F, seed):
    """The multi-scale vertex pipeline on a whole mesh (dataClasses.py:374-440 small-mesh branch, train.py:1668-1798):
    faces padded with -1 rows and re-ordered like the nodes, getVerticesFaces, normalizePointSets, then
    avg_ignore_zeros pooling, updateFacesCenter and update_position_MS driven with clean-mesh normals pooled to the
    three levels."""
    prep = np.load(os.path.join(OUT, "prep_%s.npz" % tag))
    perm = prep["permutations"]                       # old -> new
    n0 = len(perm)
    new_to_old = np.empty(n0, dtype=np.int64)
    new_to_old[perm] = np.arange(n0)
    faces_p = np.concatenate([F.astype(np.int64), -np.ones((n0 - F.shape[0], 3), dtype=np.int64)], 0)[new_to_old]
    v_faces = ref_utils.getVerticesFaces(faces_p, 25, V_noisy.shape[0])
    Vn, _ = ref_utils.normalizePointSets(V_noisy.astype(np.float32), V_noisy.astype(np.float32))
    nrm = ref_utils.computeFacesNormals(Vclean, F)
    nrm0 = np.concatenate([nrm, np.zeros((n0 - F.shape[0], 3))], 0)[new_to_old].astype(np.float32)
    n0_t = torch.tensor(nrm0, dtype=FDT)[None]
    n1_t = ref_utils.normalizeTensor(ref_model.custom_binary_tree_pooling(n0_t, steps=2, pooltype="avg_ignore_zeros"))
    n2_t = ref_utils.normalizeTensor(ref_model.custom_binary_tree_pooling(n1_t, steps=2, pooltype="avg_ignore_zeros"))
    x = torch.tensor(Vn.astype(np.float32), dtype=FDT)[None]
    faces_t = torch.tensor(faces_p[None].astype(np.int32))
    vf_t = torch.tensor(v_faces[None].astype(np.int32))
    centers = ref_train.updateFacesCenter(x, faces_t, 2)
    out = {"verts": V_noisy.astype(np.float32), "verts_norm": Vn.astype(np.float32), "faces_perm": faces_p.astype(np.int32),
           "v_faces": v_faces.astype(np.int32), "n0": nrm0, "n1": n1_t.detach().numpy()[0], "n2": n2_t.detach().numpy()[0],
           "fpos0": centers[0].detach().numpy()[0], "fpos1": centers[1].detach().numpy()[0],
           "fpos2": centers[2].detach().numpy()[0]}
    for its in ((2, 1, 1), (80, 20, 20)):
        xr, dxl = ref_train.update_position_MS(x, [n0_t, n1_t, n2_t], faces_t, vf_t, coarsening_steps=2,
                                               iter_num_list=list(its))
        key = "_".join(str(i) for i in its)
        out["x_" + key] = xr.detach().numpy()[0]
        for k, dx in enumerate(dxl):
            out["dx%d_%s" % (k, key)] = dx.detach().numpy()
    save("msvertex_%s%s.npz" % (tag, "_f64" if F64 else ""), **out)


def patch_case(tag, F, patch_size, min_patch_size, seed):
    """getGraphPatch_wMask (utils.py:1508-1696) driven as dataClasses.py:76-95 drives it: seeds drawn with
    np.random among the uncovered faces unless the previous patch proposed one; every patch is recorded."""
    import contextlib, io
    adj = ref_utils.getFacesLargeAdj(F, 23)
    fnum = F.shape[0]
    check = np.zeros(fnum)
    rng = np.arange(fnum)
    np.random.seed(seed)
    next_seed = -1
    out = {"faces": F.astype(np.int32), "adj": adj.astype(np.int32), "patch_size": patch_size,
           "min_patch_size": min_patch_size}
    k = 0
    while np.any(check == 0):
        todo = rng[check == 0]
        s0 = todo[np.random.randint(todo.shape[0])] if next_seed == -1 else next_seed
        mask_in = check.copy()
        with contextlib.redirect_stdout(io.StringIO()):
            padj, old, next_seed = ref_utils.getGraphPatch_wMask(adj, patch_size, s0, check, min_patch_size)
        check[old] = 1
        out["seed%d" % k], out["mask%d" % k] = int(s0), mask_in.astype(np.int8)
        out["adj%d" % k], out["old%d" % k], out["next%d" % k] = padj.astype(np.int32), old.astype(np.int32), int(next_seed)
        k += 1
    out["num_patches"] = k
    save("patch_%s.npz" % tag, **out)


def mesh_patch_case(tag, V, F, cases):
    """getMeshPatch (utils.py:1298-1410) for a few (faceNum, seed) pairs, plus the bounding-box slice helpers
    (utils.py:2109-2137) on the patch's vertices."""
    adj = ref_utils.getFacesLargeAdj(F, 23)
    out = {"verts": V.astype(np.float32), "faces": F.astype(np.int32), "adj": adj.astype(np.int32),
           "cases": np.asarray(cases, dtype=np.int32)}
    for k, (face_num, seed) in enumerate(cases):
        vO, fO, aO, vOld, fOld = ref_utils.getMeshPatch(V.astype(np.float32), F, adj, face_num, seed)
        bb = ref_utils.getBoundingBox(vO)
        inside = ref_utils.takePointSetSlice(V.astype(np.float32), bb)
        out.update({"v%d" % k: vO.astype(np.float32), "f%d" % k: fO.astype(np.int32), "a%d" % k: aO.astype(np.int32),
                    "vold%d" % k: vOld.astype(np.int32), "fold%d" % k: fOld.astype(np.int32),
                    "bb%d" % k: bb.astype(np.float32), "slice%d" % k: inside.astype(np.float32)})
    save("meshpatch_%s.npz" % tag, **out)


def random_klist(n, K, seed, zero_rows=(3,), dup=True):
    """Random one-indexed K-list with self slot, ragged degrees, duplicates and an isolated (all-zero) row."""
    rs = np.random.RandomState(seed)
    adj = np.zeros((n, K), dtype=np.int32)
    for i in range(n):
        d = rs.randint(0, K)  # number of neighbours besides self
        adj[i, 0] = i + 1
        nb = rs.randint(1, n + 1, size=d)
        if dup and d >= 2:
            nb[1] = nb[0]
        adj[i, 1:1 + d] = nb
    for r in zero_rows:
        adj[r, :] = 0
    # one saturated row
    adj[n - 1, :] = rs.randint(1, n + 1, size=K)
    adj[n - 1, 0] = n
    return adj


def main():
    only = sys.argv[1:]  # optional subset of parts

    def want(p):
        return not only or p in only

    V, F = icosphere(3)
    if not F64:
        x, adjs, gt, ds = preprocess_fixture("ico3", V, F, seed=0)
        Vt, Ft = torus(20, 16)
        xt, adjst, gtt, dst = preprocess_fixture("torus640", Vt, Ft, seed=5)
    else:
        z = np.load(os.path.join(OUT, "prep_ico3.npz"))
        x, adjs, gt = z["x"], [z["adj0"], z["adj1"], z["adj2"]], z["gt"]
        ds = types.SimpleNamespace(permutations=[z["permutations"]], num_faces=[int(z["num_faces"])])
        z = np.load(os.path.join(OUT, "prep_torus640.npz"))
        xt, adjst, gtt = z["x"], [z["adj0"], z["adj1"], z["adj2"]], z["gt"]

    if want("conv"):
        # C1: icosphere, one conv 6->32, raw K-list (duplicates, construction order) ...
        zz = np.load(os.path.join(OUT, "prep_ico3.npz"))
        feat = np.concatenate([zz["normals"], zz["centres"]], axis=1)[None]
        conv_case("c1_raw", feat, zz["fadj"][None], 32, 9, seed=11)
        # ... and the coarsened pipeline's level-0 adjacency (fake rows, ascending, merged)
        conv_case("c1_coarsened", x, adjs[0], 32, 9, seed=12)
        # odd shapes on random ragged K-lists
        rs = np.random.RandomState(7)
        conv_case("rand_5_7", rs.normal(size=(1, 40, 5)), random_klist(40, 23, 1)[None], 7, 9, seed=13)
        conv_case("rand_32_64", rs.normal(size=(1, 96, 32)), random_klist(96, 23, 2)[None], 64, 9, seed=14)
        conv_case("rand_128_64", rs.normal(size=(1, 64, 128)), random_klist(64, 23, 3)[None], 64, 9, seed=15)
        conv_case("rand_nomask", rs.normal(size=(1, 48, 16)), random_klist(48, 23, 4)[None], 8, 9, seed=16,
                  biasMask=False)
    if want("net"):
        net_case("ico3", x, adjs, gt, seed=0, multi_scale=False)
        if not F64:  # the float64 run (error-budget reference) is kept for one net only
            net_case("ico3", x, adjs, gt, seed=0, multi_scale=True)
            net_case("torus640", xt, adjst, gtt, seed=1, multi_scale=False)
    if want("net20k") and not F64:
        # the reference's own patch size (settings.py:20): ~800 level-0 tiles of 32 nodes, more workgroups than CUs
        net_case_patch_size("ico5_20k", 5, seed=3)
    if want("infer") and not F64:
        infer_case("ico3", x, adjs, ds, seed=0)
    if want("msvertex"):
        multiscale_vertex_case("ico3", add_noise(V, F), V, F, seed=0)
    if want("patch") and not F64:
        Vp, Fp = torus(24, 20)          # 960 faces, patches of 300 (min 120)
        patch_case("torus960", Fp, 300, 120, seed=11)
        # two disjoint components (a fresh random seed is needed when a component is exhausted)
        V2, F2 = icosphere(2)
        patch_case("two_spheres", np.concatenate([F2, F2 + V2.shape[0]]), 150, 60, seed=12)
    if want("meshpatch") and not F64:
        Vp, Fp = torus(24, 20)
        mesh_patch_case("torus960", Vp, Fp, [(300, 0), (300, 517), (100, 959), (2000, 3)])
        # open mesh with a ragged border + a second component
        Vt2, Ft2 = torus(20, 16)
        used, Fo = np.unique(Ft2[:450], return_inverse=True)
        V2, F2 = icosphere(1)
        Fo2 = np.concatenate([Fo.reshape(-1, 3), F2 + used.shape[0]]).astype(Ft2.dtype)
        Vo2 = np.concatenate([Vt2[used], V2 + 5.0])
        mesh_patch_case("open_plus_sphere", Vo2, Fo2, [(200, 10), (60, 455), (450, 449)])
    if want("vertex"):
        # closed mesh: noisy icosphere, target normals = normals of the clean sphere
        Vn = add_noise(V, F)
        vertex_case("ico3", Vn, F, ref_utils.computeFacesNormals(V, F))
        # open mesh (boundary edges have one face): the first 450 faces of the torus grid, vertices re-indexed
        # (getEdgeMap sizes its table by max(faces) + 1, i.e. assumes every vertex is used)
        Vt2, Ft2 = torus(20, 16)
        used, Fo = np.unique(Ft2[:450], return_inverse=True)
        Fo = Fo.reshape(-1, 3).astype(Ft2.dtype)
        Vo = Vt2[used]
        vertex_case("torus_open", add_noise(Vo, Fo), Fo, ref_utils.computeFacesNormals(Vo, Fo))


if __name__ == "__main__":
    main()
