"""Thin functional wrappers over the C ABI (one call = one enqueue on torch's current stream).

No autograd here (see model.py); tensors are fp32, contiguous, on the GPU.  Everything raises
if libfgc.so is missing: there is no eager/CPU fallback by design.
"""
import ctypes as C

import numpy as np
import torch

from . import _lib
from ._lib import ConvDesc, ConvBwdIO, ptr, stream_ptr, check, AG_LD, DL_LD, FGC_M

_WS = {}


def _workspace(nbytes, device, tag="ws"):
    """Grow-only scratch buffer per (device, tag); the library never allocates."""
    key = (str(device), tag)
    t = _WS.get(key)
    if t is None or t.numel() < nbytes:
        t = torch.empty(max(int(nbytes), 256), dtype=torch.uint8, device=device)
        _WS[key] = t
    return t


def _f32c(t):
    if t is None:
        return None
    if t.dtype != torch.float32 or not t.is_contiguous():
        t = t.contiguous().float()
    return t


def _req_cuda(*ts):
    for t in ts:
        if t is not None and not t.is_cuda:
            raise RuntimeError("facet_graph_convolution_amd ops need GPU tensors (no CPU fallback)")


def make_conv_desc(graph, x0, x1, shift, params, bias_mask, act, alpha):
    W0, b, u, c, v = params
    d = ConvDesc()
    d.n, d.nnz = graph.n, graph.nnz
    d.rowptr, d.col = graph.rowptr.data_ptr(), graph.col.data_ptr()
    d.x0 = x0.data_ptr()
    d.x1 = x1.data_ptr() if x1 is not None else None
    d.c0 = x0.shape[-1]
    d.c1 = x1.shape[-1] if x1 is not None else 0
    d.shift = shift
    d.cout = W0.shape[1]
    d.W0, d.b, d.u, d.c, d.v = W0.data_ptr(), b.data_ptr(), u.data_ptr(), c.data_ptr(), v.data_ptr()
    d.bias_mask = int(bool(bias_mask))
    d.act = int(act)
    d.alpha = float(alpha)
    d.max_deg = graph.max_deg
    return d


def attach_pairs(d, graph, hc):
    """Give a descriptor of a layer over a 4x-upsampled input its pair graph and the table of transformed coarse rows
    hc [(n >> 2), 9 * cout]: the library then runs the layer in the pair form if it qualifies (fgc_conv_uses_pairs)."""
    pg = graph.pairs()
    d.pair_rowptr, d.pair_col, d.pair_mul = pg.prow.data_ptr(), pg.pcol.data_ptr(), pg.pmul.data_ptr()
    d.n_pairs, d.max_pair_deg, d.max_pair_in_deg = pg.n_pairs, pg.max_deg, pg.max_in_deg
    d.hc = hc.data_ptr()
    return pg


def conv_fwd(graph, x0, x1, shift, params, bias_mask=True, act=0, alpha=0.1, want_pool=False, pairs=None):
    """Returns (y [n,cout], y_pool [n/4,cout] or None, ag [(n>>shift),24]).

    pairs: a dict - the layer (shift == 2) is described with its pair graph; on return pairs["hc"] holds the transformed
    coarse rows (pass the same dict to conv_bwd) and pairs["used"] whether the library took the pair form."""
    _req_cuda(x0, x1, *params)
    x0, x1 = _f32c(x0), _f32c(x1)
    params = [_f32c(p) for p in params]
    W0 = params[0]
    cin = x0.shape[-1] + (x1.shape[-1] if x1 is not None else 0)
    if W0.shape[0] != FGC_M or W0.shape[2] != cin:
        raise ValueError("W0 must be [9, cout, %d], got %s" % (cin, tuple(W0.shape)))
    rows = graph.n >> shift
    if x0.shape[0] != rows or (x1 is not None and x1.shape[0] != rows):
        raise ValueError("input rows %d != n>>shift = %d" % (x0.shape[0], rows))
    d = make_conv_desc(graph, x0, x1, shift, params, bias_mask, act, alpha)
    L = _lib.lib()
    dev = x0.device
    if pairs is not None:
        pairs["hc"] = torch.empty(rows, FGC_M * d.cout, dtype=torch.float32, device=dev)
        attach_pairs(d, graph, pairs["hc"])
        pairs["used"] = bool(L.fgc_conv_uses_pairs(C.byref(d)))
    ws_bytes = L.fgc_conv_workspace_bytes(C.byref(d))
    ws = _workspace(ws_bytes, dev)
    ag = torch.empty(rows, AG_LD, dtype=torch.float32, device=dev)
    y = torch.empty(graph.n, d.cout, dtype=torch.float32, device=dev)
    yp = torch.empty(graph.n // 4, d.cout, dtype=torch.float32, device=dev) if want_pool else None
    check(L.fgc_conv_fwd(C.byref(d), ptr(ag), ptr(y), ptr(yp), ptr(ws), ws.numel(), stream_ptr()), "fgc_conv_fwd")
    return y, yp, ag


def conv_bwd(graph, x0, x1, shift, params, ag, y, dy, bias_mask=True, act=0, alpha=0.1, need_dx=True,
             dx0=None, dx1=None, acc0=False, acc1=False, pairs=None):
    """Gradient of conv_fwd.  Returns (dx0, dx1, [dW0, db, du, dc, dv]).

    dx0/dx1 may be passed in (with acc flags) so that a tensor consumed by two layers
    accumulates both contributions in a fixed order; otherwise they are allocated.
    """
    _req_cuda(x0, x1, dy, *params)
    x0, x1, dy = _f32c(x0), _f32c(x1), _f32c(dy)
    y = _f32c(y)
    params = [_f32c(p) for p in params]
    d = make_conv_desc(graph, x0, x1, shift, params, bias_mask, act, alpha)
    L = _lib.lib()
    dev = x0.device
    n, cout = graph.n, d.cout
    trow, tcol, tedge = graph.transposed()
    f32 = dict(dtype=torch.float32, device=dev)
    ds = torch.empty(n, cout, **f32)
    dl = torch.empty(max(graph.nnz, 1), DL_LD, **f32)
    dag = torch.empty(n, AG_LD, **f32)
    r = torch.empty(n, FGC_M * cout + 24, **f32)
    grads = [torch.empty_like(p) for p in params]
    if need_dx:
        if dx0 is None:
            dx0 = torch.empty_like(x0)
            acc0 = False
        if x1 is not None and dx1 is None:
            dx1 = torch.empty_like(x1)
            acc1 = False
    else:
        dx0 = dx1 = None
    io = ConvBwdIO()
    io.trowptr, io.tcol, io.tedge = trow.data_ptr(), tcol.data_ptr(), tedge.data_ptr()
    io.max_in_deg = graph.max_in_deg
    io.ag, io.y, io.dy = ag.data_ptr(), (y.data_ptr() if y is not None else None), dy.data_ptr()
    io.ds, io.dl, io.dag, io.r = ds.data_ptr(), dl.data_ptr(), dag.data_ptr(), r.data_ptr()
    io.dx0 = dx0.data_ptr() if dx0 is not None else None
    io.dx1 = dx1.data_ptr() if dx1 is not None else None
    io.accumulate0, io.accumulate1 = int(bool(acc0)), int(bool(acc1))
    io.dW0, io.db, io.du, io.dc, io.dv = [g.data_ptr() for g in grads]
    if pairs is not None:
        pg = attach_pairs(d, graph, pairs["hc"])
        io.tpair_rowptr, io.tpair_col, io.tpair_edge = pg.trow.data_ptr(), pg.tcol.data_ptr(), pg.tedge.data_ptr()
        dt = torch.empty(max(pg.n_pairs, 1), cout, **f32)
        if dl.shape[0] < pg.n_pairs:
            dl = torch.empty(pg.n_pairs, DL_LD, **f32)
            io.dl = dl.data_ptr()
        io.dt = dt.data_ptr()
    ws_bytes = L.fgc_conv_bwd_workspace_bytes(C.byref(d))
    ws = _workspace(ws_bytes, dev, "bwd")
    check(L.fgc_conv_bwd(C.byref(d), C.byref(io), ptr(ws), ws.numel(), stream_ptr()), "fgc_conv_bwd")
    return dx0, dx1, grads


def mlp_fwd(x, W1, b1, W2, b2, alpha=0.1, want_abs_partial=False):
    _req_cuda(x, W1, b1, W2, b2)
    x, W1, b1, W2, b2 = map(_f32c, (x, W1, b1, W2, b2))
    n, cin = x.shape
    hidden, cout = W2.shape
    L = _lib.lib()
    ws = _workspace(L.fgc_mlp_workspace_bytes(cin, hidden, cout), x.device)
    y = torch.empty(n, cout, dtype=torch.float32, device=x.device)
    part = torch.empty(L.fgc_mlp_num_partials(n), dtype=torch.float32, device=x.device) if want_abs_partial else None
    check(L.fgc_mlp_fwd(ptr(x), n, cin, hidden, cout, ptr(W1), ptr(b1), ptr(W2), ptr(b2), alpha, ptr(y), ptr(part),
                        0, ptr(ws), ws.numel(), stream_ptr()), "fgc_mlp_fwd")
    return (y, part) if want_abs_partial else y


def mlp_bwd(x, dy, W1, b1, W2, alpha=0.1):
    _req_cuda(x, dy, W1, b1, W2)
    x, dy, W1, b1, W2 = map(_f32c, (x, dy, W1, b1, W2))
    n, cin = x.shape
    hidden, cout = W2.shape
    L = _lib.lib()
    ws = _workspace(L.fgc_mlp_bwd_workspace_bytes(n, cin, hidden, cout), x.device, "bwd")
    dx = torch.empty_like(x)
    dW1, db1, dW2 = torch.empty_like(W1), torch.empty_like(b1), torch.empty_like(W2)
    db2 = torch.empty(cout, dtype=torch.float32, device=x.device)
    check(L.fgc_mlp_bwd(ptr(x), ptr(dy), n, cin, hidden, cout, ptr(W1), ptr(b1), ptr(W2), alpha, ptr(dx), ptr(dW1),
                        ptr(db1), ptr(dW2), ptr(db2), 0, ptr(ws), ws.numel(), stream_ptr()), "fgc_mlp_bwd")
    return dx, dW1, db1, dW2, db2


def lrelu_fwd(x, alpha):
    x = _f32c(x)
    y = torch.empty_like(x)
    check(_lib.lib().fgc_lrelu_fwd(ptr(x), ptr(y), x.numel(), alpha, stream_ptr()), "fgc_lrelu_fwd")
    return y


def lrelu_bwd(y, dy, alpha):
    y, dy = _f32c(y), _f32c(dy)
    dx = torch.empty_like(y)
    check(_lib.lib().fgc_lrelu_bwd(ptr(y), ptr(dy), ptr(dx), y.numel(), alpha, stream_ptr()), "fgc_lrelu_bwd")
    return dx


def pool4_fwd(x):
    x = _f32c(x)
    n, c = x.shape
    if n % 4:
        raise ValueError("pooling needs a multiple of 4 rows, got %d" % n)
    y = torch.empty(n // 4, c, dtype=torch.float32, device=x.device)
    check(_lib.lib().fgc_pool4_fwd(ptr(x), ptr(y), n // 4, c, stream_ptr()), "fgc_pool4_fwd")
    return y


def pool4_bwd(x, y, dy, dx=None, accumulate=False):
    x, y, dy = _f32c(x), _f32c(y), _f32c(dy)
    if dx is None:
        dx = torch.empty_like(x)
        accumulate = False
    check(_lib.lib().fgc_pool4_bwd(ptr(x), ptr(y), ptr(dy), ptr(dx), y.shape[0], y.shape[1], int(accumulate),
                                   stream_ptr()), "fgc_pool4_bwd")
    return dx


def upsample4_fwd(x):
    x = _f32c(x)
    n, c = x.shape
    y = torch.empty(n * 4, c, dtype=torch.float32, device=x.device)
    check(_lib.lib().fgc_upsample4_fwd(ptr(x), ptr(y), n, c, stream_ptr()), "fgc_upsample4_fwd")
    return y


def upsample4_bwd(dy, dx=None, accumulate=False):
    dy = _f32c(dy)
    n4, c = dy.shape
    if dx is None:
        dx = torch.empty(n4 // 4, c, dtype=torch.float32, device=dy.device)
        accumulate = False
    check(_lib.lib().fgc_upsample4_bwd(ptr(dy), ptr(dx), n4 // 4, c, int(accumulate), stream_ptr()),
          "fgc_upsample4_bwd")
    return dx


def pool_fwd(x, group):
    """max over each `group` consecutive rows (custom_binary_tree_pooling with group = 2^steps, model.py:779-788)."""
    x = _f32c(x)
    n, c = x.shape
    if n % group:
        raise ValueError("pooling needs a multiple of %d rows, got %d" % (group, n))
    y = torch.empty(n // group, c, dtype=torch.float32, device=x.device)
    check(_lib.lib().fgc_pool_fwd(ptr(x), ptr(y), n // group, c, int(group), stream_ptr()), "fgc_pool_fwd")
    return y


def pool_bwd(x, y, dy, group):
    x, y, dy = _f32c(x), _f32c(y), _f32c(dy)
    dx = torch.empty_like(x)
    check(_lib.lib().fgc_pool_bwd(ptr(x), ptr(y), ptr(dy), ptr(dx), y.shape[0], y.shape[1], int(group), 0, stream_ptr()),
          "fgc_pool_bwd")
    return dx


def upsample_fwd(x, group):
    """every row repeated `group` times consecutively (custom_upsampling with group = 2^steps, model.py:817-825)."""
    x = _f32c(x)
    n, c = x.shape
    y = torch.empty(n * group, c, dtype=torch.float32, device=x.device)
    check(_lib.lib().fgc_upsample_fwd(ptr(x), ptr(y), n, c, int(group), stream_ptr()), "fgc_upsample_fwd")
    return y


def upsample_bwd(dy, group):
    dy = _f32c(dy)
    n, c = dy.shape
    dx = torch.empty(n // group, c, dtype=torch.float32, device=dy.device)
    check(_lib.lib().fgc_upsample_bwd(ptr(dy), ptr(dx), n // group, c, int(group), 0, stream_ptr()), "fgc_upsample_bwd")
    return dx


def lin_fwd(x, W, b):
    """custom_lin (model.py:763-769): x [n, cin] W [cin, cout] + b."""
    _req_cuda(x, W, b)
    x, W, b = _f32c(x), _f32c(W), _f32c(b)
    n, cin = x.shape
    cout = W.shape[1]
    if W.shape[0] != cin or b.shape[0] != cout:
        raise ValueError("custom_lin: x %s, W %s, b %s" % (tuple(x.shape), tuple(W.shape), tuple(b.shape)))
    y = torch.empty(n, cout, dtype=torch.float32, device=x.device)
    check(_lib.lib().fgc_lin_fwd(ptr(x), n, cin, cout, ptr(W), ptr(b), ptr(y), stream_ptr()), "fgc_lin_fwd")
    return y


def lin_bwd(x, dy, W, need_dx=True):
    """Returns (dx or None, dW, db)."""
    _req_cuda(x, dy, W)
    x, dy, W = _f32c(x), _f32c(dy), _f32c(W)
    n, cin = x.shape
    cout = W.shape[1]
    L = _lib.lib()
    ws = _workspace(L.fgc_lin_bwd_workspace_bytes(n, cin, cout), x.device, "lin")
    dx = torch.empty_like(x) if need_dx else None
    dW = torch.empty_like(W)
    db = torch.empty(cout, dtype=torch.float32, device=x.device)
    check(L.fgc_lin_bwd(ptr(x), ptr(dy), n, cin, cout, ptr(W), ptr(dx), ptr(dW), ptr(db), ptr(ws), ws.numel(), stream_ptr()),
          "fgc_lin_bwd")
    return dx, dW, db


def normalize_fwd(x, abs_partial=None):
    """normalizeTensor on [n,3]; returns (y, scratch) - scratch feeds normalize_bwd."""
    x = _f32c(x)
    n = x.shape[0]
    L = _lib.lib()
    scratch = torch.empty(2 + L.fgc_norm_num_partials(n), dtype=torch.float32, device=x.device)
    y = torch.empty_like(x)
    npart = abs_partial.numel() if abs_partial is not None else 0
    check(L.fgc_normalize_fwd(ptr(x), n, ptr(abs_partial), npart, ptr(y), ptr(scratch), stream_ptr()),
          "fgc_normalize_fwd")
    return y, scratch


def normalize_bwd(x, dy, scratch):
    x, dy = _f32c(x), _f32c(dy)
    dx = torch.empty_like(x)
    check(_lib.lib().fgc_normalize_bwd(ptr(x), ptr(dy), x.shape[0], ptr(dx), ptr(scratch), stream_ptr()),
          "fgc_normalize_bwd")
    return dx


def angular_loss_fwd(fn, gt, sample_ind):
    fn, gt = _f32c(fn), _f32c(gt)
    out = torch.empty(2, dtype=torch.float32, device=fn.device)
    check(_lib.lib().fgc_angular_loss_fwd(ptr(fn), ptr(gt), ptr(sample_ind), sample_ind.numel(), ptr(out),
                                          stream_ptr()), "fgc_angular_loss_fwd")
    return out


def angular_loss_bwd(fn, gt, sample_ind, loss_out, dloss=1.0):
    fn, gt = _f32c(fn), _f32c(gt)
    dfn = torch.empty_like(fn)
    check(_lib.lib().fgc_angular_loss_bwd(ptr(fn), ptr(gt), ptr(sample_ind), sample_ind.numel(), fn.shape[0],
                                          ptr(loss_out), float(dloss), ptr(dfn), stream_ptr()),
          "fgc_angular_loss_bwd")
    return dfn


def rotate_rows(x, R):
    """Every 3-vector v of every row becomes R v (train.py:439-451).  R: 3x3 array-like or device tensor."""
    x = _f32c(x)
    n, c = x.shape
    if c % 3:
        raise ValueError("channels must be a multiple of 3")
    if not isinstance(R, torch.Tensor) or not R.is_cuda:
        R = torch.as_tensor(np.asarray(R, dtype=np.float32).reshape(9)).to(x.device)
    R = R.reshape(9).float().contiguous()
    y = torch.empty_like(x)
    check(_lib.lib().fgc_rotate_rows(ptr(x), ptr(y), n, c // 3, ptr(R), stream_ptr()), "fgc_rotate_rows")
    return y


def adam_step(p, g, m, v, t, lr=1e-3, b1=0.9, b2=0.999, eps=1e-8):
    check(_lib.lib().fgc_adam_step(ptr(p), ptr(g), ptr(m), ptr(v), p.numel(), int(t), lr, b1, b2, eps,
                                   stream_ptr()), "fgc_adam_step")


def infer_epilogue(n_conv, perm, num_faces):
    n_conv = _f32c(n_conv)
    out = torch.empty(num_faces, 3, dtype=torch.float32, device=n_conv.device)
    check(_lib.lib().fgc_infer_epilogue(ptr(n_conv), ptr(perm), num_faces, ptr(out), stream_ptr()),
          "fgc_infer_epilogue")
    return out


def gather_rows(src, idx):
    src = _f32c(src)
    dst = torch.empty(idx.numel(), src.shape[1], dtype=torch.float32, device=src.device)
    check(_lib.lib().fgc_gather_rows(ptr(src), ptr(idx), idx.numel(), src.shape[1], ptr(dst), stream_ptr()),
          "fgc_gather_rows")
    return dst


def scatter_add_rows(src, idx, dst):
    src = _f32c(src)
    check(_lib.lib().fgc_scatter_add_rows(ptr(src), ptr(idx), idx.numel(), src.shape[1], ptr(dst), stream_ptr()),
          "fgc_scatter_add_rows")
    return dst


def vertex_update(x, normals, e_map, v_e_map, iters, lmbd=1.0 / 18):
    """update_position2 (train.py:1467-1557) on [V,3] positions; e_map int32 [E,4], v_e_map int32 [V,max_edges]."""
    _req_cuda(x, normals, e_map, v_e_map)
    x, normals = _f32c(x), _f32c(normals)
    e_map = e_map.to(torch.int32).contiguous()
    v_e_map = v_e_map.to(torch.int32).contiguous()
    out, tmp = torch.empty_like(x), torch.empty_like(x)
    check(_lib.lib().fgc_vertex_update(ptr(x), ptr(out), ptr(tmp), x.shape[0], ptr(normals), normals.shape[0],
                                       ptr(e_map), e_map.shape[0], ptr(v_e_map), v_e_map.shape[1], int(iters),
                                       float(lmbd), stream_ptr()), "fgc_vertex_update")
    return out


def pool4_avg_iz(x):
    """custom_binary_tree_pooling(x, steps=2, 'avg_ignore_zeros') (model.py:792-814) on [n, c] rows."""
    _req_cuda(x)
    x = _f32c(x)
    n, c = x.shape
    y = torch.empty(n // 4, c, dtype=torch.float32, device=x.device)
    check(_lib.lib().fgc_pool4_avg_iz(ptr(x), n, c, ptr(y), stream_ptr()), "fgc_pool4_avg_iz")
    return y


def face_centers(x, faces):
    """Barycentres of faces [n0,3] (int, -1 = fake) of the vertices x [V,3] (train.py:1779-1787)."""
    _req_cuda(x, faces)
    x = _f32c(x)
    faces = faces.to(torch.int32).contiguous()
    out = torch.empty(faces.shape[0], 3, dtype=torch.float32, device=x.device)
    check(_lib.lib().fgc_face_centers(ptr(x), x.shape[0], ptr(faces), faces.shape[0], ptr(out), stream_ptr()),
          "fgc_face_centers")
    return out


def vertex_update_ms(x, normals, faces, v_faces, iters=(80, 20, 20)):
    """update_position_MS (train.py:1668-1798), coarsening_steps = 2.  x [V,3]; normals = [n0 [N0,3], n1 [N0/4,3],
    n2 [N0/16,3]]; faces int [N0,3] (-1 rows = fake nodes); v_faces int [V,K].  Returns (x_out [V,3], dx [3,V,3])."""
    _req_cuda(x, faces, v_faces, *normals)
    x = _f32c(x)
    n0, n1, n2 = (_f32c(t.reshape(-1, 3)) for t in normals)
    faces = faces.to(torch.int32).contiguous()
    v_faces = v_faces.to(torch.int32).contiguous()
    nv, N0 = x.shape[0], faces.shape[0]
    if n0.shape[0] != N0 or n1.shape[0] * 4 != N0 or n2.shape[0] * 16 != N0:
        raise ValueError("normals must have N0, N0/4 and N0/16 rows (N0 = %d)" % N0)
    out = torch.empty_like(x)
    dx = torch.empty(3, nv, 3, dtype=torch.float32, device=x.device)
    nscr = 3 * (2 * nv + N0 + N0 // 4 + N0 // 16)
    scr = torch.empty(nscr, dtype=torch.float32, device=x.device)
    it = (C.c_int32 * 3)(*[int(i) for i in iters])
    check(_lib.lib().fgc_vertex_update_ms(ptr(x), ptr(out), nv, ptr(faces), N0, ptr(v_faces), v_faces.shape[1], ptr(n0),
                                          ptr(n1), ptr(n2), it, ptr(dx), ptr(scr), nscr, stream_ptr()),
          "fgc_vertex_update_ms")
    return out, dx
