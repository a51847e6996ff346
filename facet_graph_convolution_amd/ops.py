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
    return d


def conv_fwd(graph, x0, x1, shift, params, bias_mask=True, act=0, alpha=0.1, want_pool=False):
    """Returns (y [n,cout], y_pool [n/4,cout] or None, ag [(n>>shift),24])."""
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
    ws_bytes = L.fgc_conv_workspace_bytes(C.byref(d))
    ws = _workspace(ws_bytes, dev)
    ag = torch.empty(rows, AG_LD, dtype=torch.float32, device=dev)
    y = torch.empty(graph.n, d.cout, dtype=torch.float32, device=dev)
    yp = torch.empty(graph.n // 4, d.cout, dtype=torch.float32, device=dev) if want_pool else None
    check(L.fgc_conv_fwd(C.byref(d), ptr(ag), ptr(y), ptr(yp), ptr(ws), ws.numel(), stream_ptr()), "fgc_conv_fwd")
    return y, yp, ag
