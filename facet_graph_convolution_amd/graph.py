"""Facet adjacency in the layout the kernels stream: CSR (+ transposed CSR for backward).

The reference feeds K-lists ``int [1, n, 23]`` (one-indexed, 0 = empty, slot 0 = self;
utils.py:243-295 / utils.py:1799-1827) through ``feed_dict`` (train.py:568-575).  They are
converted ONCE per mesh: slot order and duplicates are preserved, so
``deg = rowptr[i+1]-rowptr[i]`` equals ``count_nonzero`` (model.py:436) and the round
trip CSR -> K-list is bit-identical.
"""
import ctypes as C

import numpy as np
import torch

from . import _lib


def _as_klist(adj):
    a = adj.detach().cpu().numpy() if isinstance(adj, torch.Tensor) else np.asarray(adj)
    if a.ndim == 3:
        if a.shape[0] != 1:
            raise ValueError("batch size must be 1 (train.py:405), got %d" % a.shape[0])
        a = a[0]
    if a.ndim != 2:
        raise ValueError("adjacency must be [n, K] or [1, n, K]")
    return np.ascontiguousarray(a.astype(np.int32))


def csr_from_klist(adj):
    """numpy K-list -> (rowptr int32 [n+1], col int32 [nnz]) via the library's host routine."""
    a = _as_klist(adj)
    n, K = a.shape
    L = _lib.lib()
    rowptr = np.empty(n + 1, dtype=np.int32)
    nnz = C.c_int64(0)
    _lib.check(L.fgc_csr_from_klist(a.ctypes.data, n, K, rowptr.ctypes.data, None, C.byref(nnz)), "csr_from_klist")
    col = np.empty(max(nnz.value, 1), dtype=np.int32)
    _lib.check(L.fgc_csr_from_klist(a.ctypes.data, n, K, rowptr.ctypes.data, col.ctypes.data, C.byref(nnz)),
               "csr_from_klist")
    return rowptr, col[:nnz.value]


def klist_from_csr(rowptr, col, K):
    n = len(rowptr) - 1
    out = np.empty((n, K), dtype=np.int32)
    rowptr = np.ascontiguousarray(rowptr, dtype=np.int32)
    col = np.ascontiguousarray(col, dtype=np.int32)
    _lib.check(_lib.lib().fgc_klist_from_csr(rowptr.ctypes.data, col.ctypes.data, n, K, out.ctypes.data),
               "klist_from_csr")
    return out


def csr_transpose(rowptr, col):
    n = len(rowptr) - 1
    nnz = int(rowptr[-1])
    rowptr = np.ascontiguousarray(rowptr, dtype=np.int32)
    col = np.ascontiguousarray(col, dtype=np.int32)
    trow = np.empty(n + 1, dtype=np.int32)
    tcol = np.empty(max(nnz, 1), dtype=np.int32)
    tedge = np.empty(max(nnz, 1), dtype=np.int32)
    _lib.check(_lib.lib().fgc_csr_transpose(rowptr.ctypes.data, col.ctypes.data, n, trow.ctypes.data,
                                            tcol.ctypes.data, tedge.ctypes.data), "csr_transpose")
    return trow, tcol[:nnz], tedge[:nnz]


class FacetGraph:
    """One adjacency level resident in HBM: CSR and (lazily) its transpose."""

    def __init__(self, adj, device="cuda"):
        self.klist_shape = tuple(_as_klist(adj).shape)
        self.rowptr_h, self.col_h = csr_from_klist(adj)
        self.n = self.klist_shape[0]
        self.K = self.klist_shape[1]
        self.nnz = int(self.rowptr_h[-1])
        self.max_deg = int(np.diff(self.rowptr_h).max()) if self.n else 0
        self.max_in_deg = int(np.bincount(self.col_h, minlength=1).max()) if self.nnz else 0
        self.device = torch.device(device)
        self.rowptr = torch.from_numpy(self.rowptr_h).to(self.device)
        self.col = torch.from_numpy(np.ascontiguousarray(self.col_h) if self.nnz else np.zeros(1, np.int32)).to(
            self.device)
        self._t = None

    def transposed(self):
        if self._t is None:
            trow, tcol, tedge = csr_transpose(self.rowptr_h, self.col_h)
            pad = (lambda a: a if len(a) else np.zeros(1, np.int32))
            self._t = tuple(torch.from_numpy(np.ascontiguousarray(pad(a))).to(self.device)
                            for a in (trow, tcol, tedge))
        return self._t

    def to_klist(self):
        return klist_from_csr(self.rowptr_h, self.col_h, self.K)


_GRAPH_CACHE = {}


def as_graph(adj, device="cuda"):
    """Accept a FacetGraph, or a K-list tensor/array (converted once and cached by identity)."""
    if isinstance(adj, FacetGraph):
        return adj
    key = (id(adj), getattr(adj, "_version", 0))
    hit = _GRAPH_CACHE.get(key)
    if hit is not None and hit[0] is adj:
        return hit[1]
    g = FacetGraph(adj, device)
    if len(_GRAPH_CACHE) > 64:
        _GRAPH_CACHE.clear()
    _GRAPH_CACHE[key] = (adj, g)
    return g
