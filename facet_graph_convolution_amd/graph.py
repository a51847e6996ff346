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


def pair_graph(rowptr, col):
    """Parent-compressed graph of a level read through a 4x upsampling (fgc_pair_graph, include/fgc.h):
    (prow int32 [n/4+1], pcol int32 [np], pmul uint32 [np])."""
    n = len(rowptr) - 1
    rowptr = np.ascontiguousarray(rowptr, dtype=np.int32)
    col = np.ascontiguousarray(col, dtype=np.int32)
    if len(col) == 0:
        col = np.zeros(1, np.int32)
    L = _lib.lib()
    prow = np.empty(n // 4 + 1, dtype=np.int32)
    cnt = C.c_int64(0)
    _lib.check(L.fgc_pair_graph(rowptr.ctypes.data, col.ctypes.data, n, prow.ctypes.data, None, None, C.byref(cnt)),
               "pair_graph")
    pcol = np.empty(max(cnt.value, 1), dtype=np.int32)
    pmul = np.empty(max(cnt.value, 1), dtype=np.uint32)
    _lib.check(L.fgc_pair_graph(rowptr.ctypes.data, col.ctypes.data, n, prow.ctypes.data, pcol.ctypes.data,
                                pmul.ctypes.data, C.byref(cnt)), "pair_graph")
    return prow, pcol[:cnt.value], pmul[:cnt.value]


class PairGraph:
    """The pair graph of a level and its transpose, resident in HBM."""

    def __init__(self, rowptr_h, col_h, device):
        prow, pcol, pmul = pair_graph(rowptr_h, col_h)
        self.n_pairs = len(pcol)
        nc = len(prow) - 1
        self.max_deg = int(np.diff(prow).max()) if nc else 0
        self.max_in_deg = int(np.bincount(pcol, minlength=1).max()) if self.n_pairs else 0
        trow, tcol, tedge = csr_transpose(prow, pcol)
        pad = (lambda a, dt: np.ascontiguousarray(a) if len(a) else np.zeros(1, dt))
        self.host = dict(prow=prow, pcol=pcol, pmul=pmul, trow=trow, tcol=tcol, tedge=tedge)
        up = lambda a: torch.from_numpy(a).to(device)
        self.prow, self.pcol = up(prow), up(pad(pcol, np.int32))
        self.pmul = up(pad(pmul, np.uint32).view(np.int32))
        self.trow, self.tcol, self.tedge = up(trow), up(pad(tcol, np.int32)), up(pad(tedge, np.int32))


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
        self._pairs = None

    def pairs(self):
        """The pair graph of this level (for a convolution whose input is a 4x-upsampled coarse tensor), built once."""
        if self._pairs is None:
            if self.n % 4:
                raise ValueError("a pair graph needs a multiple of 4 nodes, got %d" % self.n)
            self._pairs = PairGraph(self.rowptr_h, self.col_h, self.device)
        return self._pairs

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
