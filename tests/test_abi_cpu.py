"""CPU checks of the C-ABI boundary: the library loads, exports every symbol include/fgc.h declares, and its host
entry points validate their arguments (no GPU compute here)."""
import ctypes as C
import os
import re

import numpy as np
import pytest

from facet_graph_convolution_amd import _lib

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    src = open(os.path.join(REPO, "include", "fgc.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(fgc_[a-z0-9_]+)\s*\(", src)))


def test_header_symbols_are_exported_and_bound():
    names = _declared()
    assert len(names) >= 40
    L = C.CDLL(_lib.LIB_PATH)
    for n in names:
        assert hasattr(L, n), "libfgc.so does not export %s" % n
    assert set(names) == set(_lib.EXPORTS), set(names) ^ set(_lib.EXPORTS)
    assert _lib.lib().fgc_version() >= 100
    # the ctypes mirrors of the descriptor structs have the layout the library was compiled with
    assert _lib.lib().fgc_struct_size(0) == C.sizeof(_lib.ConvDesc)
    assert _lib.lib().fgc_struct_size(1) == C.sizeof(_lib.ConvBwdIO)
    assert _lib.lib().fgc_struct_size(2) == C.sizeof(_lib.PackExtra)


def test_error_reporting_across_the_boundary():
    L = _lib.lib()
    rc = L.fgc_csr_from_klist(None, 4, 23, None, None, None)
    assert rc == -22 and b"fgc_csr_from_klist" in L.fgc_last_error()
    with pytest.raises(RuntimeError, match="fgc_csr_from_klist"):
        _lib.check(rc, "fgc_csr_from_klist")
    assert L.fgc_conv_fwd(None, None, None, None, None, 0, None) == -22       # null descriptor, no launch
    assert L.fgc_mlp_fwd(None, 0, 0, 0, 0, None, None, None, None, 0.1, None, None, 0, None, 0, None) == -22


def test_missing_library_fails_loudly(monkeypatch, tmp_path):
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "LIB_PATH", str(tmp_path / "nope.so"))
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        _lib.lib()


def test_gpu_ops_refuse_cpu_tensors():
    import torch
    from facet_graph_convolution_amd import ops
    from facet_graph_convolution_amd.graph import FacetGraph
    adj = np.zeros((4, 23), dtype=np.int32)
    adj[:, 0] = np.arange(1, 5)
    g = FacetGraph(adj, "cpu")
    x = torch.zeros(4, 6)
    p = [torch.zeros(9, 8, 6), torch.zeros(8), torch.zeros(9, 6), torch.zeros(9), torch.zeros(9, 6)]
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        ops.conv_fwd(g, x, None, 0, p)
