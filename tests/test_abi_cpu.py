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
    hdr = open(os.path.join(REPO, "include", "fgc.h")).read()
    assert _lib.lib().fgc_version() == _lib.ABI_VERSION == int(re.search(r"#define FGC_ABI_VERSION (\d+)", hdr).group(1))
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


def test_options_are_a_table_not_the_environment(monkeypatch):
    """fgc_set_option / fgc_get_option: process-level switches; the environment is read ONCE (initial values), no launch path
    reads it afterwards; unknown names are refused; the sources hold one getenv (the table's initialiser)."""
    import glob
    L = _lib.lib()
    names = _lib.option_names()
    assert "NO_PAIRS" in names and "W8_DATA16_MIN_N" in names and len(names) == L.fgc_option_count()
    assert _lib.get_option("FGC_W8_DATA16_MIN_N") == _lib.get_option("W8_DATA16_MIN_N")
    old = _lib.get_option("NO_PAIRS")
    monkeypatch.setenv("FGC_NO_PAIRS", str(1 - old))          # too late: the table was filled at the first read
    assert _lib.get_option("NO_PAIRS") == old
    with _lib.options(NO_PAIRS=1 - old):
        assert _lib.get_option("NO_PAIRS") == 1 - old
    assert _lib.get_option("NO_PAIRS") == old
    assert L.fgc_set_option(b"NO_SUCH_OPTION", 1) == -22 and b"NO_SUCH_OPTION" in L.fgc_last_error()
    n_getenv = sum(open(f).read().count("getenv(") for f in glob.glob(os.path.join(REPO, "facet_graph_convolution_amd", "csrc", "*.h*")))
    assert n_getenv == 1, n_getenv


def test_a_descriptor_can_carry_its_own_option_values():
    """fgc_conv_desc.options (round 6; the round-5 review's 'process-level options are global mutable state'): a list of
    fgc_option_override in a descriptor holds for the calls made WITH that descriptor, on the calling thread, and leaves the
    process-level table alone - two descriptors in one process answer fgc_conv_uses_pairs differently at the same time, also
    from two threads at once."""
    import ctypes as C
    import threading
    L = _lib.lib()
    assert L.fgc_struct_size(0) == C.sizeof(_lib.ConvDesc)

    def pair_desc():
        d = _lib.ConvDesc()
        fake = 4096                                   # (never dereferenced by the host-side test of the shape)
        d.n, d.nnz, d.rowptr, d.col, d.x0 = 64, 256, fake, fake, fake
        d.c0, d.c1, d.shift, d.cout = 64, 0, 2, 32
        d.W0 = d.b = d.u = d.c = d.v = fake
        d.pair_rowptr = d.pair_col = d.pair_mul = d.hc = fake
        d.n_pairs, d.max_pair_deg, d.max_pair_in_deg, d.max_deg = 100, 8, 8, 13
        return d

    plain, fine = pair_desc(), pair_desc()
    over = _lib.option_overrides(NO_PAIRS=1)
    fine.options, fine.n_options = C.addressof(over), len(over)
    old = _lib.get_option("NO_PAIRS")
    assert old == 0
    assert L.fgc_conv_uses_pairs(C.byref(plain)) == 1 and L.fgc_conv_uses_pairs(C.byref(fine)) == 0
    assert _lib.get_option("NO_PAIRS") == old            # nothing global was written
    # ... and the process-level switch still rules a descriptor without a list
    with _lib.options(NO_PAIRS=1):
        assert L.fgc_conv_uses_pairs(C.byref(plain)) == 0
        back = _lib.option_overrides(NO_PAIRS=0)
        on = pair_desc()
        on.options, on.n_options = C.addressof(back), 1
        assert L.fgc_conv_uses_pairs(C.byref(on)) == 1
    # two threads, two descriptors, at the same time: the scope is per thread
    out = {}

    def ask(name, d):
        out[name] = [L.fgc_conv_uses_pairs(C.byref(d)) for _ in range(20000)]
    ts = [threading.Thread(target=ask, args=("plain", plain)), threading.Thread(target=ask, args=("fine", fine))]
    [t.start() for t in ts]
    [t.join() for t in ts]
    assert set(out["plain"]) == {1} and set(out["fine"]) == {0}
    with pytest.raises(ValueError):
        _lib.option_overrides(NO_SUCH_OPTION=1)


def test_packed_operands_carry_the_identity_of_their_layout():
    """fgc_conv_layout_id / fgc_mlp_layout_id (ABI 104; the round-5 advisor's 'setting an option between fgc_conv_pack and the
    launch computes garbage silently'): the number changes with every option that changes what fgc_conv_pack writes - pair form
    or not, the d-logits operand as split planes or fp32, the first-layer path, the MLP's split forms - follows a
    descriptor's own overrides, and does not depend on the stored packed_layout.  (That an FGC_CONV_PACKED call refuses a stale
    value is a launch-path check: tests/test_gpu_conv.py.)"""
    L = _lib.lib()
    fake = 4096

    def desc(cin, cout, shift=0, pairs=False, n=640, max_deg=12):
        d = _lib.ConvDesc()
        d.n, d.nnz, d.rowptr, d.col, d.x0 = n, 12 * n, fake, fake, fake
        d.c0, d.c1, d.shift, d.cout, d.max_deg = cin, 0, shift, cout, max_deg
        d.W0 = d.b = d.u = d.c = d.v = fake
        if pairs:
            d.pair_rowptr = d.pair_col = d.pair_mul = d.hc = fake
            d.n_pairs, d.max_pair_deg, d.max_pair_in_deg = 100, 8, 8
        return d

    ident = lambda d: L.fgc_conv_layout_id(C.byref(d))
    up, deep, first = desc(64, 32, shift=2, pairs=True), desc(32, 32), desc(6, 32)
    base = [ident(d) for d in (up, deep, first)]
    assert all(b != 0 for b in base) and len(set(base)) == 3
    with _lib.options(NO_PAIRS=1):
        assert ident(up) != base[0] and ident(deep) == base[1]
    with _lib.options(NO_K1_SPLIT=1):
        assert ident(deep) != base[1] and ident(first) == base[2]
    with _lib.options(NO_NARROW=1):
        assert ident(first) != base[2]
    assert [ident(d) for d in (up, deep, first)] == base
    over = _lib.option_overrides(NO_PAIRS=1)
    up.options, up.n_options = C.addressof(over), 1
    assert ident(up) != base[0]
    up.options, up.n_options = None, 0
    up.packed_layout = 12345
    assert ident(up) == base[0]
    bf = desc(32, 32)
    bf.flags = _lib.CONV_BF16
    assert ident(bf) != base[1]
    # the MLP's: 1 ... 255, moves with the two split switches, one form for bf16 storage
    m = L.fgc_mlp_layout_id(32, 1024, 3, 0)
    assert 0 < m < 256 and 0 < L.fgc_mlp_layout_id(32, 1024, 3, 1) < 256
    with _lib.options(NO_MLP_BWD_SPLIT=1):
        m1 = L.fgc_mlp_layout_id(32, 1024, 3, 0)
    with _lib.options(NO_MLP_SPLIT=1):
        m2 = L.fgc_mlp_layout_id(32, 1024, 3, 0)
        assert L.fgc_mlp_layout_id(32, 1024, 3, 1) == L.fgc_mlp_layout_id(32, 1024, 3, 1)
    assert len({m, m1, m2}) == 3
    assert _lib.mlp_layout(m) == m << 8
