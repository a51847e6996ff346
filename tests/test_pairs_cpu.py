"""CPU tests of the pair graph (libfgc host routine fgc_pair_graph) against the numpy restatement in oracle/prep_ref.py."""
import numpy as np
import pytest

from facet_graph_convolution_amd import graph
from oracle import prep_ref


def _random_klist(n, rs, kmax=13, window=40, isolated=0.1):
    adj = np.zeros((n, 23), dtype=np.int32)
    for i in range(n):
        if rs.uniform() < isolated:
            continue                      # a row without any slot (deg 0)
        d = rs.randint(0, kmax)
        adj[i, 0] = i + 1
        nb = np.clip(i + rs.randint(-window, window + 1, size=d), 0, n - 1)
        adj[i, 1:1 + d] = nb + 1
    return adj


@pytest.mark.parametrize("n,seed", [(16, 0), (148, 1), (1024, 2)])
def test_pair_graph_matches_the_numpy_restatement(n, seed):
    rs = np.random.RandomState(seed)
    rowptr, col = graph.csr_from_klist(_random_klist(n, rs))
    prow, pcol, pmul = graph.pair_graph(rowptr, col)
    rrow, rcol, rmul = prep_ref.pair_graph_ref(rowptr, col)
    assert np.array_equal(prow, rrow) and np.array_equal(pcol, rcol) and np.array_equal(pmul, rmul)
    # every edge is in exactly one pair: multiplicities add up to the degrees
    deg = np.diff(rowptr).reshape(-1, 4)
    got = np.zeros_like(deg)
    blk = np.repeat(np.arange(n // 4), np.diff(prow))
    for ch in range(4):
        np.add.at(got[:, ch], blk, (pmul >> (8 * ch)) & 255)
    assert np.array_equal(got, deg)


def test_pair_graph_of_a_preprocessed_mesh_is_the_next_level():
    """On a mesh coarsened by the reference's procedure the pairs of level l are the edges of level l + 1 (plus the
    self pair): the coarse graph connects two clusters iff any of their members are adjacent (coarsening.py:16-31)."""
    from facet_graph_convolution_amd.dataClasses import TrainingSet
    from facet_graph_convolution_amd.meshgen import icosphere, add_noise
    V, F = icosphere(3)
    ds = TrainingSet()
    ds.addMeshWithGT(add_noise(V, F), F, V, seed=0)
    adjs = ds.adj_list[0]
    for lvl in (0, 1):
        rowptr, col = graph.csr_from_klist(adjs[lvl])
        prow, pcol, _ = graph.pair_graph(rowptr, col)
        r1, c1 = graph.csr_from_klist(adjs[lvl + 1])
        fine = set(zip(np.repeat(np.arange(len(prow) - 1), np.diff(prow)).tolist(), pcol.tolist()))
        coarse = set(zip(np.repeat(np.arange(len(r1) - 1), np.diff(r1)).tolist(), c1.tolist()))
        assert fine == coarse


def test_pair_graph_rejects_bad_input():
    rowptr = np.array([0, 1, 2, 3], dtype=np.int32)     # n = 3
    with pytest.raises(RuntimeError):
        graph.pair_graph(rowptr, np.zeros(3, np.int32))
