"""CPU tests of the facet-sharding plan and of the exchange back end (gloo, world size 2)."""
import os
import sys

import numpy as np
import pytest
import torch

from facet_graph_convolution_amd.graph import csr_from_klist
from facet_graph_convolution_amd.shard import ShardPlan


def _graphs(golden_dir, tag="ico3"):
    z = np.load(os.path.join(golden_dir, "prep_%s.npz" % tag))
    return [csr_from_klist(z["adj%d" % l]) for l in range(3)], z


@pytest.mark.parametrize("world", [2, 3, 4, 8])
def test_plan_is_consistent(golden_dir, world):
    gh, _ = _graphs(golden_dir)
    plans = [ShardPlan(gh, r, world) for r in range(world)]
    for l in range(3):
        rowptr, col = gh[l]
        n = len(rowptr) - 1
        src = np.repeat(np.arange(n), np.diff(rowptr))
        # ownership is a partition, aligned so that pooling / upsampling stay local
        assert sum(p.levels[l].n_own for p in plans) == n
        for p in plans:
            P = p.levels[l]
            assert P.lo % (4 ** (2 - l)) == 0 and P.n_own % (4 ** (2 - l)) == 0
            ids = p.local_rows(l)
            # local CSR reproduces the global rows
            for i in range(0, P.n_own, max(1, P.n_own // 17)):
                glob = col[rowptr[P.lo + i]:rowptr[P.lo + i + 1]]
                loc = P.col[P.rowptr[i]:P.rowptr[i + 1]]
                assert np.array_equal(ids[loc], glob)
                if l < 2:
                    # the variant for a 4x-upsampled coarse source: an owned node keeps its id (id >> 2 = its parent's local
                    # row), a halo node points at the tail row of its parent among the UNIQUE halo parents
                    up = P.col_up[P.rowptr[i]:P.rowptr[i + 1]]
                    own = loc < P.n_own
                    assert np.array_equal(up[own], loc[own]) and (up[~own] % 4 == 0).all()
                    assert np.array_equal(P.pair.halo_ids[up[~own] // 4 - P.n_own // 4], ids[loc[~own]] >> 2)
            # interior / boundary tiles: a partition of the 32-row tiles; interior ones touch owned rows only
            for rp, cc, ti, tb in ((P.rowptr, P.col, P.tiles_int, P.tiles_bnd),
                                   (P.trowptr, P.tcol, P.ttiles_int, P.ttiles_bnd)):
                ntiles = (P.n_own + 31) // 32
                assert np.array_equal(np.sort(np.concatenate([ti, tb])), np.arange(ntiles))
                for t in ti:
                    assert (cc[rp[32 * t]:rp[min(32 * t + 32, P.n_own)]] < P.n_own).all()
                for t in tb:
                    assert (cc[rp[32 * t]:rp[min(32 * t + 32, P.n_own)]] >= P.n_own).any()
            # what peers send me, in their order, is exactly my halo
            got = []
            for q in range(world):
                Q = plans[q].levels[l]
                got.append(Q.send_rows[p.rank] + Q.lo)
                assert len(Q.send_rows[p.rank]) == P.recv_counts[q]
            assert np.array_equal(np.concatenate(got), P.halo_ids)
            if l < 2:
                # the level's pair graph, sharded over the coarse ranges: local pairs = the global pairs of the owned blocks
                # (columns renamed), tail = the unique parents of the halo, and what peers send is exactly that tail
                from facet_graph_convolution_amd.graph import pair_graph
                prow, pcol, pmul = pair_graph(rowptr, col)
                PP = P.pair
                assert PP.n_own == P.n_own // 4 and np.array_equal(PP.halo_ids, np.unique(P.halo_ids >> 2))
                cids = np.concatenate([np.arange(PP.lo, PP.hi), PP.halo_ids])
                e0 = prow[PP.lo]
                assert np.array_equal(cids[PP.col], pcol[e0:e0 + PP.nnz]) and np.array_equal(PP.pmul, pmul[e0:e0 + PP.nnz])
                gotp = np.concatenate([plans[q].levels[l].pair.send_rows[p.rank] + plans[q].levels[l].pair.lo
                                       for q in range(world)])
                assert np.array_equal(gotp, PP.halo_ids)
                # every in-pair of an owned coarse row is there, the cross ones behind the owned pairs in the senders' order
                assert len(PP.tedge) == int(((pcol >= PP.lo) & (pcol < PP.hi)).sum())
                sent = sum(len(plans[q].levels[l].pair.send_edges[p.rank]) for q in range(world))
                assert sent == PP.n_cross_in and PP.tedge.max(initial=-1) < PP.nnz + PP.n_cross_in
            # transposed CSR: every in-edge of an owned node, in global edge order, owned or cross
            gl_in = np.where((col >= P.lo) & (col < P.hi))[0]
            assert len(P.tedge) == len(gl_in)
            cross_sent = np.concatenate([plans[q].levels[l].send_edges[p.rank] + rowptr[plans[q].levels[l].lo]
                                         for q in range(world)]) if world > 1 else np.zeros(0, np.int64)
            assert len(cross_sent) == P.n_cross_in
            for j in range(0, P.n_own, max(1, P.n_own // 13)):
                e_glob = np.sort(np.where(col == P.lo + j)[0])
                te = P.tedge[P.trowptr[j]:P.trowptr[j + 1]]
                tc = P.tcol[P.trowptr[j]:P.trowptr[j + 1]]
                assert np.array_equal(ids[tc], src[e_glob])
                back = np.where(te < P.nnz, te + rowptr[P.lo], cross_sent[np.maximum(te - P.nnz, 0)])
                assert np.array_equal(back, e_glob)


def _worker(rank, world, port, golden_dir, out):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from facet_graph_convolution_amd.shard import DistComm
        gh, z = _graphs(golden_dir)
        plan = ShardPlan(gh, rank, world)
        comm = DistComm()
        ok = True
        # the grouped form a step uses: the rows of two tensors (two levels, different widths) for every peer packed into
        # ONE all-to-all (shard.PackedExchange).  The product runs the pack / unpack job lists through
        # fgc_copy_rows_jobs on the GPU; here a torch restatement of the same lists moves CPU tensors, so that layout,
        # split sizes and the gloo transport are checked without a GPU.  A "feature" encodes the global row id: after
        # the exchange every halo row must hold its own id.
        from facet_graph_convolution_amd.shard import PackedExchange
        ts, blocks = [], []
        for l in (0, 1, 2):
            P = plan.levels[l]
            ids = plan.local_rows(l)
            t = torch.zeros(len(ids), 2 + l)
            t[:P.n_own, 0] = torch.arange(P.lo, P.hi, dtype=torch.float32)
            t[:P.n_own, 1] = 7.0
            idx = torch.from_numpy(np.concatenate(P.send_rows).astype(np.int32))
            blocks.append((t, idx, P.send_counts, t[P.n_own:], P.recv_counts))
            ts.append((t, ids))
        px = PackedExchange(blocks, world)
        assert sum(px.send_splits) == sum(sum(b[2]) * b[0].shape[1] for b in blocks)
        for j in px.pack_jobs:
            rows = j["src"][j["idx"][j["idx_off"]:j["idx_off"] + j["rows"]].long()]
            j["dst"][j["dst_off"]:j["dst_off"] + j["rows"] * j["width"]] = rows.reshape(-1)
        comm.all_to_all_flat(px.send_buf, px.send_splits, px.recv_buf, px.recv_splits)
        for j in px.unpack_jobs:
            j["dst"][j["dst_row"]:j["dst_row"] + j["rows"]] = \
                j["src"][j["src_off"]:j["src_off"] + j["rows"] * j["width"]].view(j["rows"], j["width"])
        for t, ids in ts:
            ok &= bool(torch.equal(t[:, 0], torch.from_numpy(ids).float()))
            ok &= bool((t[:, 1] == 7.0).all())
        comm.host_staged = True
        s = torch.tensor([float(rank + 1)])
        comm.all_reduce_sum(s)
        ok &= s.item() == world * (world + 1) / 2
        out.put((rank, ok))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 8])
def test_gloo_halo_exchange(golden_dir, world):
    """World 2, and world 8 = the rank count of the driver's scaling run (8 processes over gloo on the CPU)."""
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 2000) + world
    procs = [ctx.Process(target=_worker, args=(r, world, port, golden_dir, q)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(timeout=240)
        assert p.exitcode == 0
    res = dict(q.get(timeout=5) for _ in range(world))
    assert res == {r: True for r in range(world)}


def test_a_rank_without_a_halo_still_issues_the_collective():
    """Whether to call a collective is never a rank-local decision (a shard whose level has no halo - a disconnected
    component - would otherwise skip an all-to-all its peers issue).  Only a world of one skips."""
    from facet_graph_convolution_amd.shard import DistComm

    class FakeDist:
        def __init__(self):
            self.calls = 0

        def all_to_all_single(self, recv, send, rs, ss, group=None, async_op=False):
            self.calls += 1
            return "work"

    for world, expect in ((1, 0), (2, 1), (8, 1)):
        c = DistComm.__new__(DistComm)
        c.dist, c.group, c.world, c.rank, c.host_staged = FakeDist(), None, world, 0, False
        out = c.all_to_all_flat(torch.empty(0), [0] * world, torch.empty(0), [0] * world, async_op=True)
        assert c.dist.calls == expect and (out == "work") == bool(expect)


def test_sim_run_serves_waits_per_rank_and_delivers_at_the_wait():
    """shard.sim_run: ranks may disagree on WHERE they wait (one overlaps, its peer blocks), not on the collectives; an
    overlapped exchange lands at the wait and its destination rows are NaN until then."""
    from facet_graph_convolution_amd import shard

    class FakePx:
        direct = False

        def __init__(self, log, name):
            self.log, self.name = log, name
            self.send_splits = self.recv_splits = [0, 0]
            self.send_buf = self.recv_buf = torch.zeros(0)

        def pack(self):
            self.log.append((self.name, "pack"))

        def unpack(self):
            self.log.append((self.name, "unpack"))

        def poison_tails(self):
            self.log.append((self.name, "poison"))

    log = []

    class FakeNet:
        def __init__(self, name):
            self.px = FakePx(log, name)

        def _packed(self, req):
            return self.px

    def overlapping(net):
        yield ("xchg", [], "k")
        log.append((net.px.name, "interior"))
        yield ("wait", "k")
        log.append((net.px.name, "boundary"))

    def blocking(net):
        yield ("xchg", [], None)
        log.append((net.px.name, "whole"))

    nets = [FakeNet("a"), FakeNet("b")]
    gens = {"a": overlapping, "b": blocking}
    shard.sim_run(nets, lambda n: gens[n.px.name](n))
    a = [e for n, e in log if n == "a"]
    b = [e for n, e in log if n == "b"]
    assert a == ["pack", "poison", "interior", "unpack", "boundary"] and b == ["pack", "unpack", "whole"]

    def never_waits(net):
        yield ("xchg", [], "k")

    with pytest.raises(AssertionError):
        shard.sim_run([FakeNet("a")], never_waits)


def test_pair_form_is_a_job_wide_decision():
    """A mesh on which exactly ONE shard has a coarse row with more in-pairs than the data-gradient kernel has edge slots
    (24): a rank-local test would put that rank's up-convolution in the fine form and its peer's in the pair form, and the
    two forms exchange different tensors in the backward pass.  shard.pair_form_allowed looks at the GLOBAL pair graph, which
    every rank has: the same answer on every rank, and 'no' as soon as any shard would have to refuse."""
    from facet_graph_convolution_amd.dataClasses import InferenceMesh
    from facet_graph_convolution_amd.meshgen import torus, flip_edges, add_noise
    from facet_graph_convolution_amd.shard import pair_form_allowed
    PAIR_KMAX = 24          # edge slots of the data-gradient kernel (csrc: KMAX); the library's own test is what decides
    V, F = torus(60, 50)
    F = flip_edges(F, 9000, seed=1)
    ds = InferenceMesh()
    ds.addMesh(add_noise(V, F, 0.2, seed=1), F, seed=0)
    gh = [csr_from_klist(a) for a in ds.adj_list[0]]
    plans = [ShardPlan(gh, r, 2) for r in range(2)]
    found = False
    for l in (0, 1):
        local = [p.levels[l].pair.max_in_deg for p in plans]
        glob = [p.levels[l].pair.global_max_in_deg for p in plans]
        assert glob[0] == glob[1] == max(local)
        answers = [pair_form_allowed(p.levels[l].pair, 32) for p in plans]
        assert answers[0] == answers[1] == (glob[0] <= PAIR_KMAX)
        if min(local) <= PAIR_KMAX < max(local):
            found = True                  # the case a per-rank decision gets wrong
            assert answers == [False, False]
    assert found, "the mesh no longer has a level on which the shards disagree: pick another seed"
    # the decision IS the library's (fgc_conv_pairs_allowed, the function fgc_conv_uses_pairs applies to a descriptor's own
    # counts): its limits, swept
    from facet_graph_convolution_amd import _lib
    allowed = _lib.lib().fgc_conv_pairs_allowed
    for rows, pairs, indeg, cout, want in [(1000, 7000, 24, 64, 1), (1000, 7000, 25, 64, 0), (1 << 24, 7000, 8, 32, 0),
                                           ((1 << 24) - 1, 7000, 8, 32, 0),        # rows * 9 * cout * 4 >= 2^32
                                           (1000, (1 << 24) - (1 << 20), 8, 32, 0), (1000, (1 << 24) - (1 << 20) - 1, 8, 32, 1),
                                           (3_000_000, 16_000_000, 8, 64, 0), (1_800_000, 14_000_000, 8, 64, 1), (-1, 5, 8, 32, 0)]:
        assert allowed(rows, pairs, indeg, cout) == want, (rows, pairs, indeg, cout)
    # a regular mesh: allowed on every rank
    gh, _ = _graphs_of_torus()
    for r in range(2):
        P = ShardPlan(gh, r, 2)
        assert all(pair_form_allowed(P.levels[l].pair, 64) for l in (0, 1))


def _graphs_of_torus():
    from facet_graph_convolution_amd.dataClasses import InferenceMesh
    from facet_graph_convolution_amd.meshgen import torus, add_noise
    V, F = torus(40, 32)
    ds = InferenceMesh()
    ds.addMesh(add_noise(V, F, 0.2, seed=1), F, seed=0)
    return [csr_from_klist(a) for a in ds.adj_list[0]], ds
