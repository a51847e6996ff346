"""Facet sharding: one mesh split over P ranks with a one-hop halo per graph level.

The reference has no multi-device path (single tf.Session, train.py:396); this is the MI355X-native design for
BASELINE configs 4-5.  Ownership follows the binary tree: rank r owns a contiguous range of COARSEST nodes and the
4x / 16x ranges under it, so pooling, upsampling and the skip concats stay local.  Because the coarsest level is
ordered along a Morton curve (csrc/fgc_prep.hip) a contiguous range is a compact surface patch and the halo is a
thin rim.  Per graph level and rank the plan holds

  * the owned row range [lo, hi) and the sorted global ids of the HALO rows (every remote node that is a
    neighbour or an in-neighbour of an owned node); local row index = id - lo for owned rows, n_own + position
    for halo rows, so a source tensor is simply [owned rows | halo rows] and the kernels need no change;
  * the local CSR over owned rows (two column variants: plain local ids, and "virtual" ids 4*(n_own/4 + h) for
    convolutions that read a 4x-upsampled coarse tensor whose tail rows hold the halo nodes' parents);
  * the local transposed CSR for backward, whose edge ids address [owned edges | incoming cross-shard edges];
  * send lists: which owned rows / owned edges every peer needs, in the order the peer stores them.

Exchanges per step (17 collectives): 7 grouped exchanges forward (everything a layer produces that peers gather, in one
message right behind the layer), 7 backward (per conv the s = dy/deg rows and the d-logits of cross edges together), two
scalar all-reduces for normalizeTensor and its gradient, one all-reduce of the flat gradient with the loss sum in its
tail.  A grouped exchange is one RCCL group of point-to-point sends / receives: over xGMI every peer pair has its own
link, so it is one hop, never a ring.
"""
import time

import numpy as np
import torch


def _owner_ranges(n2, world):
    """Contiguous split of the coarsest level; level l ranges are these times 4^(2-l)."""
    cuts = [(n2 * r) // world for r in range(world + 1)]
    return cuts


class LevelPlan:
    pass


TILE_ROWS = 32   # csrc/fgc_conv_core.h TILE


def _split_tiles(rowptr, edge_is_remote, n_own):
    """(interior, boundary) tile indices of a local CSR: a tile is boundary if any edge of its rows is remote."""
    ntiles = (n_own + TILE_ROWS - 1) // TILE_ROWS
    row_of_edge = np.repeat(np.arange(n_own, dtype=np.int64), np.diff(rowptr))
    bnd = np.zeros(ntiles, dtype=bool)
    bnd[row_of_edge[np.asarray(edge_is_remote, dtype=bool)] // TILE_ROWS] = True
    return np.nonzero(~bnd)[0].astype(np.int32), np.nonzero(bnd)[0].astype(np.int32)


def build_level_plan(rowptr, col, lo_hi, rank):
    """Plan of one graph level for `rank`.  rowptr/col: GLOBAL CSR (host numpy), lo_hi: list of (lo, hi) per rank."""
    world = len(lo_hi)
    lo, hi = lo_hi[rank]
    n = len(rowptr) - 1
    deg = np.diff(rowptr)
    src = np.repeat(np.arange(n, dtype=np.int64), deg)
    dst = col.astype(np.int64)
    eid = np.arange(len(col), dtype=np.int64)
    owner_of = np.zeros(n, dtype=np.int64)
    for r, (a, b) in enumerate(lo_hi):
        owner_of[a:b] = r

    P = LevelPlan()
    P.lo, P.hi, P.n_own = lo, hi, hi - lo
    e0, e1 = int(rowptr[lo]), int(rowptr[hi])
    P.nnz = e1 - e0
    out_dst = dst[e0:e1]
    in_mask = (dst >= lo) & (dst < hi)
    in_src, in_dst, in_eid = src[in_mask], dst[in_mask], eid[in_mask]
    remote_out = out_dst[(out_dst < lo) | (out_dst >= hi)]
    remote_in = in_src[(in_src < lo) | (in_src >= hi)]
    P.halo_ids = np.unique(np.concatenate([remote_out, remote_in])).astype(np.int64)
    P.n_halo = len(P.halo_ids)
    P.halo_owner = owner_of[P.halo_ids] if P.n_halo else np.zeros(0, np.int64)
    P.recv_counts = [int((P.halo_owner == q).sum()) for q in range(world)]

    def to_local(g):
        g = np.asarray(g, dtype=np.int64)
        own = (g >= lo) & (g < hi)
        out = np.where(own, g - lo, 0)
        if P.n_halo:
            h = np.searchsorted(P.halo_ids, g[~own])
            out[~own] = P.n_own + h
        return out

    P.rowptr = (rowptr[lo:hi + 1] - e0).astype(np.int32)
    loc = to_local(out_dst)
    P.col = loc.astype(np.int32)
    # (the variant for 4x-upsampled sources, col_up, is filled in by ShardPlan: owned j keeps j - j >> 2 is its parent's local
    #  row because n_own % 4 == 0 -, a halo node becomes 4 * (n_own / 4 + slot of its parent among the unique halo parents))
    P.col_up = None
    P.pair = None
    P.max_deg = int(np.diff(P.rowptr).max()) if P.n_own else 0
    # 32-row tiles (the conv kernels' workgroup granule) that gather owned rows only / that touch the halo: the first
    # kind runs while the halo rows are still travelling
    P.tiles_int, P.tiles_bnd = _split_tiles(P.rowptr, loc >= P.n_own, P.n_own)

    # rows of mine that peer q needs, in q's halo order (ascending global id)
    P.send_rows = []
    for q, (a, b) in enumerate(lo_hi):
        if q == rank:
            P.send_rows.append(np.zeros(0, np.int64))
            continue
        qe0, qe1 = int(rowptr[a]), int(rowptr[b])
        q_out = dst[qe0:qe1]
        q_in_mask = (dst >= a) & (dst < b)
        q_in_src = src[q_in_mask]
        need = np.unique(np.concatenate([q_out, q_in_src]))
        need = need[(need >= lo) & (need < hi)]
        P.send_rows.append(need - lo)
    P.send_counts = [len(s) for s in P.send_rows]

    # ---- transposed CSR over owned targets; in-edges of a node ordered by global edge id (= unsharded order)
    local_in = (in_src >= lo) & (in_src < hi)
    # incoming cross edges, stored behind the owned edges grouped by source rank in (target, source, eid) order
    cross_src, cross_dst, cross_eid = in_src[~local_in], in_dst[~local_in], in_eid[~local_in]
    cross_owner = owner_of[cross_src] if len(cross_src) else np.zeros(0, np.int64)
    order = np.lexsort((cross_eid, cross_src, cross_dst, cross_owner))
    slot_of = np.empty(len(order), dtype=np.int64)
    slot_of[order] = np.arange(len(order))
    P.n_cross_in = len(order)
    P.cross_recv_counts = [int((cross_owner == q).sum()) for q in range(world)]
    tedge_all = np.empty(len(in_eid), dtype=np.int64)
    tedge_all[local_in] = in_eid[local_in] - e0
    tedge_all[~local_in] = P.nnz + slot_of
    tcol_all = to_local(in_src)
    tkey = np.lexsort((in_eid, in_dst))
    tdst = in_dst[tkey] - lo
    P.trowptr = np.zeros(P.n_own + 1, dtype=np.int32)
    np.add.at(P.trowptr, tdst + 1, 1)
    P.trowptr = np.cumsum(P.trowptr).astype(np.int32)
    P.tcol = tcol_all[tkey].astype(np.int32)
    P.tedge = tedge_all[tkey].astype(np.int32)
    P.max_in_deg = int(np.diff(P.trowptr).max()) if P.n_own else 0
    P.ttiles_int, P.ttiles_bnd = _split_tiles(P.trowptr, P.tcol >= P.n_own, P.n_own)
    # my outgoing cross edges to peer q, in q's storage order (target, source, eid)
    P.send_edges = []
    o_src, o_dst, o_eid = src[e0:e1], out_dst, eid[e0:e1]
    for q, (a, b) in enumerate(lo_hi):
        m = (o_dst >= a) & (o_dst < b) if q != rank else np.zeros(len(o_dst), bool)
        s_, d_, e_ = o_src[m], o_dst[m], o_eid[m]
        k = np.lexsort((e_, s_, d_))
        P.send_edges.append((e_[k] - e0).astype(np.int64))
    P.cross_send_counts = [len(s) for s in P.send_edges]
    return P


class ShardPlan:
    """Everything rank `rank` of `world` needs to run its shard of one mesh."""

    def __init__(self, graphs_h, rank, world):
        """graphs_h: [(rowptr, col)] x 3 global CSRs (host numpy), levels 0..2."""
        n = [len(g[0]) - 1 for g in graphs_h]
        if n[0] != 4 * n[1] or n[1] != 4 * n[2]:
            raise ValueError("level sizes must be N0 = 4 N1 = 16 N2")
        if world > n[2]:
            raise ValueError("more ranks than coarsest nodes")
        cuts = _owner_ranges(n[2], world)
        self.rank, self.world = rank, world
        self.ranges = []
        for l in range(3):
            f = 4 ** (2 - l)
            self.ranges.append([(cuts[r] * f, cuts[r + 1] * f) for r in range(world)])
        self.levels = [build_level_plan(graphs_h[l][0], graphs_h[l][1], self.ranges[l], rank) for l in range(3)]
        self.n_total = n
        # Levels 0 and 1 are read through a 4x upsampling by the two up-convolutions (model.py:902-905,923-926): their PAIR
        # graph (graph.pair_graph: a CSR over the coarse nodes of the level above) is sharded by the same routine over the
        # coarse ranges.  Its halo is the set of UNIQUE parents of the level's halo nodes: the tail rows of the coarse source
        # tensor (d3 / d2) hold one row per such parent (not one per halo node: 2-4x fewer rows to exchange), `col_up` points
        # at them, and its transposed graph / cross-pair send lists drive the backward exchange of the pair form.
        from .graph import pair_graph
        for l in (0, 1):
            prow, pcol, pmul = pair_graph(graphs_h[l][0], graphs_h[l][1])
            PP = build_level_plan(prow, pcol, self.ranges[l + 1], rank)
            e0 = int(prow[self.ranges[l + 1][rank][0]])
            PP.pmul = np.ascontiguousarray(pmul[e0:e0 + PP.nnz])
            # Whether a layer runs in the pair form is a decision of the WHOLE job, not of a rank: the two forms exchange
            # different tensors in the backward pass (pair: dt + per-pair d-logits; fine: ds rows + per-edge d-logits), so
            # ranks that disagree send each other messages of the wrong size and content.  The quantities the library's
            # test looks at (include/fgc.h: fgc_conv_uses_pairs) are therefore taken over the GLOBAL pair graph - every
            # rank has it - and a rank's own values can only be smaller: all ranks get the same answer.
            PP.global_max_in_deg = int(np.bincount(pcol, minlength=len(prow) - 1).max()) if len(pcol) else 0
            PP.global_n_pairs = int(len(pcol))
            PP.global_rows = int(len(prow) - 1)
            L = self.levels[l]
            assert np.array_equal(PP.halo_ids, np.unique(L.halo_ids >> 2)), "pair halo != parents of the level's halo"
            loc = L.col.astype(np.int64)
            is_h = loc >= L.n_own
            slot = np.searchsorted(PP.halo_ids, L.halo_ids >> 2)
            up = loc.copy()
            up[is_h] = 4 * (L.n_own // 4 + slot[loc[is_h] - L.n_own])
            L.col_up = up.astype(np.int32)
            L.pair = PP

    def local_rows(self, level):
        """Global ids of the local rows [owned | halo] of a level."""
        P = self.levels[level]
        return np.concatenate([np.arange(P.lo, P.hi, dtype=np.int64), P.halo_ids])


def pair_form_allowed(PP, cout):
    """The job-wide half of fgc_conv_uses_pairs for a facet-sharded layer: the limits that depend on the graph, evaluated on
    the GLOBAL pair graph (ShardPlan) so that every rank decides alike - by the library's own function
    (fgc_conv_pairs_allowed, which fgc_conv_uses_pairs applies to a descriptor's counts), not a copy of its constants.
    `cout`: the layer's output width."""
    from . import _lib
    return bool(_lib.lib().fgc_conv_pairs_allowed(int(PP.global_rows), int(PP.global_n_pairs), int(PP.global_max_in_deg), int(cout)))


class LocalPairGraph:
    """Device-resident local pair graph of a level (graph.PairGraph's attributes) with its halo of unique parents: which
    coarse rows go to / come from every peer, and which pairs' dt / d-logit rows cross shards in the backward pass."""

    def __init__(self, PP, device):
        pad = (lambda a: a if len(a) else np.zeros(1, a.dtype))
        up = lambda a: torch.from_numpy(np.ascontiguousarray(pad(a))).to(device)
        self.n_pairs, self.n_halo, self.n_cross_in = PP.nnz, PP.n_halo, PP.n_cross_in
        self.max_deg, self.max_in_deg = PP.max_deg, PP.max_in_deg
        self.global_max_in_deg, self.global_n_pairs, self.global_rows = PP.global_max_in_deg, PP.global_n_pairs, PP.global_rows
        self.prow, self.pcol = up(PP.rowptr), up(PP.col)
        self.pmul = up(PP.pmul.view(np.int32))
        self.trow, self.tcol, self.tedge = up(PP.trowptr), up(PP.tcol), up(PP.tedge)
        self.send_rows = up(np.concatenate(PP.send_rows).astype(np.int32))
        self.send_counts, self.recv_counts = list(PP.send_counts), list(PP.recv_counts)
        self.send_edges = up(np.concatenate(PP.send_edges).astype(np.int32))
        self.cross_send_counts, self.cross_recv_counts = list(PP.cross_send_counts), list(PP.cross_recv_counts)
        # 32-row tiles of the COARSE rows whose in-pairs all have owned parents / that have an incoming cross-shard pair: the
        # backward data kernel of the pair form runs the first kind while the dt / d-logit rows of the others travel
        self.tiles = {k: (up(getattr(PP, k)), len(getattr(PP, k))) for k in ("ttiles_int", "ttiles_bnd")}


class LocalGraph:
    """Device-resident local CSR of one level of a shard (same attribute names as graph.FacetGraph)."""

    def __init__(self, P, device):
        self.n, self.nnz = P.n_own, P.nnz
        self.n_halo, self.n_cross_in = P.n_halo, P.n_cross_in
        self.max_deg, self.max_in_deg = P.max_deg, P.max_in_deg
        pad = (lambda a: a if len(a) else np.zeros(1, a.dtype))
        dev = torch.device(device)
        self.rowptr = torch.from_numpy(P.rowptr.copy()).to(dev)
        self.col = torch.from_numpy(pad(P.col).copy()).to(dev)
        self.col_up = torch.from_numpy(pad(P.col_up).copy()).to(dev) if P.col_up is not None else None
        self.pair = LocalPairGraph(P.pair, dev) if P.pair is not None else None
        self._t = tuple(torch.from_numpy(pad(a).copy()).to(dev) for a in (P.trowptr, P.tcol, P.tedge))
        self.tiles = {k: (torch.from_numpy(pad(getattr(P, k)).copy()).to(dev), len(getattr(P, k)))
                      for k in ("tiles_int", "tiles_bnd", "ttiles_int", "ttiles_bnd")}
        self.send_rows = torch.from_numpy(pad(np.concatenate(P.send_rows)).astype(np.int32)).to(dev)
        self.send_counts, self.recv_counts = list(P.send_counts), list(P.recv_counts)
        self.send_edges = torch.from_numpy(pad(np.concatenate(P.send_edges)).astype(np.int32)).to(dev)
        self.cross_send_counts, self.cross_recv_counts = list(P.cross_send_counts), list(P.cross_recv_counts)

    def transposed(self):
        return self._t


# ---------------------------------------------------------------------------------------------------
# one grouped exchange = one all-to-all
# ---------------------------------------------------------------------------------------------------
class PackedExchange:
    """The static plan of ONE grouped halo exchange: the rows of several tensors ("blocks") for every peer, packed
    peer-major into one send buffer, moved by a single all-to-all, and copied from the receive buffer into the halo
    tails of the tensors.  A block is (src [rows, width], idx, send_counts, recv [halo rows, width], recv_counts): rows
    idx[...] of src go out (idx sorted by peer, send_counts[q] of them to peer q); recv is the tail view the incoming
    rows land in, peer after peer.  Rows are float32 views (a bf16 row of C channels is C / 2 dwords).  Buffers, split
    sizes and the two copy-job lists are built once; every step then costs one pack launch, one collective call and one
    unpack launch (fgc_copy_rows_jobs), whatever the number of tensors and peers."""

    def __init__(self, blocks, world):
        dev = blocks[0][0].device
        w = [int(b[0].shape[1]) for b in blocks]
        self.send_splits = [int(sum(b[2][q] * w[i] for i, b in enumerate(blocks))) for q in range(world)]
        self.recv_splits = [int(sum(b[4][q] * w[i] for i, b in enumerate(blocks))) for q in range(world)]
        self.send_buf = torch.empty(sum(self.send_splits), dtype=torch.float32, device=dev)
        # ONE tensor: its halo tail IS the receive buffer - halo rows are stored in ascending global id, i.e. owner after
        # owner, which is the peer-major order the all-to-all delivers - and there is nothing to unpack (five of the seven
        # forward exchanges of a step).  Several tensors: a receive buffer and one unpack launch.
        self.direct = len(blocks) == 1 and blocks[0][3].is_contiguous()
        if self.direct:
            self.recv_buf = blocks[0][3].reshape(-1)
            assert self.recv_buf.data_ptr() == blocks[0][3].data_ptr() and self.recv_buf.numel() == sum(self.recv_splits)
        else:
            self.recv_buf = torch.empty(sum(self.recv_splits), dtype=torch.float32, device=dev)
        s_off = [np.cumsum([0] + list(b[2])) for b in blocks]
        r_off = [np.cumsum([0] + list(b[4])) for b in blocks]
        # pack job: rows idx[idx_off : idx_off + rows] of src -> send_buf[dst_off ...]; unpack job: recv_buf[src_off ...]
        # (rows consecutive) -> rows dst_row ... of the tail view
        self.pack_jobs, self.unpack_jobs = [], []
        so = ro = 0
        for q in range(world):
            for i, (src, idx, sc, recv, rc) in enumerate(blocks):
                if sc[q]:
                    self.pack_jobs.append(dict(src=src, idx=idx, idx_off=int(s_off[i][q]), dst=self.send_buf, dst_off=so,
                                               rows=int(sc[q]), width=w[i]))
                    so += int(sc[q]) * w[i]
                if rc[q] and not self.direct:
                    self.unpack_jobs.append(dict(src=self.recv_buf, src_off=ro, dst=recv, dst_row=int(r_off[i][q]),
                                                 rows=int(rc[q]), width=w[i]))
                    ro += int(rc[q]) * w[i]
        self._tables = None

    def _row_jobs(self):
        """The ctypes job tables of fgc_copy_rows_jobs (pointers are static: built once), in chunks of <= 32 jobs."""
        if self._tables is None:
            from . import _lib
            import ctypes as C

            def table(jobs, pack):
                chunks = []
                for c0 in range(0, len(jobs), 32):
                    part = jobs[c0:c0 + 32]
                    arr = (_lib.RowJob * len(part))()
                    for k, j in enumerate(part):
                        if pack:
                            arr[k].src, arr[k].idx = j["src"].data_ptr(), j["idx"].data_ptr() + 4 * j["idx_off"]
                            arr[k].dst = j["dst"].data_ptr() + 4 * j["dst_off"]
                        else:
                            arr[k].src, arr[k].idx = j["src"].data_ptr() + 4 * j["src_off"], None
                            arr[k].dst = j["dst"].data_ptr() + 4 * j["dst_row"] * j["width"]
                        arr[k].rows, arr[k].width = j["rows"], j["width"]
                    chunks.append((arr, len(part)))
                return chunks
            self._tables = (table(self.pack_jobs, True), table(self.unpack_jobs, False))
        return self._tables

    def _run(self, chunks, what):
        from . import _lib
        st = torch.cuda.current_stream().cuda_stream
        for arr, n in chunks:
            _lib.check(_lib.lib().fgc_copy_rows_jobs(arr, n, st), what)

    def pack(self):
        self._run(self._row_jobs()[0], "halo pack")

    def poison_tails(self):
        """Tests only (shard.sim_run): NaN in every row this exchange is going to deliver."""
        if self.direct:
            self.recv_buf.fill_(float("nan"))
        for j in self.unpack_jobs:
            j["dst"][j["dst_row"]:j["dst_row"] + j["rows"]].fill_(float("nan"))

    def unpack(self):
        self._run(self._row_jobs()[1], "halo unpack")


# ---------------------------------------------------------------------------------------------------
# exchange back ends
# ---------------------------------------------------------------------------------------------------
class DistComm:
    """One shard per process; RCCL (backend 'nccl') on GPUs, gloo through host staging in CPU-only tests."""

    def __init__(self, group=None):
        import torch.distributed as dist
        self.dist, self.group = dist, group
        self.world = dist.get_world_size(group)
        self.rank = dist.get_rank(group)
        self.host_staged = dist.get_backend(group) == "gloo"

    # ---- grouped exchanges: the rows of several tensors with every peer in ONE all-to-all (PackedExchange) ----------
    def all_to_all_flat(self, send, send_splits, recv, recv_splits, async_op=False):
        """1-D buffers, split sizes in elements per peer.  RCCL: one all_to_all_single (a group of sends and receives,
        every peer pair on its own xGMI link); returns the work handle when async_op.  gloo: staged through the host."""
        if self.host_staged:
            r = torch.empty(recv.shape, dtype=recv.dtype)
            self._gloo_a2a(r, send.cpu(), recv_splits, send_splits)
            recv.copy_(r)
            return None
        # Issued by EVERY rank of a world > 1 even when this rank neither sends nor receives (a shard without a halo at
        # some level): whether to call a collective must never be a rank-local decision.  A world of one has no peers.
        if self.world == 1 and send.numel() == 0 and recv.numel() == 0:
            return None
        return self.dist.all_to_all_single(recv, send, list(recv_splits), list(send_splits), group=self.group,
                                           async_op=async_op)

    def exchange_begin(self, px):
        """Start the exchange `px` (PackedExchange) and return a handle for `finish`: pack (one launch), then ONE
        collective call - on RCCL on the communicator's own stream, so kernels enqueued between begin and finish overlap
        it.  The halo tails must not be read before `finish`.  (A step has 14 of these; issued as point-to-point pairs
        per tensor and peer they were 230 isend / irecv calls per step on 8 ranks - more host time than the step has GPU
        time.)  The host-staged gloo path exchanges synchronously."""
        px.pack()
        work = self.all_to_all_flat(px.send_buf, px.send_splits, px.recv_buf, px.recv_splits, async_op=True)
        return (work, px)

    def exchange(self, px):
        """The exchange `px` as ONE blocking call, for a schedule that has nothing to launch before it needs the rows: pack,
        a SYNCHRONOUS collective, unpack.  torch (>= 2.7) runs a synchronous RCCL op on the current stream - no event for
        RCCL's own stream to wait for, none for the compute stream to wait for afterwards: 19 us per call instead of 40 on the
        compute stream's clock, 26 instead of 50 us of host time (world size 1, tools/rccl_call_cost_probe.py,
        profiles/r6_rccl_call_cost_probe.txt).  The host does not block either way."""
        px.pack()
        self.all_to_all_flat(px.send_buf, px.send_splits, px.recv_buf, px.recv_splits, async_op=False)
        px.unpack()

    def finish(self, handle):
        if handle is not None:
            work, px = handle
            if work is not None:
                work.wait()             # the current stream waits for the collective; the host does not block
            px.unpack()

    def _gloo_a2a(self, r, s, recv_counts, send_counts):
        # gloo has no all_to_all_single on every build: P2P rounds instead
        so, ro = np.cumsum([0] + list(send_counts)), np.cumsum([0] + list(recv_counts))
        reqs = []
        for q in range(self.world):
            if q == self.rank:
                continue
            if send_counts[q]:
                reqs.append(self.dist.isend(s[so[q]:so[q + 1]].contiguous(), q, group=self.group))
        for q in range(self.world):
            if q == self.rank or not recv_counts[q]:
                continue
            buf = torch.empty((recv_counts[q],) + tuple(r.shape[1:]), dtype=r.dtype)
            self.dist.recv(buf, q, group=self.group)
            r[ro[q]:ro[q + 1]] = buf
        for q in reqs:
            q.wait()

    def all_reduce_sum(self, t):
        if self.host_staged:
            c = t.cpu()
            self.dist.all_reduce(c, group=self.group)
            t.copy_(c)
        else:
            self.dist.all_reduce(t, group=self.group)




def graphs_to_host_csr(adjs):
    """3 K-lists (or FacetGraphs) -> [(rowptr, col)] host CSRs for ShardPlan."""
    from .graph import FacetGraph, csr_from_klist
    out = []
    for a in adjs:
        if isinstance(a, FacetGraph):
            out.append((a.rowptr_h, a.col_h))
        else:
            out.append(csr_from_klist(a))
    return out


def make_sim_shards(x, adjs, gt, world, device="cuda", seed=0, dtype="f32", multi_scale=False):
    """`world` shard networks of one mesh inside ONE process (parity tests of the sharded schedule on a single GPU)."""
    from .net import FacetDenoiser
    gh = graphs_to_host_csr(adjs)
    nets = []
    for r in range(world):
        plan = ShardPlan(gh, r, world)
        nets.append(FacetDenoiser(device, seed=seed, dtype=dtype, multi_scale=multi_scale).bind_mesh(x, adjs, gt=gt, plan=plan))
    return nets


class SimLatency:
    """Probe infrastructure for sim_run (tools/shard_latency_probe.py): gives the stand-in exchanges the LATENCY of a real
    collective, which one process on one GPU otherwise never shows.  Every collective is a spin kernel of one wave, `us`
    microseconds long.  A blocking exchange or all-reduce spins in the compute stream itself (a synchronous RCCL op runs on the
    current stream); an overlapped exchange spins on a side stream of its shard that waits for the compute stream (as RCCL's
    own stream does for an asynchronous op) from the moment the shard resumes computing, and the shard's ("wait", key) makes
    the compute stream wait for it - so the launches in between hide as much of it as they last, like on a rank of its own,
    at the price of the two cross-stream dependencies.  All shards share the one compute stream, in
    order: a stall of one shard is not filled by another's kernels."""

    _streams = None      # side streams that were SEEN to run beside the compute stream (found once per process)

    def __init__(self, us, n_shards, sync_on_side_stream=False):
        self.us = float(us)
        self.sync_on_side_stream = bool(sync_on_side_stream)
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda._sleep(1000)
        torch.cuda.synchronize()
        a.record()
        torch.cuda._sleep(4_000_000)
        b.record()
        torch.cuda.synchronize()
        self.cycles_per_us = 4_000_000 / (a.elapsed_time(b) * 1e3)
        self.cycles = int(self.us * self.cycles_per_us)
        if SimLatency._streams is None:
            # HIP streams share a few hardware queues, and two streams on one queue run one after the other: a side stream that
            # happens to share the compute stream's queue would show every overlapped exchange as exposed.  Keep the streams on
            # which a spin demonstrably overlaps a spin on the compute stream (both take ~200 us: together ~200, not ~400).
            good, c = [], int(2000 * self.cycles_per_us)
            self.overlap_test_ms = []
            for _ in range(12):
                st = torch.cuda.Stream()
                with torch.cuda.stream(st):
                    torch.cuda._sleep(1000)          # (the first launch on a stream sets its queue up)
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                torch.cuda._sleep(c)
                with torch.cuda.stream(st):
                    torch.cuda._sleep(c)
                torch.cuda.synchronize()
                ms = (time.perf_counter() - t0) * 1e3
                self.overlap_test_ms.append(round(ms, 2))
                if ms < 3.0:                         # (two 2 ms spins: ~2 ms together, ~4 ms one after the other)
                    good.append(st)
            SimLatency._streams = good
        if not SimLatency._streams:
            raise RuntimeError("SimLatency: no side stream runs beside the compute stream on this device (two 2 ms spins took %s ms)"
                               % getattr(self, "overlap_test_ms", "?"))
        self.side = [SimLatency._streams[i % len(SimLatency._streams)] for i in range(n_shards)]
        self.n_concurrent_streams = len(SimLatency._streams)

    def start(self, i):
        """The clock of shard i's overlapped exchange starts now (the shard resumes); returns the event its wait needs."""
        ev = torch.cuda.Event()
        ev.record()
        with torch.cuda.stream(self.side[i]):
            self.side[i].wait_event(ev)
            if self.cycles > 0:
                torch.cuda._sleep(self.cycles)
            done = torch.cuda.Event()
            done.record()
        return done

    def stall(self, i=0):
        """A blocking collective of shard i (an exchange served as one blocking call, an all-reduce): torch runs a synchronous
        RCCL op on the CURRENT stream, so it is a spin kernel in the compute stream itself - no cross-stream dependency.
        (`sync_on_side_stream`: the pre-2.7 behaviour and what shard.DistComm did until round 6 - the op on another stream,
        which waits for the compute stream and which the compute stream then waits for.)"""
        if self.sync_on_side_stream:
            torch.cuda.current_stream().wait_event(self.start(i))
        elif self.cycles > 0:
            torch.cuda._sleep(self.cycles)


def sim_run(nets, make_gen, poison=True, latency=None):
    """Advance the schedules of all shards; every yielded exchange / all-reduce request is served among them.

    Exchanges and sums are COLLECTIVE: all shards must arrive at them in the same order.  A ("wait", key) is
    rank-local (one shard may overlap an exchange with its interior tiles while a peer with a small interior blocks), so
    every generator is advanced to its next collective request on its own.  An overlapped exchange is modelled the way
    the hardware may behave at worst: the rows are packed when it begins, but they only land in the halo tails at the
    shard's ("wait", key) - and until then the tails hold NaN (`poison`), so a kernel of the overlapped stretch that
    lets a halo row reach a result fails the parity tests instead of reading last step's values."""
    gens = [make_gen(n) for n in nets]
    pending = [dict() for _ in nets]
    clocks = [dict() for _ in nets]      # (latency is not None: key -> the event an overlapped exchange completes with)

    def advance(i):
        if latency is not None:
            for key in pending[i]:
                if key not in clocks[i]:
                    clocks[i][key] = latency.start(i)
        while True:
            try:
                r = next(gens[i])
            except StopIteration:
                assert not pending[i], "a begun exchange was never awaited"
                return None
            if r[0] == "wait":
                px = pending[i].pop(r[1])
                if latency is not None:
                    torch.cuda.current_stream().wait_event(clocks[i].pop(r[1]))
                if getattr(px, "_sim_late", None) is not None:
                    px.recv_buf.copy_(px._sim_late)
                    px._sim_late = None
                px.unpack()
                continue
            return r

    while True:
        reqs = [advance(i) for i in range(len(nets))]
        if all(r is None for r in reqs):
            return
        assert all(r is not None for r in reqs), "shards disagree on the exchange schedule"
        assert len({r[0] for r in reqs}) == 1, "shards disagree on the exchange schedule"
        if reqs[0][0] == "call":       # launches a shard makes outside its captured graphs (see net._loss_backward_gen)
            for r in reqs:
                r[1]()
            continue
        if reqs[0][0] == "sum":
            tot = reqs[0][1].clone()
            for r in reqs[1:]:
                tot += r[1]
            for i, r in enumerate(reqs):
                r[1].copy_(tot)
                if latency is not None:
                    latency.stall(i)     # (an all-reduce blocks every rank: one stall per shard on the shared stream)
            continue
        pxs = [n._packed(r) for n, r in zip(nets, reqs)]
        for px in pxs:
            px.pack()
        for dst, pd in enumerate(pxs):
            ro = np.cumsum([0] + pd.recv_splits)
            # an overlapped exchange that receives straight into a halo tail (PackedExchange.direct) is held back in a
            # staging buffer until the shard's wait, like the unpack of the others
            late = pd.direct and reqs[dst][2] is not None
            target = torch.empty_like(pd.recv_buf) if late else pd.recv_buf
            for src, ps in enumerate(pxs):
                if src == dst or not pd.recv_splits[src]:
                    continue
                so = np.cumsum([0] + ps.send_splits)
                assert ps.send_splits[dst] == pd.recv_splits[src], (src, dst)
                target[ro[src]:ro[src + 1]].copy_(ps.send_buf[so[dst]:so[dst + 1]])
            pd._sim_late = target if late else None
        for i, (px, r) in enumerate(zip(pxs, reqs)):
            if r[2] is None:
                if latency is not None:
                    latency.stall(i)
                px.unpack()
            else:
                assert r[2] not in pending[i], "two exchanges in flight under one key"
                pending[i][r[2]] = px
                if poison:
                    px.poison_tails()


def sim_forward_backward(nets, rotate=True, latency=None):
    sim_run(nets, lambda n: n._forward_gen(rotate, n._fused_loss_now()), latency=latency)
    sim_run(nets, lambda n: n._loss_backward_gen(rotate), latency=latency)


def sim_forward_multi_scale(nets, rotate=False):
    sim_run(nets, lambda n: n._forward_ms_gen(rotate))


def sim_forward_backward_captured(nets, rotate=True, latency=None):
    """The step of sim_forward_backward with every shard's launches replayed from its hipGraph segments (one graph per
    stretch between two exchanges, net._capture_segments), the requests served between the replays - what
    `forward_backward(capture=True)` does on a sharded rank, with the simulated exchange in place of the communicator.
    Run one eager step first: one-time set-up inside the library must not happen under capture."""
    for n in nets:
        if n._graph_fb is None:
            n._own_step_inputs()
            n._graph_fb = ((n._capture_segments(lambda: n._forward_gen(rotate, n._fused_loss_now())),
                            n._capture_segments(lambda: n._loss_backward_gen(rotate))), rotate)

    def replay(segs):
        for g, req in segs:
            if g is not None:        # (a stretch without launches is no graph: net._capture_segments)
                g.replay()
            if req is not None:
                yield req

    sim_run(nets, lambda n: replay(n._graph_fb[0][0]), latency=latency)
    sim_run(nets, lambda n: replay(n._graph_fb[0][1]), latency=latency)

