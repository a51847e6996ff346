"""The facet-sharded SCHEDULE the multi-GPU benchmark runs, held against the unsharded network.

At 100k facets per rank every big layer takes the interior / boundary split with an overlapped exchange
(net.py: `split_min_tiles`, FGC_SPLIT_MIN_TILES, default 1024 interior tiles): interior tiles (and the logit rows of
owned sources) while the halo rows travel, the boundary tiles after the wait; backward the same with the data-gradient
kernel.  The fixtures' meshes have a few dozen tiles per shard and never reach that branch on their own, so here it is
(a) forced onto them (FGC_SPLIT_MIN_TILES = 0 and 8), eager and replayed from hipGraph segments, fp32 and bf16, regular
and irregular meshes; (b) reached naturally: a 200k-facet torus in 2 shards; (c) run at the sizes of BASELINE configs 4 and
5: the 1M-facet torus and the 500k-facet multi-scale forward as 8 shards in one process, inside guard zones.

`shard.sim_run` delivers an overlapped exchange only at the shard's wait and keeps NaN in the halo tails until then: a
kernel of the overlapped stretch that lets a halo row reach a result cannot pass."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _prep(golden_dir, tag):
    prep = np.load(os.path.join(golden_dir, "prep_%s.npz" % tag))
    return prep["x"], [prep["adj0"], prep["adj1"], prep["adj2"]], prep["gt"]


def _mesh(nu, nv, flips=0, seed=0):
    from facet_graph_convolution_amd.dataClasses import TrainingSet
    from facet_graph_convolution_amd.meshgen import torus, add_noise, flip_edges
    V, F = torus(nu, nv)
    if flips:
        F = flip_edges(F, flips, seed=1)
    ds = TrainingSet()
    ds.addMeshWithGT(add_noise(V, F, 0.2, seed=1 + seed), F, V, seed=seed)
    return ds.in_list[0], ds.adj_list[0], ds.gt_list[0]


def _took_the_split_branch(nets):
    """(forward layers, backward layers) of shard 0 that run interior | exchange | boundary."""
    n = nets[0]
    g = n._mesh["graphs"]
    fwd = sum(1 for lay in n.layers[1:] if g[lay.level].tiles["tiles_int"][1] >= n.split_min_tiles)
    bwd = sum(1 for lay in n.layers[1:] if g[lay.level].tiles["ttiles_int"][1] >= n.split_min_tiles)
    return fwd, bwd


def _compare(ref, nets, tol_n, tol_loss, tol_g):
    torch.cuda.synchronize()
    full = ref.buffers["nconv"].cpu().numpy()
    assert np.isfinite(full).all()
    worst = 0.0
    for n in nets:
        P = n._mesh["plan"].levels[0]
        got = n.buffers["nconv"].cpu().numpy()
        assert np.isfinite(got).all(), "NaN reached the output: a kernel of an overlapped stretch read a halo row"
        np.testing.assert_allclose(got, full[P.lo:P.hi], rtol=0, atol=tol_n)
        assert abs(n.buffers["loss"][0].item() - ref.buffers["loss"][0].item()) < tol_loss * max(1.0, abs(ref.buffers["loss"][0].item()))
        assert torch.equal(n.params.grad, nets[0].params.grad), "ranks hold different all-reduced gradients"
    for i, (g, gr) in enumerate(zip(nets[0].params.grads, ref.params.grads)):
        a, b = g.cpu().numpy(), gr.cpu().numpy()
        assert np.isfinite(a).all(), "grad %d" % i
        err = np.abs(a - b).max() / max(np.abs(b).max(), 1e-3)
        worst = max(worst, err)
        assert err < tol_g, "grad %d: rel err %.3e" % (i, err)
    return worst


def _step_pair(x, adjs, gt, world, dtype, samp, R):
    from facet_graph_convolution_amd.net import FacetDenoiser
    from facet_graph_convolution_amd.shard import make_sim_shards
    ref = FacetDenoiser("cuda:0", seed=0, dtype=dtype).bind_mesh(x, adjs, gt=gt)
    nets = make_sim_shards(x, adjs, gt, world, "cuda:0", 0, dtype=dtype)
    for n in [ref] + nets:
        n.set_rotation(R)
        n.set_samples(samp)
    ref.forward_backward(rotate=True)
    return ref, nets


TOL = {"f32": (1e-6, 1e-4, 1e-3), "bf16": (1e-5, 1e-3, 1e-2)}     # test_gpu_net.py / test_gpu_bf16.py: sharded vs unsharded


def _eager_then_segments(ref, nets, dtype):
    from facet_graph_convolution_amd.shard import sim_forward_backward, sim_forward_backward_captured
    sim_forward_backward(nets, rotate=True)
    worst = _compare(ref, nets, *TOL[dtype])
    eager = [n.params.grad.clone() for n in nets]
    for _ in range(2):
        sim_forward_backward_captured(nets, rotate=True)
    _compare(ref, nets, *TOL[dtype])
    for n, g in zip(nets, eager):
        assert torch.equal(n.params.grad, g), "hipGraph segments differ from the eager schedule"
    return worst


@pytest.mark.parametrize("dtype", ["f32", "bf16"])
@pytest.mark.parametrize("tag,world", [("ico3", 2), ("torus640", 3)])
def test_split_schedule_with_empty_interiors(golden_dir, tag, world, dtype, monkeypatch):
    """The fixture meshes keep the reference's node order (not the Morton order of the native preprocessing): every tile
    of a shard touches its halo.  Threshold 0 forces the split anyway: an EMPTY interior call (logits of the owned rows
    only), every tile after the wait; eager and replayed from hipGraph segments."""
    monkeypatch.setenv("FGC_SPLIT_MIN_TILES", "0")
    x, adjs, gt = _prep(golden_dir, tag)
    z = np.load(os.path.join(golden_dir, "net_%s.npz" % tag))
    ref, nets = _step_pair(x, adjs, gt, world, dtype, z["sample_ind"], z["R"])
    assert nets[0].split_min_tiles == 0 and _took_the_split_branch(nets) == (7, 7)
    assert nets[0]._mesh["graphs"][0].tiles["tiles_int"][1] == 0
    _eager_then_segments(ref, nets, dtype)


@pytest.mark.parametrize("dtype", ["f32", "bf16"])
@pytest.mark.parametrize("nu,nv,world,min_tiles", [(64, 48, 2, "0"), (64, 48, 4, "8"), (96, 64, 3, "8")])
def test_forced_split_schedule_matches_the_unsharded_network(nu, nv, world, min_tiles, dtype, monkeypatch):
    """Natively preprocessed tori (Morton order: 20-100 interior tiles per shard at level 0, a dozen at level 1).
    Threshold 0: every layer splits.  Threshold 8: level 0 splits, level 1 splits on SOME shards only (12 / 4 / 13
    interior tiles on the three shards of the 96 x 64 torus), level 2 blocks - shards overlap and block side by side."""
    monkeypatch.setenv("FGC_SPLIT_MIN_TILES", min_tiles)
    x, adjs, gt = _mesh(nu, nv, seed=0)
    samp = np.random.RandomState(4).randint(x.shape[1], size=4000)
    from facet_graph_convolution_amd.utils import rand_rotation_matrix
    R = rand_rotation_matrix(randnums=np.random.RandomState(3).uniform(size=3))
    ref, nets = _step_pair(x, adjs, gt, world, dtype, samp, R)
    took = [_took_the_split_branch([n]) for n in nets]
    assert all(f >= 2 and b >= 2 for f, b in took), took
    if (nu, world) == (96, 3):
        assert len(set(took)) > 1, "the shards were meant to disagree on which layers split: %s" % took
    assert nets[0]._mesh["graphs"][0].tiles["tiles_int"][1] >= 8
    _eager_then_segments(ref, nets, dtype)


def test_sharded_step_with_the_separate_loss_launches(monkeypatch):
    """FGC_NO_FUSED_LOSS=1: the loss end of a sharded step through the separate entry points (normalise / rotate / loss /
    gradients, torch glue for the scalars, the loss sum riding in the gradient all-reduce) instead of fgc_loss_shard_*:
    same results, and a rank WITHOUT samples takes part in both forms."""
    from facet_graph_convolution_amd.shard import sim_forward_backward
    x, adjs, gt = _mesh(64, 48, seed=0)
    n0 = x.shape[1]
    hi0 = ((n0 // 16) // 2) * 16                                        # shard 0 owns rows [0, hi0) (shard._owner_ranges)
    samp = np.random.RandomState(4).randint(hi0, size=4000)            # every sample in shard 0: shard 1 has none
    out = {}
    for mode in ("1", "0"):
        monkeypatch.setenv("FGC_NO_FUSED_LOSS", mode)
        ref, nets = _step_pair(x, adjs, gt, 2, "f32", samp, np.eye(3))
        assert nets[0].fused_loss == (mode == "0")
        sim_forward_backward(nets, rotate=True)
        assert nets[1].buffers["sample_ind_local"].numel() == 0
        _compare(ref, nets, *TOL["f32"])
        out[mode] = (nets[0].params.grad.clone(), nets[0].buffers["loss"][0].item())
    assert abs(out["0"][1] - out["1"][1]) < 1e-4 * abs(out["1"][1])
    assert (out["0"][0] - out["1"][0]).abs().max().item() < 1e-5 * out["1"][0].abs().max().item()


@pytest.mark.parametrize("dtype", ["f32", "bf16"])
def test_forced_split_schedule_on_an_irregular_mesh(dtype, monkeypatch):
    """Facet degrees up to K = 23 (24-slot kernels, the LONG d-logits form) under the split schedule, 3 shards."""
    from facet_graph_convolution_amd.shard import sim_forward_backward
    monkeypatch.setenv("FGC_SPLIT_MIN_TILES", "4")
    # (900 flips: out-degrees up to 20, in-degrees up to 23 - the bf16 data-gradient kernel takes at most 24 in-edges)
    x, adjs, gt = _mesh(48, 40, flips=900 if dtype == "bf16" else 1600, seed=0)
    assert max(int((a[0] > 0).sum(1).max()) for a in adjs) > 16
    samp = np.random.RandomState(4).randint(x.shape[1], size=4000)
    ref, nets = _step_pair(x, adjs, gt, 3, dtype, samp, np.eye(3))
    assert min(_took_the_split_branch(nets)) >= 1
    sim_forward_backward(nets, rotate=True)
    tol = TOL[dtype] if dtype == "f32" else (1e-5, 1e-3, 2e-2)    # (few rows per tensor: test_gpu_bf16.py's irregular bound)
    _compare(ref, nets, *tol)


def test_split_schedule_stays_inside_its_buffers(monkeypatch):
    """Guard zones round every buffer of three shards under the forced split schedule (tile lists, packed exchange
    buffers, halo tails), eager and hipGraph segments."""
    from test_gpu_guard import _Guarded
    from facet_graph_convolution_amd.shard import make_sim_shards, sim_forward_backward, sim_forward_backward_captured
    monkeypatch.setenv("FGC_SPLIT_MIN_TILES", "8")
    x, adjs, gt = _mesh(96, 64, seed=3)
    with _Guarded() as g:
        nets = make_sim_shards(x, adjs, gt, 3, "cuda:0", seed=0)
        samp = np.random.RandomState(2).randint(x.shape[1], size=4000)
        for n in nets:
            n.set_rotation(np.eye(3))
            n.set_samples(samp)
        assert min(_took_the_split_branch(nets)) >= 1
        sim_forward_backward(nets, rotate=True)
        g.check("split schedule, eager")
        for _ in range(2):
            sim_forward_backward_captured(nets, rotate=True)
        g.check("split schedule, hipGraph segments")


@pytest.mark.parametrize("dtype", ["f32", "bf16"])
def test_200k_facets_in_two_shards_reach_the_split_branch_on_their_own(dtype):
    """The benchmark's own shape (100k facets per rank, default threshold): the level-0 layers split, the coarse levels
    block - forward, loss and all 44 gradients against the unsharded network, eager and hipGraph segments."""
    assert "FGC_SPLIT_MIN_TILES" not in os.environ
    x, adjs, gt = _mesh(500, 200, seed=0)
    samp = np.random.RandomState(100).randint(x.shape[1], size=4000)
    from facet_graph_convolution_amd.utils import rand_rotation_matrix
    R = rand_rotation_matrix(randnums=np.random.RandomState(3).uniform(size=3))
    ref, nets = _step_pair(x, adjs, gt, 2, dtype, samp, R)
    assert nets[0].split_min_tiles == 1024
    fwd, bwd = _took_the_split_branch(nets)
    assert fwd == 2 and bwd == 2, (fwd, bwd)          # upconv1, dconv1 (conv1 has no producer to wait for; backward: no exchange)
    worst = _eager_then_segments(ref, nets, dtype)
    print("200k facets, 2 shards, %s: worst rel gradient difference to the unsharded network %.2e" % (dtype, worst))


def test_weak_scaling_mesh_of_the_eight_gpu_run_in_eight_shards():
    """What `bench.py --gpus 8` (the driver's scaling run, config c2 weak) binds: ONE torus of 8 x 100 000 facets
    (2000 x 200), 100k facets per shard, default split threshold, hipGraph segments as the bench replays them - against
    the unsharded network, inside guard zones."""
    from test_gpu_guard import _Guarded
    from facet_graph_convolution_amd.utils import rand_rotation_matrix
    x, adjs, gt = _mesh(2000, 200, seed=0)
    samp = np.random.RandomState(100).randint(x.shape[1], size=4000)
    R = rand_rotation_matrix(randnums=np.random.RandomState(3).uniform(size=3))
    with _Guarded() as g:
        ref, nets = _step_pair(x, adjs, gt, 8, "f32", samp, R)
        assert all(_took_the_split_branch([n]) == (2, 2) for n in nets)
        worst = _eager_then_segments(ref, nets, "f32")
        g.check("800k facets, 8 shards, eager + hipGraph segments")
    print("800k facets (c2 x 8), 8 shards: worst rel gradient difference %.2e" % worst)


def test_config4_one_million_facets_in_eight_shards():
    """BASELINE config 4 at its own size: the 1 000 000-facet torus (1000 x 500) as 8 shards against the unsharded
    network - unit normals within 1e-6, loss 1e-4, every gradient within 1e-3 of its tensor's largest entry - with every
    buffer of the nine networks between guard zones."""
    from test_gpu_guard import _Guarded
    from facet_graph_convolution_amd.shard import sim_forward_backward
    from facet_graph_convolution_amd.utils import rand_rotation_matrix
    x, adjs, gt = _mesh(1000, 500, seed=0)
    assert (np.abs(gt[0]).sum(1) > 1e-3).sum() == 1000000
    samp = np.random.RandomState(100).randint(x.shape[1], size=4000)
    R = rand_rotation_matrix(randnums=np.random.RandomState(3).uniform(size=3))
    with _Guarded() as g:
        ref, nets = _step_pair(x, adjs, gt, 8, "f32", samp, R)
        fwd, bwd = _took_the_split_branch(nets)
        assert fwd >= 2 and bwd >= 2, (fwd, bwd)
        sim_forward_backward(nets, rotate=True)
        worst = _compare(ref, nets, 1e-6, 1e-4, 1e-3)
        g.check("1M facets, 8 shards")
    halo = [nets[0]._mesh["nh"][l] / nets[0]._mesh["ns"][l] for l in range(3)]
    print("1M facets, 8 shards: worst rel gradient difference %.2e; halo / owned rows on shard 0: %s" % (
        worst, ", ".join("%.3f" % h for h in halo)))


def test_config5_multiscale_500k_facets_in_eight_shards():
    """BASELINE config 5 at its own size: the multi-scale denoising forward (three heads, each through normalizeTensor)
    of the 500 000-facet torus as 8 shards against the unsharded network, inside guard zones."""
    from test_gpu_guard import _Guarded
    from facet_graph_convolution_amd.net import FacetDenoiser
    from facet_graph_convolution_amd.shard import make_sim_shards, sim_forward_multi_scale
    x, adjs, gt = _mesh(500, 500, seed=0)
    with _Guarded() as g:
        ref = FacetDenoiser("cuda:0", seed=0, multi_scale=True).bind_mesh(x, adjs)
        outs = [t.cpu().numpy() for t in ref.forward_multi_scale()]
        nets = make_sim_shards(x, adjs, None, 8, "cuda:0", seed=0, multi_scale=True)
        assert _took_the_split_branch(nets)[0] >= 2
        sim_forward_multi_scale(nets)
        torch.cuda.synchronize()
        for n in nets:
            for lvl, (key, full) in enumerate(zip(("nconv", "nconv1", "nconv2"), outs)):
                P = n._mesh["plan"].levels[lvl]
                got = n.buffers[key].cpu().numpy()
                assert np.isfinite(got).all()
                np.testing.assert_allclose(got, full[P.lo:P.hi], rtol=0, atol=1e-6, err_msg="head of level %d" % lvl)
        g.check("500k facets multi-scale, 8 shards")


def test_no_schedule_segment_is_an_empty_graph_and_back_to_back_requests_run(golden_dir, monkeypatch):
    """The invariant the segment replay rests on (net._capture_segments): the runtime answers how many nodes a captured
    stretch holds (hipGraphGetNodes - a failure raises, nothing is guessed), a stretch without launches - the fused sharded
    loss end issues ("call", samples) and ("sum", table) back to back, and nothing follows the last all-reduce - is
    recorded as no graph at all, and every graph that IS instantiated and replayed holds at least one node.  The schedule
    with those back-to-back requests then replays twice, bit-identical to eager."""
    from facet_graph_convolution_amd import net as netmod
    monkeypatch.setenv("FGC_SPLIT_MIN_TILES", "0")
    x, adjs, gt = _prep(golden_dir, "ico3")
    z = np.load(os.path.join(golden_dir, "net_ico3.npz"))
    ref, nets = _step_pair(x, adjs, gt, 2, "f32", z["sample_ind"], z["R"])
    assert all(n.fused_loss for n in nets)
    _eager_then_segments(ref, nets, "f32")
    for n in nets:
        fwd, bwd = n._graph_fb[0]
        segs = fwd + bwd
        assert len(n.segment_nodes) == len(segs) and min(n.segment_nodes) >= 0
        for (g, req), nodes in zip(segs, n.segment_nodes):
            assert (g is None) == (nodes == 0), "a segment with %d nodes was %s" % (nodes, "kept" if g is not None else "dropped")
            if g is not None:
                assert netmod._graph_node_count(g) == nodes >= 1
        # the back-to-back pair of the fused sharded loss end is there, as a stretch without a graph
        kinds = [(req[0] if req else None, g is None) for g, req in bwd]
        i = kinds.index(("call", False)) if ("call", False) in kinds else kinds.index(("call", True))
        assert kinds[i + 1] == ("sum", True), kinds[i:i + 2]
        assert bwd[-1][1] is None and bwd[-1][0] is None, "nothing is launched behind the last all-reduce"


def test_capture_is_not_interrupted_by_the_garbage_collector():
    """The round-3 abort, as the stack of its round-4 recurrence shows it: Python's cyclic collector ran inside a segment
    capture (`Garbage-collecting` under net._capture_segments) and freed a dead cycle that owned an earlier network's
    hipGraphs; hipGraphDestroy / hipFree are not allowed on a capturing stream, torch raised from a destructor, abort.
    `net._no_gc_while_capturing` collects before a capture and keeps the collector off until it ends.  The probe sets the
    situation up on purpose - dead cycles holding captured networks, then a capture with the collector's thresholds at 1 -
    in a process of its own, with the guard on (tools/capture_gc_probe.py; its `noguard` mode is the experiment, not a test)."""
    import subprocess, sys
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, os.path.join(repo, "tools", "capture_gc_probe.py"), "guard"], cwd=repo,
                         capture_output=True, text=True, timeout=600)
    assert out.returncode == 0 and "probe ok (guard)" in out.stdout, out.stdout[-1500:] + out.stderr[-3000:]
    # in-process: the collector is off inside the guard and back on after it
    import gc
    from facet_graph_convolution_amd.net import _no_gc_while_capturing
    assert gc.isenabled()
    with _no_gc_while_capturing():
        assert not gc.isenabled()
    assert gc.isenabled()


@pytest.mark.parametrize("dtype", ["f32", "bf16"])
def test_weight_gradients_inside_the_next_layers_exchange_window(dtype, monkeypatch):
    """Round 6: a layer's weight-gradient stage (GEMM over its r rows + partial sums) is launched between the BEGIN of the next
    layer's backward exchange and its wait - the only independent work a sequential backward pass has to put under a
    collective's latency (tools/shard_latency_probe.py).  Same launches in another order: the gradients equal those of the
    old order (FGC_NO_DW_IN_WINDOW=1, GEMM right behind its own data kernel: every exchange then a blocking call) and the
    unsharded network's within the usual bound; every backward exchange of a layer that follows another is overlapped (begun
    under a key, awaited later), with launches in between; and r is never overwritten before its reader ran - the halo tails and the r buffer hold NaN until
    written (sim_run poisons the tails), so a GEMM that ran too late would read the next layer's rows."""
    from facet_graph_convolution_amd.shard import sim_forward_backward, sim_run
    x, adjs, gt = _mesh(64, 48, seed=0)
    samp = np.random.RandomState(4).randint(x.shape[1], size=4000)
    from facet_graph_convolution_amd.utils import rand_rotation_matrix
    R = rand_rotation_matrix(randnums=np.random.RandomState(3).uniform(size=3))
    out = {}
    for off in ("1", "0"):
        monkeypatch.setenv("FGC_NO_DW_IN_WINDOW", off)
        ref, nets = _step_pair(x, adjs, gt, 2, dtype, samp, R)
        assert nets[0].dw_in_window == (off == "0")
        # (windowed: only the first layer, which has no exchange, keeps its GEMM for the grouped launch)
        assert (len(nets[0].grouped_dw_layers) == 1) == (off == "0")
        sim_forward_backward(nets, rotate=True)
        _compare(ref, nets, *TOL[dtype])
        out[off] = [n.params.grad.clone() for n in nets]
        if off == "0":
            _eager_then_segments(ref, nets, dtype)
            # the schedule itself: every backward exchange is begun under a key and awaited later
            reqs = [r for r in nets[0]._loss_backward_gen(True)]
            torch.cuda.synchronize()
            x_ = [r for r in reqs if r[0] == "xchg"]
            # (the first layer of the backward pass has no predecessor whose GEMM could fill its window, and this mesh is too
            #  small to split: its exchange is ONE blocking call - a synchronous collective on the compute stream)
            assert len(x_) == 7 and x_[0][2] is None and all(r[2] == "bwd" for r in x_[1:])
            assert sum(1 for r in reqs if r[0] == "wait") == 6
    for a, b in zip(out["1"], out["0"]):
        # (the two runs group different layers' GEMMs - one launch per kernel form or one per layer: the same sums in the same
        #  order, tests/test_gpu_net.py::test_grouped_weight_gradient_launches_equal_the_per_layer_launches)
        assert torch.equal(a, b), "the windowed order changed a gradient: max diff %.3e" % (a - b).abs().max().item()
