"""Developer helper (GPU box): what a facet-sharded step costs when its collectives have LATENCY - the one thing two shards in
one process never show (shard.sim_run's exchanges are device copies of ~5 us).  shard.SimLatency stalls the compute stream for
X us at every blocking exchange / all-reduce and lets an overlapped exchange run X us on a side stream under whatever the shard
launches before its wait.  For each variant - weight-gradient stage inside the next layer's exchange window or right behind its
own data kernel (FGC_NO_DW_IN_WINDOW), eager launches or hipGraph segments, split threshold (net.split_min_tiles) - and each X the
probe prints the GPU time per shard and step, and the slope d ms / d X = how many of a step's 17 collectives are EXPOSED.  All
variants run interleaved in ONE process, three passes, the minimum kept.  This is the measurement behind the default threshold,
bench.py's `split_tune` candidates, the windowed backward order and the prediction table of DESIGN.md section 7.
usage: python tools/shard_latency_probe.py [nu nv world steps]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from bench import build_mesh
from facet_graph_convolution_amd.shard import make_sim_shards, sim_forward_backward, sim_forward_backward_captured, SimLatency
nu, nv, world, steps = (int(a) for a in (sys.argv[1:5] + ["500", "200", "2", "20"][len(sys.argv) - 1:]))
ds, F = build_mesh(nu, nv, 0)
x, adjs, gt = ds.in_list[0], ds.adj_list[0], ds.gt_list[0]
samp = np.random.RandomState(0).randint(x.shape[1], size=4000)
sets = {}
for window in ("window", "behind"):
    os.environ["FGC_NO_DW_IN_WINDOW"] = "0" if window == "window" else "1"
    sets[window] = make_sim_shards(x, adjs, gt, world, "cuda:0", seed=0)
    for n in sets[window]:
        n.set_samples(samp); n.set_rotation(np.eye(3))
g = sets["window"][0]._mesh["graphs"]
print("%d facets in %d shards on one GPU; shard 0: interior tiles per level fwd %s, bwd %s, coarse-row tiles of the pair graphs %s"
      % (F, world, [g[l].tiles["tiles_int"][1] for l in range(3)], [g[l].tiles["ttiles_int"][1] for l in range(3)],
         [g[l].pair.tiles["ttiles_int"][1] for l in (0, 1) if g[l].pair is not None]))
lats = (0, 10, 20, 40)
cal = SimLatency(10, world)
print("spin kernel: %.1f cycles per us; %d of 12 side streams run beside the compute stream" % (cal.cycles_per_us, cal.n_concurrent_streams))
thrs = (1 << 30, 1024, 256, 64)
modes = (("eager", sim_forward_backward), ("segments", sim_forward_backward_captured))
# (weight-gradient stage, blocking collectives on a side stream?) - the last one is what shard.DistComm did until round 6:
# every exchange an asynchronous op awaited at once
variants = (("window", False), ("behind", False), ("behind", True))
best = {}
for rep in range(3):
    for window, side in variants:
        nets = sets[window]
        for thr in thrs:
            for n in nets:
                n.split_min_tiles = thr
                n._graph_fb = None
            for mode, fn in modes:
                if mode == "segments" and thr not in (1024,):
                    continue
                for X in lats:
                    lat = SimLatency(X, world, sync_on_side_stream=side) if (X or side) else None
                    for _ in range(3):
                        fn(nets, rotate=True, latency=lat)
                    torch.cuda.synchronize(); t = time.perf_counter()
                    for _ in range(steps):
                        fn(nets, rotate=True, latency=lat)
                    torch.cuda.synchronize()
                    k = (window, side, mode, thr, X)
                    best[k] = min(best.get(k, 1e9), (time.perf_counter() - t) / steps / world * 1e3)
print("ms per shard and step at X = %s us   | exposed collectives of 17 (slope between %d and %d us)" % ("/".join(str(v) for v in lats), lats[1], lats[-1]))
for mode, _ in modes:
    for thr in thrs:
        for window, side in variants:
            if (window, side, mode, thr, lats[0]) not in best:
                continue
            row = [best[(window, side, mode, thr, X)] for X in lats]
            slope = (row[-1] - row[1]) / (lats[-1] - lats[1]) * 1e3
            print("%-8s threshold %-6s dW %-6s blocking collectives %-22s %s   | %.1f"
                  % (mode, "none" if thr == 1 << 30 else thr, window, "on a side stream" if side else "in the compute stream",
                     "  ".join("%.3f" % v for v in row), slope))
print("loss %.4f / %.4f" % tuple(s_[0].buffers["loss"][0].item() for s_ in sets.values()))
