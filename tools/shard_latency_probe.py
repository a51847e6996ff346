"""Developer helper (GPU box): what a facet-sharded step costs when its collectives have LATENCY - the one thing two shards in
one process never show (shard.sim_run's exchanges are device copies of ~5 us).  shard.SimLatency stalls the compute stream for
X us at every blocking exchange / all-reduce and lets an overlapped exchange run X us on a side stream under the shard's
interior tiles.  For each split threshold (net.split_min_tiles: layers with at least that many interior tiles run as interior |
exchange | boundary) and each X the probe prints the GPU time per shard and step, and per threshold the slope d ms / d X =
how many of a step's 17 collectives are EXPOSED.  This is the measurement behind the default threshold, behind bench.py's
`split_tune` candidates and behind the prediction table of DESIGN.md section 7.
usage: python tools/shard_latency_probe.py [nu nv world steps]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from bench import build_mesh
from facet_graph_convolution_amd.shard import make_sim_shards, sim_forward_backward_captured, SimLatency
nu, nv, world, steps = (int(a) for a in (sys.argv[1:5] + ["500", "200", "2", "20"][len(sys.argv) - 1:]))
ds, F = build_mesh(nu, nv, 0)
x, adjs, gt = ds.in_list[0], ds.adj_list[0], ds.gt_list[0]
samp = np.random.RandomState(0).randint(x.shape[1], size=4000)
nets = make_sim_shards(x, adjs, gt, world, "cuda:0", seed=0)
for n in nets:
    n.set_samples(samp); n.set_rotation(np.eye(3))
g = nets[0]._mesh["graphs"]
print("%d facets in %d shards on one GPU; shard 0: interior tiles per level fwd %s, bwd %s; hipGraph segments between the requests"
      % (F, world, [g[l].tiles["tiles_int"][1] for l in range(3)], [g[l].tiles["ttiles_int"][1] for l in range(3)]))
lats = (0, 10, 20, 40)
cal = SimLatency(10, world)
print("spin kernel: %.1f cycles per us; %d of 12 side streams run beside the compute stream" % (cal.cycles_per_us, cal.n_concurrent_streams))
thrs = (1 << 30, 1024, 256, 64)
best = {(thr, X): 1e9 for thr in thrs for X in lats}
nsplit = {}
for rep in range(3):             # (interleaved and repeated, the minimum kept: thresholds are compared inside ONE process)
    for thr in thrs:
        for n in nets:
            n.split_min_tiles = thr
            n._graph_fb = None
        pg = [g[l].pair for l in (0, 1)]
        nsplit[thr] = (sum(1 for lay in nets[0].layers[1:] if g[lay.level].tiles["tiles_int"][1] >= thr),
                       sum(1 for lay in nets[0].layers[1:] if g[lay.level].tiles["ttiles_int"][1] >= thr),
                       sum(1 for q in pg if q is not None and q.tiles["ttiles_int"][1] >= nets[0].pair_split_min_tiles))
        for X in lats:
            lat = SimLatency(X, world) if X else None
            for _ in range(3):
                sim_forward_backward_captured(nets, rotate=True, latency=lat)
            torch.cuda.synchronize(); t = time.perf_counter()
            for _ in range(steps):
                sim_forward_backward_captured(nets, rotate=True, latency=lat)
            torch.cuda.synchronize()
            best[(thr, X)] = min(best[(thr, X)], (time.perf_counter() - t) / steps / world * 1e3)
for thr in thrs:
    row = [best[(thr, X)] for X in lats]
    slope = (row[-1] - row[1]) / (lats[-1] - lats[1]) * 1e3
    print("threshold %10d (shard 0: %d fwd / %d bwd layers by their level's tiles, %d pair layers split their backward exchange): "
          "ms per shard and step at X = %s us: %s   exposed collectives (slope between %d and %d us): %.1f of 17"
          % (thr, nsplit[thr][0], nsplit[thr][1], nsplit[thr][2], "/".join(str(v) for v in lats), "  ".join("%.3f" % v for v in row),
             lats[1], lats[-1], slope))
print("loss %.4f" % nets[0].buffers["loss"][0].item())
