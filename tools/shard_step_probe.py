"""Developer helper: GPU time of a facet-sharded step as N shards in ONE process (shard.sim_run: the exchanges are device
copies), with the fused loss end (fgc_loss_shard_*) and with the separate launches (FGC_NO_FUSED_LOSS=1).
usage: python tools/shard_step_probe.py [nu nv world steps]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from bench import build_mesh
from facet_graph_convolution_amd.shard import make_sim_shards, sim_forward_backward, sim_forward_backward_captured
nu, nv, world, steps = (int(a) for a in (sys.argv[1:5] + ["500", "200", "2", "20"][len(sys.argv) - 1:]))
ds, F = build_mesh(nu, nv, 0)
x, adjs, gt = ds.in_list[0], ds.adj_list[0], ds.gt_list[0]
samp = np.random.RandomState(0).randint(x.shape[1], size=4000)
for mode in ("1", "0"):
    os.environ["FGC_NO_FUSED_LOSS"] = mode
    nets = make_sim_shards(x, adjs, gt, world, "cuda:0", seed=0)
    for n in nets:
        n.set_samples(samp); n.set_rotation(np.eye(3))
    out = {}
    for name, fn in (("eager", sim_forward_backward), ("hipGraph segments", sim_forward_backward_captured)):
        for _ in range(3):
            fn(nets, rotate=True)
        torch.cuda.synchronize(); t = time.perf_counter()
        for _ in range(steps):
            fn(nets, rotate=True)
        torch.cuda.synchronize(); out[name] = (time.perf_counter() - t) / steps * 1e3
    print("FGC_NO_FUSED_LOSS=%s: %d facets in %d shards on one GPU: %.3f ms per step eager, %.3f ms hipGraph segments (all shards, "
          "simulated exchanges included), loss %.4f" % (mode, F, world, out["eager"], out["hipGraph segments"], nets[0].buffers["loss"][0].item()))
    del nets
