import os, sys
sys.path.insert(0, '/root/repo')
import numpy as np, torch
from bench import build_mesh
from facet_graph_convolution_amd.shard import make_sim_shards, sim_forward_backward
from facet_graph_convolution_amd.net import FacetDenoiser
ds, F = build_mesh(500, 200, 0)
x, adjs, gt = ds.in_list[0], ds.adj_list[0], ds.gt_list[0]
samp = np.random.RandomState(0).randint(x.shape[1], size=4000)
nets = make_sim_shards(x, adjs, gt, 2, "cuda:0", seed=0)
for n in nets:
    n.set_samples(samp); n.set_rotation(np.eye(3))
for _ in range(3): sim_forward_backward(nets, rotate=True)
steps = 10
for n in nets: n.profile = True
nets[0].L.fgc_profile_enable(1)
for _ in range(steps): sim_forward_backward(nets, rotate=True)
torch.cuda.synchronize()
prof = nets[0].profile_stop()
tot = 0
rows = sorted(((ms, k, c) for k, (c, ms) in prof.items()), reverse=True)
agg = {}
for ms, k, c in rows:
    kern = k.split("/", 1)[1].split("<")[0]
    a = agg.setdefault(kern, [0, 0.0]); a[0] += c; a[1] += ms
print("per shard and step (2 shards, 100k facets each):")
for kern, (c, ms) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    print("%-36s launches %5.1f  %8.1f us" % (kern, c / steps / 2, ms / steps / 2 * 1e3))
print("sum %.1f us, launches %.1f" % (sum(v[1] for v in agg.values()) / steps / 2 * 1e3, sum(v[0] for v in agg.values()) / steps / 2))
