"""Developer probe: step time on an irregular 100k-facet mesh (torus with random edge flips) next to the regular one."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from facet_graph_convolution_amd.net import FacetDenoiser
from facet_graph_convolution_amd.dataClasses import TrainingSet
from facet_graph_convolution_amd.meshgen import torus, flip_edges, add_noise

for nflips in (0, 30000):
    V, F = torus(250, 200)
    t0 = time.time()
    if nflips:
        F = flip_edges(F, nflips, seed=1)
    ds = TrainingSet()
    ds.addMeshWithGT(add_noise(V, F), F, V, seed=0)
    x, adjs, gt = ds.in_list[0], ds.adj_list[0], ds.gt_list[0]
    deg = [(a[0] > 0).sum(1) for a in adjs]
    net = FacetDenoiser("cuda:0", seed=0).bind_mesh(x, adjs, gt=gt)
    rs = np.random.RandomState(0)
    for _ in range(3):
        net.train_step(sample_ind=rs.randint(x.shape[1], size=4000), R=np.eye(3))
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    for _ in range(20):
        net.train_step(sample_ind=rs.randint(x.shape[1], size=4000), R=np.eye(3))
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t1) / 20 * 1e3
    print("flips %6d: N0 %d, max degree per level %s, mean %s -> %.3f ms/step (prep %.1f s)" % (
        nflips, x.shape[1], [int(d.max()) for d in deg], ["%.1f" % d.mean() for d in deg], ms, time.time() - t0))
