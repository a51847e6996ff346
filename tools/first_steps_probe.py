"""Developer probe: GPU time of each of the first steps after a network is bound (hipEvents around every step), eager
launches or one replayed hipGraph per step.  usage: python tools/first_steps_probe.py [graph 0|1] [steps] [idle seconds] [preheat ms of unrelated GPU work]"""
import sys, time
import numpy as np
import torch
sys.path.insert(0, ".")
import bench
from facet_graph_convolution_amd.net import FacetDenoiser
from facet_graph_convolution_amd.utils import rand_rotation_matrix

graph = int(sys.argv[1]) if len(sys.argv) > 1 else 1
nsteps = int(sys.argv[2]) if len(sys.argv) > 2 else 40
idle = float(sys.argv[3]) if len(sys.argv) > 3 else 0.0
ds, F = bench.build_mesh(250, 200, 0)
dev = torch.device("cuda:0")
net = FacetDenoiser(dev).bind_mesh(ds.in_list[0], ds.adj_list[0], gt=ds.gt_list[0])
n0 = ds.in_list[0].shape[1]
rs = np.random.RandomState(100)
samp = [rs.randint(n0, size=4000) for _ in range(nsteps)]
rot = [rand_rotation_matrix(randnums=rs.uniform(size=3)) for _ in range(nsteps)]
SR = FacetDenoiser.pack_step_inputs(samp, rot, dev)
torch.cuda.synchronize()
if idle:
    time.sleep(idle)
preheat = float(sys.argv[4]) if len(sys.argv) > 4 else 0.0
if preheat:      # unrelated work (fp32 matrix products on other memory) right in front of the first step
    a = torch.randn(4096, 4096, device=dev)
    t0 = time.perf_counter()
    while (time.perf_counter() - t0) * 1e3 < preheat:
        for _ in range(4):
            b = a @ a
        torch.cuda.synchronize()
ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(nsteps)]
wall = []
for k in range(nsteps):
    t0 = time.perf_counter()
    ev[k][0].record()
    net.set_step_inputs_packed(SR[k], in_place=True)
    net.forward_backward(rotate=True, capture=bool(graph))
    net.adam_step()
    ev[k][1].record()
    wall.append((time.perf_counter() - t0) * 1e3)
torch.cuda.synchronize()
gpu = [a.elapsed_time(b) for a, b in ev]
if len(sys.argv) > 5:      # a second batch of steps after an idle pause (same network, training continues)
    time.sleep(float(sys.argv[5]))
    ev2 = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(nsteps)]
    for k in range(nsteps):
        ev2[k][0].record()
        net.set_step_inputs_packed(SR[k], in_place=True)
        net.forward_backward(rotate=True, capture=bool(graph))
        net.adam_step()
        ev2[k][1].record()
    torch.cuda.synchronize()
    print("after %s s idle  :" % sys.argv[5], " ".join("%.3f" % a.elapsed_time(b) for a, b in ev2))
print("graph", graph, "idle", idle, "preheat", preheat)
print("gpu ms per step :", " ".join("%.3f" % v for v in gpu))
print("host ms per step:", " ".join("%.3f" % v for v in wall))
