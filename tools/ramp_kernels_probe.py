"""Developer probe: per-kernel hipEvent durations of the steps right behind an idle GPU against the same kernels in steady
state (eager launches, library profiling on).  usage: python tools/ramp_kernels_probe.py"""
import sys, time
import numpy as np
import torch
sys.path.insert(0, ".")
import bench
from facet_graph_convolution_amd.net import FacetDenoiser
from facet_graph_convolution_amd.utils import rand_rotation_matrix

ds, F = bench.build_mesh(250, 200, 0)
dev = torch.device("cuda:0")
net = FacetDenoiser(dev).bind_mesh(ds.in_list[0], ds.adj_list[0], gt=ds.gt_list[0])
n0 = ds.in_list[0].shape[1]
rs = np.random.RandomState(100)
N = 64
samp = [rs.randint(n0, size=4000) for _ in range(N)]
rot = [rand_rotation_matrix(randnums=rs.uniform(size=3)) for _ in range(N)]
SR = FacetDenoiser.pack_step_inputs(samp, rot, dev)

def steps(k0, n):
    for k in range(k0, k0 + n):
        net.set_step_inputs_packed(SR[k % N], in_place=True)
        net.forward_backward(rotate=True, capture=False)
        net.adam_step()

steps(0, 30)                      # everything loaded and warm
torch.cuda.synchronize()
time.sleep(0.3)                   # idle
net.profile_start()
steps(30, 3)                      # the first three steps behind the idle
cold = net.profile_stop()
steps(33, 25)
net.profile_start()
steps(58, 3)
warm = net.profile_stop()
torch.cuda.synchronize()
rows = []
for k, (c, ms) in cold.items():
    if k in warm and warm[k][1] > 0:
        rows.append((ms / c * 1e3, warm[k][1] / warm[k][0] * 1e3, k))
rows.sort(reverse=True)
print("%-64s %9s %9s %7s" % ("kernel", "cold us", "warm us", "ratio"))
for c, w, k in rows[:40]:
    print("%-64s %9.2f %9.2f %7.3f" % (k[:64], c, w, c / w))
print("sum: cold %.1f us  warm %.1f us  ratio %.3f" % (sum(r[0] for r in rows), sum(r[1] for r in rows), sum(r[0] for r in rows) / sum(r[1] for r in rows)))
