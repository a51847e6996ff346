"""Developer helper: a handful of forward+backward(+Adam) steps of the bench workload (for rocprofv3 --pmc passes).
usage: python tools/pmc_steps.py [steps] [f32|bf16]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from bench import build_mesh
from facet_graph_convolution_amd.net import FacetDenoiser
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 3
dtype = sys.argv[2] if len(sys.argv) > 2 else "f32"
ds, F = build_mesh(250, 200, 0)
net = FacetDenoiser("cuda:0", seed=0, dtype=dtype).bind_mesh(ds.in_list[0], ds.adj_list[0], gt=ds.gt_list[0])
net.set_samples(np.random.RandomState(0).randint(ds.in_list[0].shape[1], size=4000))
net.set_rotation(np.eye(3))
torch.cuda.synchronize()
for _ in range(steps):
    net.forward_backward(rotate=True)
    net.adam_step()
torch.cuda.synchronize()
print("done")
