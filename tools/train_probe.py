import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from bench import build_mesh
from facet_graph_convolution_amd.net import FacetDenoiser
from facet_graph_convolution_amd.utils import rand_rotation_matrix
nu, nv = int(sys.argv[1]), int(sys.argv[2])
ds, F = build_mesh(nu, nv, 0)
net = FacetDenoiser("cuda:0", seed=0).bind_mesh(ds.in_list[0], ds.adj_list[0], gt=ds.gt_list[0])
n0 = ds.in_list[0].shape[1]
rs = np.random.RandomState(100)
for it in range(int(sys.argv[3])):
    net.set_samples(rs.randint(n0, size=4000))
    R = rand_rotation_matrix(randnums=rs.uniform(size=3)) if sys.argv[4] == "rot" else np.eye(3)
    net.set_rotation(R)
    net.forward_backward(rotate=True, capture=(len(sys.argv) > 5 and sys.argv[5] == 'graph'))
    nosync = len(sys.argv) > 6 and sys.argv[6] == "nosync"
    g = 0.0 if nosync else net.params.grad.norm().item()
    net.adam_step()
    if nosync and it + 1 < int(sys.argv[3]):
        continue
    print(it, "loss %.3f" % net.buffers["loss"][0].item(), "|g| %.3e" % g, "det %.3f" % np.linalg.det(R), "|y0| %.3e" % net.buffers["y0"].abs().mean().item())
