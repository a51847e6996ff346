"""Developer probe: the bf16-storage network against the fp32 one on the same mesh and weights, buffer by buffer."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from facet_graph_convolution_amd.net import FacetDenoiser
from facet_graph_convolution_amd.dataClasses import TrainingSet
from facet_graph_convolution_amd.meshgen import icosphere, torus, add_noise, flip_edges

which = sys.argv[1] if len(sys.argv) > 1 else "ico"
if which == "ico":
    V, F = icosphere(3)
elif which == "irr":
    V, F = torus(24, 20)
    F = flip_edges(F, 400, seed=1)
else:
    V, F = torus(60, 40)
ds = TrainingSet()
ds.addMeshWithGT(add_noise(V, F), F, V, seed=0)
x, adjs, gt = ds.in_list[0], ds.adj_list[0], ds.gt_list[0]
samp = np.random.RandomState(2).randint(x.shape[1], size=4000)
nets = {}
for dt in ("f32", "bf16"):
    net = FacetDenoiser("cuda:0", seed=0, dtype=dt).bind_mesh(x, adjs, gt=gt)
    net.set_samples(samp)
    net.set_rotation(np.eye(3))
    net.forward_backward(rotate=True)
    torch.cuda.synchronize()
    nets[dt] = net
a, b = nets["f32"].buffers, nets["bf16"].buffers
for k in ["h1", "p1", "h2", "p2", "h3", "d3", "u2", "d2", "u1", "d1", "y0", "nconv", "g_y0", "g_d1", "g_u1", "g_h1", "g_d2", "g_u2",
          "g_h2", "g_d3", "g_h3", "g_p2", "g_p1"]:
    ra, rb = a[k].float().cpu().numpy(), b[k].float().cpu().numpy()
    sc = max(np.abs(ra).max(), 1e-12)
    print("%-8s max|f32| %.3e  max err %.3e  rel %.3e" % (k, sc, np.abs(ra - rb).max(), np.abs(ra - rb).max() / sc))
print("loss f32 %.5f bf16 %.5f" % (a["loss"][0].item(), b["loss"][0].item()))
spec = nets["f32"].params.spec
for i, (ga, gb) in enumerate(zip(nets["f32"].params.grads, nets["bf16"].params.grads)):
    ra, rb = ga.cpu().numpy(), gb.cpu().numpy()
    sc = max(np.abs(ra).max(), 1e-12)
    print("grad %2d %-10s %-16s max %.3e rel err %.3e" % (i, spec[i][0], spec[i][1], sc, np.abs(ra - rb).max() / sc))
