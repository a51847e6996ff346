import sys, os
sys.path.insert(0, os.getcwd())
import numpy as np, torch
from bench import build_mesh
from facet_graph_convolution_amd.net import FacetDenoiser
from facet_graph_convolution_amd.utils import rand_rotation_matrix
mode, inputs = sys.argv[1], sys.argv[2]
variant = sys.argv[4] if len(sys.argv) > 4 else ""
ds, F = build_mesh(250, 200, 0)
net = FacetDenoiser("cuda:0", seed=0).bind_mesh(ds.in_list[0], ds.adj_list[0], gt=ds.gt_list[0])
n0 = ds.in_list[0].shape[1]
rs = np.random.RandomState(100)
N = 23
samp = [rs.randint(n0, size=4000) for _ in range(N)]
rot = [rand_rotation_matrix(randnums=rs.uniform(size=3)) for _ in range(N)]
S_all = torch.from_numpy(np.stack(samp).astype(np.int32)).cuda()
R_all = torch.from_numpy(np.stack(rot).astype(np.float32).reshape(N, 9)).cuda()
torch.cuda.synchronize()
for k in range(N):
    if inputs == "dev":
        net.set_step_inputs_device(S_all[k], R_all[k])
    else:
        net.set_samples(samp[k]); net.set_rotation(rot[k])
    net.forward_backward(rotate=True, capture=(mode == "graph"))
    if variant == "fence":      # an explicit (semantically empty) dependency between the replay and what follows
        e = torch.cuda.Event(); e.record(); torch.cuda.current_stream().wait_event(e)
    net.adam_step()
    if len(sys.argv) > 3 and k == int(sys.argv[3]) - 1:
        if variant == "streamsync":
            torch.cuda.current_stream().synchronize()
        elif variant == "eventsync":
            e = torch.cuda.Event(); e.record(); e.synchronize()
        else:
            torch.cuda.synchronize()
        if variant == "recapture":
            net._graph_fb = None
torch.cuda.synchronize()
print(mode, inputs, variant, "loss %.4f" % net.buffers["loss"][0].item())
