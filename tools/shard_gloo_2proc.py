"""Developer check: 2 processes on ONE GPU, gloo (host-staged) exchange, real kernels; compares the sharded
loss with the single-process run.  python -m torch.distributed.run --nproc-per-node 2 tools/shard_gloo_2proc.py
FGC_TOOL_BACKEND=nccl with --nproc-per-node 1 drives the RCCL code path (async all_to_all_single, all_reduce) on the
one GPU of a test box: a world of one, so every exchange is empty, but the calls are the ones a multi-GPU run makes."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, torch.distributed as dist
from bench import build_mesh
from facet_graph_convolution_amd.net import FacetDenoiser
from facet_graph_convolution_amd.shard import ShardPlan, DistComm, graphs_to_host_csr
os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
backend = os.environ.get("FGC_TOOL_BACKEND", "gloo")
if backend == "nccl":
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", device_id=torch.device("cuda", 0))
else:
    dist.init_process_group("gloo")
rank, world = dist.get_rank(), dist.get_world_size()
ds, F = build_mesh(60, 40, 0)
plan = ShardPlan(graphs_to_host_csr(ds.adj_list[0]), rank, world)
net = FacetDenoiser("cuda:0", seed=0).bind_mesh(ds.in_list[0], ds.adj_list[0], gt=ds.gt_list[0], plan=plan, comm=DistComm())
n0 = ds.in_list[0].shape[1]
samp = np.random.RandomState(0).randint(n0, size=4000)
net.set_samples(samp); net.set_rotation(np.eye(3))
net.forward_backward(rotate=True)
torch.cuda.synchronize()
loss = net.buffers["loss"][0].item()
gn = net.params.grad.norm().item()
# the same schedule replayed from hipGraphs (one per stretch of launches between two exchanges): bit-identical
g_eager = net.params.grad.clone()
for it in range(3):       # first call: eager warm-up + capture; then two replays
    net.forward_backward(rotate=True, capture=True)
torch.cuda.synchronize()
assert net.buffers["loss"][0].item() == loss and torch.equal(net.params.grad, g_eager), "graph replay differs from eager"
nseg = sum(1 for s in net._graph_fb[0] for g, _ in s if g is not None)
if rank == 0:
    print("captured schedule: %d graphs per step" % nseg)
if rank == 0:
    ref = FacetDenoiser("cuda:0", seed=0).bind_mesh(ds.in_list[0], ds.adj_list[0], gt=ds.gt_list[0])
    ref.set_samples(samp); ref.set_rotation(np.eye(3)); ref.forward_backward(rotate=True); torch.cuda.synchronize()
    print("sharded loss %.6f |g| %.6f  vs single %.6f |g| %.6f" % (loss, gn, ref.buffers["loss"][0].item(), ref.params.grad.norm().item()))
    assert abs(loss - ref.buffers["loss"][0].item()) < 1e-3 and abs(gn - ref.params.grad.norm().item()) < 1e-3 * gn
    for i, (a, b) in enumerate(zip(net.params.grads, ref.params.grads)):
        err = (a - b).abs().max().item() / max(b.abs().max().item(), 1e-3)
        assert err < 1e-3, "grad %d differs from the unsharded network: %.3e" % (i, err)
    g = net._mesh["graphs"]
    print("split threshold %d: interior tiles per level %s" % (net.split_min_tiles, [g[l].tiles["tiles_int"][1] for l in range(3)]))
    print("OK")
dist.destroy_process_group()
