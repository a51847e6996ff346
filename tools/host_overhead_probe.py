"""Developer probe: host-side enqueue time of one sharded step (rank 0 of an 8-way plan, exchanges replaced by no-ops)
against the GPU time of the same step.  Tells whether a multi-GPU run would be host-bound."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from bench import build_mesh
from facet_graph_convolution_amd.net import FacetDenoiser
from facet_graph_convolution_amd.shard import ShardPlan, graphs_to_host_csr


class NullComm:
    """No wire: everything a rank does on the host and on the GPU for an exchange except the collective call itself."""
    world, rank, host_staged = 8, 0, False
    n = 0
    def exchange(self, px): px.pack(); px.unpack(); NullComm.n += 1
    def exchange_begin(self, px): px.pack(); NullComm.n += 1; return px
    def finish(self, h):
        if h is not None: h.unpack()
    def all_reduce_sum(self, t): NullComm.n += 1


world = int(sys.argv[1]) if len(sys.argv) > 1 else 8
ds, F = build_mesh(250 * world, 200, 0)
plan = ShardPlan(graphs_to_host_csr(ds.adj_list[0]), 0, world)
net = FacetDenoiser("cuda:0", seed=0).bind_mesh(ds.in_list[0], ds.adj_list[0], gt=ds.gt_list[0], plan=plan, comm=NullComm())
n0 = ds.in_list[0].shape[1]
rs = np.random.RandomState(0)
samp = [rs.randint(n0, size=4000) for _ in range(8)]
for k in range(3):
    net.set_samples(samp[k]); net.set_rotation(np.eye(3)); net.forward_backward(rotate=True); net.adam_step()
torch.cuda.synchronize()
steps = 30
t0 = time.perf_counter()
host = 0.0
for k in range(steps):
    h0 = time.perf_counter()
    net.set_samples(samp[k % 8]); net.set_rotation(np.eye(3)); net.forward_backward(rotate=True); net.adam_step()
    host += time.perf_counter() - h0
    torch.cuda.synchronize()
tot = time.perf_counter() - t0
print("world %d shard: host enqueue %.3f ms/step, step with sync %.3f ms, %d collective calls per step" % (
    world, host / steps * 1e3, tot / steps * 1e3, NullComm.n // (steps + 3)))
# the same schedule replayed from per-segment hipGraphs
for k in range(3):
    net.set_samples(samp[k]); net.set_rotation(np.eye(3)); net.forward_backward(rotate=True, capture=True); net.adam_step()
torch.cuda.synchronize()
host = 0.0
t0 = time.perf_counter()
for k in range(steps):
    h0 = time.perf_counter()
    net.set_samples(samp[k % 8]); net.set_rotation(np.eye(3)); net.forward_backward(rotate=True, capture=True); net.adam_step()
    host += time.perf_counter() - h0
    torch.cuda.synchronize()
tot = time.perf_counter() - t0
print("world %d shard, hipGraph segments: host enqueue %.3f ms/step, step with sync %.3f ms" % (world, host / steps * 1e3, tot / steps * 1e3))
# unsharded for comparison
ds1, _ = build_mesh(250, 200, 0)
net1 = FacetDenoiser("cuda:0", seed=0).bind_mesh(ds1.in_list[0], ds1.adj_list[0], gt=ds1.gt_list[0])
n1 = ds1.in_list[0].shape[1]
for k in range(3):
    net1.set_samples(rs.randint(n1, size=4000)); net1.forward_backward(rotate=True); net1.adam_step()
torch.cuda.synchronize()
host = 0.0
t0 = time.perf_counter()
for k in range(steps):
    h0 = time.perf_counter()
    net1.set_samples(samp[k % 8] % n1); net1.set_rotation(np.eye(3)); net1.forward_backward(rotate=True); net1.adam_step()
    host += time.perf_counter() - h0
    torch.cuda.synchronize()
tot = time.perf_counter() - t0
print("unsharded: host enqueue %.3f ms/step, step with sync %.3f ms" % (host / steps * 1e3, tot / steps * 1e3))
