"""Developer probe (GPU box, run it ONCE per question): does Python's cyclic garbage collector, running in the middle of a
stream capture, abort the process when the garbage owns hipGraphs?  The mechanism behind the round-3 abort as the round-4
stack shows it: a dead cycle [network -> its captured segment graphs] is collected inside
net._capture_segments; its destructors call hipGraphExecDestroy / hipGraphDestroy / hipFree while the stream captures.

usage: python tools/capture_gc_probe.py [guard|noguard]
  guard    (default) the capture runs under net._no_gc_while_capturing: must print "probe ok"
  noguard  FGC_NO_CAPTURE_GC_GUARD=1: the collector is forced to run inside the capture (gc.set_threshold(1))"""
import gc, os, sys
mode = sys.argv[1] if len(sys.argv) > 1 else "guard"
if mode == "noguard":
    os.environ["FGC_NO_CAPTURE_GC_GUARD"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from facet_graph_convolution_amd.shard import make_sim_shards, sim_forward_backward, sim_forward_backward_captured
from facet_graph_convolution_amd.dataClasses import TrainingSet
from facet_graph_convolution_amd.meshgen import torus, add_noise

V, F = torus(48, 40)
ds = TrainingSet()
ds.addMeshWithGT(add_noise(V, F, 0.2, seed=1), F, V, seed=0)
x, adjs, gt = ds.in_list[0], ds.adj_list[0], ds.gt_list[0]
samp = np.random.RandomState(0).randint(x.shape[1], size=4000)


def shards():
    nets = make_sim_shards(x, adjs, gt, 2, "cuda:0", seed=0)
    for n in nets:
        n.set_samples(samp)
        n.set_rotation(np.eye(3))
    sim_forward_backward(nets, rotate=True)
    return nets


# 1. networks with captured segment graphs, then dead but kept alive by a reference cycle (the collector stays off while
#    they are made, and afterwards its thresholds are out of reach: nothing is collected before the capture by chance)
gc.disable()
for _ in range(3):
    nets = shards()
    sim_forward_backward_captured(nets, rotate=True)
    torch.cuda.synchronize()
    cyc = [nets]
    cyc.append(cyc)
    del nets, cyc
nets = shards()
gc.set_threshold(10 ** 9, 10 ** 9, 10 ** 9)
gc.enable()
# 2. the collector "fires" at the first schedule tag INSIDE a capture - if it is enabled there, as an allocation-count trigger
#    would find it (net._no_gc_while_capturing turns it off for the capture and has collected before it)
from facet_graph_convolution_amd.net import FacetDenoiser
fired = []
orig_tag = FacetDenoiser._tag


def tag(self, name):
    if not fired and torch.cuda.is_current_stream_capturing():
        fired.append(gc.collect() if gc.isenabled() else -1)
    return orig_tag(self, name)


FacetDenoiser._tag = tag
sim_forward_backward_captured(nets, rotate=True)
torch.cuda.synchronize()
gc.set_threshold(700, 10, 10)
print("probe ok (%s): loss %.4f; inside the capture the collector %s" % (
    mode, nets[0].buffers["loss"][0].item(),
    "was off" if fired == [-1] else "ran and freed %d objects" % fired[0] if fired else "never got its turn"))
