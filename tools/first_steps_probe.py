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
# the card's shader clock and power (sysfs, read-only) sampled by a thread while the steps run
import glob, os, threading
samples, stop = [], [False]
def _card():
    pr = torch.cuda.get_device_properties(0)
    want = "%04x:%02x:%02x.0" % (getattr(pr, "pci_domain_id", 0), getattr(pr, "pci_bus_id", 0), getattr(pr, "pci_device_id", 0))
    for c in glob.glob("/sys/class/drm/card*/device"):
        if os.path.basename(os.path.realpath(c)) == want:
            h = glob.glob(c + "/hwmon/hwmon*")
            return h[0] if h else None
    return None
hw = _card()
def _sample():
    f1, pw = hw + "/freq1_input", hw + "/power1_input"
    while not stop[0]:
        try:
            samples.append((time.perf_counter(), int(open(f1).read()) / 1e6, int(open(pw).read()) / 1e6))
        except Exception:
            pass
        time.sleep(0.0005)
th = threading.Thread(target=_sample) if hw else None
if th:
    th.start()
ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(nsteps)]
wall = []
t_start = time.perf_counter()
for k in range(nsteps):
    t0 = time.perf_counter()
    ev[k][0].record()
    net.set_step_inputs_packed(SR[k], in_place=True)
    net.forward_backward(rotate=True, capture=bool(graph))
    net.adam_step()
    ev[k][1].record()
    wall.append((time.perf_counter() - t0) * 1e3)
torch.cuda.synchronize()
t_end = time.perf_counter()
stop[0] = True
if th:
    th.join()
gpu = [a.elapsed_time(b) for a, b in ev]
if samples:
    # (the steps run back to back on the GPU from the end of step 0's capture: sample times are mapped onto steps by the
    #  cumulative GPU time counted back from the end of the run)
    ends = np.cumsum(gpu[::-1])[::-1]
    print("hwmon:", hw, "samples", len(samples))
    out = []
    for k in range(nsteps):
        lo, hi = t_end - ends[k] * 1e-3, t_end - (ends[k] - gpu[k]) * 1e-3
        v = [(f, p) for t, f, p in samples if lo <= t < hi]
        out.append("%d:%s" % (k, ("%.0fMHz/%.0fW" % (np.mean([a for a, _ in v]), np.mean([b for _, b in v]))) if v else "-"))
    print("clock / power per step:", " ".join(out))
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
