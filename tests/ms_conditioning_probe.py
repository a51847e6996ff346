"""Developer helper (CPU): why the three-head training step at 100k facets sits at 1e-3 of the float64 oracle while the
single-head step sits at 2e-5 (round-5 review, item 4c).

Runs oracle/model_csr_ref.train_loss_ms on the headline mesh twice - float64 and float32, the SAME closed form, nothing of the
HIP path - and prints (a) per tensor the float32-vs-float64 gradient error, i.e. what ANY fp32 evaluation of this objective
gets, (b) per head the samples with the largest |d loss / d cos| = 1 / sqrt(1 - cos^2): a sampled row whose prediction is
within 1e-6 of (anti)parallel to its target carries a gradient ~1000x a typical row's and its fp32 cosine is uncertain by
6e-8 / (1 - cos^2).  usage: python tests/ms_conditioning_probe.py [nu nv]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from oracle import model_csr_ref as C
from oracle import model_ref as R
from facet_graph_convolution_amd.dataClasses import TrainingSet
from facet_graph_convolution_amd.meshgen import torus, add_noise
from facet_graph_convolution_amd.utils import rand_rotation_matrix

nu, nv = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (250, 200)
torch.set_num_threads(min(len(os.sched_getaffinity(0)), 32))
V, F = torus(nu, nv)
ds = TrainingSet()
ds.addMeshWithGT(add_noise(V, F, 0.2, seed=1), F, V, seed=0)
x, adjs, gt = ds.in_list[0], ds.adj_list[0], ds.gt_list[0]
samp = np.random.RandomState(2).randint(x.shape[1], size=4000)
Rm = rand_rotation_matrix(randnums=np.random.RandomState(3).uniform(size=3)).astype(np.float32)
names = None
res = {}
for dt in (torch.float64, torch.float32):
    C.DT = dt
    params = C.init_params(0, multi_scale=True)
    tot, losses, nconvs = C.train_loss_ms(x.astype(np.float32), adjs, gt.astype(np.float32), params, samp, Rm)
    tot.backward()
    res[dt] = ([p.grad.double() for p in params], [l.item() for l in losses], [n[0].detach().double() for n in nconvs])
    del tot, losses, nconvs, params
g64, l64, n64 = res[torch.float64]
g32, l32, n32 = res[torch.float32]
print("losses f64 %s  f32 %s" % (l64, l32))
rows = []
for i, (a, b) in enumerate(zip(g32, g64)):
    big = b.abs().max().item()
    rows.append((i, tuple(b.shape), big, (a - b).abs().max().item() / max(big, 1e-3)))
print("float32 closed form vs float64 closed form, per tensor (worst ten):")
for i, sh, big, rel in sorted(rows, key=lambda t: -t[3])[:10]:
    print("  %3d %-18s max|ref| %.4e  rel err %.3e" % (i, sh, big, rel))
# per head: the ill-conditioned samples
gtt = torch.as_tensor(gt.astype(np.float32), dtype=torch.float64).reshape(1, -1, 3)
Rt = torch.as_tensor(Rm, dtype=torch.float64)
gts = [torch.matmul(gtt, Rt.t())]
g = gtt
for k in (1, 2):
    g = R.pooled_gt(g)
    gts.append(torch.matmul(g, Rt.t()))
idx = torch.as_tensor(samp, dtype=torch.long)
for k in range(3):
    ik = idx % n64[k].shape[0]
    c64 = (n64[k][ik] * gts[k][0, ik]).sum(-1)
    c32 = (n32[k][ik] * gts[k][0, ik]).sum(-1)
    real = gts[k][0, ik].abs().sum(-1) > 1e-3
    w = torch.where(real & (c64.abs() < 0.9999999), 1.0 / torch.sqrt((1 - c64 * c64).clamp_min(1e-30)), torch.zeros_like(c64))
    top = torch.argsort(-w)[:5]
    print("head %d: %d real samples, sum |dloss/dcos| %.1f, largest five: %s" % (
        k, int(real.sum()), w.sum().item(),
        ", ".join("cos %.9f w %.0f (fp32 cos off by %.1e)" % (c64[t].item(), w[t].item(), (c32[t] - c64[t]).item()) for t in top)))
