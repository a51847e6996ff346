"""Developer helper (GPU box): for fgc_mlp_bwd rows of dx that differ from float64, which hidden unit explains the difference."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from facet_graph_convolution_amd import ops
n, cin, seed = [int(a) for a in sys.argv[1:4]] if len(sys.argv) > 3 else (7639, 32, 5)
rs = np.random.RandomState(seed)
f = lambda *a, **k: torch.from_numpy(rs.normal(*a, **k).astype(np.float32))
x, dy = f(size=(n, cin)), f(size=(n, 3))
W1, b1, W2 = f(0, 0.05, (cin, 1024)), f(0, 0.01, 1024), f(0, 0.05, (1024, 3))
xd, W1d, b1d, W2d, dyd = x.double(), W1.double(), b1.double(), W2.double(), dy.double()
h = xd @ W1d + b1d
g = dyd @ W2d.t()
slope = torch.where(h > 0, torch.ones_like(h), torch.full_like(h, 0.1))
dx_ref = (g * slope) @ W1d.t()
got = ops.mlp_bwd(x.cuda(), dy.cuda(), W1.cuda(), b1.cuda(), W2.cuda(), 0.1)
dx = got[0].cpu().double()
res = dx - dx_ref
bad = (res.abs().max(dim=1).values > 1e-5).nonzero().flatten().tolist()
print("rows of dx off by more than 1e-5:", bad)
for i in bad:
    r = res[i]
    # least squares over all hidden units of this row: r = sum_k c_k * (0.9 g_ik sign) W1[:, k]; look at single-unit fits
    fits = []
    for k in range(1024):
        v = g[i, k] * ((0.1 if h[i, k] > 0 else 1.0) - slope[i, k]) * W1d[:, k]
        c = (r @ v) / (v @ v)
        fits.append(((r - c * v).norm().item(), k, c.item(), h[i, k].item()))
    fits.sort()
    print("row %d: |res| %.3e; best single-unit fits (remaining |res|, k, coefficient, h64): %s" % (
        i, r.norm().item(), [(("%.1e" % a), k, round(c, 3), "%.2e" % hh) for a, k, c, hh in fits[:3]]))
    hs = h[i].abs().sort()
    print("        smallest |h64| in the row:", ["%.2e" % v for v in hs.values[:4].tolist()])
