"""Developer helper (GPU box): fgc_mlp_fwd / fgc_mlp_bwd of the multi-scale heads' shapes (64 / 128 input channels) at the
row counts of a 100k-facet mesh against float64 (round-5 review item 4c: which tensor gives the three-head step its 1e-3)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from facet_graph_convolution_amd import ops

for n, cin in [(333, 128), (2000, 128), (7632, 128), (7639, 128), (7639, 64), (30556, 64), (30556, 128), (7639, 32)]:
    rs = np.random.RandomState(n + cin)
    f = lambda *a, **k: torch.from_numpy(rs.normal(*a, **k).astype(np.float32))
    x = f(size=(n, cin)); dy = f(size=(n, 3))
    W1 = f(0, 0.05, (cin, 1024)); b1 = f(0, 0.01, 1024); W2 = f(0, 0.05, (1024, 3)); b2 = f(0, 0.01, 3)
    xd = x.double().requires_grad_(True)
    pd = [t.double().requires_grad_(True) for t in (W1, b1, W2)]
    h = xd @ pd[0] + pd[1]
    y = torch.where(h > 0, h, 0.1 * h) @ pd[2] + b2.double()
    (y * dy.double()).sum().backward()
    yg = ops.mlp_fwd(x.cuda(), W1.cuda(), b1.cuda(), W2.cuda(), b2.cuda(), 0.1)
    yg = yg[0] if isinstance(yg, tuple) else yg
    got = ops.mlp_bwd(x.cuda(), dy.cuda(), W1.cuda(), b1.cuda(), W2.cuda(), 0.1)
    refs = [xd.grad, pd[0].grad, pd[1].grad, pd[2].grad, dy.double().sum(0)]
    errs = ["%s %.2e" % (nm, (g.cpu().double() - r).abs().max().item() / max(1e-3, r.abs().max().item()))
            for nm, g, r in zip(["dx", "dW1", "db1", "dW2", "db2"], got, refs)]
    print("n %6d cin %3d: y %.2e  %s" % (n, cin, (yg.cpu().double() - y.detach()).abs().max().item(), "  ".join(errs)), flush=True)
