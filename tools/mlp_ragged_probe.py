"""Developer helper (GPU box): fgc_mlp_bwd over every residue of the row count modulo 64 (round 6: a ragged last tile of
23 rows gave the 128-wide head dx / dW1 / db1 errors of 1e-2 at n = 7639)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from facet_graph_convolution_amd import ops
base = int(sys.argv[1]) if len(sys.argv) > 1 else 7616
for cin in (32, 64, 128):
    bad = []
    for r in range(0, 64):
        n = base + r
        rs = np.random.RandomState(n + cin)
        f = lambda *a, **k: torch.from_numpy(rs.normal(*a, **k).astype(np.float32))
        x = f(size=(n, cin)); dy = f(size=(n, 3))
        W1 = f(0, 0.05, (cin, 1024)); b1 = f(0, 0.01, 1024); W2 = f(0, 0.05, (1024, 3))
        xd = x.double().requires_grad_(True)
        pd = [t.double().requires_grad_(True) for t in (W1, b1, W2)]
        h = xd @ pd[0] + pd[1]
        y = torch.where(h > 0, h, 0.1 * h) @ pd[2]
        (y * dy.double()).sum().backward()
        got = ops.mlp_bwd(x.cuda(), dy.cuda(), W1.cuda(), b1.cuda(), W2.cuda(), 0.1)
        refs = [xd.grad, pd[0].grad, pd[1].grad, pd[2].grad, dy.double().sum(0)]
        e = [(g.cpu().double() - rr).abs().max().item() / max(1e-3, rr.abs().max().item()) for g, rr in zip(got, refs)]
        if max(e) > 5e-6:
            rows = ((got[0].cpu().double() - xd.grad).abs().max(dim=1).values > 1e-5).nonzero().flatten().tolist()
            bad.append((r, "%.1e" % max(e), rows[:4], len(rows)))
    print("cin %3d base %d: bad residues (r, worst err, first wrong dx rows, count): %s" % (cin, base, bad), flush=True)
