"""Developer probe: max error of fgc_mlp_fwd against float64 (run with and without FGC_NO_MLP_SPLIT=1)."""
import numpy as np, torch, sys
sys.path.insert(0, ".")
from facet_graph_convolution_amd import ops
rs = np.random.RandomState(5)
x = torch.from_numpy(rs.normal(size=(4096, 32)).astype(np.float32))
W1 = torch.from_numpy(rs.normal(0, 0.05, (32, 1024)).astype(np.float32)); b1 = torch.from_numpy(rs.normal(0, 0.01, 1024).astype(np.float32))
W2 = torch.from_numpy(rs.normal(0, 0.05, (1024, 3)).astype(np.float32)); b2 = torch.from_numpy(rs.normal(0, 0.01, 3).astype(np.float32))
h = x.double() @ W1.double() + b1.double()
ref = torch.where(h > 0, h, 0.1 * h) @ W2.double() + b2.double()
y = ops.mlp_fwd(x.cuda(), W1.cuda(), b1.cuda(), W2.cuda(), b2.cuda(), 0.1)
print("ERR %.6e  (|y| max %.3f)" % ((y.cpu().double() - ref).abs().max().item(), ref.abs().max().item()))
