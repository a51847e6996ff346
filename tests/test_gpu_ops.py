"""GPU parity of the MLP and the element-wise / reduction ops against the oracle (torch fp32/fp64 on CPU)."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _t(a):
    return torch.tensor(np.asarray(a, dtype=np.float32))


@pytest.mark.parametrize("n,cin", [(1000, 32), (77, 32), (300, 64), (130, 128)])
def test_mlp_forward(n, cin):
    from facet_graph_convolution_amd import ops
    from oracle import model_ref as R
    rs = np.random.RandomState(n)
    x = _t(rs.normal(size=(n, cin)))
    W1, b1 = _t(rs.normal(0, 0.05, (cin, 1024))), _t(rs.normal(0, 0.01, 1024))
    W2, b2 = _t(rs.normal(0, 0.05, (1024, 3))), _t(rs.normal(0, 0.01, 3))
    ref = R.custom_lin(R.lrelu(R.custom_lin(x.double(), W1.double(), b1.double())), W2.double(), b2.double())
    y, part = ops.mlp_fwd(x.to(DEV), W1.to(DEV), b1.to(DEV), W2.to(DEV), b2.to(DEV), 0.1, want_abs_partial=True)
    np.testing.assert_allclose(y.cpu().numpy(), ref.numpy(), atol=3e-6)
    assert abs(part.sum().item() - ref.abs().sum().item()) < 1e-4 * ref.abs().sum().item()


def test_mlp_forward_split_operands_are_as_accurate_as_the_fp32_mfma():
    """fgc_mlp_fwd runs its 1024-wide product on the bf16 matrix pipe with three-term operand splits (six bf16 MFMAs
    per product, fp32 accumulation; fgc_mlp_bf16.hip) and keeps the fp32-MFMA kernel behind FGC_NO_MLP_SPLIT=1.  Both
    against a float64 reference on the same inputs: the split form must be as close as the fp32 MFMA is."""
    import subprocess, sys, textwrap
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = textwrap.dedent("""
        import numpy as np, torch
        from facet_graph_convolution_amd import ops
        rs = np.random.RandomState(5)
        x = torch.from_numpy(rs.normal(size=(4096, 32)).astype(np.float32))
        W1 = torch.from_numpy(rs.normal(0, 0.05, (32, 1024)).astype(np.float32)); b1 = torch.from_numpy(rs.normal(0, 0.01, 1024).astype(np.float32))
        W2 = torch.from_numpy(rs.normal(0, 0.05, (1024, 3)).astype(np.float32)); b2 = torch.from_numpy(rs.normal(0, 0.01, 3).astype(np.float32))
        h = x.double() @ W1.double() + b1.double()
        ref = torch.where(h > 0, h, 0.1 * h) @ W2.double() + b2.double()
        y = ops.mlp_fwd(x.cuda(), W1.cuda(), b1.cuda(), W2.cuda(), b2.cuda(), 0.1)
        y = y[0] if isinstance(y, tuple) else y
        print("ERR %.6e" % (y.cpu().double() - ref).abs().max().item())
    """)
    errs = {}
    for tag, env in (("split", {}), ("fp32", {"FGC_NO_MLP_SPLIT": "1"})):
        out = subprocess.run([sys.executable, "-c", code], cwd=repo, env=dict(os.environ, **env), capture_output=True,
                             text=True, timeout=300)
        assert out.returncode == 0, out.stderr[-2000:]
        errs[tag] = float([l for l in out.stdout.splitlines() if l.startswith("ERR")][0].split()[1])
    assert errs["fp32"] < 3e-6 and errs["split"] < 3e-6
    assert errs["split"] < 2.0 * errs["fp32"] + 2e-7, errs


def test_mlp_backward_split_operands_are_as_accurate_as_the_fp32_mfma():
    """fgc_mlp_bwd with 32 input channels (the network's head) runs every 1024-wide product on the bf16 matrix pipe with
    three-term operand splits (mlp_bwd_dx_split_kernel / mlp_bwd_w_split_kernel, fgc_mlp_bf16.hip) and keeps the fused
    fp32-MFMA kernel behind FGC_NO_MLP_BWD_SPLIT=1.  Both against float64 on the same inputs, 5 000 rows (several tiles per
    walker, a ragged last tile): the split form must be as close as the fp32 MFMA is."""
    import subprocess, sys, textwrap
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = textwrap.dedent("""
        import numpy as np, torch
        from facet_graph_convolution_amd import ops
        rs = np.random.RandomState(11)
        n = 5003
        f = lambda *a, **k: torch.from_numpy(rs.normal(*a, **k).astype(np.float32))
        x = f(size=(n, 32)); dy = f(size=(n, 3))
        W1 = f(0, 0.05, (32, 1024)); b1 = f(0, 0.01, 1024); W2 = f(0, 0.05, (1024, 3))
        xd = x.double().requires_grad_(True)
        pd = [t.double().requires_grad_(True) for t in (W1, b1, W2)]
        h = xd @ pd[0] + pd[1]
        y = torch.where(h > 0, h, 0.1 * h) @ pd[2]
        (y * dy.double()).sum().backward()
        got = ops.mlp_bwd(x.cuda(), dy.cuda(), W1.cuda(), b1.cuda(), W2.cuda(), 0.1)
        refs = [xd.grad, pd[0].grad, pd[1].grad, pd[2].grad, dy.double().sum(0)]
        for name, g, r in zip(["dx", "dW1", "db1", "dW2", "db2"], got, refs):
            print("ERR", name, "%.6e" % ((g.cpu().double() - r).abs().max().item() / max(1.0, r.abs().max().item())))
    """)
    errs = {}
    for tag, env in (("split", {}), ("fp32", {"FGC_NO_MLP_BWD_SPLIT": "1"})):
        out = subprocess.run([sys.executable, "-c", code], cwd=repo, env=dict(os.environ, **env), capture_output=True,
                             text=True, timeout=300)
        assert out.returncode == 0, out.stderr[-2000:]
        errs[tag] = {l.split()[1]: float(l.split()[2]) for l in out.stdout.splitlines() if l.startswith("ERR")}
    for name in ("dx", "dW1", "db1", "dW2", "db2"):
        assert errs["fp32"][name] < 5e-6 and errs["split"][name] < 5e-6, (name, errs)
        assert errs["split"][name] < 2.0 * errs["fp32"][name] + 3e-7, (name, errs)


@pytest.mark.parametrize("n,cin", [(1000, 32), (300, 64), (130, 128), (70, 48)])
def test_mlp_operands_packed_by_the_step_prologue(n, cin):
    """fgc_conv_pack with an fgc_pack_extra leaves the MLP's operands (split planes or the fp32 operand, whichever the shape
    takes) and the rotated input rows in ONE launch; fgc_mlp_fwd / fgc_mlp_bwd with FGC_MLP_PACKED then give bit for bit what
    they give when they pack themselves, and the rotation what fgc_rotate_rows gives."""
    import ctypes as C
    from facet_graph_convolution_amd import _lib, ops
    L, p, st = _lib.lib(), _lib.ptr, _lib.stream_ptr()
    rs = np.random.RandomState(n)
    x = _t(rs.normal(size=(n, cin))).to(DEV)
    W1, b1 = _t(rs.normal(0, 0.05, (cin, 1024))).to(DEV), _t(rs.normal(0, 0.01, 1024)).to(DEV)
    W2, b2 = _t(rs.normal(0, 0.05, (1024, 3))).to(DEV), _t(rs.normal(0, 0.01, 3)).to(DEV)
    dy = _t(rs.normal(size=(n, 3))).to(DEV)
    rows = _t(rs.normal(size=(n, 6))).to(DEV)
    Rm = _t(np.linalg.qr(rs.normal(size=(3, 3)))[0].reshape(9)).to(DEV)
    y_ref = ops.mlp_fwd(x, W1, b1, W2, b2, 0.1)
    y_ref = y_ref[0] if isinstance(y_ref, tuple) else y_ref
    g_ref = ops.mlp_bwd(x, dy, W1, b1, W2, 0.1)
    rot_ref = torch.empty_like(rows)
    _lib.check(L.fgc_rotate_rows(p(rows), p(rot_ref), n, 2, p(Rm), st))
    wsf = torch.empty(L.fgc_mlp_workspace_bytes(cin, 1024, 3) + 256, dtype=torch.uint8, device=DEV)
    wsb = torch.empty(L.fgc_mlp_bwd_workspace_bytes(n, cin, 1024, 3) + 256, dtype=torch.uint8, device=DEV)
    rot = torch.empty_like(rows)
    # ... and the first layer's logit table of the rotated rows (fgc_pack_extra.rot_ag): ag[:, 0:9] = u x + c, [12:21] = v x
    u, v = _t(rs.normal(0, 0.3, (9, 6))).to(DEV), _t(rs.normal(0, 0.3, (9, 6))).to(DEV)
    cc = _t(rs.normal(0, 0.3, 9)).to(DEV)
    ag = torch.full((n, 24), 7.0, device=DEV)
    ex = _lib.PackExtra(rot_x=p(rows), rot_y=p(rot), rot_R=p(Rm), rot_rows=n, rot_vecs=2, rot_ag=p(ag), rot_u=p(u), rot_c=p(cc),
                        rot_v=p(v), mlp_bf16=0, mlp_W1=p(W1),
                        mlp_W2=p(W2), mlp_n=n, mlp_cin=cin, mlp_hidden=1024, mlp_cout=3, mlp_fwd_ws=p(wsf), mlp_bwd_ws=p(wsb))
    _lib.check(L.fgc_conv_pack(None, None, None, None, 0, C.byref(ex), st))
    y = torch.empty(n, 3, device=DEV)
    _lib.check(L.fgc_mlp_fwd(p(x), n, cin, 1024, 3, p(W1), p(b1), p(W2), p(b2), 0.1, p(y), None, _lib.MLP_PACKED, p(wsf),
                             wsf.numel(), st))
    dx = torch.empty_like(x)
    g = [torch.empty_like(t) for t in (W1, b1, W2, b2)]
    _lib.check(L.fgc_mlp_bwd(p(x), p(dy), n, cin, 1024, 3, p(W1), p(b1), p(W2), 0.1, p(dx), p(g[0]), p(g[1]), p(g[2]), p(g[3]),
                             _lib.MLP_PACKED, p(wsb), wsb.numel(), st))
    torch.cuda.synchronize()
    assert torch.equal(rot, rot_ref)
    ag_ref = torch.zeros(n, 24, dtype=torch.float64)
    ag_ref[:, 0:9] = rot_ref.cpu().double() @ u.cpu().double().t() + cc.cpu().double()
    ag_ref[:, 12:21] = rot_ref.cpu().double() @ v.cpu().double().t()
    assert (ag.cpu().double() - ag_ref).abs().max().item() < 2e-6
    assert torch.equal(y, y_ref)
    for a, b in zip([dx] + g, g_ref):
        assert torch.equal(a, b)


@pytest.mark.parametrize("n,cin", [(1000, 32), (77, 32), (5000, 32), (6, 32), (900, 64), (333, 128), (70, 48)])
def test_mlp_backward(n, cin):
    """cin = 32 is the network head; 64 / 128 are the multi-scale heads (model.py:894-899, 915-920)."""
    from facet_graph_convolution_amd import ops
    from oracle import model_ref as R
    rs = np.random.RandomState(n + 1)
    x = _t(rs.normal(size=(n, cin))).double().requires_grad_(True)
    ps = [_t(rs.normal(0, 0.05, (cin, 1024))), _t(rs.normal(0, 0.01, 1024)), _t(rs.normal(0, 0.05, (1024, 3))),
          _t(rs.normal(0, 0.01, 3))]
    pd = [p.double().requires_grad_(True) for p in ps]
    dy = _t(rs.normal(size=(n, 3)))
    y = R.custom_lin(R.lrelu(R.custom_lin(x, pd[0], pd[1])), pd[2], pd[3])
    (y * dy.double()).sum().backward()
    dx, dW1, db1, dW2, db2 = ops.mlp_bwd(x.detach().float().to(DEV), dy.to(DEV), ps[0].to(DEV), ps[1].to(DEV),
                                         ps[2].to(DEV), 0.1)
    for got, ref, name in [(dx, x.grad, "dx"), (dW1, pd[0].grad, "dW1"), (db1, pd[1].grad, "db1"),
                           (dW2, pd[2].grad, "dW2"), (db2, pd[3].grad, "db2")]:
        scale = max(1.0, ref.abs().max().item())
        err = (got.cpu().double() - ref).abs().max().item() / scale
        assert err < 5e-6, (name, err)


@pytest.mark.parametrize("n,cin,seed", [(7639, 128, 0), (7639, 128, 9), (7639, 32, 5), (30556, 64, 1)])
def test_mlp_backward_at_a_coarse_head_size_is_exact_up_to_the_sign_of_zero_preactivations(n, cin, seed):
    """The round-5 three-head step at 100k facets sat at 1e-3 of the float64 oracle in the level-2 head (7 639 rows x 128
    channels) while every other tensor sat at 2e-5.  Cause: the leaky ReLU's kink.  Of the n x 1024 hidden pre-activations a
    handful lie within fp32 rounding of zero (|h| < 1e-7); whether such a unit takes slope 1 or 0.1 - or 0, when the fp32 sum
    is exactly zero - is decided by the last bit of an fp32 sum, and ONE unit on another branch moves its row of dx, its
    column of dW1 and its entry of db1 by 0.9 g_ik (g_ik): 1e-3 ... 1e-2 of a tensor whose entries sum over a few thousand
    rows only.  Plain torch fp32 against torch float64 does the same on these seeds.  So: every unit
    whose float64 pre-activation is within 1e-6 of zero is AMBIGUOUS; for each one the reference is moved to the branch the
    kernel took (read off db1, which the unit touches in one entry, and off the unit's row of dx); after that the kernel must
    match float64 to 5e-6 like at every other size - a wrong row tile, a dropped ragged tail or a misplaced slab would not
    survive that."""
    from facet_graph_convolution_amd import ops
    rs = np.random.RandomState(seed)
    f = lambda *a, **k: torch.from_numpy(rs.normal(*a, **k).astype(np.float32))
    x, dy = f(size=(n, cin)), f(size=(n, 3))
    W1, b1, W2 = f(0, 0.05, (cin, 1024)), f(0, 0.01, 1024), f(0, 0.05, (1024, 3))
    xd, W1d, b1d, W2d, dyd = x.double(), W1.double(), b1.double(), W2.double(), dy.double()
    h = xd @ W1d + b1d
    g = dyd @ W2d.t()                                           # d loss / d lrelu(h)
    slope = torch.where(h > 0, torch.ones_like(h), torch.full_like(h, 0.1))
    dh = g * slope
    ref = {"dx": dh @ W1d.t(), "dW1": xd.t() @ dh, "db1": dh.sum(0), "dW2": torch.where(h > 0, h, 0.1 * h).t() @ dyd,
           "db2": dyd.sum(0)}
    got = dict(zip(["dx", "dW1", "db1", "dW2", "db2"],
                   [t.cpu().double() for t in ops.mlp_bwd(x.to(DEV), dy.to(DEV), W1.to(DEV), b1.to(DEV), W2.to(DEV), 0.1)]))
    amb = (h.abs() < 1e-6).nonzero().tolist()
    assert len(amb) < 400, "ambiguous units must be few"
    import itertools

    def moves(i, k):
        """what dh[i, k] changes by if the unit takes one of the two slopes the reference did not: the other side of the kink,
        or 0 - the gradient of relu(x) - alpha relu(-x) at a pre-activation that is EXACTLY zero in fp32 (model.py:828-830)"""
        s0 = slope[i, k].item()
        return [0.0] + [g[i, k].item() * (s1 - s0) for s1 in (1.0, 0.1, 0.0) if s1 != s0]

    by_col = {}
    for i, k in amb:
        by_col.setdefault(k, []).append(i)
    moved = 0
    for k, rows in by_col.items():
        assert len(rows) <= 6
        res = got["db1"][k].item() - ref["db1"][k].item()
        # the choice per ambiguous unit of the column that explains the residual of db1[k] best (usually one unit, in or out)
        best = min(itertools.product(*[moves(i, k) for i in rows]), key=lambda c: abs(res - sum(c)))
        for i, delta in zip(rows, best):
            if delta != 0.0:
                moved += 1
                ref["db1"][k] += delta
                ref["dW1"][:, k] += delta * xd[i]
    # dx comes from a kernel of its own where the backward is two launches (32 input channels: mlp_bwd_dx_split_kernel and
    # mlp_bwd_w_split_kernel each recompute the hidden layer, in different summation orders): its decision per unit is read
    # off the unit's row of dx - the component of the row's residual along W1[:, k]
    moved_dx = 0
    for i, k in amb:
        w = W1d[:, k]
        c = ((got["dx"][i] - ref["dx"][i]) @ w).item() / (w @ w).item()
        delta = min(moves(i, k), key=lambda d: abs(c - d))
        if delta != 0.0:
            moved_dx += 1
            ref["dx"][i] += delta * w
    moved = (moved, moved_dx)
    print("n = %d, cin = %d: %d pre-activations within 1e-6 of zero, the kernel took the other slope at %d (dW1 / db1) and %d (dx) of them" % ((n, cin, len(amb)) + moved))
    for name in ("dx", "dW1", "db1", "dW2", "db2"):
        err = (got[name] - ref[name]).abs().max().item() / max(1.0, ref[name].abs().max().item())
        assert err < 5e-6, (name, err, "ambiguous units %d, moved %s" % (len(amb), moved))


def test_copy_rows_jobs_moves_every_width_and_alignment():
    """fgc_copy_rows_jobs (the pack / unpack launches of a facet-sharded step): several jobs in one launch - gathered by an
    index list or consecutive, widths that are multiples of four dwords (16-byte pieces) and widths that are not, aligned and
    misaligned buffers, a job of more than one workgroup with a ragged last piece, an empty job - bit for bit against torch."""
    from facet_graph_convolution_amd import _lib
    L = _lib.lib()
    rs = np.random.RandomState(0)
    jobs, checks = [], []
    keep = []
    for rows, width, gather, mis_src, mis_dst in [(700, 64, True, 0, 0), (33, 12, True, 0, 0), (5, 3, True, 0, 0), (64, 32, False, 0, 0),
                                                  (300, 16, True, 1, 0), (300, 16, True, 0, 1), (0, 32, True, 0, 0), (129, 128, False, 0, 0),
                                                  (2049, 4, True, 0, 0)]:
        n_src = max(rows * 2, 8)
        src_all = torch.from_numpy(rs.normal(size=n_src * width + 4).astype(np.float32)).to(DEV)
        dst_all = torch.full((max(rows, 1) * width + 4,), -7.0, device=DEV)
        src = src_all[mis_src:mis_src + n_src * width].view(n_src, width)
        dst = dst_all[mis_dst:mis_dst + max(rows, 1) * width].view(max(rows, 1), width)
        idx = torch.from_numpy(rs.randint(0, n_src, size=max(rows, 1)).astype(np.int32)).to(DEV) if gather else None
        keep += [src_all, dst_all, idx]
        jobs.append((src, idx, dst, rows, width))
        checks.append((src, idx, dst, rows, dst_all, mis_dst, width))
    arr = (_lib.RowJob * len(jobs))()
    for k, (src, idx, dst, rows, width) in enumerate(jobs):
        arr[k].src, arr[k].idx, arr[k].dst = src.data_ptr(), (idx.data_ptr() if idx is not None else None), dst.data_ptr()
        arr[k].rows, arr[k].width = rows, width
    _lib.check(L.fgc_copy_rows_jobs(arr, len(jobs), _lib.stream_ptr()))
    torch.cuda.synchronize()
    for src, idx, dst, rows, dst_all, off, width in checks:
        want = src[idx.long()[:rows]] if idx is not None else src[:rows]
        assert torch.equal(dst[:rows], want), (rows, width)
        # nothing outside the job's rows was written
        assert (dst_all[:off] == -7.0).all() and (dst_all[off + rows * width:] == -7.0).all(), (rows, width)


def test_elementwise_ops():
    from facet_graph_convolution_amd import ops
    from oracle import model_ref as R
    rs = np.random.RandomState(0)
    x = _t(rs.normal(size=(64, 48)))
    x[3, 5] = 0.0
    xd = x.to(DEV)
    y = ops.lrelu_fwd(xd, 0.1)
    assert torch.equal(y.cpu(), R.lrelu(x, 0.1))
    dy = _t(rs.normal(size=(64, 48)))
    xr = x.clone().requires_grad_(True)
    (R.lrelu(xr, 0.1) * dy).sum().backward()
    assert torch.equal(ops.lrelu_bwd(y, dy.to(DEV), 0.1).cpu(), xr.grad)
    # pooling with ties (fake rows produce identical values)
    x[8:12] = x[8]
    x[16:18] = x[16]
    xd = x.to(DEV)
    p = ops.pool4_fwd(xd)
    assert torch.equal(p.cpu(), R.custom_binary_tree_pooling(x[None], 2)[0])
    xr = x.clone().requires_grad_(True)
    dp = _t(rs.normal(size=(16, 48)))
    (R.custom_binary_tree_pooling(xr[None], 2)[0] * dp).sum().backward()
    np.testing.assert_allclose(ops.pool4_bwd(xd, p, dp.to(DEV)).cpu().numpy(), xr.grad.numpy(), atol=1e-7)
    u = ops.upsample4_fwd(p)
    assert torch.equal(u.cpu(), R.custom_upsampling(p.cpu()[None], 2)[0])
    du = _t(rs.normal(size=(64, 48)))
    np.testing.assert_allclose(ops.upsample4_bwd(du.to(DEV)).cpu().numpy(),
                               du.reshape(16, 4, 48).sum(1).numpy(), atol=1e-6)


@pytest.mark.parametrize("n,cin,cout", [(1000, 32, 1024), (333, 1024, 3), (70, 48, 5), (5000, 64, 96), (1, 6, 3)])
def test_custom_lin_forward_backward(n, cin, cout):
    """fgc_lin_fwd / fgc_lin_bwd (custom_lin, model.py:763-769, on its own) against float64: y within 3e-6 of the largest
    output, gradients within 5e-6 of each tensor's largest entry (exact fp32 products, fp32 sums; 5 000 rows = three
    partial slabs summed in a fixed order; ragged tiles in every dimension)."""
    from facet_graph_convolution_amd import ops
    rs = np.random.RandomState(n + cin)
    x, W, b = _t(rs.normal(size=(n, cin)) * 0.5), _t(rs.normal(size=(cin, cout)) * 0.05), _t(rs.normal(size=cout) * 0.01)
    dy = _t(rs.normal(size=(n, cout)))
    X, Wd, Bd = x.double().requires_grad_(True), W.double().requires_grad_(True), b.double().requires_grad_(True)
    Y = X @ Wd + Bd
    Y.backward(dy.double())
    y = ops.lin_fwd(x.to(DEV), W.to(DEV), b.to(DEV))
    dx, dW, db = ops.lin_bwd(x.to(DEV), dy.to(DEV), W.to(DEV))
    dx2, dW2, db2 = ops.lin_bwd(x.to(DEV), dy.to(DEV), W.to(DEV), need_dx=False)
    torch.cuda.synchronize()
    assert dx2 is None and torch.equal(dW, dW2) and torch.equal(db, db2)       # fixed summation order: bitwise repeatable
    assert (y.cpu().double() - Y.detach()).abs().max().item() < 3e-6 * max(1.0, Y.abs().max().item())
    for got, ref in ((dx, X.grad), (dW, Wd.grad), (db, Bd.grad)):
        assert (got.cpu().double() - ref).abs().max().item() < 5e-6 * max(1.0, ref.abs().max().item()), tuple(ref.shape)


@pytest.mark.parametrize("steps", [0, 1, 2, 3])
def test_pooling_and_upsampling_for_any_step_count(steps):
    """custom_binary_tree_pooling / custom_upsampling with steps other than the network's 2 (model.py:779-788,817-825): one
    2^steps : 1 launch each way, against the oracle; ties in a group share the pooling gradient evenly."""
    from facet_graph_convolution_amd import ops
    from oracle import model_ref as R
    rs = np.random.RandomState(steps)
    g = 2 ** steps
    x = _t(rs.normal(size=(16 * g, 24)))
    if g > 1:
        x[1] = x[0]                        # a tie in every column of the first group
    xr = x.clone().requires_grad_(True)
    ref = R.custom_binary_tree_pooling(xr[None], steps)[0]
    dp = _t(rs.normal(size=(16, 24)))
    (ref * dp).sum().backward()
    xd = x.to(DEV)
    p = ops.pool_fwd(xd, g)
    assert torch.equal(p.cpu(), ref.detach())
    np.testing.assert_allclose(ops.pool_bwd(xd, p, dp.to(DEV), g).cpu().numpy(), xr.grad.numpy(), atol=1e-7)
    u = ops.upsample_fwd(p, g)
    assert torch.equal(u.cpu(), R.custom_upsampling(p.cpu()[None], steps)[0])
    du = _t(rs.normal(size=(16 * g, 24)))
    np.testing.assert_allclose(ops.upsample_bwd(du.to(DEV), g).cpu().numpy(), du.reshape(16, g, 24).sum(1).numpy(), atol=1e-6)


@pytest.mark.parametrize("n", [100, 5000])
def test_normalize_and_loss(n):
    from facet_graph_convolution_amd import ops
    from oracle import model_ref as R
    rs = np.random.RandomState(n)
    x = _t(rs.normal(size=(n, 3)) * 0.3)
    x[7] = 0.0  # zero row -> inv = 0 branch
    gt = _t(rs.normal(size=(n, 3)))
    gt = gt / gt.norm(dim=1, keepdim=True)
    gt[5] = 0.0  # fake row
    idx = torch.tensor(rs.randint(n, size=4000).astype(np.int32))
    xr = x.double().requires_grad_(True)
    nref = R.normalizeTensor(xr[None])
    loss_ref = R.faceNormalsLoss(nref[:, idx.long()], gt.double()[None][:, idx.long()])
    loss_ref.backward()
    xd, gtd, idxd = x.to(DEV), gt.to(DEV), idx.to(DEV)
    y, scratch = ops.normalize_fwd(xd)
    np.testing.assert_allclose(y.cpu().numpy(), nref[0].detach().numpy(), atol=2e-6)
    out = ops.angular_loss_fwd(y, gtd, idxd)
    assert abs(out[0].item() - loss_ref.item()) < 1e-4 * abs(loss_ref.item())
    dfn = ops.angular_loss_bwd(y, gtd, idxd, out, 1.0)
    dx = ops.normalize_bwd(xd, dfn, scratch)
    ref = xr.grad.numpy()
    np.testing.assert_allclose(dx.cpu().numpy(), ref, atol=2e-4 * np.abs(ref).max())


@pytest.mark.parametrize("n,ns,rotate", [(100, 4000, True), (5000, 4000, False), (70000, 9000, True)])
def test_fused_loss_step_matches_oracle_and_the_separate_entry_points(n, ns, rotate):
    """fgc_loss_step (two launches) against the float64 oracle - normalizeTensor, rotated ground truth, sampled angular
    loss, gradient back to the network output - and against the seven-launch chain it replaces; duplicate samples, a fake
    (all-zero ground truth) row, a zero output row, more samples than one sweep of the kernel; called twice: its scratch
    must be zero again after every call."""
    import ctypes as C
    from facet_graph_convolution_amd import _lib, ops
    from oracle import model_ref as R
    L = _lib.lib()
    rs = np.random.RandomState(n)
    x = _t(rs.normal(size=(n, 3)) * 0.3)
    x[7] = 0.0
    gt = _t(rs.normal(size=(n, 3)))
    gt = gt / gt.norm(dim=1, keepdim=True)
    gt[5] = 0.0
    idx = rs.randint(n, size=ns).astype(np.int32)
    idx[:3] = 5          # the fake row, sampled three times
    idx[3:6] = 11        # duplicates of a real row
    idx = torch.tensor(idx)
    Rm = _t(np.linalg.qr(rs.normal(size=(3, 3)))[0]) if rotate else torch.eye(3)
    xr = x.double().requires_grad_(True)
    nref = R.normalizeTensor(xr[None])
    gtr = (gt.double() @ Rm.double().t())
    loss_ref = R.faceNormalsLoss(nref[:, idx.long()], gtr[None][:, idx.long()])
    loss_ref.backward()
    xd, gtd, idxd, Rd = x.to(DEV), gt.to(DEV), idx.to(DEV), Rm.reshape(9).contiguous().to(DEV)
    # |y| partials as fgc_mlp_fwd leaves them: any split of the sum (here: one partial per 64 rows)
    part = torch.stack([c.abs().sum() for c in xd.split(64)]).contiguous()
    gacc = torch.zeros(n, 3, device=DEV)
    nconv, dy = torch.empty(n, 3, device=DEV), torch.empty(n, 3, device=DEV)
    loss, scratch = torch.zeros(2, device=DEV), torch.zeros(L.fgc_loss_step_scratch_floats(ns), device=DEV)
    p = _lib.ptr
    for call in range(2):
        _lib.check(L.fgc_loss_step(p(xd), n, p(part), part.numel(), p(gtd), p(Rd) if rotate else None, p(idxd), ns, p(gacc),
                                   p(nconv), p(dy), p(loss), p(scratch), _lib.stream_ptr()), "fgc_loss_step")
        torch.cuda.synchronize()
        assert not bool(gacc.any()), "the scratch must be zero again after the call"
        np.testing.assert_allclose(nconv.cpu().numpy(), nref[0].detach().numpy(), atol=2e-6)
        assert abs(loss[0].item() - loss_ref.item()) < 1e-4 * abs(loss_ref.item())
        assert loss[1].item() == float((gtr[idx.long()].abs().sum(1) > 1e-3).sum())
        ref = xr.grad.numpy()
        np.testing.assert_allclose(dy.cpu().numpy(), ref, atol=2e-4 * np.abs(ref).max())
    # the chain of separate launches on the same inputs: same arithmetic per row, sums in another order
    y2, sc2 = ops.normalize_fwd(xd)
    gt2 = ops.rotate_rows(gtd, Rm.numpy())
    out2 = ops.angular_loss_fwd(y2, gt2, idxd)
    dx2 = ops.normalize_bwd(xd, ops.angular_loss_bwd(y2, gt2, idxd, out2, 1.0), sc2)
    assert torch.equal(nconv, y2) or (nconv - y2).abs().max().item() < 1e-6
    assert abs(loss[0].item() - out2[0].item()) < 1e-5 * abs(out2[0].item()) and loss[1].item() == out2[1].item()
    assert (dy - dx2).abs().max().item() < 1e-5 * dx2.abs().max().item()


def test_rotate_adam_epilogue_gather():
    from facet_graph_convolution_amd import ops
    from oracle import model_ref as R
    rs = np.random.RandomState(3)
    x = _t(rs.normal(size=(50, 6)))
    Rm = _t(np.linalg.qr(rs.normal(size=(3, 3)))[0])
    xr, _ = R.rotate_inputs(x[None], None, Rm)
    np.testing.assert_allclose(ops.rotate_rows(x.to(DEV), Rm.numpy()).cpu().numpy(), xr[0].numpy(), atol=1e-6)
    # Adam, three steps
    p = _t(rs.normal(size=1000))
    m, v = torch.zeros(1000), torch.zeros(1000)
    pd, md, vd = p.to(DEV), m.to(DEV), v.to(DEV)
    pr, mr, vr = [p.clone().double()], [m.clone().double()], [v.clone().double()]
    for t in range(1, 4):
        g = _t(rs.normal(size=1000))
        ops.adam_step(pd, g.to(DEV), md, vd, t)
        R.adam_step_tf1(pr, [g.double()], mr, vr, t)
    np.testing.assert_allclose(pd.cpu().numpy(), pr[0].numpy(), atol=1e-6)
    # inference epilogue
    nc = _t(rs.normal(size=(40, 3)))
    perm = rs.permutation(40).astype(np.int32)
    ref = R.infer_epilogue(nc[None], perm, 30)
    got = ops.infer_epilogue(nc.to(DEV), torch.tensor(perm).to(DEV), 30)
    np.testing.assert_allclose(got.cpu().numpy(), ref.numpy(), atol=1e-6)
    # halo pack/unpack
    src = _t(rs.normal(size=(40, 8))).to(DEV)
    idx = torch.tensor(rs.permutation(40)[:10].astype(np.int32)).to(DEV)
    gathered = ops.gather_rows(src, idx)
    assert torch.equal(gathered, src[idx.long()])
    dst = torch.zeros(40, 8, device=DEV)
    ops.scatter_add_rows(gathered, idx, dst)
    assert torch.equal(dst[idx.long()], gathered)


@pytest.mark.parametrize("tag", ["ico3", "torus_open"])
def test_vertex_update_matches_reference(golden_dir, tag):
    """update_position2 (train.py:1467-1557), 1 and 60 iterations, closed mesh and mesh with boundary edges.
    fp32 tolerance: 2e-6 absolute on coordinates of a unit-size mesh (the reference's own fp32 run is 5e-7 away
    from its float64 run)."""
    import os
    from facet_graph_convolution_amd import train as T
    z = np.load(os.path.join(golden_dir, "vertex_%s.npz" % tag))
    z64 = np.load(os.path.join(golden_dir, "vertex_%s_f64.npz" % tag))
    x = torch.tensor(z["verts"], device=DEV)[None]
    fn = torch.tensor(z["normals"], device=DEV)[None]
    em = torch.tensor(z["edge_map"], device=DEV)[None]
    vem = torch.tensor(z["v_e_map"], device=DEV)[None]
    for it in (0, 1, 2, 60):
        out = T.update_position2(x, fn, em, vem, iter_num=it, max_edges=20)
        assert out.shape == x.shape
        if it == 0:
            assert torch.equal(out, x)
        elif it in (1, 60):
            np.testing.assert_allclose(out[0].cpu().numpy(), z["x_%d" % it], rtol=0, atol=2e-6)
    err = np.abs(out[0].cpu().numpy().astype(np.float64) - z64["x_60"]).max()
    print("%s: |gpu - f64| after 60 iterations %.2e" % (tag, err))
    assert err < 2e-6


def test_vertex_update_large_mesh_against_oracle():
    """50 000-vertex torus: the kernel against the oracle on the same inputs, and the full inference driver."""
    from facet_graph_convolution_amd import ops, utils
    from facet_graph_convolution_amd.meshgen import torus, add_noise
    from oracle import model_ref as R
    V, F = torus(250, 100)
    Vn = add_noise(V, F)
    nrm = utils.computeFacesNormals(V, F).astype(np.float32)
    em, vem = utils.getEdgeMap(F, maxEdges=20)
    ref = R.update_position2(torch.tensor(Vn.astype(np.float32)), nrm, em, vem, 20).numpy()
    got = ops.vertex_update(torch.tensor(Vn.astype(np.float32), device=DEV), torch.tensor(nrm, device=DEV),
                            torch.tensor(em, device=DEV), torch.tensor(vem, device=DEV), 20).cpu().numpy()
    np.testing.assert_allclose(got, ref, rtol=0, atol=2e-6)
    # the update moves the noisy vertices towards the clean surface
    assert np.abs(got - V).mean() < 0.8 * np.abs(Vn - V).mean()


def test_multiscale_vertex_update_matches_reference(golden_dir):
    """avg_ignore_zeros pooling (model.py:792-814), updateFacesCenter and update_position_MS (train.py:1668-1798) on
    the coarsened icosphere (fake nodes included), (2,1,1) and (80,20,20) iterations.  fp32 tolerance 3e-6 on
    coordinates of a unit-size mesh (reference fp32 vs its float64 run: 1.2e-7)."""
    import os
    from facet_graph_convolution_amd import model as M, train as T
    z = np.load(os.path.join(golden_dir, "msvertex_ico3.npz"))
    z64 = np.load(os.path.join(golden_dir, "msvertex_ico3_f64.npz"))
    dev = DEV
    n0 = torch.tensor(z["n0"], device=dev)[None]
    p1 = M.custom_binary_tree_pooling(n0, steps=2, pooltype="avg_ignore_zeros")
    n1 = M.normalizeTensor(p1)
    n2 = M.normalizeTensor(M.custom_binary_tree_pooling(n1, steps=2, pooltype="avg_ignore_zeros"))
    np.testing.assert_allclose(n1[0].cpu().numpy(), z["n1"], atol=2e-6)
    np.testing.assert_allclose(n2[0].cpu().numpy(), z["n2"], atol=2e-6)
    x = torch.tensor(z["verts_norm"], device=dev)[None]
    faces = torch.tensor(z["faces_perm"], device=dev)[None]
    vf = torch.tensor(z["v_faces"], device=dev)[None]
    c = T.updateFacesCenter(x, faces, 2)
    for k in range(3):
        np.testing.assert_allclose(c[k][0].cpu().numpy(), z["fpos%d" % k], atol=1e-7)
    nl = [torch.tensor(z["n%d" % k], device=dev)[None] for k in range(3)]
    for its in ((2, 1, 1), (80, 20, 20)):
        key = "_".join(map(str, its))
        xo, dxl = T.update_position_MS(x, nl, faces, vf, 2, iter_num_list=list(its))
        assert xo.shape == x.shape and len(dxl) == 3
        np.testing.assert_allclose(xo[0].cpu().numpy(), z["x_" + key], atol=3e-6)
        for k in range(3):
            np.testing.assert_allclose(dxl[k].cpu().numpy(), z["dx%d_%s" % (k, key)], atol=3e-6)
    assert np.abs(xo[0].cpu().numpy().astype(np.float64) - z64["x_80_20_20"]).max() < 3e-6
