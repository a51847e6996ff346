"""BASELINE config 3: bf16 STORAGE / fp32 accumulate (opt-in build extension; the reference is fp32 only,
train.py:409-427).  Activations and their gradients live in HBM as bf16, the dense contractions run on
v_mfma_f32_16x16x32_bf16 with fp32 accumulators, master weights, Adam, logit tables and d-logits stay fp32.

Stated tolerances against the reference's fp32 run (fixtures made by executing the reference source), measured values
in brackets (icosphere / torus fixtures):
  unit normals   max abs  5e-3   [1.3e-3]     (bf16 keeps 8 significant bits: 2^-9 = 2e-3 relative per stored value)
  loss           relative 1e-2   [1e-5]
  gradients      6e-2 of each tensor's largest entry  [conv parameters 1.2e-2; first MLP layer 3.8e-2: a pre-activation
                 whose sign changes under bf16 rounding switches the leaky-ReLU slope between 1 and 0.1]"""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

TOL_NORMALS, TOL_LOSS, TOL_GRAD = 5e-3, 1e-2, 6e-2


def _bind(golden_dir, tag, seed, dtype):
    from facet_graph_convolution_amd.net import FacetDenoiser
    prep = np.load(os.path.join(golden_dir, "prep_%s.npz" % tag))
    net = FacetDenoiser("cuda:0", seed=seed, dtype=dtype)
    net.bind_mesh(prep["x"], [prep["adj0"], prep["adj1"], prep["adj2"]], gt=prep["gt"])
    return net, prep


@pytest.mark.parametrize("name,tag,seed", [("net_ico3", "ico3", 0), ("net_torus640", "torus640", 1)])
def test_bf16_train_step_within_stated_tolerance_of_the_reference(golden_dir, name, tag, seed):
    z = np.load(os.path.join(golden_dir, name + ".npz"))
    net, _ = _bind(golden_dir, tag, seed, "bf16")
    assert net.buffers["d1"].dtype == torch.bfloat16 and net.buffers["g_h1"].dtype == torch.bfloat16
    assert net.params.theta.dtype == torch.float32 and net.buffers["y0"].dtype == torch.float32
    net.set_rotation(z["R"])
    net.set_samples(z["sample_ind"])
    loss = net.forward_backward(rotate=True)
    torch.cuda.synchronize()
    err_n = np.abs(net.buffers["nconv"].cpu().numpy() - z["n_conv"][0]).max()
    assert err_n < TOL_NORMALS, err_n
    assert abs(loss[0].item() - float(z["loss"])) < TOL_LOSS * float(z["loss"])
    worst = 0.0
    for i, g in enumerate(net.params.grads):
        ref = z["g%02d" % i]
        err = np.abs(g.cpu().numpy() - ref).max() / max(np.abs(ref).max(), 1e-3)
        worst = max(worst, err)
        assert err < TOL_GRAD, "grad %d (%s): rel err %.3e" % (i, net.params.spec[i], err)
    print("bf16 vs reference fp32: normals %.2e, worst rel grad err %.2e" % (err_n, worst))


@pytest.mark.parametrize("tag,seed", [("ico3", 0), ("torus640", 1)])
def test_bf16_fused_ds_prologue_matches_the_separate_launch(golden_dir, tag, seed, fgc_option):
    """The bf16 d-logits kernel computes s = dy * lrelu'(y) / deg (and folds the pooling gradient in) in its prologue;
    the library option NO_FUSED_DS_BF16 = 1 (fgc_set_option) runs ds_db_kernel instead.  Same operations on the same inputs: every tensor that does not
    pass through the bias-gradient partial sums is identical bit for bit, the bias gradients agree to fp32 rounding."""
    z = np.load(os.path.join(golden_dir, ("net_%s" % tag) + ".npz"))
    grads = {}
    for mode in ("0", "1"):
        fgc_option("NO_FUSED_DS_BF16", int(mode))
        net, _ = _bind(golden_dir, tag, seed, "bf16")
        net.set_rotation(z["R"])
        net.set_samples(z["sample_ind"])
        net.forward_backward(rotate=True)
        torch.cuda.synchronize()
        grads[mode] = [g.clone() for g in net.params.grads]
    nbias = 0
    for i, (a, b) in enumerate(zip(grads["0"], grads["1"])):
        if net.params.spec[i][0] == "bias":
            nbias += 1
            assert (a - b).abs().max().item() <= 2e-6 * max(b.abs().max().item(), 1e-6), (i, net.params.spec[i])
        else:
            assert torch.equal(a, b), (i, net.params.spec[i])
    assert nbias >= 8


def test_bf16_irregular_mesh_matches_the_fp32_network():
    """Facet degrees up to 23: the 24-slot conv kernels and the LONG d-logits form in bf16."""
    from facet_graph_convolution_amd.net import FacetDenoiser
    from facet_graph_convolution_amd.dataClasses import TrainingSet
    from facet_graph_convolution_amd.meshgen import torus, flip_edges, add_noise
    V, F = torus(24, 20)
    F = flip_edges(F, 400, seed=1)
    ds = TrainingSet()
    ds.addMeshWithGT(add_noise(V, F), F, V, seed=0)
    x, adjs, gt = ds.in_list[0], ds.adj_list[0], ds.gt_list[0]
    assert max(int((a[0] > 0).sum(1).max()) for a in adjs) > 16
    samp = np.random.RandomState(2).randint(x.shape[1], size=4000)
    out = {}
    for dt in ("f32", "bf16"):
        net = FacetDenoiser("cuda:0", seed=0, dtype=dt).bind_mesh(x, adjs, gt=gt)
        net.set_samples(samp)
        net.set_rotation(np.eye(3))
        net.forward_backward(rotate=True)
        torch.cuda.synchronize()
        out[dt] = net
    a, b = out["f32"], out["bf16"]
    assert (a.buffers["nconv"] - b.buffers["nconv"]).abs().max().item() < TOL_NORMALS
    assert abs(a.buffers["loss"][0].item() - b.buffers["loss"][0].item()) < TOL_LOSS * a.buffers["loss"][0].item()
    # 960 facets: fewer rows to average the sign flips of the leaky-ReLU derivative over, so twice the gradient tolerance
    errs = [(ga - gb).abs().max().item() / max(ga.abs().max().item(), 1e-3) for ga, gb in zip(a.params.grads, b.params.grads)]
    print("worst rel grad err, bf16 vs fp32 network: %.3e (grad %d)" % (max(errs), int(np.argmax(errs))))
    assert max(errs) < 2 * TOL_GRAD, errs


def test_bf16_mlp_kernels_against_torch(golden_dir):
    """fgc_mlp_fwd_bf16 / fgc_mlp_bwd_bf16 through the C ABI against a float64 torch MLP fed the SAME bf16-rounded x
    and weights: what is left is the fp32 accumulation order and the bf16 rounding of dh (dx, dW1, db1) and of the hidden
    activation (dW2: the K = nodes product hact^T dy runs on the bf16 matrix pipe, dy as a hi + lo pair; 1.6e-3 at 1000
    rows, the rounding errors are unbiased and average out over the rows)."""
    import ctypes as C
    from facet_graph_convolution_amd import _lib
    L = _lib.lib()
    dev = "cuda:0"
    g = torch.Generator().manual_seed(0)
    for n, cin in ((1000, 32), (333, 64)):
        x = (torch.randn(n, cin, generator=g) * 0.5).to(torch.bfloat16)
        W1 = (torch.randn(cin, 1024, generator=g) * 0.05).to(torch.bfloat16).float()
        b1 = torch.randn(1024, generator=g) * 0.01
        W2 = torch.randn(1024, 3, generator=g) * 0.05
        b2 = torch.randn(3, generator=g) * 0.01
        dy = torch.randn(n, 3, generator=g)
        xd, W1d, b1d, W2d, b2d, dyd = (t.to(dev).contiguous() for t in (x, W1, b1, W2, b2, dy))
        y = torch.empty(n, 3, device=dev)
        ws = torch.empty(L.fgc_mlp_bwd_bf16_workspace_bytes(n, cin, 1024, 3) + 256, dtype=torch.uint8, device=dev)
        st = _lib.stream_ptr()
        p = _lib.ptr
        _lib.check(L.fgc_mlp_fwd_bf16(p(xd), n, cin, 1024, 3, p(W1d), p(b1d), p(W2d), p(b2d), 0.1, p(y), None, 0, p(ws),
                                      ws.numel(), st))
        dx = torch.empty(n, cin, dtype=torch.bfloat16, device=dev)
        dW1, db1, dW2, db2 = (torch.empty_like(t) for t in (W1d, b1d, W2d, b2d))
        _lib.check(L.fgc_mlp_bwd_bf16(p(xd), p(dyd), n, cin, 1024, 3, p(W1d), p(b1d), p(W2d), 0.1, p(dx), p(dW1), p(db1),
                                      p(dW2), p(db2), 0, p(ws), ws.numel(), st))
        # the same two calls on operands packed ahead of time by fgc_conv_pack's launch (FGC_MLP_PACKED): bit-identical
        wsf = torch.empty(L.fgc_mlp_bf16_workspace_bytes(cin, 1024, 3) + 256, dtype=torch.uint8, device=dev)
        wsb = torch.empty_like(ws)
        ex = _lib.PackExtra(mlp_bf16=1, mlp_W1=p(W1d), mlp_W2=p(W2d), mlp_n=n, mlp_cin=cin, mlp_hidden=1024, mlp_cout=3,
                            mlp_fwd_ws=p(wsf), mlp_bwd_ws=p(wsb))
        _lib.check(L.fgc_conv_pack(None, None, None, None, 0, C.byref(ex), st))
        y2, dx2 = torch.empty_like(y), torch.empty_like(dx)
        g2 = [torch.empty_like(t) for t in (W1d, b1d, W2d, b2d)]
        _lib.check(L.fgc_mlp_fwd_bf16(p(xd), n, cin, 1024, 3, p(W1d), p(b1d), p(W2d), p(b2d), 0.1, p(y2), None,
                                      _lib.MLP_PACKED, p(wsf), wsf.numel(), st))
        _lib.check(L.fgc_mlp_bwd_bf16(p(xd), p(dyd), n, cin, 1024, 3, p(W1d), p(b1d), p(W2d), 0.1, p(dx2), p(g2[0]), p(g2[1]),
                                      p(g2[2]), p(g2[3]), _lib.MLP_PACKED, p(wsb), wsb.numel(), st))
        torch.cuda.synchronize()
        assert torch.equal(y, y2) and torch.equal(dx, dx2)
        for a, b in zip((dW1, db1, dW2, db2), g2):
            assert torch.equal(a, b)
        X = x.double().requires_grad_(True)
        P = [t.double().requires_grad_(True) for t in (W1, b1, W2, b2)]
        h = X @ P[0] + P[1]
        h = torch.relu(h) - 0.1 * torch.relu(-h)
        Y = h @ P[2] + P[3]
        Y.backward(dy.double())
        assert (y.cpu().double() - Y.detach()).abs().max().item() < 1e-5
        for got, ref, tol in ((dx.float(), X.grad, 1.5e-2), (dW1, P[0].grad, 1.5e-2), (db1, P[1].grad, 1.5e-2),
                              (dW2, P[2].grad, 4e-3), (db2, P[3].grad, 1e-5)):
            err = (got.cpu().double() - ref).abs().max().item() / ref.abs().max().item()
            assert err < tol, (n, cin, tuple(ref.shape), err)


def test_bf16_training_reduces_loss_and_graph_replay_matches(golden_dir):
    net, prep = _bind(golden_dir, "ico3", 0, "bf16")
    rs = np.random.RandomState(0)
    losses = []
    for it in range(30):
        loss = net.train_step(sample_ind=rs.randint(prep["x"].shape[1], size=4000), R=np.eye(3))
        losses.append(loss[0].item())
    assert np.isfinite(losses).all() and losses[-1] < 0.6 * losses[0], losses
    # bitwise reproducible, eager == hipGraph replay
    net.forward_backward(rotate=True)
    g1, n1 = net.params.grad.clone(), net.buffers["nconv"].clone()
    net.forward_backward(rotate=True, capture=True)
    net.forward_backward(rotate=True, capture=True)
    torch.cuda.synchronize()
    assert torch.equal(g1, net.params.grad) and torch.equal(n1, net.buffers["nconv"])


def test_bf16_multiscale_heads_forward(golden_dir):
    z = np.load(os.path.join(golden_dir, "net_ico3_ms.npz"))
    from facet_graph_convolution_amd.net import FacetDenoiser
    prep = np.load(os.path.join(golden_dir, "prep_ico3.npz"))
    net = FacetDenoiser("cuda:0", multi_scale=True, seed=0, dtype="bf16")
    net.bind_mesh(prep["x"], [prep["adj0"], prep["adj1"], prep["adj2"]], gt=prep["gt"])
    net.set_rotation(z["R"])
    net.forward(rotate=True)
    torch.cuda.synchronize()
    for k in ("y0", "y1", "y2"):
        ref = z[k][0]
        assert np.abs(net.buffers[k].cpu().numpy() - ref).max() < 8e-3 * max(np.abs(ref).max(), 1e-6), k


@pytest.mark.parametrize("tag,world,seed", [("ico3", 2, 0), ("torus640", 3, 1)])
def test_bf16_facet_sharded_step_matches_the_unsharded_bf16_network(golden_dir, tag, world, seed):
    """Facet sharding of the bf16-storage network: halo rows and the rows of s travel as bf16 (half the bytes of the fp32
    exchange), the per-edge d-logits as fp32.  `world` shards in one process against the unsharded bf16 network: the
    forward pass does the same arithmetic per row (equal up to the order of the global-mean all-reduce); gradients differ
    where partial sums are rounded to bf16 in a different grouping (measured 3e-3 of a tensor's largest entry)."""
    from facet_graph_convolution_amd.net import FacetDenoiser
    from facet_graph_convolution_amd.shard import make_sim_shards, sim_forward_backward
    z = np.load(os.path.join(golden_dir, "net_%s.npz" % tag))
    ref, prep = _bind(golden_dir, tag, seed, "bf16")
    ref.set_rotation(z["R"])
    ref.set_samples(z["sample_ind"])
    ref.forward_backward(rotate=True)
    nets = make_sim_shards(prep["x"], [prep["adj0"], prep["adj1"], prep["adj2"]], prep["gt"], world, "cuda:0", seed,
                           dtype="bf16")
    for n in nets:
        n.set_rotation(z["R"])
        n.set_samples(z["sample_ind"])
    sim_forward_backward(nets, rotate=True)
    torch.cuda.synchronize()
    full = ref.buffers["nconv"].cpu().numpy()
    for n in nets:
        P = n._mesh["plan"].levels[0]
        np.testing.assert_allclose(n.buffers["nconv"].cpu().numpy(), full[P.lo:P.hi], rtol=0, atol=1e-5)
        assert abs(n.buffers["loss"][0].item() - ref.buffers["loss"][0].item()) < 1e-3
        for i, (g, gr) in enumerate(zip(n.params.grads, ref.params.grads)):
            a, b = g.cpu().numpy(), gr.cpu().numpy()
            assert np.abs(a - b).max() / max(np.abs(b).max(), 1e-3) < 1e-2, "grad %d" % i


def test_bf16_rejects_what_it_does_not_cover(golden_dir):
    from facet_graph_convolution_amd.net import FacetDenoiser
    prep = np.load(os.path.join(golden_dir, "prep_ico3.npz"))
    adjs = [prep["adj0"], prep["adj1"], prep["adj2"]]
    net = FacetDenoiser("cuda:0", multi_scale=True, dtype="bf16").bind_mesh(prep["x"], adjs, gt=prep["gt"])
    with pytest.raises(NotImplementedError):        # training the coarse heads: fp32 only
        net.train_step(sample_ind=np.arange(100), R=np.eye(3))
    with pytest.raises(ValueError):
        FacetDenoiser("cuda:0", dtype="fp8")


def _bf16_conv_fwd_bwd(g, x0, x1, shift, params, dy, act, pairs=False, options=None):
    """One conv layer with FGC_CONV_BF16 straight through the C ABI: bf16 x / y / dy / ds / r / dx, fp32 everything else.
    pairs: the layer (shift == 2) gets its pair graph, a bf16 hc table and a bf16 dt scratch (include/fgc.h).
    Returns (y bf16, dx0, dx1, [dW0, db, du, dc, dv])."""
    import ctypes as C
    from facet_graph_convolution_amd import _lib, ops
    from facet_graph_convolution_amd._lib import ConvBwdIO, AG_LD, DL_LD, FGC_M, ptr, stream_ptr, check
    L = _lib.lib()
    dev = x0.device
    d = ops.make_conv_desc(g, x0, x1, shift, params, True, act, 0.1)
    d.flags = _lib.CONV_BF16
    d.src_rows = x0.shape[0]
    if options:          # per-descriptor library options (fgc_conv_desc.options): this layer only
        over = _lib.option_overrides(**options)
        d.options, d.n_options = C.addressof(over), len(over)
    n, cout = g.n, d.cout
    bf = dict(dtype=torch.bfloat16, device=dev)
    f32 = dict(dtype=torch.float32, device=dev)
    pg = None
    if pairs:
        hc = torch.empty(x0.shape[0], FGC_M * cout, **bf)
        pg = ops.attach_pairs(d, g, hc)
        assert L.fgc_conv_uses_pairs(C.byref(d)) == 1, (pg.max_deg, pg.max_in_deg)
    ws = torch.empty(L.fgc_conv_workspace_bytes(C.byref(d)) + 256, dtype=torch.uint8, device=dev)
    ag = torch.empty(x0.shape[0], AG_LD, **f32)
    y = torch.empty(n, cout, **bf)
    check(L.fgc_conv_fwd(C.byref(d), ptr(ag), ptr(y), None, ptr(ws), ws.numel(), stream_ptr()), "fgc_conv_fwd bf16")
    trow, tcol, tedge = g.transposed()
    io = ConvBwdIO()
    io.trowptr, io.tcol, io.tedge = trow.data_ptr(), tcol.data_ptr(), tedge.data_ptr()
    io.max_in_deg = g.max_in_deg
    ds = torch.empty(n, cout, **bf)
    dl = torch.empty(max(g.nnz, 1), DL_LD, **f32)
    dag = torch.empty(n, AG_LD, **f32)
    r = torch.empty(n, FGC_M * cout + 24, **bf)
    grads = [torch.empty_like(p) for p in params]
    dx0 = torch.empty_like(x0)
    dx1 = torch.empty_like(x1) if x1 is not None else None
    io.ag, io.y, io.dy = ag.data_ptr(), y.data_ptr(), dy.data_ptr()
    io.ds, io.dl, io.dag, io.r = ds.data_ptr(), dl.data_ptr(), dag.data_ptr(), r.data_ptr()
    io.dx0, io.dx1 = dx0.data_ptr(), (dx1.data_ptr() if dx1 is not None else None)
    io.dW0, io.db, io.du, io.dc, io.dv = [t.data_ptr() for t in grads]
    if pg is not None:
        io.tpair_rowptr, io.tpair_col, io.tpair_edge = pg.trow.data_ptr(), pg.tcol.data_ptr(), pg.tedge.data_ptr()
        dt = torch.empty(pg.n_pairs, cout, **bf)
        io.dt = dt.data_ptr()
    wsb = torch.empty(L.fgc_conv_bwd_workspace_bytes(C.byref(d)) + 256, dtype=torch.uint8, device=dev)
    check(L.fgc_conv_bwd(C.byref(d), C.byref(io), ptr(wsb), wsb.numel(), stream_ptr()), "fgc_conv_bwd bf16")
    torch.cuda.synchronize()
    return y, dx0, dx1, grads


@pytest.mark.parametrize("mode,cin,cout", [("plain", 32, 64), ("plain", 64, 128), ("concat", 128, 64), ("upsample", 128, 64),
                                           ("plain", 128, 128), ("plain", 64, 32), ("concat", 64, 32),
                                           ("upsample_pairs", 128, 64), ("upsample_pairs", 64, 32)])
def test_bf16_conv_kernels_against_float64_on_the_same_rounded_operands(mode, cin, cout):
    """Kernel error isolated from storage error: one conv layer, forward and backward, through the bf16 kernels against
    the float64 oracle fed the SAME bf16-rounded x and dy (and, for lrelu', the bf16 kernel's own stored y).  What is
    left is what the kernels themselves round: the output y and dx to bf16 (2^-9 relative per element), s and r to bf16
    inside the backward (each a sum's operand: the errors average out over the nodes) - a wrong term, a dropped edge or
    a misplaced column would show as O(1e-1).  Bounds, as fractions of each tensor's largest entry (measured on the five
    shapes in brackets): y 4e-3 [2.6-3.0e-3], dx 5e-3 [2.5-3.7e-3], dW0 3e-3 [2.0-2.4e-3], db 1e-5 [1e-7], du / dv 6e-3
    [2.4-4.2e-3: da | dg travel as bf16 behind the r row], dc 1e-2 [1.4-6.6e-3: a sum over all edges of signed d-logits that
    cancel per edge].  The end-to-end comparisons in this file keep their 10x looser bounds for STORAGE error."""
    from facet_graph_convolution_amd.graph import FacetGraph
    from oracle import model_ref as R
    dev = torch.device("cuda:0")
    rs = np.random.RandomState(11)
    n = 2048
    adj = np.zeros((n, 23), dtype=np.int32)
    for i in range(n):
        deg = rs.randint(3, 13)
        adj[i, 0] = i + 1
        win = 24 if mode == "upsample_pairs" else 60      # (pair form: at most 24 in-pairs per coarse row)
        adj[i, 1:1 + deg] = rs.randint(max(1, i - win), min(n, i + win) + 1, size=deg)
    g = FacetGraph(adj, dev)
    pairs = mode == "upsample_pairs"
    if pairs:
        mode = "upsample"       # the same layer, run on its coarse source rows: bf16 hc and dt are its extra roundings
    rows = n // 4 if mode == "upsample" else n
    widths = [cin // 2, cin // 2] if mode == "concat" else [cin]
    xs = [torch.tensor(rs.normal(size=(rows, w)).astype(np.float32)).to(torch.bfloat16) for w in widths]
    dy = torch.tensor(rs.normal(size=(n, cout)).astype(np.float32)).to(torch.bfloat16)
    p = R.conv_params(cin, cout, 21)
    p[0] = p[0].to(torch.bfloat16).float()       # the kernels pack W0 as a bf16 MFMA operand: exact for these values
    xd = [t.to(dev).contiguous() for t in xs]
    y, dx0, dx1, grads = _bf16_conv_fwd_bwd(g, xd[0], xd[1] if len(xd) > 1 else None, 2 if mode == "upsample" else 0,
                                            [t.to(dev) for t in p], dy.to(dev).contiguous(), act=1, pairs=pairs)
    # float64 oracle on the rounded operands
    X = [t.double().requires_grad_(True) for t in xs]
    P = [t.double().requires_grad_(True) for t in p]
    xin = torch.cat(X, 1)[None]
    if mode == "upsample":
        xin = R.custom_upsampling(xin, 2)
    pre = R.custom_conv2d(xin, torch.tensor(adj[None]), P)[0]
    yref = torch.where(pre > 0, pre, 0.1 * pre)
    ygpu = y.cpu().double()
    e_y = (ygpu - yref.detach()).abs().max().item() / yref.abs().max().item()
    # the backward kernels take lrelu'(y) from the stored (bf16) y: give the oracle the same slopes
    slope = torch.where(ygpu > 0, torch.ones_like(ygpu), torch.full_like(ygpu, 0.1))
    (pre * slope * dy.double()).sum().backward()
    errs = {"y": e_y}
    for name, got, ref in [("dx0", dx0, X[0].grad)] + ([("dx1", dx1, X[1].grad)] if dx1 is not None else []) + \
            list(zip(["dW0", "db", "du", "dc", "dv"], grads, [t.grad for t in P])):
        errs[name] = (got.cpu().double() - ref).abs().max().item() / max(ref.abs().max().item(), 1e-6)
    print("%s %d->%d: %s" % (mode, cin, cout, ", ".join("%s %.1e" % kv for kv in errs.items())))
    bound = {"y": 4e-3, "dx0": 5e-3, "dx1": 5e-3, "dW0": 3e-3, "db": 1e-5, "du": 6e-3, "dv": 6e-3, "dc": 1e-2}
    for k, v in errs.items():
        assert v < bound[k], (k, errs)



@pytest.mark.parametrize("n,cin,cout,mode", [(37, 32, 32, "plain"), (100, 64, 32, "concat"), (250, 128, 128, "plain"),
                                             (1001, 64, 64, "plain"), (404, 128, 64, "upsample"), (16, 32, 64, "plain")])
def test_bf16_matrix_pipe_aggregation_equals_the_vector_form_on_ragged_sizes(n, cin, cout, mode):
    """conv_bfm_kernel (round 6: rows gathered straight into LDS, the aggregation on v_mfma_f32_16x16x32_bf16 with q as two
    bf16 terms) against conv_w8_kernel<.., BF> (the vector form: option NO_BFM = 1, set for ONE descriptor) on the same
    operands, forward and backward: node counts that fill neither a 16- nor a 32-node tile, rows without edges, degrees 1 ..
    16, two sources, an upsampled source.  The two forms round the same aggregates to bf16, computed from q with 16
    (hi + lo) against 24 significand bits: an element may land on the neighbouring bf16 value (2^-8 relative), nothing else
    may differ - bounds 1.5e-2 of each tensor's largest entry for y / dx, 6e-3 for the parameter gradients (sums over
    nodes)."""
    from facet_graph_convolution_amd.graph import FacetGraph
    from oracle import model_ref as R
    dev = torch.device("cuda:0")
    rs = np.random.RandomState(n + cin)
    adj = np.zeros((n, 23), dtype=np.int32)
    for i in range(n):
        deg = rs.randint(0, 16) if i % 7 else 0          # every seventh row: the self slot only
        adj[i, 0] = i + 1
        adj[i, 1:1 + deg] = rs.randint(max(1, i - 40), min(n, i + 40) + 1, size=deg)
    g = FacetGraph(adj, dev)
    assert g.max_deg <= 16
    rows = n // 4 if mode == "upsample" else n
    widths = [cin // 2, cin // 2] if mode == "concat" else [cin]
    xs = [torch.tensor(rs.normal(size=(rows, w)).astype(np.float32)).to(torch.bfloat16).to(dev).contiguous() for w in widths]
    dy = torch.tensor(rs.normal(size=(n, cout)).astype(np.float32)).to(torch.bfloat16).to(dev).contiguous()
    p = [t.to(dev) for t in R.conv_params(cin, cout, 5)]
    out = []
    for options in (None, {"NO_BFM": 1}):
        # (no activation: an output within bf16 rounding of zero would take different leaky-ReLU slopes in the two forms
        #  and move every gradient by one row's worth - the kink of DESIGN.md section 4, not a property of the kernels)
        out.append(_bf16_conv_fwd_bwd(g, xs[0], xs[1] if len(xs) > 1 else None, 2 if mode == "upsample" else 0, p, dy, act=0,
                                      options=options))
    (ya, dxa0, dxa1, ga), (yb, dxb0, dxb1, gb) = out
    rel = lambda a, b: (a.float() - b.float()).abs().max().item() / max(b.float().abs().max().item(), 1e-6)
    errs = {"y": rel(ya, yb), "dx0": rel(dxa0, dxb0)}
    if dxa1 is not None:
        errs["dx1"] = rel(dxa1, dxb1)
    for name, a, b in zip(["dW0", "db", "du", "dc", "dv"], ga, gb):
        errs[name] = rel(a, b)
    print("n = %d, %s %d -> %d: %s" % (n, mode, cin, cout, ", ".join("%s %.1e" % kv for kv in errs.items())))
    for k, v in errs.items():
        assert v < (1.5e-2 if k in ("y", "dx0", "dx1") else 6e-3), (k, errs)
