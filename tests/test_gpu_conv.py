"""GPU parity: libfgc graph convolution vs the golden fixtures (reference source) and the oracle."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

CONV_CASES = ["c1_raw", "c1_coarsened", "rand_5_7", "rand_32_64", "rand_128_64", "rand_nomask"]


def _params(cin, cout, seed, device):
    from oracle import model_ref as R
    return [p.to(device) for p in R.conv_params(cin, cout, seed)]


@pytest.mark.parametrize("case", CONV_CASES)
def test_conv_forward_matches_golden(golden_dir, case):
    from facet_graph_convolution_amd import ops
    from facet_graph_convolution_amd.graph import FacetGraph
    z = np.load(os.path.join(golden_dir, "conv_%s.npz" % case))
    z64 = np.load(os.path.join(golden_dir, "conv_%s_f64.npz" % case))
    dev = torch.device("cuda:0")
    x = torch.tensor(z["x"][0], device=dev)
    g = FacetGraph(z["adj"], dev)
    params = _params(x.shape[1], int(z["cout"]), int(z["seed"]), dev)
    y, _, _ = ops.conv_fwd(g, x, None, 0, params, bias_mask=(case != "rand_nomask"))
    torch.cuda.synchronize()
    got = y.cpu().numpy()
    ref32, ref64 = z["y"][0], z64["y"][0]
    err_ref = np.abs(ref32 - ref64).max()
    err_gpu = np.abs(got - ref64).max()
    print("%s: |gpu-f64| %.3e  |ref32-f64| %.3e  |gpu-ref32| %.3e" % (case, err_gpu, err_ref, np.abs(got - ref32).max()))
    # fp32 tolerance: both fp32 evaluations must sit within 2e-6 of the float64 result
    assert err_gpu < 2e-6


@pytest.mark.parametrize("case", CONV_CASES)
def test_conv_backward_matches_golden(golden_dir, case):
    """All six gradients of custom_conv2d against autograd through the reference source."""
    from facet_graph_convolution_amd import ops
    from facet_graph_convolution_amd.graph import FacetGraph
    z = np.load(os.path.join(golden_dir, "conv_%s.npz" % case))
    z64 = np.load(os.path.join(golden_dir, "conv_%s_f64.npz" % case))
    dev = torch.device("cuda:0")
    x = torch.tensor(z["x"][0], device=dev)
    g = FacetGraph(z["adj"], dev)
    params = _params(x.shape[1], int(z["cout"]), int(z["seed"]), dev)
    mask = case != "rand_nomask"
    y, _, ag = ops.conv_fwd(g, x, None, 0, params, bias_mask=mask)
    dy = torch.tensor(z["dy"][0], device=dev)
    dx, _, grads = ops.conv_bwd(g, x, None, 0, params, ag, y, dy, bias_mask=mask)
    torch.cuda.synchronize()
    for key, got in zip(["dW0", "db", "du", "dc", "dv", "dx"], grads + [dx]):
        got = got.cpu().numpy()
        ref32 = z[key].reshape(got.shape)
        ref64 = z64[key].reshape(got.shape)
        scale = max(1.0, np.abs(ref64).max())
        e_gpu = np.abs(got - ref64).max() / scale
        e_ref = np.abs(ref32 - ref64).max() / scale
        print("%s %-3s |gpu-f64| %.2e |ref32-f64| %.2e (scale %.1f)" % (case, key, e_gpu, e_ref, scale))
        # gradients: relative to the tensor's max magnitude, within 5e-6 of the float64 gradient
        assert e_gpu < 5e-6, key


def test_conv_concat_upsample_and_pool_match_oracle():
    """The fused addressing modes (two-source concat, 4x upsampled input, fused activation + pooling) against
    the oracle's materialised tf.concat / custom_upsampling / lrelu / custom_binary_tree_pooling."""
    from facet_graph_convolution_amd import ops
    from facet_graph_convolution_amd.graph import FacetGraph
    from oracle import model_ref as R
    dev = torch.device("cuda:0")
    rs = np.random.RandomState(5)
    n = 96
    adj = np.zeros((n, 23), dtype=np.int32)
    for i in range(n):
        d = rs.randint(1, 14)
        adj[i, 0] = i + 1
        adj[i, 1:1 + d] = rs.randint(1, n + 1, size=d)
    g = FacetGraph(adj, dev)
    xa = torch.tensor(rs.normal(size=(n, 32)).astype(np.float32))
    xb = torch.tensor(rs.normal(size=(n, 32)).astype(np.float32))
    xc = torch.tensor(rs.normal(size=(n // 4, 64)).astype(np.float32))
    adj_t = torch.tensor(adj[None])
    # concat
    p = R.conv_params(64, 32, 3)
    ref = R.lrelu(R.custom_conv2d(torch.cat([xa, xb], 1)[None], adj_t, p))
    ref_pool = R.custom_binary_tree_pooling(ref, 2)
    y, yp, _ = ops.conv_fwd(g, xa.to(dev), xb.to(dev), 0, [t.to(dev) for t in p], act=1, alpha=0.1, want_pool=True)
    np.testing.assert_allclose(y.cpu().numpy(), ref[0].numpy(), atol=2e-6)
    np.testing.assert_allclose(yp.cpu().numpy(), ref_pool[0].numpy(), atol=2e-6)
    # upsampled single source
    p = R.conv_params(64, 32, 4)
    ref = R.custom_conv2d(R.custom_upsampling(xc[None], 2), adj_t, p)
    y, _, _ = ops.conv_fwd(g, xc.to(dev), None, 2, [t.to(dev) for t in p])
    np.testing.assert_allclose(y.cpu().numpy(), ref[0].numpy(), atol=2e-6)


def _oracle_conv_grads(x_list, shift, adj, params, dy, act):
    from oracle import model_ref as R
    xs = [t.clone().requires_grad_(True) for t in x_list]
    ps = [t.clone().requires_grad_(True) for t in params]
    xin = torch.cat(xs, 1)[None]
    if shift:
        xin = R.custom_upsampling(xin, 2)
    y = R.custom_conv2d(xin, torch.tensor(adj[None]), ps)
    if act:
        y = R.lrelu(y)
    (y[0] * dy).sum().backward()
    return [t.grad for t in xs], [t.grad for t in ps]


@pytest.mark.parametrize("mode", ["concat", "upsample"])
def test_conv_backward_fused_addressing_matches_oracle(mode):
    from facet_graph_convolution_amd import ops
    from facet_graph_convolution_amd.graph import FacetGraph
    from oracle import model_ref as R
    dev = torch.device("cuda:0")
    rs = np.random.RandomState(9)
    n = 128
    adj = np.zeros((n, 23), dtype=np.int32)
    for i in range(n):
        d = rs.randint(0, 12)
        adj[i, 0] = i + 1
        adj[i, 1:1 + d] = rs.randint(1, n + 1, size=d)
    g = FacetGraph(adj, dev)
    if mode == "concat":
        xs = [torch.tensor(rs.normal(size=(n, 64)).astype(np.float32)) for _ in range(2)]
        shift, cin, cout = 0, 128, 64
    else:
        xs = [torch.tensor(rs.normal(size=(n // 4, 128)).astype(np.float32))]
        shift, cin, cout = 2, 128, 64
    p = R.conv_params(cin, cout, 21)
    dy = torch.tensor(rs.normal(size=(n, cout)).astype(np.float32))
    gx_ref, gp_ref = _oracle_conv_grads(xs, shift, adj, p, dy, act=True)
    xd = [t.to(dev) for t in xs]
    pd = [t.to(dev) for t in p]
    y, _, ag = ops.conv_fwd(g, xd[0], xd[1] if len(xd) > 1 else None, shift, pd, act=1, alpha=0.1)
    dx0, dx1, grads = ops.conv_bwd(g, xd[0], xd[1] if len(xd) > 1 else None, shift, pd, ag, y, dy.to(dev), act=1,
                                   alpha=0.1)
    got_x = [dx0] + ([dx1] if dx1 is not None else [])
    for a, b in zip(got_x, gx_ref):
        np.testing.assert_allclose(a.cpu().numpy(), b.numpy(), atol=5e-6 * max(1.0, b.abs().max().item()))
    for a, b, name in zip(grads, gp_ref, ["dW0", "db", "du", "dc", "dv"]):
        np.testing.assert_allclose(a.cpu().numpy(), b.numpy(), atol=5e-6 * max(1.0, b.abs().max().item()),
                                   err_msg=name)


def test_logit_gradients_on_split_operands_match_the_fp32_form(fgc_option):
    """The dz GEMM of the half-tile d-logits kernel on the bf16 matrix pipe with three-term operand splits (option
    NO_K1_SPLIT = 0, the default for 32 outputs on regular graphs): every gradient against the oracle at the fp32 tolerance
    of the other conv tests, and against the fp32-MFMA form of the same kernel at the size of a summation-order difference."""
    from facet_graph_convolution_amd import _lib, ops
    from facet_graph_convolution_amd.graph import FacetGraph
    from oracle import model_ref as R
    dev = torch.device("cuda:0")
    rs = np.random.RandomState(31)
    n, cin, cout = 400, 64, 32
    adj = np.zeros((n, 23), dtype=np.int32)
    for i in range(n):
        d = rs.randint(3, 13)                      # degree <= 16: the half-tile kernel
        adj[i, 0] = i + 1
        adj[i, 1:1 + d] = rs.randint(1, n + 1, size=d)
    g = FacetGraph(adj, dev)
    x = torch.tensor(rs.normal(size=(n, cin)).astype(np.float32))
    p = R.conv_params(cin, cout, 5)
    dy = torch.tensor(rs.normal(size=(n, cout)).astype(np.float32))
    gx_ref, gp_ref = _oracle_conv_grads([x], 0, adj, p, dy, act=True)
    pd = [t.to(dev) for t in p]
    out = {}
    for no_split in (0, 1):
        fgc_option("NO_K1_SPLIT", no_split)
        y, _, ag = ops.conv_fwd(g, x.to(dev), None, 0, pd, act=1, alpha=0.1)
        dx0, _, grads = ops.conv_bwd(g, x.to(dev), None, 0, pd, ag, y, dy.to(dev), act=1, alpha=0.1)
        torch.cuda.synchronize()
        out[no_split] = [dx0.cpu().numpy()] + [t.cpu().numpy() for t in grads]
    assert _lib.get_option("NO_K1_SPLIT") == 1
    refs = [gx_ref[0].numpy()] + [t.numpy() for t in gp_ref]
    for a, b, r, name in zip(out[0], out[1], refs, ["dx", "dW0", "db", "du", "dc", "dv"]):
        scale = max(1.0, np.abs(r).max())
        np.testing.assert_allclose(a, r, atol=5e-6 * scale, err_msg=name)
        np.testing.assert_allclose(a, b, atol=2e-6 * scale, err_msg=name + " (split vs fp32 MFMA)")
    # the two forms are different kernels: the logit gradients are not bit-identical (or the option did nothing)
    assert any(not np.array_equal(a, b) for a, b in zip(out[0][3:], out[1][3:]))


def test_partial_calls_compose_to_the_whole_layer(golden_dir):
    """fgc_conv_desc.tile_list / proj_rows / FGC_CONV_PACKED and fgc_conv_bwd_io.data_tile_list / stages: a layer
    computed as two partial calls (what a facet-sharded run does around its halo exchange) is bit-identical to the
    single call, forward and backward."""
    import ctypes as C
    from facet_graph_convolution_amd import ops, _lib
    from facet_graph_convolution_amd.graph import FacetGraph
    from facet_graph_convolution_amd.ops import make_conv_desc, ptr, stream_ptr
    z = np.load(os.path.join(golden_dir, "conv_c1_coarsened.npz"))
    dev = torch.device("cuda:0")
    x = torch.tensor(z["x"][0], device=dev)
    g = FacetGraph(z["adj"], dev)
    n, cin, cout = g.n, x.shape[1], int(z["cout"])
    params = _params(cin, cout, int(z["seed"]), dev)
    L = _lib.lib()
    y_ref, _, ag_ref = ops.conv_fwd(g, x, None, 0, params)
    dy = torch.tensor(np.random.RandomState(3).normal(size=(n, cout)).astype(np.float32), device=dev)
    dx_ref, _, grads_ref = ops.conv_bwd(g, x, None, 0, params, ag_ref, y_ref, dy)

    ntiles = (n + 31) // 32
    rs = np.random.RandomState(4)
    first = np.sort(rs.choice(ntiles, ntiles // 3, replace=False)).astype(np.int32)
    second = np.setdiff1d(np.arange(ntiles, dtype=np.int32), first).astype(np.int32)
    lists = [torch.tensor(a, device=dev) for a in (first, second)]
    d = make_conv_desc(g, x, None, 0, params, True, 0, 0.1)
    ws = torch.empty(max(L.fgc_conv_workspace_bytes(C.byref(d)), L.fgc_conv_bwd_workspace_bytes(C.byref(d))) + 256,
                     dtype=torch.uint8, device=dev)
    ag = torch.full((n, 24), float("nan"), device=dev)
    y = torch.full((n, cout), float("nan"), device=dev)
    split_row = 700
    # call 1: every logit row + the first tile list; call 2: no logits, the other tiles, operands still packed
    for tl, pr, fl in [(lists[0], 0, 0), (lists[1], -1, 1)]:
        d.tile_list, d.n_tiles, d.proj_row0, d.proj_rows, d.flags = tl.data_ptr(), tl.numel(), 0, pr, fl
        _lib.check(L.fgc_conv_fwd(C.byref(d), ptr(ag), ptr(y), None, ptr(ws), ws.numel(), stream_ptr()), "partial fwd")
    assert torch.equal(y, y_ref) and torch.equal(ag, ag_ref)
    # logits in two row ranges
    ag2 = torch.full((n, 24), float("nan"), device=dev)
    empty = torch.zeros(1, dtype=torch.int32, device=dev)
    for r0, nr, fl in [(0, split_row, 0), (split_row, n - split_row, 1)]:
        d.tile_list, d.n_tiles, d.proj_row0, d.proj_rows, d.flags = empty.data_ptr(), 0, r0, nr, fl
        _lib.check(L.fgc_conv_fwd(C.byref(d), ptr(ag2), ptr(y), None, ptr(ws), ws.numel(), stream_ptr()), "logits only")
    assert torch.equal(ag2, ag_ref)
    d.tile_list, d.n_tiles, d.proj_row0, d.proj_rows, d.flags = None, 0, 0, 0, 0

    # backward: stages 1, 2, then the data kernel over two tile lists, then the weight gradients
    trow, tcol, tedge = g.transposed()
    f32 = dict(dtype=torch.float32, device=dev)
    io = _lib.ConvBwdIO()
    io.trowptr, io.tcol, io.tedge, io.max_in_deg = trow.data_ptr(), tcol.data_ptr(), tedge.data_ptr(), g.max_in_deg
    bufs = dict(ds=torch.empty(n, cout, **f32), dl=torch.empty(g.nnz, 12, **f32), dag=torch.empty(n, 24, **f32),
                r=torch.empty(n, 9 * cout + 24, **f32), dx=torch.full((n, cin), float("nan"), **f32))
    grads = [torch.empty_like(p) for p in params]
    io.ag, io.y, io.dy = ag_ref.data_ptr(), y_ref.data_ptr(), dy.data_ptr()
    io.ds, io.dl, io.dag, io.r = (bufs[k].data_ptr() for k in ("ds", "dl", "dag", "r"))
    io.dx0 = bufs["dx"].data_ptr()
    io.dW0, io.db, io.du, io.dc, io.dv = [t.data_ptr() for t in grads]
    for stages, tl, fl in [(1, None, 0), (2, None, 0), (4, lists[1], 1), (4 | 8, lists[0], 1)]:
        io.stages, io.flags = stages, fl
        io.data_tile_list, io.n_data_tiles = (tl.data_ptr(), tl.numel()) if tl is not None else (None, 0)
        _lib.check(L.fgc_conv_bwd(C.byref(d), C.byref(io), ptr(ws), ws.numel(), stream_ptr()), "partial bwd")
    torch.cuda.synchronize()
    assert torch.equal(bufs["dx"], dx_ref)
    for a, b in zip(grads, grads_ref):
        assert torch.equal(a, b)


@pytest.mark.parametrize("case", ["c1_raw", "c1_coarsened"])
def test_first_layer_backward_without_input_gradient(golden_dir, case):
    """need_dx=False on a narrow input (cin = 6: the network's conv1) takes the vector-ALU first-layer path
    (fgc_conv_narrow.hip: no transposed graph, no per-edge buffer).  Same five parameter gradients as the reference."""
    from facet_graph_convolution_amd import ops
    from facet_graph_convolution_amd.graph import FacetGraph
    z = np.load(os.path.join(golden_dir, "conv_%s.npz" % case))
    z64 = np.load(os.path.join(golden_dir, "conv_%s_f64.npz" % case))
    dev = torch.device("cuda:0")
    x = torch.tensor(z["x"][0], device=dev)
    g = FacetGraph(z["adj"], dev)
    params = _params(x.shape[1], int(z["cout"]), int(z["seed"]), dev)
    y, _, ag = ops.conv_fwd(g, x, None, 0, params)
    dy = torch.tensor(z["dy"][0], device=dev)
    dx, _, grads = ops.conv_bwd(g, x, None, 0, params, ag, y, dy, need_dx=False)
    assert dx is None
    for key, got in zip(["dW0", "db", "du", "dc", "dv"], grads):
        got = got.cpu().numpy()
        ref64 = z64[key].reshape(got.shape)
        scale = max(1.0, np.abs(ref64).max())
        assert np.abs(got - ref64).max() / scale < 5e-6, key
    # activation + pooled output through the narrow forward kernel
    yp_ref = None
    y2, yp, _ = ops.conv_fwd(g, x, None, 0, params, act=1, alpha=0.1, want_pool=(g.n % 4 == 0))
    ref = torch.where(y > 0, y, 0.1 * y)
    assert torch.allclose(y2, ref, atol=1e-7)
    if yp is not None:
        assert torch.equal(yp, y2.view(-1, 4, y2.shape[1]).amax(1))


def test_the_stride_of_r_and_the_layout_of_packed_operands_are_stated_not_implied(golden_dir):
    """ABI 104.  (1) fgc_conv_bwd_io.r_ld: the row stride of r is a property of the buffer - a staged backward whose calls never
    carry FGC_CONV_R_PAD but whose io states the padded stride is bit-identical to the single call (the writer, stage 4, and the
    reader, stage 8, used to re-derive the stride from each call's own flags: one forgotten bit gave silently wrong weight
    gradients); a stride that cannot be one is refused.  (2) fgc_conv_desc.packed_layout: an FGC_CONV_PACKED call made after
    an option moved the operand layout returns FGC_EINVAL instead of multiplying by the stale operand."""
    import ctypes as C
    from facet_graph_convolution_amd import ops, _lib
    from facet_graph_convolution_amd.graph import FacetGraph
    from facet_graph_convolution_amd.ops import make_conv_desc, ptr, stream_ptr
    z = np.load(os.path.join(golden_dir, "conv_c1_coarsened.npz"))
    dev = torch.device("cuda:0")
    g = FacetGraph(z["adj"], dev)
    n, cin, cout = g.n, 32, 32
    x = torch.tensor(np.random.RandomState(0).normal(size=(n, cin)).astype(np.float32), device=dev)
    params = _params(cin, cout, 5, dev)
    L = _lib.lib()
    y_ref, _, ag_ref = ops.conv_fwd(g, x, None, 0, params, act=1)
    dy = torch.tensor(np.random.RandomState(3).normal(size=(n, cout)).astype(np.float32), device=dev)
    dx_ref, _, grads_ref = ops.conv_bwd(g, x, None, 0, params, ag_ref, y_ref, dy, act=1)

    d = make_conv_desc(g, x, None, 0, params, True, 1, 0.1)
    ws = torch.empty(L.fgc_conv_bwd_workspace_bytes(C.byref(d)) + 256, dtype=torch.uint8, device=dev)
    trow, tcol, tedge = g.transposed()
    f32 = dict(dtype=torch.float32, device=dev)
    rld = L.fgc_conv_r_ld(cout, 1, 0)
    assert rld % 32 == 0 and rld > 9 * cout + 24
    io = _lib.ConvBwdIO()
    io.trowptr, io.tcol, io.tedge, io.max_in_deg = trow.data_ptr(), tcol.data_ptr(), tedge.data_ptr(), g.max_in_deg
    bufs = dict(ds=torch.empty(n, cout, **f32), dl=torch.empty(g.nnz, 12, **f32), dag=torch.empty(n, 24, **f32),
                r=torch.full((n, rld), float("nan"), **f32), dx=torch.full((n, cin), float("nan"), **f32))
    grads = [torch.empty_like(p) for p in params]
    io.ag, io.y, io.dy = ag_ref.data_ptr(), y_ref.data_ptr(), dy.data_ptr()
    io.ds, io.dl, io.dag, io.r = (bufs[k].data_ptr() for k in ("ds", "dl", "dag", "r"))
    io.dx0 = bufs["dx"].data_ptr()
    io.dW0, io.db, io.du, io.dc, io.dv = [t.data_ptr() for t in grads]
    io.r_ld = rld
    for stages, fl in [(1 | 2, 0), (4, _lib.CONV_PACKED), (8, _lib.CONV_PACKED)]:      # (no call carries CONV_R_PAD)
        io.stages, io.flags = stages, fl
        _lib.check(L.fgc_conv_bwd(C.byref(d), C.byref(io), ptr(ws), ws.numel(), stream_ptr()), "staged bwd")
    torch.cuda.synchronize()
    assert torch.equal(bufs["dx"], dx_ref)
    for a, b in zip(grads, grads_ref):
        assert torch.equal(a, b)
    r = bufs["r"]
    assert torch.isfinite(r[:, :9 * cout + 24]).all() and torch.isnan(r[:, 9 * cout + 24:]).all()   # pad columns untouched
    # ... and the flag still works for a caller that sets no r_ld (ABI 103 behaviour)
    io.r_ld = 0
    grads2 = [torch.empty_like(p) for p in params]
    io.dW0, io.db, io.du, io.dc, io.dv = [t.data_ptr() for t in grads2]
    io.stages, io.flags = 0, _lib.CONV_R_PAD
    _lib.check(L.fgc_conv_bwd(C.byref(d), C.byref(io), ptr(ws), ws.numel(), stream_ptr()), "bwd, flag form")
    torch.cuda.synchronize()
    for a, b in zip(grads2, grads_ref):
        assert torch.equal(a, b)
    for bad in (9 * cout + 23, 9 * cout + 25, 9 * cout + 26):      # too short / not congruent to 9 cout + 24 modulo 4
        io.r_ld = bad
        assert L.fgc_conv_bwd(C.byref(d), C.byref(io), ptr(ws), ws.numel(), stream_ptr()) == -22
        assert b"r_ld" in L.fgc_last_error()
    io.r_ld = 0

    # (2) the operands in ws were packed with the split d-logits operand (the default for a 32-wide layer on this graph)
    ident = L.fgc_conv_layout_id(C.byref(d))
    with _lib.options(NO_K1_SPLIT=1):
        other = L.fgc_conv_layout_id(C.byref(d))
    if other == ident:
        pytest.skip("this shape has one d-logits operand form on this graph")
    d.packed_layout = ident
    io.stages, io.flags = 0, _lib.CONV_PACKED | _lib.CONV_R_PAD
    _lib.check(L.fgc_conv_bwd(C.byref(d), C.byref(io), ptr(ws), ws.numel(), stream_ptr()), "packed bwd, same options")
    with _lib.options(NO_K1_SPLIT=1):
        assert L.fgc_conv_bwd(C.byref(d), C.byref(io), ptr(ws), ws.numel(), stream_ptr()) == -22
        assert b"packed in layout" in L.fgc_last_error()
        io.flags = _lib.CONV_R_PAD           # not told PACKED: the call packs for itself and runs
        _lib.check(L.fgc_conv_bwd(C.byref(d), C.byref(io), ptr(ws), ws.numel(), stream_ptr()), "bwd packs again")
    torch.cuda.synchronize()
    for a, b in zip(grads2, grads_ref):
        assert torch.allclose(a, b, rtol=1e-4, atol=1e-5)      # (the fp32-MFMA form of the d-logits GEMM: not bit-identical)
    # the MLP's flags carry theirs
    from facet_graph_convolution_amd.net import HIDDEN
    m = L.fgc_mlp_layout_id(32, HIDDEN, 3, 0)
    xs = torch.tensor(np.random.RandomState(1).normal(size=(512, 32)).astype(np.float32), device=dev)
    W1 = torch.randn(32, HIDDEN, device=dev) * 0.1
    b1, W2, b2 = torch.zeros(HIDDEN, device=dev), torch.randn(HIDDEN, 3, device=dev) * 0.1, torch.zeros(3, device=dev)
    wsm = torch.empty(L.fgc_mlp_workspace_bytes(32, HIDDEN, 3) + 256, dtype=torch.uint8, device=dev)
    ym = torch.empty(512, 3, device=dev)
    call = lambda flags: L.fgc_mlp_fwd(ptr(xs), 512, 32, HIDDEN, 3, ptr(W1), ptr(b1), ptr(W2), ptr(b2), 0.1, ptr(ym), None, flags,
                                       ptr(wsm), wsm.numel(), stream_ptr())
    assert call(0) == 0
    y0 = ym.clone()
    assert call(_lib.MLP_PACKED | _lib.mlp_layout(m)) == 0 and torch.equal(ym, y0)
    with _lib.options(NO_MLP_SPLIT=1):
        assert call(_lib.MLP_PACKED | _lib.mlp_layout(m)) == -22 and b"packed in layout" in L.fgc_last_error()
    torch.cuda.synchronize()
