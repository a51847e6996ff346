"""GPU parity of the PAIR form of a convolution over a 4x-upsampled coarse tensor (include/fgc.h: fgc_conv_desc.pair_rowptr;
csrc/fgc_conv_pair.hip) against the oracle's materialised custom_upsampling -> custom_conv2d (model.py:817-825,427-504)
and against the fine form of the same layer."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _local_klist(n, rs, kmax, window, isolated=0.08, fake_tail=0):
    """Random K-list with spatial locality (so that pair degrees stay moderate), rows without any slot, self-only
    (fake) rows at the end, duplicates allowed: siblings of a block get different neighbour sets."""
    adj = np.zeros((n, 23), dtype=np.int32)
    for i in range(n):
        if i >= n - fake_tail:
            adj[i, 0] = i + 1                                # fake node: self only (dataClasses.py:136-146)
            continue
        if rs.uniform() < isolated:
            continue                                         # no slot at all: deg 0, no bias with biasMask
        d = rs.randint(0, kmax)
        adj[i, 0] = i + 1
        adj[i, 1:1 + d] = np.clip(i + rs.randint(-window, window + 1, size=d), 0, n - 1) + 1
    return adj


def _oracle(xc, adj, params, dy, act):
    from oracle import model_ref as R
    x = xc.clone().requires_grad_(True)
    ps = [t.clone().requires_grad_(True) for t in params]
    y = R.custom_conv2d(R.custom_upsampling(x[None], 2), torch.tensor(adj[None]), ps)
    if act:
        y = R.lrelu(y)
    (y[0] * dy).sum().backward()
    return y[0].detach(), x.grad, [t.grad for t in ps]


CASES = [
    # n, cin, cout, act, kmax, window, fake_tail, seed
    (148, 64, 32, 0, 13, 24, 7, 1),      # upconv1's shape; blocks do not fill the last workgroup
    (256, 128, 64, 0, 13, 24, 16, 2),    # upconv2's shape
    (512, 64, 32, 1, 22, 30, 0, 3),      # blocks with more than 16 pairs (chunked), in-degree above 16, activation
    (192, 32, 64, 1, 10, 12, 4, 4),
    (96, 128, 32, 0, 6, 8, 0, 5),
]


@pytest.mark.parametrize("n,cin,cout,act,kmax,window,fake_tail,seed", CASES)
def test_pair_form_matches_oracle_and_fine_form(n, cin, cout, act, kmax, window, fake_tail, seed):
    from facet_graph_convolution_amd import ops
    from facet_graph_convolution_amd.graph import FacetGraph
    from oracle import model_ref as R
    dev = torch.device("cuda:0")
    rs = np.random.RandomState(seed)
    adj = _local_klist(n, rs, kmax, window, fake_tail=fake_tail)
    g = FacetGraph(adj, dev)
    pg = g.pairs()
    xc = torch.tensor(rs.normal(size=(n // 4, cin)).astype(np.float32))
    dy = torch.tensor(rs.normal(size=(n, cout)).astype(np.float32))
    p = R.conv_params(cin, cout, 30 + seed)
    y_ref, gx_ref, gp_ref = _oracle(xc, adj, p, dy, act)
    pd = [t.to(dev) for t in p]
    pairs = {}
    y, _, ag = ops.conv_fwd(g, xc.to(dev), None, 2, pd, act=act, alpha=0.1, pairs=pairs)
    assert pairs["used"], "pair form not taken (max pair degree %d / in %d)" % (pg.max_deg, pg.max_in_deg)
    dx, _, grads = ops.conv_bwd(g, xc.to(dev), None, 2, pd, ag, y, dy.to(dev), act=act, alpha=0.1, pairs=pairs)
    yf, _, agf = ops.conv_fwd(g, xc.to(dev), None, 2, pd, act=act, alpha=0.1)
    dxf, _, gradsf = ops.conv_bwd(g, xc.to(dev), None, 2, pd, agf, yf, dy.to(dev), act=act, alpha=0.1)
    torch.cuda.synchronize()
    print("pairs/block max %d in %d" % (pg.max_deg, pg.max_in_deg))
    assert torch.equal(ag.cpu()[:, :9], agf.cpu()[:, :9]) or (ag.cpu() - agf.cpu()).abs().max() < 2e-6
    np.testing.assert_allclose(y.cpu().numpy(), y_ref.numpy(), atol=3e-6)
    np.testing.assert_allclose(y.cpu().numpy(), yf.cpu().numpy(), atol=3e-6)
    for name, got, fine, ref in zip(["dx", "dW0", "db", "du", "dc", "dv"], [dx] + grads, [dxf] + gradsf, [gx_ref] + gp_ref):
        scale = max(1.0, ref.abs().max().item())
        e_ref = (got.cpu() - ref.reshape(got.shape)).abs().max().item() / scale
        e_fine = (got.cpu() - fine.cpu()).abs().max().item() / scale
        print("%-3s |pair-oracle| %.2e |pair-fine| %.2e (scale %.1f)" % (name, e_ref, e_fine, scale))
        assert e_ref < 5e-6 and e_fine < 5e-6, name


def test_pair_form_accumulates_into_dx_and_respects_no_bias_mask():
    """accumulate0 (the multi-scale heads write g_d3 / g_d2 first) and biasMask = False."""
    from facet_graph_convolution_amd import ops
    from facet_graph_convolution_amd.graph import FacetGraph
    from oracle import model_ref as R
    dev = torch.device("cuda:0")
    rs = np.random.RandomState(11)
    n, cin, cout = 160, 64, 32
    adj = _local_klist(n, rs, 12, 20, fake_tail=8)
    g = FacetGraph(adj, dev)
    xc = torch.tensor(rs.normal(size=(n // 4, cin)).astype(np.float32)).to(dev)
    dy = torch.tensor(rs.normal(size=(n, cout)).astype(np.float32)).to(dev)
    pd = [t.to(dev) for t in R.conv_params(cin, cout, 77)]
    pairs = {}
    y, _, ag = ops.conv_fwd(g, xc, None, 2, pd, bias_mask=False, pairs=pairs)
    yf, _, agf = ops.conv_fwd(g, xc, None, 2, pd, bias_mask=False)
    assert pairs["used"]
    base = torch.tensor(rs.normal(size=(n // 4, cin)).astype(np.float32)).to(dev)
    dx, _, grads = ops.conv_bwd(g, xc, None, 2, pd, ag, y, dy, bias_mask=False, dx0=base.clone(), acc0=True, pairs=pairs)
    dxf, _, gradsf = ops.conv_bwd(g, xc, None, 2, pd, agf, yf, dy, bias_mask=False, dx0=base.clone(), acc0=True)
    torch.cuda.synchronize()
    assert (y - yf).abs().max().item() < 3e-6
    assert (dx - dxf).abs().max().item() < 5e-6 * max(1.0, dxf.abs().max().item())
    for a, b in zip(grads, gradsf):
        assert (a - b).abs().max().item() < 5e-6 * max(1.0, b.abs().max().item())


def test_no_pairs_switch_and_unsupported_shapes_fall_back_to_the_fine_form():
    from facet_graph_convolution_amd import ops
    from facet_graph_convolution_amd.graph import FacetGraph
    from oracle import model_ref as R
    dev = torch.device("cuda:0")
    rs = np.random.RandomState(12)
    n = 128
    adj = _local_klist(n, rs, 12, 20)
    g = FacetGraph(adj, dev)
    xc = torch.tensor(rs.normal(size=(n // 4, 64)).astype(np.float32)).to(dev)
    pd = [t.to(dev) for t in R.conv_params(64, 32, 5)]
    from facet_graph_convolution_amd import _lib
    with _lib.options(NO_PAIRS=1):
        pairs = {}
        y0, _, _ = ops.conv_fwd(g, xc, None, 2, pd, pairs=pairs)
        assert not pairs["used"]
    pairs = {}
    y1, _, _ = ops.conv_fwd(g, xc, None, 2, pd, pairs=pairs)
    assert pairs["used"]
    torch.cuda.synchronize()
    assert (y0 - y1).abs().max().item() < 3e-6
    # a width the pair kernels do not cover (cout = 16): the fine form, silently
    pd16 = [t.to(dev) for t in R.conv_params(64, 16, 6)]
    pairs = {}
    ops.conv_fwd(g, xc, None, 2, pd16, pairs=pairs)
    assert not pairs["used"]


def test_network_uses_the_pair_form_for_both_up_convolutions():
    from facet_graph_convolution_amd.net import FacetDenoiser
    from facet_graph_convolution_amd.dataClasses import TrainingSet
    from facet_graph_convolution_amd.meshgen import icosphere, add_noise
    import ctypes as C
    V, F = icosphere(3)
    ds = TrainingSet()
    ds.addMeshWithGT(add_noise(V, F), F, V, seed=0)
    net = FacetDenoiser("cuda:0", seed=0).bind_mesh(ds.in_list[0], ds.adj_list[0], gt=ds.gt_list[0])
    M = net._mesh
    for name in ("upconv1", "upconv2"):
        assert net.L.fgc_conv_uses_pairs(C.byref(M["descs"][name])) == 1
        assert ("hc_" + name) in M["B"]
    for name in ("conv2", "dconv1"):
        assert net.L.fgc_conv_uses_pairs(C.byref(M["descs"][name])) == 0


def test_facet_sharded_network_uses_the_pair_form_too(golden_dir):
    """A facet-sharded rank runs its up-convolutions on the LOCAL pair graph (columns = [owned coarse rows | unique halo
    parents], shard.ShardPlan), fp32 and bf16 storage."""
    import ctypes as C
    from facet_graph_convolution_amd.shard import make_sim_shards
    prep = np.load(os.path.join(golden_dir, "prep_torus640.npz"))
    adjs = [prep["adj0"], prep["adj1"], prep["adj2"]]
    for dtype in ("f32", "bf16"):
        nets = make_sim_shards(prep["x"], adjs, prep["gt"], 3, "cuda:0", 0, dtype=dtype)
        for n in nets:
            for name in ("upconv1", "upconv2"):
                d = n._mesh["descs"][name]
                assert n.L.fgc_conv_uses_pairs(C.byref(d)) == 1
                g = n._mesh["graphs"][0 if name == "upconv1" else 1]
                assert d.src_rows == d.n // 4 + g.pair.n_halo and g.pair.n_halo <= g.n_halo


def test_network_in_the_pair_form_equals_the_network_in_the_fine_form(fgc_option):
    """The whole training step with the two up-convolutions in the pair form against the same step with the library option NO_PAIRS = 1 (the
    round-3 kernels): same sums in another order - normals within 1e-6, loss 1e-6 relative, every gradient within 2e-5 of its
    tensor's largest entry - on a natively preprocessed torus (Morton order, fake rows in most blocks)."""
    from facet_graph_convolution_amd.net import FacetDenoiser
    from facet_graph_convolution_amd.dataClasses import TrainingSet
    from facet_graph_convolution_amd.meshgen import torus, add_noise
    from facet_graph_convolution_amd.utils import rand_rotation_matrix
    V, F = torus(96, 64)
    ds = TrainingSet()
    ds.addMeshWithGT(add_noise(V, F, 0.2, seed=1), F, V, seed=0)
    x, adjs, gt = ds.in_list[0], ds.adj_list[0], ds.gt_list[0]
    samp = np.random.RandomState(2).randint(x.shape[1], size=4000)
    Rm = rand_rotation_matrix(randnums=np.random.RandomState(3).uniform(size=3))
    out = {}
    for mode in ("0", "1"):
        fgc_option("NO_PAIRS", int(mode))
        net = FacetDenoiser("cuda:0", seed=0).bind_mesh(x, adjs, gt=gt)
        assert bool(net.pair_dims()) == (mode == "0")
        net.set_samples(samp)
        net.set_rotation(Rm)
        loss = net.forward_backward(rotate=True)
        torch.cuda.synchronize()
        out[mode] = (net.buffers["nconv"].clone(), loss[0].item(), [g.clone() for g in net.params.grads])
        del net
    (n0, l0, g0), (n1, l1, g1) = out["0"], out["1"]
    assert (n0 - n1).abs().max().item() < 1e-6
    assert abs(l0 - l1) < 1e-6 * abs(l1)
    for i, (a, b) in enumerate(zip(g0, g1)):
        assert (a - b).abs().max().item() < 2e-5 * max(b.abs().max().item(), 1e-3), i
