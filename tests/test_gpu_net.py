"""GPU parity of the full denoising network (forward, loss, all 44 gradients, inference epilogue, Adam)
against the fixtures produced by executing the reference source."""
import os

import numpy as np
import pytest
import torch

from parity_report import FP32_GRAD_TOL, FP32_NORMAL_TOL, check_gradients, check_normals

pytestmark = pytest.mark.gpu


def _bind(golden_dir, tag, seed, multi_scale=False):
    from facet_graph_convolution_amd.net import FacetDenoiser
    prep = np.load(os.path.join(golden_dir, "prep_%s.npz" % tag))
    net = FacetDenoiser("cuda:0", multi_scale=multi_scale, seed=seed)
    net.bind_mesh(prep["x"], [prep["adj0"], prep["adj1"], prep["adj2"]], gt=prep["gt"])
    return net, prep


@pytest.mark.parametrize("name,tag,seed", [("net_ico3", "ico3", 0), ("net_torus640", "torus640", 1)])
def test_train_forward_backward_matches_reference(golden_dir, name, tag, seed):
    z = np.load(os.path.join(golden_dir, name + ".npz"))
    net, prep = _bind(golden_dir, tag, seed)
    assert net.params.num_parameters() == 474199 and len(net.params.spec) == int(z["n_vars"])
    net.set_rotation(z["R"])
    net.set_samples(z["sample_ind"])
    loss = net.forward_backward(rotate=True)
    torch.cuda.synchronize()
    B = net.buffers
    np.testing.assert_allclose(B["xr"].cpu().numpy(), z["fn_rot"][0], atol=1e-6)
    y0 = B["y0"].cpu().numpy()
    ref = z["y0"][0]
    print("y0 max abs err %.3e (scale %.3f)" % (np.abs(y0 - ref).max(), np.abs(ref).max()))
    np.testing.assert_allclose(y0, ref, rtol=0, atol=3e-6 * max(1.0, np.abs(ref).max()))
    # denoised normals: SURVEY section 8c allows 2e-5 abs on unit vectors (~1e-3 degree); held to what the kernels achieve
    check_normals(B["nconv"], z["n_conv"][0], FP32_NORMAL_TOL, name)
    assert abs(loss[0].item() - float(z["loss"])) < 1e-4 * float(z["loss"])
    check_gradients(net.params.spec, net.params.grads, [z["g%02d" % i] for i in range(len(net.params.grads))],
                    FP32_GRAD_TOL, name)


def test_train_step_matches_the_reference_at_its_patch_size(golden_dir):
    """The HIP step against the FIXTURE (not the oracle) at the reference's own patch size (settings.py:20): net_ico5_20k.npz
    is the reference source - preprocessing with its own coarsening draw, network, loss, backward - executed on an icosphere
    of 20 480 faces (N0 = 25 024: 782 level-0 tiles, more workgroups than the chip has CUs).  Same fp32 tolerances as the
    small fixtures."""
    from facet_graph_convolution_amd.net import FacetDenoiser
    z = np.load(os.path.join(golden_dir, "net_ico5_20k.npz"))
    adjs = [z["adj%d" % l].astype(np.int32) for l in range(3)]
    net = FacetDenoiser("cuda:0", seed=int(z["seed"]))
    net.bind_mesh(z["x"], adjs, gt=z["gt"])
    assert len(net.params.spec) == int(z["n_vars"])
    net.set_rotation(z["R"])
    net.set_samples(z["sample_ind"].astype(np.int64))
    loss = net.forward_backward(rotate=True)
    torch.cuda.synchronize()
    B = net.buffers
    ref = z["y0"][0]
    np.testing.assert_allclose(B["y0"].cpu().numpy(), ref, rtol=0, atol=3e-6 * max(1.0, np.abs(ref).max()))
    check_normals(B["nconv"], z["n_conv"][0], FP32_NORMAL_TOL, "net_ico5_20k")
    assert abs(loss[0].item() - float(z["loss"])) < 1e-4 * float(z["loss"])
    check_gradients(net.params.spec, net.params.grads, [z["g%02d" % i] for i in range(len(net.params.grads))],
                    FP32_GRAD_TOL, "net_ico5_20k (reference fixture, 20 480 faces)")


def test_gradients_vs_float64_truth(golden_dir):
    """Error budget: the GPU fp32 gradients are as close to the float64 gradients as the reference's own fp32 run."""
    z32 = np.load(os.path.join(golden_dir, "net_ico3.npz"))
    z64 = np.load(os.path.join(golden_dir, "net_ico3_f64.npz"))
    net, _ = _bind(golden_dir, "ico3", 0)
    net.set_rotation(z32["R"])
    net.set_samples(z32["sample_ind"])
    net.forward_backward(rotate=True)
    torch.cuda.synchronize()
    e_gpu = e_ref = 0.0
    for i, g in enumerate(net.params.grads):
        t = z64["g%02d" % i]
        scale = max(np.abs(t).max(), 1e-3)
        e_gpu = max(e_gpu, np.abs(g.cpu().numpy() - t).max() / scale)
        e_ref = max(e_ref, np.abs(z32["g%02d" % i] - t).max() / scale)
    print("max rel grad error vs float64: gpu %.3e, reference fp32 %.3e" % (e_gpu, e_ref))
    assert e_gpu < max(4 * e_ref, 1e-3)


def test_forward_is_bitwise_reproducible_and_graph_capture_matches(golden_dir):
    z = np.load(os.path.join(golden_dir, "net_ico3.npz"))
    net, _ = _bind(golden_dir, "ico3", 0)
    net.set_rotation(z["R"])
    net.set_samples(z["sample_ind"])
    net.forward_backward(rotate=True)
    g1 = net.params.grad.clone()
    n1 = net.buffers["nconv"].clone()
    net.forward_backward(rotate=True)
    assert torch.equal(g1, net.params.grad) and torch.equal(n1, net.buffers["nconv"])
    net.forward_backward(rotate=True, capture=True)   # hipGraph capture + replay
    net.forward_backward(rotate=True, capture=True)
    torch.cuda.synchronize()
    assert torch.equal(g1, net.params.grad) and torch.equal(n1, net.buffers["nconv"])


def test_inference_epilogue_matches_reference(golden_dir):
    z = np.load(os.path.join(golden_dir, "infer_ico3.npz"))
    net, prep = _bind(golden_dir, "ico3", 0)
    out = net.infer_normals(prep["permutations"], int(prep["num_faces"]))
    torch.cuda.synchronize()
    check_normals(net.buffers["nconv"], z["n_conv"][0], FP32_NORMAL_TOL, "infer_ico3")
    got = out.cpu().numpy()
    ref = z["predicted_normals"]
    assert got.shape == (1280, 3)
    ang = np.degrees(np.arccos(np.clip((got * ref).sum(1), -1, 1)))
    print("max angular deviation vs reference: %.2e deg" % ang.max())
    assert ang.max() < 0.05 and np.abs(got - ref).max() < FP32_NORMAL_TOL


def test_multiscale_heads_forward(golden_dir):
    z = np.load(os.path.join(golden_dir, "net_ico3_ms.npz"))
    net, _ = _bind(golden_dir, "ico3", 0, multi_scale=True)
    assert len(net.params.spec) == 52
    net.set_rotation(z["R"])
    net.forward(rotate=True)
    torch.cuda.synchronize()
    for k in ("y0", "y1", "y2"):
        ref = z[k][0]
        np.testing.assert_allclose(net.buffers[k].cpu().numpy(), ref, rtol=0, atol=3e-6 * max(1.0, np.abs(ref).max()))


def test_adam_training_reduces_loss(golden_dir):
    net, prep = _bind(golden_dir, "ico3", 0)
    rs = np.random.RandomState(0)
    losses = []
    for it in range(30):
        loss = net.train_step(sample_ind=rs.randint(prep["x"].shape[1], size=4000), R=np.eye(3))
        losses.append(loss[0].item())
    assert np.isfinite(losses).all() and losses[-1] < 0.6 * losses[0], losses


@pytest.mark.parametrize("tag,world", [("ico3", 2), ("ico3", 4), ("torus640", 3)])
def test_facet_sharded_step_matches_single_gpu(golden_dir, tag, world):
    """The sharded schedule (halo rows, cross-edge d-logits, scalar and gradient all-reduces), run as `world`
    shards inside one process, against the unsharded network on the same mesh."""
    from facet_graph_convolution_amd.shard import make_sim_shards, sim_forward_backward
    z = np.load(os.path.join(golden_dir, "net_%s.npz" % tag))
    seed = 0 if tag == "ico3" else 1
    ref, prep = _bind(golden_dir, tag, seed)
    ref.set_rotation(z["R"])
    ref.set_samples(z["sample_ind"])
    ref.forward_backward(rotate=True)
    nets = make_sim_shards(prep["x"], [prep["adj0"], prep["adj1"], prep["adj2"]], prep["gt"], world, "cuda:0", seed)
    for n in nets:
        n.set_rotation(z["R"])
        n.set_samples(z["sample_ind"])
    sim_forward_backward(nets, rotate=True)
    torch.cuda.synchronize()
    full = ref.buffers["nconv"].cpu().numpy()
    for n in nets:
        P = n._mesh["plan"].levels[0]
        got = n.buffers["nconv"].cpu().numpy()
        # forward: same per-row arithmetic in the same order -> identical up to the global-mean all-reduce order
        np.testing.assert_allclose(got, full[P.lo:P.hi], rtol=0, atol=1e-6)
        assert abs(n.buffers["loss"][0].item() - ref.buffers["loss"][0].item()) < 1e-4
        for i, (g, gr) in enumerate(zip(n.params.grads, ref.params.grads)):
            a, b = g.cpu().numpy(), gr.cpu().numpy()
            scale = max(np.abs(b).max(), 1e-3)
            assert np.abs(a - b).max() / scale < 1e-3, "grad %d" % i


def _run_ranks(nproc, backend, port, **extra_env):
    import subprocess, sys
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, FGC_TOOL_BACKEND=backend, MASTER_ADDR="127.0.0.1", **extra_env)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(nproc),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.join(repo, "tools", "shard_gloo_2proc.py")]
    out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0 and "OK" in out.stdout, out.stdout[-2000:] + out.stderr[-4000:]


def test_two_process_sharded_step_over_gloo():
    """One process per shard (two of them on this one GPU), exchanges through torch.distributed: the same DistComm a
    multi-GPU run uses, host-staged because RCCL wants one device per rank."""
    _run_ranks(2, "gloo", 29631)


@pytest.mark.parametrize("min_tiles", ["0", "8"])
def test_two_process_split_schedule_over_gloo(min_tiles):
    """The same two processes with the interior / boundary split forced (FGC_SPLIT_MIN_TILES): exchange_begin / finish
    of the real DistComm round every big layer, eager and replayed from hipGraph segments, all 44 gradients against the
    unsharded network."""
    _run_ranks(2, "gloo", 29633 + int(min_tiles), FGC_SPLIT_MIN_TILES=min_tiles)


def test_rccl_code_path_world_of_one():
    """backend 'nccl' (= RCCL): the async all_to_all_single / all_reduce calls of the sharded schedule, world size 1."""
    _run_ranks(1, "nccl", 29632)


def test_training_alternates_between_cached_meshes(golden_dir):
    """bind_cached: two meshes stay bound in HBM; switching back and forth gives the same losses as re-binding."""
    prep_a = np.load(os.path.join(golden_dir, "prep_ico3.npz"))
    prep_b = np.load(os.path.join(golden_dir, "prep_torus640.npz"))
    meshes = [(p["x"], [p["adj%d" % l] for l in range(3)], p["gt"]) for p in (prep_a, prep_b)]

    from facet_graph_convolution_amd.net import FacetDenoiser

    def run(cached):
        net = FacetDenoiser("cuda:0", seed=4)
        rs = np.random.RandomState(1)
        out = []
        for it in range(8):
            b = it % 2
            x, adjs, gt = meshes[b]
            if cached:
                net.bind_cached(b, x, adjs, gt=gt)
            else:
                net.bind_mesh(x, adjs, gt=gt)
            loss = net.train_step(sample_ind=rs.randint(x.shape[1], size=4000), R=np.eye(3))
            out.append(loss[0].item())
        return out, net
    a, net = run(True)
    b, _ = run(False)
    assert a == b and len(net._mesh_cache) == 2


def _irregular_mesh():
    from facet_graph_convolution_amd.dataClasses import TrainingSet
    from facet_graph_convolution_amd.meshgen import torus, flip_edges, add_noise
    V, F = torus(24, 20)
    F = flip_edges(F, 400, seed=1)          # vertex valences 3..12, facet degrees 11..23 (K-list saturation included)
    ds = TrainingSet()
    ds.addMeshWithGT(add_noise(V, F), F, V, seed=0)
    return ds


def test_irregular_mesh_train_step_matches_oracle():
    """Facet degrees up to K = 23: the d-logits kernel's second edge sweep, three 8-slot batches in the conv
    kernels, ragged rows everywhere.  One train step against the oracle on the same inputs."""
    from facet_graph_convolution_amd.net import FacetDenoiser
    from facet_graph_convolution_amd.utils import rand_rotation_matrix
    from oracle import model_ref as R
    ds = _irregular_mesh()
    x, adjs, gt = ds.in_list[0], ds.adj_list[0], ds.gt_list[0]
    assert max(int((a[0] > 0).sum(1).max()) for a in adjs) > 16
    net = FacetDenoiser("cuda:0", seed=0).bind_mesh(x, adjs, gt=gt)
    samp = np.random.RandomState(2).randint(x.shape[1], size=4000)
    Rm = rand_rotation_matrix(randnums=np.random.RandomState(3).uniform(size=3))
    net.set_samples(samp)
    net.set_rotation(Rm)
    loss = net.forward_backward(rotate=True)
    torch.cuda.synchronize()
    params = [p.requires_grad_(True) for p in R.init_params(0)]
    ref_loss, n_conv = R.train_loss(torch.tensor(x.astype(np.float32)), [torch.tensor(a.astype(np.int32)) for a in adjs],
                                    torch.tensor(gt.astype(np.float32)), params, samp,
                                    torch.tensor(Rm.astype(np.float32)))
    ref_loss.backward()
    check_normals(net.buffers["nconv"], n_conv[0], FP32_NORMAL_TOL, "irregular mesh")
    assert abs(loss[0].item() - ref_loss.item()) < 1e-4 * abs(ref_loss.item())
    check_gradients(net.params.spec, net.params.grads, [p.grad for p in params], FP32_GRAD_TOL, "irregular mesh vs model_ref")


def test_irregular_mesh_sharded_matches_single():
    from facet_graph_convolution_amd.net import FacetDenoiser
    from facet_graph_convolution_amd.shard import make_sim_shards, sim_forward_backward
    ds = _irregular_mesh()
    x, adjs, gt = ds.in_list[0], ds.adj_list[0], ds.gt_list[0]
    samp = np.random.RandomState(4).randint(x.shape[1], size=4000)
    ref = FacetDenoiser("cuda:0", seed=0).bind_mesh(x, adjs, gt=gt)
    ref.set_samples(samp)
    ref.set_rotation(np.eye(3))
    ref.forward_backward(rotate=True)
    nets = make_sim_shards(x, adjs, gt, 3, "cuda:0", seed=0)
    for n in nets:
        n.set_samples(samp)
        n.set_rotation(np.eye(3))
    sim_forward_backward(nets, rotate=True)
    torch.cuda.synchronize()
    for n in nets:
        assert abs(n.buffers["loss"][0].item() - ref.buffers["loss"][0].item()) < 1e-4
        for i, (g, gr) in enumerate(zip(n.params.grads, ref.params.grads)):
            a, b = g.cpu().numpy(), gr.cpu().numpy()
            assert np.abs(a - b).max() / max(np.abs(b).max(), 1e-3) < 1e-3, "grad %d" % i


def test_tiny_mesh_one_coarsest_node():
    """Octahedron: 8 facets -> N0 = 16, N1 = 4, N2 = 1 (every tile partially filled, one coarsest node): one train step
    against the oracle."""
    from facet_graph_convolution_amd.dataClasses import TrainingSet
    from facet_graph_convolution_amd.net import FacetDenoiser
    from oracle import model_ref as R
    V = np.array([[1, 0, 0], [-1, 0, 0], [0, 1, 0], [0, -1, 0], [0, 0, 1], [0, 0, -1]], dtype=np.float32)
    F = np.array([[0, 2, 4], [2, 1, 4], [1, 3, 4], [3, 0, 4], [2, 0, 5], [1, 2, 5], [3, 1, 5], [0, 3, 5]], dtype=np.int32)
    ds = TrainingSet()
    ds.addMeshWithGT(V * np.float32(1.05), F, V, seed=0)
    x, adjs, gt = ds.in_list[0], ds.adj_list[0], ds.gt_list[0]
    assert [a.shape[1] for a in adjs] == [16, 4, 1]
    net = FacetDenoiser("cuda:0", seed=0).bind_mesh(x, adjs, gt=gt)
    samp = np.random.RandomState(0).randint(16, size=4000)
    net.set_samples(samp)
    net.set_rotation(np.eye(3))
    loss = net.forward_backward(rotate=True)
    net.adam_step()
    torch.cuda.synchronize()
    params = [p.requires_grad_(True) for p in R.init_params(0)]
    ref_loss, n_conv = R.train_loss(torch.tensor(x.astype(np.float32)), [torch.tensor(a.astype(np.int32)) for a in adjs],
                                    torch.tensor(gt.astype(np.float32)), params, samp, torch.eye(3))
    ref_loss.backward()
    check_normals(net.buffers["nconv"], n_conv[0], FP32_NORMAL_TOL, "octahedron")
    assert abs(loss[0].item() - ref_loss.item()) < 1e-4 * max(abs(ref_loss.item()), 1.0)
    check_gradients(net.params.spec, net.params.grads, [p.grad for p in params], FP32_GRAD_TOL, "octahedron vs model_ref")


@pytest.mark.parametrize("switches", [
    {"FGC_NO_W8": "1", "FGC_NO_K1DEEP": "1", "FGC_NO_TNSTREAM": "1", "FGC_NO_NARROW": "1"},   # 4-wave / generic kernels
    {"FGC_NO_W8FAST": "1", "FGC_NO_K1M": "1"},                                               # 8-wave generic, VALU logits
    {"FGC_NO_BATCHED": "1", "FGC_NO_FUSED_DS": "1"},                                         # per-layer packs / reductions, ds_db launches
    {"FGC_NO_NARROW_MMA": "1", "FGC_NO_SAVE_Z": "1"},                                        # first layer: vector-ALU products, z recomputed
    {"FGC_NO_FUSED_LOSS": "1"},                                                              # normalise / rotate / loss / gradients as seven launches
])
def test_fallback_kernels_pass_the_smoke_check(switches):
    """The fast paths have switches (FGC_NO_*); with them off the same train step runs on the fallback kernels that
    odd shapes and long edge lists take, and must still match the oracle (__graft_entry__.smoke)."""
    import subprocess, sys
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, "-c", "import __graft_entry__ as g; g.smoke()"], cwd=repo,
                         env=dict(os.environ, **switches), capture_output=True, text=True, timeout=600)
    assert out.returncode == 0 and "smoke ok" in out.stdout, out.stdout[-1500:] + out.stderr[-3000:]


def test_whole_network_pack_and_reduce_match_the_per_layer_path():
    """fgc_conv_pack + FGC_CONV_PACKED / FGC_CONV_DEFER_REDUCE + fgc_conv_bwd_reduce (one launch for what every layer
    used to launch itself) give bit-identical outputs, losses and gradients to per-layer packing and reduction."""
    from facet_graph_convolution_amd.net import FacetDenoiser
    from facet_graph_convolution_amd.dataClasses import TrainingSet
    from facet_graph_convolution_amd.meshgen import icosphere, add_noise
    from facet_graph_convolution_amd.utils import rand_rotation_matrix
    V, F = icosphere(3)
    ds = TrainingSet()
    ds.addMeshWithGT(add_noise(V, F), F, V, seed=0)
    x, adjs, gt = ds.in_list[0], ds.adj_list[0], ds.gt_list[0]
    samp = np.random.RandomState(2).randint(x.shape[1], size=4000)
    Rm = rand_rotation_matrix(randnums=np.random.RandomState(3).uniform(size=3))
    out = []
    for batched in (True, False):
        net = FacetDenoiser("cuda:0", seed=0).bind_mesh(x, adjs, gt=gt)
        net.batched = batched
        losses = [net.train_step(sample_ind=samp, R=Rm)[0].item() for _ in range(3)]
        out.append((losses, net.params.grad.clone(), net.params.theta.clone(), net.buffers["nconv"].clone()))
    assert out[0][0] == out[1][0]
    for a, b in zip(out[0][1:], out[1][1:]):
        assert torch.equal(a, b)


@pytest.mark.parametrize("dtype", ["f32", "bf16"])
def test_first_layer_backward_computes_s_itself(dtype, fgc_option):
    """The first layer's backward kernel computes s = (dy + pooled gradient) * lrelu'(y) / deg in its prologue and leaves one
    bias-gradient partial per workgroup (no ds_db launch); the library option NO_NARROW_FUSED_DS = 1 keeps the separate launch.  Same
    operations per element: every gradient is bit-identical except the first layer's bias gradient, whose partial sums group
    the rows differently."""
    from facet_graph_convolution_amd.net import FacetDenoiser
    from facet_graph_convolution_amd.dataClasses import TrainingSet
    from facet_graph_convolution_amd.meshgen import icosphere, add_noise
    from facet_graph_convolution_amd.utils import rand_rotation_matrix
    V, F = icosphere(3)
    ds = TrainingSet()
    ds.addMeshWithGT(add_noise(V, F), F, V, seed=0)
    x, adjs, gt = ds.in_list[0], ds.adj_list[0], ds.gt_list[0]
    samp = np.random.RandomState(2).randint(x.shape[1], size=4000)
    Rm = rand_rotation_matrix(randnums=np.random.RandomState(3).uniform(size=3))
    out = []
    for off in ("0", "1"):
        fgc_option("NO_NARROW_FUSED_DS", int(off))
        net = FacetDenoiser("cuda:0", seed=0, dtype=dtype).bind_mesh(x, adjs, gt=gt)
        loss = net.train_step(sample_ind=samp, R=Rm)[0].item()
        out.append((loss, [g.clone() for g in net.params.grads]))
    assert out[0][0] == out[1][0]
    sb = 1          # conv1's variables in creation order: W0, b, u, c, v (model.py:427-460)
    for i, (a, b) in enumerate(zip(out[0][1], out[1][1])):
        if i == sb:
            assert (a - b).abs().max().item() <= 2e-6 * max(b.abs().max().item(), 1e-6), "db of the first layer"
            assert not torch.equal(a, torch.zeros_like(a))
        else:
            assert torch.equal(a, b), "gradient %d" % i


def test_packed_step_inputs_are_copied_unless_the_caller_opts_into_reading_in_place():
    """set_step_inputs_packed copies the caller's row by default (the caller may overwrite it right after the call: checked by
    poisoning the row before the step is enqueued); in_place=True lets eager steps read the row where it is (no copy launch)
    until a hipGraph holds the own buffer.  Same steps every way, and set_rotation afterwards must not write into the
    caller's window."""
    from facet_graph_convolution_amd.net import FacetDenoiser
    from facet_graph_convolution_amd.dataClasses import TrainingSet
    from facet_graph_convolution_amd.meshgen import icosphere, add_noise
    from facet_graph_convolution_amd.utils import rand_rotation_matrix
    V, F = icosphere(3)
    ds = TrainingSet()
    ds.addMeshWithGT(add_noise(V, F), F, V, seed=0)
    x, adjs, gt = ds.in_list[0], ds.adj_list[0], ds.gt_list[0]
    rs = np.random.RandomState(4)
    samples = [rs.randint(x.shape[1], size=4000) for _ in range(4)]
    rots = [rand_rotation_matrix(randnums=rs.uniform(size=3)) for _ in range(4)]
    window = FacetDenoiser.pack_step_inputs(samples, rots, "cuda:0")
    keep = window.clone()
    out = []
    for mode in ("alias", "copy", "graph"):
        net = FacetDenoiser("cuda:0", seed=0).bind_mesh(x, adjs, gt=gt)
        losses = []
        for k in range(4):
            if mode == "copy":
                row = window[k].clone()
                net.set_step_inputs_packed(row)
                row.zero_()                    # the old staging-buffer pattern: refill the row right after the call
            else:
                net.set_step_inputs_packed(window[k], in_place=True)
            own = net.buffers["step_in"].data_ptr() == net.buffers["step_in_own"].data_ptr()
            assert own == (mode == "copy" or net._graph_fb is not None)
            net.forward_backward(rotate=True, capture=(mode == "graph"))
            net.adam_step()
            losses.append(net.buffers["loss"][0].item())
        out.append((losses, net.params.theta.clone()))
        net.set_rotation(np.eye(3))
        torch.cuda.synchronize()
        assert torch.equal(window, keep)
    for o in out[1:]:
        assert o[0] == out[0][0] and torch.equal(o[1], out[0][1])


@pytest.mark.parametrize("dtype", ["f32", "bf16"])
def test_grouped_weight_gradient_launches_equal_the_per_layer_launches(dtype, monkeypatch):
    """FGC_CONV_DEFER_DW: fgc_conv_bwd_reduce runs the weight-gradient GEMMs of all layers in one launch per kernel form (the
    bf16 network's default; every layer keeps its own `r` until then).  Same tiles, same slabs, same sums: every gradient is
    bit-identical to the per-layer launches, over several steps."""
    from facet_graph_convolution_amd.net import FacetDenoiser
    from facet_graph_convolution_amd.dataClasses import TrainingSet
    from facet_graph_convolution_amd.meshgen import icosphere, add_noise
    from facet_graph_convolution_amd.utils import rand_rotation_matrix
    V, F = icosphere(3)
    ds = TrainingSet()
    ds.addMeshWithGT(add_noise(V, F), F, V, seed=0)
    x, adjs, gt = ds.in_list[0], ds.adj_list[0], ds.gt_list[0]
    samp = np.random.RandomState(2).randint(x.shape[1], size=4000)
    Rm = rand_rotation_matrix(randnums=np.random.RandomState(3).uniform(size=3))
    out = []
    for grouped in ("1", "0"):
        monkeypatch.setenv("FGC_GROUPED_DW", grouped)
        net = FacetDenoiser("cuda:0", seed=0, dtype=dtype).bind_mesh(x, adjs, gt=gt)
        assert net.grouped_dw == (grouped == "1") and len(net.grouped_dw_layers) == (8 if grouped == "1" else 0)
        losses = [net.train_step(sample_ind=samp, R=Rm)[0].item() for _ in range(3)]
        out.append((losses, net.params.grad.clone(), net.params.theta.clone()))
    assert out[0][0] == out[1][0]
    assert torch.equal(out[0][1], out[1][1]) and torch.equal(out[0][2], out[1][2])


def test_two_networks_in_one_process_run_different_kernel_forms():
    """FacetDenoiser(options=...): per-network library options travel in every layer's descriptor (fgc_conv_desc.options), the
    process-level table stays untouched.  One network in the pair form and one in the fine form of the up-convolutions, and
    one with the 32-node conv kernels, side by side and interleaved step by step: each keeps its own form, all agree."""
    from facet_graph_convolution_amd import _lib
    from facet_graph_convolution_amd.net import FacetDenoiser
    from facet_graph_convolution_amd.dataClasses import TrainingSet
    from facet_graph_convolution_amd.meshgen import icosphere, add_noise
    from facet_graph_convolution_amd.utils import rand_rotation_matrix
    V, F = icosphere(4)
    ds = TrainingSet()
    ds.addMeshWithGT(add_noise(V, F), F, V, seed=0)
    x, adjs, gt = ds.in_list[0], ds.adj_list[0], ds.gt_list[0]
    samp = np.random.RandomState(2).randint(x.shape[1], size=4000)
    Rm = rand_rotation_matrix(randnums=np.random.RandomState(3).uniform(size=3))
    before = {n: _lib.get_option(n) for n in ("NO_PAIRS", "W8_NT16")}
    nets = [FacetDenoiser("cuda:0", seed=0).bind_mesh(x, adjs, gt=gt),
            FacetDenoiser("cuda:0", seed=0, options={"NO_PAIRS": 1}).bind_mesh(x, adjs, gt=gt),
            FacetDenoiser("cuda:0", seed=0, options={"W8_NT16": 0}).bind_mesh(x, adjs, gt=gt)]
    assert set(nets[0].pair_dims()) == {"upconv1", "upconv2"} and nets[1].pair_dims() == {} and len(nets[2].pair_dims()) == 2
    for n in nets:
        n.set_samples(samp)
        n.set_rotation(Rm)
    for _ in range(2):                       # interleaved: no network's form leaks into another's calls
        for n in nets:
            n.forward_backward(rotate=True)
    torch.cuda.synchronize()
    assert {n: _lib.get_option(n) for n in before} == before
    ref, names = nets[0], _lib.option_names()
    for other in nets[1:]:
        assert abs(other.buffers["loss"][0].item() - ref.buffers["loss"][0].item()) < 1e-5 * abs(ref.buffers["loss"][0].item())
        label = "per-network options %s vs the default forms" % {names[o.index]: o.value for o in other.option_overrides}
        check_gradients(ref.params.spec, other.params.grads, [g.clone() for g in ref.params.grads], 2e-5, label)
    # the 32-node form really ran: bitwise it cannot equal the half-tile form's sums in every tensor
    assert any(not torch.equal(a, b) for a, b in zip(nets[2].params.grads, ref.params.grads))


def test_pool_gradient_folded_into_the_conv_backward_equals_the_separate_pass():
    """fgc_conv_bwd_io.pool_y / pool_dy: the gradient of the 4:1 max pooling behind conv1 and conv2 is added to dy inside
    stage 1 of those layers (the d-logits kernel's prologue for conv2, ds_db_kernel for the narrow first layer) instead of
    two fgc_pool4_bwd launches: same additions in the same order, so every gradient is bit-identical."""
    from facet_graph_convolution_amd.net import FacetDenoiser
    from facet_graph_convolution_amd.dataClasses import TrainingSet
    from facet_graph_convolution_amd.meshgen import icosphere, add_noise
    from facet_graph_convolution_amd.utils import rand_rotation_matrix
    V, F = icosphere(3)
    ds = TrainingSet()
    ds.addMeshWithGT(add_noise(V, F), F, V, seed=0)
    x, adjs, gt = ds.in_list[0], ds.adj_list[0], ds.gt_list[0]
    samp = np.random.RandomState(2).randint(x.shape[1], size=4000)
    Rm = rand_rotation_matrix(randnums=np.random.RandomState(3).uniform(size=3))
    out = []
    for fused in (True, False):
        net = FacetDenoiser("cuda:0", seed=0)
        net.fused_pool = fused
        net.bind_mesh(x, adjs, gt=gt)
        losses = [net.train_step(sample_ind=samp, R=Rm)[0].item() for _ in range(3)]
        out.append((losses, net.params.grad.clone(), net.params.theta.clone()))
    assert out[0][0] == out[1][0]
    assert torch.equal(out[0][1], out[1][1]) and torch.equal(out[0][2], out[1][2])


def test_graph_replay_survives_a_device_synchronise():
    """forward_backward(capture=True) replays one hipGraph per step.  With ROCm's pre-built graph packets a replay
    enqueued after a hipDeviceSynchronize went wrong (loss 88 deg instead of 33); the package turns that runtime
    feature off at import.  Here: the same 8 training steps eager and replayed, with a device synchronise after
    step 3, must agree bit for bit."""
    import subprocess, sys, textwrap
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = textwrap.dedent("""
        import numpy as np, torch
        import facet_graph_convolution_amd
        from facet_graph_convolution_amd.net import FacetDenoiser
        from facet_graph_convolution_amd.dataClasses import TrainingSet
        from facet_graph_convolution_amd.meshgen import icosphere, add_noise
        from facet_graph_convolution_amd.utils import rand_rotation_matrix
        V, F = icosphere(4)
        ds = TrainingSet(); ds.addMeshWithGT(add_noise(V, F), F, V, seed=0)
        x, adjs, gt = ds.in_list[0], ds.adj_list[0], ds.gt_list[0]
        rs = np.random.RandomState(1)
        S = torch.tensor(np.stack([rs.randint(x.shape[1], size=4000) for _ in range(8)]).astype(np.int32)).cuda()
        R = torch.tensor(np.stack([rand_rotation_matrix(randnums=rs.uniform(size=3)).reshape(9) for _ in range(8)])
                         .astype(np.float32)).cuda()
        out = []
        for capture in (False, True):
            net = FacetDenoiser("cuda:0", seed=0).bind_mesh(x, adjs, gt=gt)
            for k in range(8):
                net.set_step_inputs_device(S[k], R[k])
                net.forward_backward(rotate=True, capture=capture)
                net.adam_step()
                if k == 2:
                    torch.cuda.synchronize()
            torch.cuda.synchronize()
            out.append((net.buffers["loss"][0].item(), net.params.theta.clone()))
        assert out[0][0] == out[1][0] and torch.equal(out[0][1], out[1][1]), (out[0][0], out[1][0])
        print("replay ok", out[0][0])
    """)
    r = subprocess.run([sys.executable, "-c", code], cwd=repo, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "replay ok" in r.stdout, r.stdout[-1500:] + r.stderr[-3000:]


def test_multiscale_training_step_matches_oracle(golden_dir):
    """Training the three heads (build extension: angular loss per level against the pooled ground truth; the
    reference's point-set objective is out of scope): losses and all 52 gradients against torch autograd through the
    oracle's multi-scale network."""
    from oracle import model_ref as R
    from facet_graph_convolution_amd.utils import rand_rotation_matrix
    net, prep = _bind(golden_dir, "ico3", 0, multi_scale=True)
    x, adjs, gt = prep["x"], [prep["adj0"], prep["adj1"], prep["adj2"]], prep["gt"]
    samp = np.random.RandomState(2).randint(x.shape[1], size=4000)
    Rm = rand_rotation_matrix(randnums=np.random.RandomState(3).uniform(size=3))
    net.set_samples(samp)
    net.set_rotation(Rm)
    net.forward_backward(rotate=True)
    torch.cuda.synchronize()
    params = [p.requires_grad_(True) for p in R.init_params(0, multi_scale=True)]
    tot, losses = R.train_loss_ms(torch.tensor(x.astype(np.float32)), [torch.tensor(a.astype(np.int32)) for a in adjs],
                                  torch.tensor(gt.astype(np.float32)), params, samp, torch.tensor(Rm.astype(np.float32)))
    tot.backward()
    got = [net.buffers["loss"][0].item(), net.buffers["loss1"][0].item(), net.buffers["loss2"][0].item()]
    for a, b in zip(got, losses):
        assert abs(a - b.item()) < 1e-4 * abs(b.item()), (got, [l.item() for l in losses])
    assert len(net.params.grads) == 52
    check_gradients(net.params.spec, net.params.grads, [p.grad for p in params], FP32_GRAD_TOL,
                    "three heads, ico3 vs model_ref")
    # and it trains
    rs = np.random.RandomState(0)
    first = None
    for it in range(25):
        loss = net.train_step(sample_ind=rs.randint(x.shape[1], size=4000), R=np.eye(3))
        tot_l = loss[0].item() + net.buffers["loss1"][0].item() + net.buffers["loss2"][0].item()
        first = tot_l if first is None else first
    assert tot_l < 0.7 * first
