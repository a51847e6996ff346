"""No kernel of a training step writes outside the buffers it was given.

Every device tensor that `FacetDenoiser` allocates while it is built and bound (parameters, activations and their gradient
twins, the shared backward scratch, every per-layer workspace) is placed between two 64 KB guard zones filled with a
byte pattern; after a forward + backward + Adam step - eager, and replayed from a hipGraph - the guard zones must be
untouched.  (Found the hard way at the end of round 2: the first layer's backward scratch was 8 KB short and the fixed-order
sum of its bias gradient wrote past the workspace, silently for a whole round - DESIGN.md section 6.)"""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

GUARD = 64 * 1024
PATTERN = 0xA5


class _Guarded:
    """Context manager: torch.empty / zeros / empty_like / zeros_like / eye on a CUDA device return views into padded
    storage for as long as it is active."""

    def __init__(self, pattern=PATTERN):
        self.bases = []
        self.saved = {}
        self.pattern = pattern

    def _alloc(self, shape, dtype, device, zero):
        dtype = dtype or torch.float32
        nbytes = int(np.prod(shape, dtype=np.int64)) * torch.empty((), dtype=dtype).element_size()
        pad = (nbytes + 255) // 256 * 256
        base = self.saved["empty"](pad + 2 * GUARD, dtype=torch.uint8, device=device)
        base.fill_(self.pattern)
        view = base[GUARD:GUARD + nbytes].view(dtype).view(*shape)
        if zero:
            view.zero_()
        self.bases.append((base, nbytes))
        return view

    @staticmethod
    def _is_cuda(device):
        return device is not None and str(device).startswith("cuda")

    def __enter__(self):
        for name in ("empty", "zeros", "empty_like", "zeros_like", "eye"):
            self.saved[name] = getattr(torch, name)

        def make(name, zero):
            real = self.saved[name]

            def f(*size, **kw):
                if not self._is_cuda(kw.get("device")) or kw.get("pin_memory"):
                    return real(*size, **kw)
                shape = tuple(size[0]) if len(size) == 1 and isinstance(size[0], (tuple, list, torch.Size)) else tuple(size)
                return self._alloc(shape, kw.get("dtype"), kw["device"], zero)
            return f

        def make_like(name, zero):
            real = self.saved[name]

            def f(t, **kw):
                dev = kw.get("device", t.device)
                if not self._is_cuda(dev):
                    return real(t, **kw)
                return self._alloc(tuple(t.shape), kw.get("dtype", t.dtype), dev, zero)
            return f

        def eye(n, **kw):
            if not self._is_cuda(kw.get("device")):
                return self.saved["eye"](n, **kw)
            out = self._alloc((n, n), kw.get("dtype"), kw["device"], True)
            out.copy_(self.saved["eye"](n, dtype=out.dtype))
            return out

        torch.empty, torch.zeros = make("empty", False), make("zeros", True)
        torch.empty_like, torch.zeros_like = make_like("empty_like", False), make_like("zeros_like", True)
        torch.eye = eye
        return self

    def __exit__(self, *exc):
        for name, fn in self.saved.items():
            setattr(torch, name, fn)

    def check(self, what):
        torch.cuda.synchronize()
        bad = []
        for k, (base, nbytes) in enumerate(self.bases):
            pad = (nbytes + 255) // 256 * 256
            lo, hi = base[:GUARD], base[GUARD + pad:]
            P = self.pattern
            if not bool((lo == P).all()) or not bool((hi == P).all()):
                first = int((hi != P).nonzero()[0]) if not bool((hi == P).all()) else -1
                bad.append((k, nbytes, first, int((hi != P).sum()), int((lo != P).sum())))
        assert not bad, "%s: guard zones overwritten (allocation #, bytes, first byte past the end, count after, count before): %s" % (
            what, bad[:5])


def _mesh(nu, nv, flips=0):
    from facet_graph_convolution_amd.dataClasses import TrainingSet
    from facet_graph_convolution_amd.meshgen import torus, add_noise, flip_edges
    V, F = torus(nu, nv)
    if flips:
        F = flip_edges(F, flips, seed=1)     # irregular: facet degrees above 16 (24-slot kernels, LONG d-logits form)
    ds = TrainingSet()
    ds.addMeshWithGT(add_noise(V, F, 0.2, seed=3), F, V, seed=7)
    return ds.in_list[0], ds.adj_list[0], ds.gt_list[0]


@pytest.mark.parametrize("dtype", ["f32", "bf16"])
@pytest.mark.parametrize("size", [(24, 20), (140, 140), (320, 256), (100, 64, 3000), (24, 20, 400)])
def test_training_step_stays_inside_its_buffers(dtype, size):
    """960 facets (a handful of tiles), 39 200 (the oracle-parity size) and 163 840 (levels beyond 131k nodes: the sizes the
    missing scratch was found at); 12 800 and 960 facets with flipped edges (irregular graphs: the 24-slot kernels)."""
    from facet_graph_convolution_amd.net import FacetDenoiser
    x, adjs, gt = _mesh(*size)
    with _Guarded() as g:
        net = FacetDenoiser("cuda:0", seed=0, dtype=dtype).bind_mesh(x, adjs, gt=gt)
        assert len(g.bases) > 40
        g.check("bind")
        rs = np.random.RandomState(1)
        for step in range(2):
            net.train_step(sample_ind=rs.randint(x.shape[1], size=4000), R=np.eye(3))
        g.check("eager steps")
        for step in range(3):
            net.train_step(sample_ind=rs.randint(x.shape[1], size=4000), R=np.eye(3), capture=True)
        g.check("hipGraph steps")
        net.forward(rotate=False)
        g.check("inference forward")


@pytest.mark.parametrize("dtype", ["f32", "bf16"])
def test_results_do_not_depend_on_what_lies_around_the_buffers(dtype):
    """Reads: the same two training steps with the guard zones filled with 0x00 and with 0xFF (NaN as a float, -1 as an
    index).  A kernel that reads past a buffer and lets the value reach a result - or an address - shows up as a different
    gradient (or a fault)."""
    from facet_graph_convolution_amd.net import FacetDenoiser
    x, adjs, gt = _mesh(140, 140)
    out = []
    for pattern in (0x00, 0xFF):
        with _Guarded(pattern) as g:
            net = FacetDenoiser("cuda:0", seed=0, dtype=dtype).bind_mesh(x, adjs, gt=gt)
            rs = np.random.RandomState(1)
            for step in range(2):
                loss = net.train_step(sample_ind=rs.randint(x.shape[1], size=4000), R=np.eye(3))
            g.check("pattern %#x" % pattern)
            out.append(([t.clone() for t in net.params.grads], net.params.theta.clone(), loss.clone(),
                        net.buffers["nconv"].clone()))
            del net
    (ga, ta, la, na), (gb, tb, lb, nb) = out
    assert torch.equal(la, lb) and torch.equal(na, nb) and torch.equal(ta, tb)
    for i, (a, b) in enumerate(zip(ga, gb)):
        assert torch.equal(a, b), "gradient %d depends on the bytes around the buffers" % i
    assert bool(torch.isfinite(ta).all())


def test_multi_scale_training_step_stays_inside_its_buffers():
    from facet_graph_convolution_amd.net import FacetDenoiser
    x, adjs, gt = _mesh(64, 48)
    with _Guarded() as g:
        net = FacetDenoiser("cuda:0", seed=0, multi_scale=True).bind_mesh(x, adjs, gt=gt)
        rs = np.random.RandomState(1)
        for step in range(2):
            net.train_step(sample_ind=rs.randint(x.shape[1], size=4000), R=np.eye(3))
        net.forward_multi_scale()
        g.check("multi-scale steps")


@pytest.mark.parametrize("world", [2, 3])
def test_facet_sharded_step_stays_inside_its_buffers(world):
    """N shards in one process (shard.sim_run): halo tails, packed exchange buffers - the blocking schedule (these shards
    have fewer interior tiles than the split threshold).  The interior / boundary split with its tile lists runs inside
    guard zones in tests/test_gpu_shard_sched.py (forced, and at 200k / 800k / 1M facets)."""
    from facet_graph_convolution_amd.shard import make_sim_shards, sim_forward_backward
    x, adjs, gt = _mesh(96, 64)
    with _Guarded() as g:
        nets = make_sim_shards(x, adjs, gt, world, "cuda:0", seed=0)
        samp = np.random.RandomState(2).randint(x.shape[1], size=4000)
        for n in nets:
            n.set_rotation(np.eye(3))
            n.set_samples(samp)
        sim_forward_backward(nets, rotate=True)
        g.check("sharded step")


def _conv_cases():
    import test_gpu_conv
    return list(test_gpu_conv.CONV_CASES)


OP_TESTS = [
    ("test_gpu_ops", "test_mlp_forward", dict(n=1000, cin=32)),
    ("test_gpu_ops", "test_mlp_forward", dict(n=130, cin=128)),
    ("test_gpu_ops", "test_mlp_backward", dict(n=5000, cin=32)),
    ("test_gpu_ops", "test_mlp_backward", dict(n=6, cin=32)),
    ("test_gpu_ops", "test_mlp_backward", dict(n=333, cin=128)),
    ("test_gpu_ops", "test_mlp_backward", dict(n=70, cin=48)),
    ("test_gpu_ops", "test_elementwise_ops", {}),
    ("test_gpu_ops", "test_normalize_and_loss", dict(n=5000)),
    ("test_gpu_ops", "test_rotate_adam_epilogue_gather", {}),
    ("test_gpu_ops", "test_vertex_update_matches_reference", dict(tag="ico3")),
    ("test_gpu_ops", "test_vertex_update_matches_reference", dict(tag="torus_open")),
    ("test_gpu_ops", "test_multiscale_vertex_update_matches_reference", {}),
    ("test_gpu_conv", "test_conv_forward_matches_golden", "ALL_CASES"),
    ("test_gpu_conv", "test_conv_backward_matches_golden", "ALL_CASES"),
    ("test_gpu_conv", "test_conv_concat_upsample_and_pool_match_oracle", {}),
    ("test_gpu_conv", "test_conv_backward_fused_addressing_matches_oracle", dict(mode="concat")),
    ("test_gpu_conv", "test_conv_backward_fused_addressing_matches_oracle", dict(mode="upsample")),
    ("test_gpu_conv", "test_partial_calls_compose_to_the_whole_layer", {}),
    ("test_gpu_conv", "test_first_layer_backward_without_input_gradient", dict(case="c1_raw")),
    ("test_gpu_conv", "test_first_layer_backward_without_input_gradient", dict(case="c1_coarsened")),
    ("test_gpu_model_api", "test_custom_conv2d_signature_and_return", {}),
    ("test_gpu_model_api", "test_multiscale_heads_train_through_the_operator_api", {}),
    ("test_gpu_bf16", "test_bf16_mlp_kernels_against_torch", {}),
]


@pytest.mark.parametrize("module,name,kw", OP_TESTS, ids=["%s-%s" % (n, "-".join(str(v) for v in k.values()) if isinstance(k, dict) else "all")
                                                        for _, n, k in OP_TESTS])
def test_operator_level_tests_stay_inside_their_buffers(golden_dir, module, name, kw):
    """The operator-level parity tests (conv forward / backward cases, MLP shapes incl. the odd ones, elementwise ops,
    vertex update, the reference-style operator API) once more with every torch.empty / zeros(..., device=cuda) they and
    the package make between guard zones: outputs, workspaces and gradient buffers of the C-ABI calls."""
    import importlib
    import inspect
    fn = getattr(importlib.import_module(module), name)
    runs = [dict(case=c) for c in _conv_cases()] if kw == "ALL_CASES" else [dict(kw)]
    for args in runs:
        if "golden_dir" in inspect.signature(fn).parameters:
            args["golden_dir"] = golden_dir
        with _Guarded() as g:
            fn(**args)
            assert g.bases, "nothing was allocated through the guarded constructors"
            g.check("%s(%s)" % (name, {k: v for k, v in args.items() if k != "golden_dir"}))
