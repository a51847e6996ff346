"""FacetDenoiser: the reference's 3-level graph U-Net + per-facet MLP, hand-scheduled on libfgc.

Replaces, for one mesh / patch (batch size is 1 in the reference, train.py:405):
  * ``get_model_reg_multi_scale``  (model.py:837-946)        -> ``forward``
  * ``normalizeTensor``            (utils.py:1700-1715)      -> fused after the MLP
  * the training objective         (train.py:439-451,503-520) -> ``loss_and_grad`` / ``train_step``
  * the inference epilogue         (train.py:115-121,136)    -> ``infer_normals``

Design (MI355X-first, nothing traced or compiled at run time):
  * every activation, gradient and scratch buffer of a mesh lives in HBM for the whole run and is
    allocated ONCE per mesh size (a 100k-facet mesh needs ~0.7 GB of 288 GB);
  * parameters, gradients and Adam moments are three flat fp32 buffers (16-byte aligned views per
    variable, creation order of the reference), so Adam is one kernel and a data-parallel
    all-reduce is one RCCL call;
  * pooling, upsampling, concat and leaky-ReLU never exist as tensors of their own: they are
    address modes / epilogues of the conv kernels (see include/fgc.h);
  * the backward pass is scheduled by hand in a fixed order (no autograd tape, no float atomics):
    results are bitwise reproducible;
  * the whole forward+backward enqueue is capturable into a hipGraph (``capture=True``).
"""
import contextlib
import ctypes as C
import gc
import os
import warnings
import math

import numpy as np
import torch

from . import _lib
from ._lib import ConvDesc, ConvBwdIO, AG_LD, DL_LD, FGC_M
from .graph import FacetGraph, as_graph

LRELU_ALPHA = 0.1      # model.py:846
STD_W, STD_B = 0.05, 0.01  # model.py:17-18
HIDDEN = 1024          # model.py:936
COST_SAMPLES = 4000    # train.py:411


def param_spec(multi_scale=False, in_channels=6):
    """(kind, shape) per variable in the reference's creation order (model.py:430-433,447,767-768,855-941)."""
    M = FGC_M

    def conv(cin, cout):
        return [("weight", (M, cout, cin)), ("bias", (cout,)), ("assignment", (M, cin)), ("assignment", (M,)),
                ("assignment", (M, cin))]

    def lin(cin, cout):
        return [("weight", (cin, cout)), ("bias", (cout,))]

    spec = conv(in_channels, 32) + conv(32, 64) + conv(64, 128) + conv(128, 128)
    if multi_scale:
        spec += lin(128, HIDDEN) + lin(HIDDEN, 3)
    spec += conv(128, 64) + conv(128, 64)
    if multi_scale:
        spec += lin(64, HIDDEN) + lin(HIDDEN, 3)
    spec += conv(64, 32) + conv(64, 32) + lin(32, HIDDEN) + lin(HIDDEN, 3)
    return spec


class FlatParams:
    """All variables as 16-byte aligned views into one flat fp32 buffer (plus grad / Adam moment twins)."""

    def __init__(self, spec, device):
        self.spec = spec
        self.offsets = []
        off = 0
        for _, shape in spec:
            self.offsets.append(off)
            off += (int(np.prod(shape)) + 3) // 4 * 4
        self.total = off
        self.device = torch.device(device)
        self.theta = torch.zeros(off, dtype=torch.float32, device=self.device)
        # the gradient buffer has a 4-float tail: a facet-sharded step puts its loss sum there, so that one all-reduce
        # carries the gradient and the loss
        self.grad_ext = torch.zeros(off + 4, dtype=torch.float32, device=self.device)
        self.grad = self.grad_ext[:off]
        self.m = torch.zeros_like(self.theta)
        self.v = torch.zeros_like(self.theta)
        self.step = 0

    def _views(self, flat):
        return [flat[o:o + int(np.prod(s))].view(*s) for o, (_, s) in zip(self.offsets, self.spec)]

    @property
    def values(self):
        return self._views(self.theta)

    @property
    def grads(self):
        return self._views(self.grad)

    def init_random(self, seed=0):
        """N(0, 0.05^2) weights/assignments, N(0, 0.01^2) biases from RandomState(seed) in creation order."""
        rs = np.random.RandomState(seed)
        host = np.zeros(self.total, dtype=np.float32)
        for o, (kind, shape) in zip(self.offsets, self.spec):
            std = STD_B if kind == "bias" else STD_W
            host[o:o + int(np.prod(shape))] = rs.normal(0.0, std, size=shape).astype(np.float32).reshape(-1)
        self.theta.copy_(torch.from_numpy(host))

    def load(self, tensors):
        assert len(tensors) == len(self.spec)
        for view, t in zip(self.values, tensors):
            view.copy_(torch.as_tensor(t, dtype=torch.float32).reshape(view.shape))

    def num_parameters(self):
        return sum(int(np.prod(s)) for _, s in self.spec)


def _p(t):
    return C.c_void_p(t.data_ptr()) if t is not None else C.c_void_p(0)


_HIP = [None]


def _graph_node_count(g):
    """Nodes of a captured (not yet instantiated) torch.cuda.CUDAGraph(keep_graph=True), through hipGraphGetNodes.  Raises
    when the runtime cannot say: whether a stretch of the schedule is a graph at all must not be a guess."""
    if _HIP[0] is None:
        _HIP[0] = C.CDLL("libamdhip64.so")
    n = C.c_size_t(0)
    rc = _HIP[0].hipGraphGetNodes(C.c_void_p(g.raw_cuda_graph()), None, C.byref(n))
    if rc != 0:
        raise RuntimeError("hipGraphGetNodes failed (%d): cannot tell whether a schedule segment has launches" % rc)
    return int(n.value)


@contextlib.contextmanager
def _no_gc_while_capturing():
    """A stream capture must not be interrupted by Python's cyclic garbage collector.  A dead reference cycle that still owns
    device objects - the hipGraphs and pool tensors of an earlier network, say - is freed whenever the collector happens to
    run, and it runs on allocation counts: in the middle of a capture its destructors call hipGraphDestroy / hipFree, which
    a capturing stream refuses; torch raises from a destructor and the process ABORTS (the round-3 abort, caught in round 4
    with its stack: `Garbage-collecting` under `_capture_segments`).  torch.cuda.graph stopped
    collecting in its __enter__ (torch.compiler.config.force_cudagraph_gc, default off), so: collect once before the
    capture, keep the collector off until it has ended.  FGC_NO_CAPTURE_GC_GUARD=1 (developer switch): no guard."""
    if os.environ.get("FGC_NO_CAPTURE_GC_GUARD", "0") == "1":
        yield
        return
    gc.collect()
    was = gc.isenabled()
    gc.disable()
    try:
        yield
    finally:
        if was:
            gc.enable()


class _ConvLayer:
    """Static description of one conv: which buffers it reads/writes and its parameter slots."""

    def __init__(self, name, level, x0, x1, shift, pidx, y, pool, act):
        self.name, self.level, self.x0, self.x1, self.shift = name, level, x0, x1, shift
        self.pidx, self.y, self.pool, self.act = pidx, y, pool, act


class FacetDenoiser:
    def __init__(self, device="cuda", multi_scale=False, in_channels=6, seed=0, dtype="f32", options=None):
        if not torch.cuda.is_available():
            raise RuntimeError("FacetDenoiser needs an MI355X (no CPU fallback)")
        if dtype not in ("f32", "bf16"):
            raise ValueError("dtype must be 'f32' or 'bf16' (storage of the activations; weights stay fp32)")
        self.dtype = dtype
        self.L = _lib.lib()
        # options: {name: value} of library options (fgc_option_name) that hold for THIS network's conv layers only - they
        # travel in every layer's descriptor (fgc_conv_desc.options), the process-level table (fgc_set_option) is not touched:
        # two networks in one process can run different kernel forms.  (The MLP entry points take no descriptor: their
        # options stay process-level.)
        self.option_overrides = _lib.option_overrides(**options) if options else None
        self.device = torch.device(device)
        self.multi_scale = multi_scale
        self.in_channels = in_channels
        self.params = FlatParams(param_spec(multi_scale, in_channels), self.device)
        self.params.init_random(seed)
        self._mesh = None
        self._graph_fb = None
        self.profile = False   # when True every enqueue is labelled for fgc_profile_collect
        self.comm = None       # exchange back end of a facet-sharded run (shard.DistComm)
        # facet-sharded runs: compute the interior tiles of a layer while its halo rows travel (FGC_NO_OVERLAP=1: the
        # whole layer after a blocking exchange, for A/B timing)
        self.overlap = os.environ.get("FGC_NO_OVERLAP", "0") != "1"
        # ... only where the interior is big enough to hide an exchange and to fill the GPU on its own: a layer of a
        # coarse level has a few hundred tiles in all, splitting it costs more than the exchange it would cover
        self.split_min_tiles = int(os.environ.get("FGC_SPLIT_MIN_TILES", "1024"))
        # ... and a layer's weight-gradient stage runs inside the NEXT layer's backward exchange window (_loss_backward_gen)
        self.dw_in_window = os.environ.get("FGC_NO_DW_IN_WINDOW", "0") != "1"
        # one launch packs the weight operands of all layers, one pair sums all parameter gradients (each small launch
        # costs ~5 us on an idle MI355X, and there were 37 of them per step); FGC_NO_BATCHED=1 = per-layer housekeeping
        self.batched = os.environ.get("FGC_NO_BATCHED", "0") != "1"
        self.step_prologue = os.environ.get("FGC_NO_STEP_PROLOGUE", "0") != "1"
        # The weight-gradient GEMMs of several layers in ONE launch per kernel form at the end of the backward pass
        # (FGC_CONV_DEFER_DW; a deferring layer keeps an `r` of its own until then).  bf16 storage: all layers - a layer's GEMM is
        # mostly ramp and tail there (100k facets: 1.095 -> 1.070 ms per step, 50k: 0.726 -> 0.705).  fp32: the GEMMs are
        # matrix-bound and grouping ALL of them loses the shared `r` whose lines stay in the Infinity Cache (1.761 -> 1.773 ms
        # at 100k facets), so a layer defers only if its `r` is small (grouped_dw_max_r_bytes, bind_mesh decides per layer):
        # on a 100k-facet mesh the level-2 layers and the up-convolutions' coarse rows, on a mesh of 25k facets or a shard of
        # a strong-scaling run every layer - where a step is its launches, not its FLOPs.  FGC_GROUPED_DW=0 / 1: none / all.
        self.grouped_dw_mode = os.environ.get("FGC_GROUPED_DW", "1" if dtype == "bf16" else "auto") if self.batched else "0"
        self.grouped_dw_max_r_bytes = int(os.environ.get("FGC_GROUPED_DW_MAX_R_MB", "40")) << 20
        self.grouped_dw = self.grouped_dw_mode == "1"       # (all layers; bind_mesh fills grouped_dw_layers)
        self.grouped_dw_layers = frozenset()
        self.save_z = os.environ.get("FGC_NO_SAVE_Z", "0") != "1"
        # rows of the backward aggregate r padded to whole 128-byte lines (include/fgc.h: FGC_CONV_R_PAD); FGC_NO_R_PAD=1: the
        # packed rows of M*cout + 24 elements
        self.r_pad = 0 if os.environ.get("FGC_NO_R_PAD", "0") == "1" else _lib.CONV_R_PAD
        # the gradient of the 4:1 max pooling behind conv1 / conv2 is a term of those layers' backward stage 1
        self.fused_pool = os.environ.get("FGC_NO_FUSED_POOL", "0") != "1"
        # the loss end of an unsharded training step (normalise, rotate the ground truth, sampled loss, both gradients) in
        # two launches instead of seven (fgc_loss_step); FGC_NO_FUSED_LOSS=1: the separate entry points
        self.fused_loss = os.environ.get("FGC_NO_FUSED_LOSS", "0") != "1"
        # the two up-convolutions on their coarse source rows (pair form, include/fgc.h); library option NO_PAIRS = 1
        # (fgc_set_option; initial value from FGC_NO_PAIRS): the fine form
        self.pairs = self._option("NO_PAIRS") != 1
        # parameter slots
        k = 0
        self.slot = {}
        for name in ["conv1", "conv2", "conv3", "dconv3"]:
            self.slot[name] = k
            k += 5
        if multi_scale:
            self.slot["head2"] = k
            k += 4
        for name in ["upconv2", "dconv2"]:
            self.slot[name] = k
            k += 5
        if multi_scale:
            self.slot["head1"] = k
            k += 4
        for name in ["upconv1", "dconv1"]:
            self.slot[name] = k
            k += 5
        self.slot["head0"] = k
        self.layers = [
            _ConvLayer("conv1", 0, "xr", None, 0, self.slot["conv1"], "h1", "p1", 1),
            _ConvLayer("conv2", 1, "p1", None, 0, self.slot["conv2"], "h2", "p2", 1),
            _ConvLayer("conv3", 2, "p2", None, 0, self.slot["conv3"], "h3", None, 1),
            _ConvLayer("dconv3", 2, "h3", None, 0, self.slot["dconv3"], "d3", None, 1),
            _ConvLayer("upconv2", 1, "d3", None, 2, self.slot["upconv2"], "u2", None, 0),
            _ConvLayer("dconv2", 1, "u2", "h2", 0, self.slot["dconv2"], "d2", None, 1),
            _ConvLayer("upconv1", 0, "d2", None, 2, self.slot["upconv1"], "u1", None, 0),
            _ConvLayer("dconv1", 0, "u1", "h1", 0, self.slot["dconv1"], "d1", None, 1),
        ]

    # ------------------------------------------------------------------------------------------
    # mesh binding: graphs + every buffer, once
    # ------------------------------------------------------------------------------------------
    def bind_mesh(self, x, adjs, gt=None, plan=None, comm=None):
        """x: [1,N0,C] / [N0,C] features (float64 numpy as in in_list, or tensor); adjs: 3 K-lists or FacetGraphs;
        gt: [1,N0,3] ground-truth normals (training only).  Casts once (float64->float32, int64->int32 as the
        TF feed does, train.py:409-427).

        plan / comm (facet sharding, shard.py): x, adjs, gt are still the WHOLE mesh; this rank keeps only its
        shard ([owned rows | halo rows] per level) and exchanges halos through `comm` between the layers."""
        dev = self.device
        bf16 = self.dtype == "bf16"
        xt = torch.as_tensor(np.asarray(x.cpu() if isinstance(x, torch.Tensor) else x, dtype=np.float32))
        xt = xt.reshape(-1, xt.shape[-1]).contiguous()
        if xt.shape[1] != self.in_channels:
            raise ValueError("expected %d input channels" % self.in_channels)
        gtt = None
        if gt is not None:
            gtt = torch.as_tensor(np.asarray(gt.cpu() if isinstance(gt, torch.Tensor) else gt, dtype=np.float32))
            gtt = gtt.reshape(-1, 3).contiguous()
        if plan is None:
            graphs = [as_graph(a, dev) for a in adjs]
            if len(graphs) != 3:
                raise ValueError("the network needs exactly 3 adjacency levels (model.py:858-931)")
            nh = [0, 0, 0]
            nhp = [0, 0]
            n_total = [g.n for g in graphs]
            own_lo = 0
        else:
            from .shard import LocalGraph
            graphs = [LocalGraph(P, dev) for P in plan.levels]
            nh = [g.n_halo for g in graphs]
            # tail rows of the coarse tensors read 4x-upsampled by the level above: one per UNIQUE parent of that level's
            # halo nodes (shard.ShardPlan)
            nhp = [graphs[0].pair.n_halo, graphs[1].pair.n_halo]
            n_total = list(plan.n_total)
            own_lo = plan.levels[0].lo
            xt = xt[torch.from_numpy(plan.local_rows(0))]
            real_flag = None
            if gtt is not None:
                # which rows of the WHOLE mesh count in the loss (train.py:1283-1292): every rank can then count the
                # real rows among a step's samples without asking the others
                real_flag = (gtt.abs().sum(1) > 1e-3).to(torch.float32)
                gtt = gtt[plan.levels[0].lo:plan.levels[0].hi]
        n0, n1, n2 = (g.n for g in graphs)
        if n0 != 4 * n1 or n1 != 4 * n2 or xt.shape[0] != n0 + nh[0]:
            raise ValueError("level sizes must be N0 = 4 N1 = 16 N2 and match x (got %d, %d, %d, x %d)" %
                             (n0, n1, n2, xt.shape[0]))
        xt = xt.contiguous().to(dev)
        f = dict(dtype=torch.float32, device=dev)
        ns = [n0, n1, n2]
        B = {"x": xt, "xr": torch.empty_like(xt)}
        # rows: owned + the halo rows a consumer gathers (d3 / d2 are read 4x-upsampled by the level above, their tail
        # rows hold the parents of THAT level's halo nodes)
        shapes = {"h1": (n0 + nh[0], 32), "p1": (n1 + nh[1], 32), "h2": (n1 + nh[1], 64), "p2": (n2 + nh[2], 64),
                  "h3": (n2 + nh[2], 128), "d3": (n2 + nhp[1], 128), "u2": (n1 + nh[1], 64), "d2": (n1 + nhp[0], 64),
                  "u1": (n0 + nh[0], 32), "d1": (n0, 32), "y0": (n0, 3), "nconv": (n0, 3)}
        # dtype "bf16": every activation that crosses a layer boundary (and its gradient) is STORED as bf16; the network
        # input, the 3-channel outputs, logit tables, per-edge d-logits and all parameters stay fp32 (include/fgc.h:
        # FGC_CONV_BF16)
        act = dict(dtype=torch.bfloat16, device=dev) if bf16 else f
        for k, sh in shapes.items():
            kw = f if k in ("y0", "nconv") else act
            B[k] = torch.zeros(*sh, **kw)
            B["g_" + k] = torch.zeros(*sh, **kw)   # gradient twin
        for lay in self.layers:
            B["ag_" + lay.name] = torch.empty(B[lay.x0].shape[0], AG_LD, **f)
        if self.multi_scale:
            for k, nk in (("1", n1), ("2", n2)):
                B["y" + k] = torch.empty(nk, 3, **f)
                B["nconv" + k] = torch.empty(nk, 3, **f)
                B["abs_part" + k] = torch.empty(self.L.fgc_mlp_num_partials(nk), **f)
                B["norm_scratch" + k] = torch.zeros(2 + self.L.fgc_norm_num_partials(nk), **f)
                if gtt is not None and plan is None:
                    B["g_y" + k] = torch.zeros(nk, 3, **f)
                    B["g_nconv" + k] = torch.zeros(nk, 3, **f)
                    B["loss" + k] = torch.zeros(2, **f)
        # shared backward scratch, sized for the largest user
        max_ds = max((ns[l.level] + nh[l.level]) * self._cout(l) for l in self.layers)
        rld = lambda l: self.L.fgc_conv_r_ld(self._cout(l), 1 if self.r_pad else 0, 1 if bf16 else 0)   # elements per row of r
        max_r = max(ns[l.level] * rld(l) for l in self.layers)
        if bf16:
            max_r = (max_r + 1) // 2           # (B["r"] is allocated in fp32 words)
        max_dl = max(max(g.nnz + getattr(g, "n_cross_in", 0) for g in graphs),
                     max((g.pair.n_pairs + g.pair.n_cross_in for g in graphs[:2] if getattr(g, "pair", None) is not None), default=0))
        B["ds"] = torch.zeros(max_ds, **f)
        B["dl"] = torch.zeros(max(max_dl, 1) * DL_LD, **f)
        B["dag"] = torch.empty(max(ns) * AG_LD, **f)
        B["r"] = torch.empty(max_r, **f)
        grouped = []
        if self.grouped_dw_mode != "0" and gt is not None:
            grouped.append(self.layers[0].name)   # (no r to keep apart: its GEMM can always wait for the grouped launch)
            # Facet-sharded: a layer's weight-gradient GEMM is what the NEXT layer's backward exchange hides behind
            # (_loss_backward_gen: dw_in_window), worth a collective's latency per layer - more than the ramp and tail a grouped
            # launch saves; only the first layer, which has no exchange, still waits for the grouped launch.  (An explicit
            # FGC_GROUPED_DW decides otherwise.)
            windowed = plan is not None and self.dw_in_window and "FGC_GROUPED_DW" not in os.environ
            for lay in ([] if windowed else self.layers[1:]):   # (the first layer has no r: its GEMM reads the saved aggregates and ds)
                cnt = ns[lay.level] * rld(lay)
                # (a layer over a 4x-upsampled tensor runs on its n / 4 coarse rows - the pair form - unless refused)
                used = cnt // 4 if (lay.shift == 2 and self.pairs) else cnt
                if self.grouped_dw_mode == "1" or used * (2 if bf16 else 4) <= self.grouped_dw_max_r_bytes:
                    B["r_" + lay.name] = torch.empty((cnt + 1) // 2 if bf16 else cnt, **f)
                    grouped.append(lay.name)
        self.grouped_dw_layers = frozenset(grouped)
        B["abs_part"] = torch.empty(self.L.fgc_mlp_num_partials(n0), **f)
        B["norm_scratch"] = torch.zeros(2 + self.L.fgc_norm_num_partials(n0), **f)
        B["loss"] = torch.zeros(2, **f)
        B["loss_gacc"] = torch.zeros(n0, 3, **f)     # scatter accumulator of the fused loss end (zero between steps)
        self._alloc_step_inputs(B, COST_SAMPLES)
        if gtt is not None:
            B["gt"] = gtt.contiguous().to(dev)
            B["gtr"] = torch.empty_like(B["gt"])
            if plan is not None:
                B["real_flag"] = real_flag.to(dev)
            elif self.multi_scale:
                # ground truth of the coarse heads (build extension, see train_step): the fine normals pooled with
                # "average ignoring zero rows" (model.py:792-814) and renormalised; rows of fake nodes only stay zero
                from . import ops
                g = B["gt"]
                for k in ("1", "2"):
                    g = ops.pool4_avg_iz(g)
                    nrm = g.norm(dim=1, keepdim=True)
                    g = torch.where(nrm > 0, g / nrm.clamp_min(1e-20), torch.zeros_like(g)).contiguous()
                    B["gt" + k] = g
                    B["gtr" + k] = torch.empty_like(g)
        # descriptors + workspace
        vals, grads = self.params.values, self.params.grads
        descs, ios, ws_f, ws_b = {}, {}, 0, 0
        pair_ios = []
        layer_flags = {}
        wsf, wsb = {}, {}    # a workspace of its own per layer: packed operands and partial sums stay put for the step
        for lay in self.layers:
            g = graphs[lay.level]
            W0, b, u, c, v = vals[lay.pidx:lay.pidx + 5]
            d = ConvDesc()
            d.n, d.nnz = g.n, g.nnz
            col = g.col_up if (plan is not None and lay.shift) else g.col
            d.rowptr, d.col = g.rowptr.data_ptr(), col.data_ptr()
            d.x0 = B[lay.x0].data_ptr()
            d.x1 = B[lay.x1].data_ptr() if lay.x1 else None
            d.c0 = B[lay.x0].shape[1]
            d.c1 = B[lay.x1].shape[1] if lay.x1 else 0
            d.shift, d.cout = lay.shift, W0.shape[1]
            d.W0, d.b, d.u, d.c, d.v = (t.data_ptr() for t in (W0, b, u, c, v))
            d.bias_mask, d.act, d.alpha = 1, lay.act, LRELU_ALPHA
            d.src_rows = B[lay.x0].shape[0]
            d.max_deg = g.max_deg
            # a narrow first layer leaves its aggregates in the forward workspace for the backward pass (training only)
            d.flags = _lib.CONV_SAVE_Z if (gt is not None and lay.name == "conv1" and self.save_z) else 0
            if bf16:
                d.flags |= _lib.CONV_BF16
            layer_flags[lay.name] = d.flags
            if self.option_overrides is not None:
                d.options, d.n_options = C.addressof(self.option_overrides), len(self.option_overrides)
            descs[lay.name] = d
            # a layer over a 4x-upsampled input (the two up-convolutions) runs on its COARSE source rows: the pair graph
            # of its level + the table of transformed coarse rows (include/fgc.h: fgc_conv_desc.pair_rowptr)
            pg = None
            use_pairs = lay.shift == 2 and self.pairs
            if use_pairs and plan is not None:
                # facet-sharded: the graph-dependent limits of the pair form on the GLOBAL pair graph - a rank-local test
                # lets ranks disagree (one shard with a coarse row of 25 in-pairs), and the two forms exchange different
                # tensors in the backward pass
                from .shard import pair_form_allowed
                use_pairs = pair_form_allowed(g.pair, W0.shape[1])
            if use_pairs:
                # (facet-sharded: the level's LOCAL pair graph, whose columns address [owned coarse rows | unique halo parents])
                pg = g.pairs() if plan is None else g.pair
                hc = torch.empty(B[lay.x0].shape[0], FGC_M * d.cout, **act)      # (bf16 storage: a bf16 table)
                d.pair_rowptr, d.pair_col, d.pair_mul = pg.prow.data_ptr(), pg.pcol.data_ptr(), pg.pmul.data_ptr()
                d.n_pairs, d.max_pair_deg, d.max_pair_in_deg = pg.n_pairs, pg.max_deg, pg.max_in_deg
                d.hc = hc.data_ptr()
                if self.L.fgc_conv_uses_pairs(C.byref(d)):
                    B["hc_" + lay.name] = hc
                elif plan is not None and not any(self._option(k) == 1 for k in ("NO_PAIRS", "NO_W8", "NO_W8FAST")):
                    # the job-wide test passed and the switches are on: a rank that still refuses would leave the others
                    # waiting for messages of the pair form
                    raise RuntimeError("layer %s: the pair form is allowed job-wide but refused on this rank (n_pairs=%d, "
                                       "max in-degree %d)" % (lay.name, pg.n_pairs, pg.max_in_deg))
                else:
                    d.pair_rowptr = d.pair_col = d.pair_mul = d.hc = None
                    d.n_pairs = d.max_pair_deg = d.max_pair_in_deg = 0
                    pg = None
            wsf[lay.name] = torch.empty(self.L.fgc_conv_workspace_bytes(C.byref(d)) + 256, dtype=torch.uint8, device=dev)
            if gt is not None:
                wsb[lay.name] = torch.empty(self.L.fgc_conv_bwd_workspace_bytes(C.byref(d)) + 256, dtype=torch.uint8,
                                            device=dev)
            trow, tcol, tedge = g.transposed()
            io = ConvBwdIO()
            io.trowptr, io.tcol, io.tedge = trow.data_ptr(), tcol.data_ptr(), tedge.data_ptr()
            io.max_in_deg = g.max_in_deg
            io.stages = 0
            io.ag, io.y, io.dy = B["ag_" + lay.name].data_ptr(), B[lay.y].data_ptr(), B["g_" + lay.y].data_ptr()
            io.ds, io.dl, io.dag, io.r = (B[k].data_ptr() for k in ("ds", "dl", "dag", "r"))
            if ("r_" + lay.name) in B:
                io.r = B["r_" + lay.name].data_ptr()
            # the stride of r belongs to the buffer: stated once here, not re-derived from every staged call's flags
            io.r_ld = rld(lay)
            gW0, gb, gu, gc, gv = grads[lay.pidx:lay.pidx + 5]
            io.dW0, io.db, io.du, io.dc, io.dv = (t.data_ptr() for t in (gW0, gb, gu, gc, gv))
            if pg is not None:
                io.tpair_rowptr, io.tpair_col, io.tpair_edge = pg.trow.data_ptr(), pg.tcol.data_ptr(), pg.tedge.data_ptr()
                if gt is not None:
                    # (one row per owned pair, then the incoming cross-shard pairs of a facet-sharded rank)
                    need = max(pg.n_pairs + getattr(pg, "n_cross_in", 0), 1) * d.cout
                    if "dt" not in B or B["dt"].numel() < need:
                        B["dt"] = torch.empty(need, **act)
                pair_ios.append(io)
            ios[lay.name] = io
        for io in pair_ios:
            io.dt = B["dt"].data_ptr() if "dt" in B else None
        # who writes each activation gradient first (write) / second (accumulate): fixed backward order
        #   g_h1: dconv1 (x1, write) then pool1 backward (accumulate);  g_h2: dconv2 (x1) then pool2 backward
        for name in ["dconv1", "upconv1", "dconv2", "upconv2", "dconv3", "conv3", "conv2"]:
            lay = next(l for l in self.layers if l.name == name)
            io = ios[name]
            io.dx0 = B["g_" + lay.x0].data_ptr()
            io.dx1 = B["g_" + lay.x1].data_ptr() if lay.x1 else None
            io.accumulate0, io.accumulate1 = 0, 0
        ios["conv1"].dx0 = None
        ios["conv1"].dx1 = None
        if self.fused_pool and gt is not None:
            ios["conv1"].pool_y, ios["conv1"].pool_dy = B["p1"].data_ptr(), B["g_p1"].data_ptr()
            ios["conv2"].pool_y, ios["conv2"].pool_dy = B["p2"].data_ptr(), B["g_p2"].data_ptr()
        if self.multi_scale and gt is not None and plan is None:
            # training the three heads: the coarse heads write their input gradients into g_d3 / g_d2 first, the
            # up-convolutions then add theirs
            ios["upconv2"].accumulate0 = 1
            ios["upconv1"].accumulate0 = 1
        if (layer_flags["conv1"] & _lib.CONV_SAVE_Z) and not self.L.fgc_conv_bwd_needs_exchange(
                C.byref(descs["conv1"]), C.byref(ios["conv1"])):
            ios["conv1"].z_saved = wsf["conv1"].data_ptr()      # (the library took the narrow path: see FGC_CONV_SAVE_Z)
        else:
            layer_flags["conv1"] = descs["conv1"].flags = layer_flags["conv1"] & ~_lib.CONV_SAVE_Z
        ws_f = max(ws_f, self.L.fgc_mlp_workspace_bytes(128, HIDDEN, 3))
        ws_b = max(ws_b, self.L.fgc_mlp_bwd_workspace_bytes(n0, 32, HIDDEN, 3))
        if self.multi_scale:
            ws_b = max(ws_b, self.L.fgc_mlp_bwd_workspace_bytes(n1, 64, HIDDEN, 3),
                       self.L.fgc_mlp_bwd_workspace_bytes(n2, 128, HIDDEN, 3))
        if bf16:
            ws_b = max(ws_b, self.L.fgc_mlp_bwd_bf16_workspace_bytes(n0, 32, HIDDEN, 3))
        B["ws"] = torch.empty(max(ws_f, ws_b) + 256, dtype=torch.uint8, device=dev)     # the MLP heads
        # the last head's forward operands have a workspace of their own: the step's first launch (fgc_conv_pack with an
        # fgc_pack_extra) packs them and the backward operands (at the start of B["ws"]) beside the conv operands
        B["ws_mlp_f"] = torch.empty(max(self.L.fgc_mlp_workspace_bytes(32, HIDDEN, 3),
                                        self.L.fgc_mlp_bf16_workspace_bytes(32, HIDDEN, 3)) + 256, dtype=torch.uint8, device=dev)
        for k, t in wsf.items():
            B["wsf_" + k] = t
        for k, t in wsb.items():
            B["wsb_" + k] = t
        names = [lay.name for lay in self.layers]
        nl = len(names)
        arrays = dict(
            descs=(C.POINTER(ConvDesc) * nl)(*[C.pointer(descs[k]) for k in names]),
            ios=(C.POINTER(ConvBwdIO) * nl)(*[C.pointer(ios[k]) for k in names]),
            wsf=(C.c_void_p * nl)(*[wsf[k].data_ptr() for k in names]),
            wsb=(C.c_void_p * nl)(*[wsb[k].data_ptr() for k in names]) if wsb else None, count=nl)
        self._mesh = dict(graphs=graphs, B=B, descs=descs, ios=ios, ns=ns, nh=nh, has_gt=gt is not None,
                          plan=plan, n_total=n_total, own_lo=own_lo, arrays=arrays, layer_flags=layer_flags)
        self.comm = comm
        self._graph_fb = None
        return self

    def _alloc_step_inputs(self, B, ns):
        """The per-step inputs - sample indices (train.py:561) and the rotation (train.py:563-565) - live in ONE device
        buffer [ns ints | 9 floats | pad]: a step refreshes them with a single device-to-device copy
        (set_step_inputs_packed)."""
        B["step_in"] = B["step_in_own"] = torch.zeros(ns + 12, dtype=torch.int32, device=self.device)
        B["loss_scratch"] = torch.zeros(self.L.fgc_loss_step_scratch_floats(ns), dtype=torch.float32, device=self.device)
        # facet-sharded steps: [0] the all-reduced sum of |y|, [4:] the all-reduced per-256-samples partial table
        B["loss_sums"] = torch.zeros(self.L.fgc_loss_shard_floats(ns), dtype=torch.float32, device=self.device)
        self._bind_step_inputs(B, B["step_in"])
        B["R"].copy_(torch.eye(3, dtype=torch.float32).reshape(9))

    @staticmethod
    def _bind_step_inputs(B, row):
        ns = row.numel() - 12
        B["step_in"] = row
        B["sample_ind"] = row[:ns]
        B["R"] = row[ns:ns + 9].view(torch.float32)

    def _own_step_inputs(self):
        """The step inputs back in the network's own buffer (set_step_inputs_packed may have let the steps read a caller's
        row in place): before anything writes them, and before a hipGraph records their address."""
        B = self._mesh["B"]
        own = B["step_in_own"]
        if B["step_in"] is not own:
            own.copy_(B["step_in"])
            self._bind_step_inputs(B, own)

    def bind_cached(self, key, x, adjs, gt=None, max_bytes=64 << 30):
        """bind_mesh with the bound state (graphs, activations, descriptors: ~7 KB per facet) kept in HBM under `key`,
        so that a training loop that alternates between meshes (train.py:556 draws one per iteration) switches
        between them without re-uploading or re-allocating anything.  States are kept while they fit in max_bytes."""
        cache = self.__dict__.setdefault("_mesh_cache", {})
        if key in cache:
            self._mesh = cache[key]
            self._graph_fb = None
            return self
        self.bind_mesh(x, adjs, gt=gt)
        used = sum(sum(t.numel() * t.element_size() for t in m["B"].values()) for m in cache.values())
        mine = sum(t.numel() * t.element_size() for t in self._mesh["B"].values())
        if used + mine <= max_bytes:
            cache[key] = self._mesh
        return self

    def _cout(self, lay):
        return self.params.spec[lay.pidx][1][1]

    # ------------------------------------------------------------------------------------------
    # enqueue helpers (no allocation, no sync).  The schedules are GENERATORS: they yield an exchange request
    # wherever a facet-sharded run must talk to its peers; an unsharded run just drains them.
    #   ("xchg", items, key)     ONE grouped exchange with every peer; items:
    #        ("rows", level, tensor, parent)   halo rows of `tensor` (parent: rows of the coarse parents)
    #        ("edges", level)                  d-logits of incoming cross-shard edges
    #      key None: blocking.  Otherwise the kernels enqueued up to the matching ("wait", key) run while it travels
    #   ("sum", tensor)          all-reduce
    #   ("call", fn)             launches whose arguments change from step to step (a rank's own loss samples): run by
    #                            an eager call, also between the hipGraphs of a captured schedule (_capture_segments)
    # Per step: 7 exchanges forward (everything a layer produces that a peer gathers goes out in one message right
    # behind it: conv1 -> {h1, p1}, conv2 -> {h2, p2}), 7 backward (a layer's s rows and cross-edge d-logits together),
    # two scalar all-reduces for normalizeTensor and its gradient, one all-reduce of gradient + loss sum: 17.
    # ------------------------------------------------------------------------------------------
    @property
    def pair_split_min_tiles(self):
        """The split threshold of a pair-form layer's backward exchange, in 32-row tiles of its COARSE rows: the same NUMBER as
        `split_min_tiles`, i.e. four times stricter in rows.  Its boundary launch - a handful of workgroups of the data kernel -
        costs that kernel's single-workgroup latency (~20 us at level 0 of a 100k-facet shard, measured: tools/
        shard_latency_probe.py), so the split pays only where a collective's latency is well above that; at the default
        threshold no pair layer of a 100k-facet shard splits (782 coarse tiles), at bench.py's tuning candidates 256 / 64 they do."""
        return self.split_min_tiles

    def _option(self, name):
        """The value a library option has for THIS network: its own override (options=...) or the process-level value."""
        if self.option_overrides is not None:
            names = _lib.option_names()
            for o in self.option_overrides:
                if names[o.index] == name:
                    return int(o.value)
        return _lib.get_option(name)

    def _st(self):
        return C.c_void_p(torch.cuda.current_stream().cuda_stream)

    # entry points that differ between the fp32 and the bf16-storage network
    @property
    def _mlp_fwd(self):
        return self.L.fgc_mlp_fwd_bf16 if self.dtype == "bf16" else self.L.fgc_mlp_fwd

    @property
    def _mlp_bwd(self):
        return self.L.fgc_mlp_bwd_bf16 if self.dtype == "bf16" else self.L.fgc_mlp_bwd

    @property
    def _pool4_bwd(self):
        return self.L.fgc_pool4_bwd_bf16 if self.dtype == "bf16" else self.L.fgc_pool4_bwd

    def _tag(self, name):
        if self.profile:
            self.L.fgc_profile_tag(name.encode())

    @property
    def sharded(self):
        return self._mesh["plan"] is not None

    def _forward_gen(self, rotate, defer_normalize=False):
        """defer_normalize: an unsharded TRAINING step leaves normalizeTensor to fgc_loss_step (which writes n_conv too)."""
        M, L, st = self._mesh, self.L, self._st()
        B, ws = M["B"], M["B"]["ws"]
        n0 = M["ns"][0]
        rows0 = B["x"].shape[0]
        vals = self.params.values
        # one housekeeping launch per step: conv operands, the rotation of the input rows, the MLP's operands
        # (FGC_NO_STEP_PROLOGUE=1: the rotation and the MLP packs as launches of their own)
        prologue = self.batched and self.step_prologue
        mlp_packed = _lib.MLP_PACKED if (prologue and not self.multi_scale) else 0
        M["mlp_packed"] = mlp_packed
        self._tag("fwd:input")
        if rotate and not prologue:
            _lib.check(L.fgc_rotate_rows(_p(B["x"]), _p(B["xr"]), rows0, self.in_channels // 3, _p(B["R"]), st),
                       "rotate")
        elif not rotate:
            B["xr"].copy_(B["x"])
        # what a layer's output feeds on OTHER ranks: sent in one grouped exchange right behind the layer; the consumer
        # waits for it between its interior and its boundary tiles
        send_after = {"conv1": [("rows", 0, "h1", False), ("rows", 1, "p1", False)],
                      "conv2": [("rows", 1, "h2", False), ("rows", 2, "p2", False)],
                      "conv3": [("rows", 2, "h3", False)], "dconv3": [("rows", 1, "d3", True)],
                      "upconv2": [("rows", 1, "u2", False)], "dconv2": [("rows", 0, "d2", True)],
                      "upconv1": [("rows", 0, "u1", False)]}
        wait_before = {"conv2": "conv1", "conv3": "conv2", "dconv3": "conv3", "upconv2": "dconv3", "dconv2": "upconv2",
                       "upconv1": "dconv2", "dconv1": "upconv1"}
        split = self.sharded and self.overlap
        # A layer's output exchange is begun under a key only if its consumer launches something before it waits (interior
        # tiles, the transform of the owned coarse rows, a multi-scale head); otherwise it is served as ONE blocking call - a
        # synchronous collective runs on the compute stream itself, without the two cross-stream dependencies of an
        # asynchronous one (20 - 25 us per call on this platform: tools/rccl_call_cost_probe.py)
        in_flight = set()

        def consumer_overlaps(nxt):
            if nxt is None or not split:
                return False
            dn = M["descs"][nxt.name]
            if L.fgc_conv_uses_pairs(C.byref(dn)):
                return True
            return M["graphs"][nxt.level].tiles["tiles_int"][1] >= self.split_min_tiles
        packed = 0
        table_done = False      # the first layer's logit table came with the housekeeping launch
        if self.batched:
            A = M["arrays"]
            self._tag("fwd:pack")
            ex = None
            if prologue:
                ex = _lib.PackExtra()
                if rotate:
                    ex.rot_x, ex.rot_y, ex.rot_R = _p(B["x"]), _p(B["xr"]), _p(B["R"])
                    ex.rot_rows, ex.rot_vecs = rows0, self.in_channels // 3
                    d1 = M["descs"]["conv1"]
                    if self.in_channels <= 6 and (d1.src_rows or d1.n) == rows0:
                        # ... and the first layer's logit table of the rotated rows in the same pass
                        s1 = self.slot["conv1"]
                        ex.rot_ag = _p(B["ag_conv1"])
                        ex.rot_u, ex.rot_c, ex.rot_v = _p(vals[s1 + 2]), _p(vals[s1 + 3]), _p(vals[s1 + 4])
                        table_done = True
                if mlp_packed:
                    s0 = self.slot["head0"]
                    ex.mlp_bf16 = 1 if self.dtype == "bf16" else 0
                    ex.mlp_W1, ex.mlp_W2 = _p(vals[s0]), _p(vals[s0 + 2])
                    ex.mlp_n, ex.mlp_cin, ex.mlp_hidden, ex.mlp_cout = n0, 32, HIDDEN, 3
                    ex.mlp_fwd_ws = _p(B["ws_mlp_f"])
                    ex.mlp_bwd_ws = _p(ws) if M["has_gt"] else None
            _lib.check(L.fgc_conv_pack(A["descs"], A["ios"], A["wsf"], A["wsb"], A["count"],
                                       C.byref(ex) if ex is not None else None, st), "pack")
            packed = _lib.CONV_PACKED
            # the operands carry the identity of their layout: a library option that moves between this launch and a layer's
            # FGC_CONV_PACKED / FGC_MLP_PACKED calls is then an error, not a product with the wrong operand
            for dd in M["descs"].values():
                dd.packed_layout = L.fgc_conv_layout_id(C.byref(dd))
            if mlp_packed:
                mlp_packed = M["mlp_packed"] = _lib.MLP_PACKED | _lib.mlp_layout(
                    L.fgc_mlp_layout_id(32, HIDDEN, 3, 1 if self.dtype == "bf16" else 0))
        for li, lay in enumerate(self.layers):
            d = M["descs"][lay.name]
            lws = B["wsf_" + lay.name]
            lflags = M["layer_flags"][lay.name]
            d.flags = packed | lflags
            args = (C.byref(d), _p(B["ag_" + lay.name]), _p(B[lay.y]), _p(B[lay.pool]) if lay.pool else None, _p(lws),
                    lws.numel(), st)
            need = wait_before.get(lay.name) if self.sharded else None
            if need is not None and need not in in_flight:
                need = None          # (its producer's exchange was a blocking call: the halo rows are there)
            if self.sharded and need and L.fgc_conv_uses_pairs(C.byref(d)):
                # pair form: the owned coarse rows are transformed while the halo parents travel; then the tail rows and
                # every block (include/fgc.h: partial forward calls)
                g = M["graphs"][lay.level]
                own_src = d.n >> 2
                self._tag("fwd:" + lay.name)
                if split and need:
                    d.tile_list, d.n_tiles = g.tiles["tiles_int"][0].data_ptr(), 0
                    d.proj_row0, d.proj_rows = 0, own_src
                    _lib.check(L.fgc_conv_fwd(*args), lay.name)
                    yield ("wait", need)
                    d.tile_list, d.n_tiles = None, 0
                    d.proj_row0, d.proj_rows = own_src, (d.src_rows - own_src) or -1
                    d.flags = _lib.CONV_PACKED | lflags
                    _lib.check(L.fgc_conv_fwd(*args), lay.name)
                    d.proj_row0, d.proj_rows, d.flags = 0, 0, packed | lflags
                else:
                    if need:
                        yield ("wait", need)
                    _lib.check(L.fgc_conv_fwd(*args), lay.name)
            elif split and need and M["graphs"][lay.level].tiles["tiles_int"][1] >= self.split_min_tiles:
                # interior tiles (they gather owned rows only) run while the halo rows travel; then the rest
                g = M["graphs"][lay.level]
                own_src = d.n >> d.shift
                self._tag("fwd:" + lay.name)
                d.tile_list, d.n_tiles = g.tiles["tiles_int"][0].data_ptr(), g.tiles["tiles_int"][1]
                d.proj_row0, d.proj_rows = 0, own_src
                _lib.check(L.fgc_conv_fwd(*args), lay.name)
                yield ("wait", need)
                d.tile_list, d.n_tiles = g.tiles["tiles_bnd"][0].data_ptr(), g.tiles["tiles_bnd"][1]
                d.proj_row0, d.proj_rows = own_src, (d.src_rows - own_src) or -1
                d.flags = _lib.CONV_PACKED | lflags
                _lib.check(L.fgc_conv_fwd(*args), lay.name)
                d.tile_list, d.n_tiles, d.proj_row0, d.proj_rows, d.flags = None, 0, 0, 0, packed | lflags
            else:
                if need:
                    yield ("wait", need)
                self._tag("fwd:" + lay.name)
                if lay.name == "conv1" and table_done:
                    d.proj_rows = -1
                _lib.check(L.fgc_conv_fwd(*args), lay.name)
                if lay.name == "conv1" and table_done:
                    d.proj_rows = 0
            if self.sharded and lay.name in send_after:
                items = [(k, lv, B[t], par) for k, lv, t, par in send_after[lay.name]]
                nxt = self.layers[li + 1] if li + 1 < len(self.layers) else None
                if consumer_overlaps(nxt) or (self.multi_scale and lay.name in ("dconv3", "dconv2")):
                    in_flight.add(lay.name)
                    yield ("xchg", items, lay.name)
                else:
                    yield ("xchg", items, None)
            if self.multi_scale and lay.name in ("dconv3", "dconv2"):
                head, out = ("head2", "y2") if lay.name == "dconv3" else ("head1", "y1")
                W1, b1, W2, b2 = vals[self.slot[head]:self.slot[head] + 4]
                xin = B[lay.y]
                nrows = M["ns"][lay.level]
                _lib.check(self._mlp_fwd(_p(xin), nrows, xin.shape[1], HIDDEN, 3, _p(W1), _p(b1), _p(W2),
                                         _p(b2), LRELU_ALPHA, _p(B[out]), _p(B["abs_part" + out[1]]), 0, _p(ws),
                                         ws.numel(), st), head)
        self._tag("fwd:mlp")
        W1, b1, W2, b2 = vals[self.slot["head0"]:self.slot["head0"] + 4]
        wsm = B["ws_mlp_f"]
        _lib.check(self._mlp_fwd(_p(B["d1"]), n0, 32, HIDDEN, 3, _p(W1), _p(b1), _p(W2), _p(b2), LRELU_ALPHA,
                                 _p(B["y0"]), _p(B["abs_part"]), mlp_packed, _p(wsm), wsm.numel(), st), "head0")
        self._tag("fwd:normalize")
        if defer_normalize and self.sharded:
            # training: the global mean |y| (utils.py:1705 takes it over the whole tensor) is all that is needed here -
            # one launch sums this rank's partials, one scalar all-reduce; the rows are normalised by fgc_loss_shard_rows
            LS = B["loss_sums"]
            _lib.check(L.fgc_loss_shard_abs_sum(_p(B["abs_part"]), B["abs_part"].numel(), _p(LS), st), "abs sum")
            yield ("sum", LS[0:1])
            return
        if defer_normalize:
            return
        if not self.sharded:
            _lib.check(L.fgc_normalize_fwd(_p(B["y0"]), n0, _p(B["abs_part"]), B["abs_part"].numel(),
                                           _p(B["nconv"]), _p(B["norm_scratch"]), st), "normalize")
        else:
            # global mean |y| (utils.py:1705 takes it over the whole tensor): one scalar all-reduce
            tot = B["abs_part"].sum().reshape(1)
            yield ("sum", tot)
            B["norm_scratch"][0:1] = tot / (3.0 * M["n_total"][0]) + 1e-5
            _lib.check(L.fgc_normalize_apply(_p(B["y0"]), n0, _p(B["norm_scratch"]), _p(B["nconv"]), st),
                       "normalize")

    def _loss_backward_gen(self, rotate):
        M, L, st = self._mesh, self.L, self._st()
        B, ws, ns = M["B"], M["B"]["ws"], M["ns"]
        n0 = ns[0]
        gt = B["gt"]
        self._tag("bwd:loss")
        fused = self._fused_loss_now()
        if rotate and not fused:
            _lib.check(L.fgc_rotate_rows(_p(B["gt"]), _p(B["gtr"]), n0, 1, _p(B["R"]), st), "rotate gt")
            gt = B["gtr"]

        def loss_rows():
            stl = self._st()
            samp = B["sample_ind_local"] if self.sharded else B["sample_ind"]
            ns_samp = samp.numel()
            if ns_samp:
                _lib.check(L.fgc_angular_loss_fwd(_p(B["nconv"]), _p(gt), _p(samp), ns_samp, _p(B["loss"]), stl), "loss")
            else:
                B["loss"].zero_()
            if self.sharded:
                # loss = sum over ranks of (sum of angles) / (real rows among ALL samples).  The count is known to every
                # rank (flags of the whole mesh were kept at bind time), so the backward pass can start at once; the sum
                # of angles rides in the tail of the gradient all-reduce at the end of the step
                ext = self.params.grad_ext
                ext[-4:-3] = torch.nan_to_num(B["loss"][0:1] * B["loss"][1:2])
                B["loss"][1:2] = B["real_flag"][B["sample_ind"].long()].sum().reshape(1)
            if ns_samp:
                _lib.check(L.fgc_angular_loss_bwd(_p(B["nconv"]), _p(gt), _p(samp), ns_samp, n0, _p(B["loss"]), 1.0,
                                                  _p(B["g_nconv"]), stl), "loss bwd")
            else:
                B["g_nconv"].zero_()

        if fused and self.sharded:
            # the same in three launches with the step's second scalar all-reduce (the partial table {sum of angles, real
            # samples, sum(d xs . y)} per 256 samples) between them; every rank leaves with the loss of the whole step
            LS = B["loss_sums"]
            ns_total = B["sample_ind"].numel()
            count = 3.0 * M["n_total"][0]

            def shard_samples():
                samp = B["sample_ind_local"]
                _lib.check(L.fgc_loss_shard_samples(_p(B["y0"]), count, _p(B["gt"]), _p(B["R"]) if rotate else None,
                                                    _p(samp) if samp.numel() else None, samp.numel(), ns_total,
                                                    _p(B["loss_gacc"]), _p(LS), self._st()), "loss samples")
            # (a rank's own samples are a list of another length - and another tensor - every step: a request of its own,
            #  served by an eager call also when the schedule is replayed from hipGraphs)
            yield ("call", shard_samples)
            yield ("sum", LS[4:])
            _lib.check(L.fgc_loss_shard_rows(_p(B["y0"]), n0, count, ns_total, _p(LS), _p(B["loss_gacc"]), _p(B["nconv"]),
                                             _p(B["g_y0"]), _p(B["loss"]), st), "loss rows")
        elif fused:
            # normalise + rotate the sampled ground-truth rows + loss + both gradients: two launches (loss_gacc is the
            # zero-on-entry / zero-on-exit scratch of fgc_loss_step: a buffer of its own, never the g_nconv that the
            # separate-launch path fills with real gradients)
            samp = B["sample_ind"]
            _lib.check(L.fgc_loss_step(_p(B["y0"]), n0, _p(B["abs_part"]), B["abs_part"].numel(), _p(B["gt"]),
                                       _p(B["R"]) if rotate else None, _p(samp), samp.numel(), _p(B["loss_gacc"]),
                                       _p(B["nconv"]), _p(B["g_y0"]), _p(B["loss"]), _p(B["loss_scratch"]), st), "loss step")
        elif self.sharded:
            # a rank's own samples are a list of another length (and another tensor) every step: these few launches are
            # a request of their own, served by an eager call also when the schedule is replayed from hipGraphs
            yield ("call", loss_rows)
        else:
            loss_rows()
        if fused:
            pass
        elif not self.sharded:
            _lib.check(L.fgc_normalize_bwd(_p(B["y0"]), _p(B["g_nconv"]), n0, _p(B["g_y0"]), _p(B["norm_scratch"]),
                                           st), "normalize bwd")
        else:
            sc = B["norm_scratch"]
            _lib.check(L.fgc_normalize_bwd_partial(_p(B["y0"]), _p(B["g_nconv"]), n0, _p(sc), _p(B["g_y0"]),
                                                   C.c_void_p(sc.data_ptr() + 8), st), "normalize bwd 1")
            tot = sc[2:].sum().reshape(1)
            yield ("sum", tot)
            sc[1:2] = -tot / (sc[0:1] * sc[0:1])
            _lib.check(L.fgc_normalize_bwd_apply(_p(B["y0"]), n0, 3.0 * M["n_total"][0], _p(sc), _p(B["g_y0"]), st),
                       "normalize bwd 2")
        vals, grads = self.params.values, self.params.grads
        if self.multi_scale:
            if self.sharded or self.dtype != "f32":
                raise NotImplementedError("training the multi-scale heads: unsharded fp32 network only")
            # Build extension: the reference trains the coarse heads through a point-set loss on the vertex update
            # (train.py:1075-1105; SURVEY.md section 2: out of scope).  Here every head gets the angular loss of the fine
            # head (train.py:1272-1294) against the pooled ground truth, on the same sampled rows modulo the level's size.
            for k, name, head in (("2", "dconv3", "head2"), ("1", "dconv2", "head1")):
                lvl = int(k)
                nk = ns[lvl]
                y, nc, part, sc = B["y" + k], B["nconv" + k], B["abs_part" + k], B["norm_scratch" + k]
                self._tag("bwd:head" + k)
                _lib.check(L.fgc_normalize_fwd(_p(y), nk, _p(part), part.numel(), _p(nc), _p(sc), st), "normalize")
                gtk = B["gt" + k]
                if rotate:
                    _lib.check(L.fgc_rotate_rows(_p(gtk), _p(B["gtr" + k]), nk, 1, _p(B["R"]), st), "rotate gt")
                    gtk = B["gtr" + k]
                sk = torch.remainder(B["sample_ind"], nk)
                B["sample_ind" + k] = sk        # (kept alive until the kernels have run)
                _lib.check(L.fgc_angular_loss_fwd(_p(nc), _p(gtk), _p(sk), sk.numel(), _p(B["loss" + k]), st), "loss")
                _lib.check(L.fgc_angular_loss_bwd(_p(nc), _p(gtk), _p(sk), sk.numel(), nk, _p(B["loss" + k]), 1.0,
                                                  _p(B["g_nconv" + k]), st), "loss bwd")
                _lib.check(L.fgc_normalize_bwd(_p(y), _p(B["g_nconv" + k]), nk, _p(B["g_y" + k]), _p(sc), st),
                           "normalize bwd")
                lay = next(l for l in self.layers if l.name == name)
                xin = B[lay.y]
                hs = self.slot[head]
                _lib.check(L.fgc_mlp_bwd(_p(xin), _p(B["g_y" + k]), nk, xin.shape[1], HIDDEN, 3, _p(vals[hs]),
                                         _p(vals[hs + 1]), _p(vals[hs + 2]), LRELU_ALPHA, _p(B["g_" + lay.y]),
                                         _p(grads[hs]), _p(grads[hs + 1]), _p(grads[hs + 2]), _p(grads[hs + 3]), 0, _p(ws),
                                         ws.numel(), st), head + " bwd")
        s = self.slot["head0"]
        self._tag("bwd:mlp")
        _lib.check(self._mlp_bwd(_p(B["d1"]), _p(B["g_y0"]), n0, 32, HIDDEN, 3, _p(vals[s]), _p(vals[s + 1]),
                                 _p(vals[s + 2]), LRELU_ALPHA, _p(B["g_d1"]), _p(grads[s]), _p(grads[s + 1]),
                                 _p(grads[s + 2]), _p(grads[s + 3]), M.get("mlp_packed", 0), _p(ws), ws.numel(), st),
                   "head0 bwd")
        # (the stride of r - padded to whole 128-byte lines unless FGC_NO_R_PAD=1 - is stated in every io's r_ld at bind time:
        #  the staged calls below may reset io.flags freely)
        # Facet-sharded: a layer's weight-gradient stage (8: the GEMM over its r rows and inputs, the partial sums) depends on
        # nothing that follows it, and r stays intact until the NEXT layer's data kernel writes it.  So it is launched inside
        # the next layer's exchange window - between the begin of that layer's exchange (s + d-logit rows, or dt + d-logit
        # rows) and its wait - where it hides 20 - 55 us of a collective's latency that nothing else in a strictly sequential
        # backward pass can hide (tools/shard_latency_probe.py; FGC_NO_DW_IN_WINDOW=1: right behind its own data kernel).
        # Same launches, same arithmetic, another order.
        pending_dw = [None]

        def flush_dw():
            if pending_dw[0] is not None:
                pending_dw[0]()
                pending_dw[0] = None

        def defer_dw(name, d, io, lws, flags):
            def run():
                self._tag("bwd:" + name)
                io.stages, io.flags = 8, flags
                io.data_tile_list, io.n_data_tiles = None, 0
                _lib.check(L.fgc_conv_bwd(C.byref(d), C.byref(io), _p(lws), lws.numel(), st), name + " bwd/weights")
                io.flags = 0
            pending_dw[0] = run
            if not self.dw_in_window:
                flush_dw()

        for name in ["dconv1", "upconv1", "dconv2", "upconv2", "dconv3", "conv3", "conv2", "conv1"]:
            self._tag("bwd:" + name)
            # (g_h2 += d pool2 and g_h1 += d pool1 are folded into stage 1 of conv2 / conv1: fgc_conv_bwd_io.pool_y / pool_dy;
            #  FGC_NO_FUSED_POOL=1: by launches of their own)
            if name == "conv2" and not self.fused_pool:
                _lib.check(self._pool4_bwd(_p(B["h2"]), _p(B["p2"]), _p(B["g_p2"]), _p(B["g_h2"]), ns[2], 64, 1, st),
                           "pool2 bwd")
            if name == "conv1" and not self.fused_pool:
                _lib.check(self._pool4_bwd(_p(B["h1"]), _p(B["p1"]), _p(B["g_p1"]), _p(B["g_h1"]), ns[1], 32, 1, st),
                           "pool1 bwd")
            d, io = M["descs"][name], M["ios"][name]
            lws = B["wsb_" + name]
            base = (_lib.CONV_PACKED | _lib.CONV_DEFER_REDUCE) if self.batched else 0
            if self.batched and name in self.grouped_dw_layers:
                base |= _lib.CONV_DEFER_DW
            if not self.sharded:
                io.stages, io.flags = 0, base
                _lib.check(L.fgc_conv_bwd(C.byref(d), C.byref(io), _p(lws), lws.numel(), st), name + " bwd")
                continue
            lay = next(l for l in self.layers if l.name == name)
            cout = d.cout
            nloc = ns[lay.level] + M["nh"][lay.level]
            g = M["graphs"][lay.level]
            call = lambda what: _lib.check(L.fgc_conv_bwd(C.byref(d), C.byref(io), _p(lws), lws.numel(), st),
                                           name + " bwd/" + what)
            io.flags = base
            if not L.fgc_conv_bwd_needs_exchange(C.byref(d), C.byref(io)):
                # first layer over a narrow input: its parameter gradients are sums over owned nodes, nothing to
                # exchange (the flat-gradient all-reduce adds the ranks)
                flush_dw()
                self._tag("bwd:" + name)
                io.stages = 1 | 2 | 8
                call("params")
                continue
            if L.fgc_conv_uses_pairs(C.byref(d)):
                # pair form: dt and the d-logits of the owned pairs; the rows of the pairs whose parent a peer owns travel to it
                # (ONE grouped exchange, no rows of s), then the data kernel over the owned coarse rows and the weight gradients
                pg = g.pair
                npl = pg.n_pairs + pg.n_cross_in
                io.stages = 1 | 2
                call("pair logits")
                dtb = B["dt"][:npl * cout].view(npl, cout)
                dlb = B["dl"][:npl * DL_LD].view(npl, DL_LD)
                items = [("pedges", lay.level, dtb), ("pedges", lay.level, dlb)]
                if self.overlap and pg.tiles["ttiles_int"][1] >= self.pair_split_min_tiles:
                    # ... it travels under the data kernel of the coarse-row tiles whose in-pairs all have owned parents; the
                    # tiles with an incoming cross-shard pair and the weight gradients follow (round 6: the pair layers'
                    # backward exchange used to block - tools/shard_latency_probe.py)
                    yield ("xchg", items, "bwd")
                    flush_dw()
                    self._tag("bwd:" + name)
                    io.stages, io.flags = 4, base | _lib.CONV_PACKED
                    io.data_tile_list, io.n_data_tiles = pg.tiles["ttiles_int"][0].data_ptr(), pg.tiles["ttiles_int"][1]
                    call("pair data/interior")
                    yield ("wait", "bwd")
                    io.data_tile_list, io.n_data_tiles = pg.tiles["ttiles_bnd"][0].data_ptr(), pg.tiles["ttiles_bnd"][1]
                    call("pair data/boundary")
                    io.data_tile_list, io.n_data_tiles, io.flags = None, 0, 0
                    defer_dw(name, d, io, lws, base | _lib.CONV_PACKED)
                    continue
                # (nothing to put under it - no weight-gradient stage pending -: ONE blocking call, a synchronous collective on
                #  the compute stream, without the cross-stream dependencies of an asynchronous one)
                key = "bwd" if pending_dw[0] is not None else None
                yield ("xchg", items, key)
                flush_dw()
                if key:
                    yield ("wait", key)
                self._tag("bwd:" + name)
                io.stages, io.flags = 4, base | _lib.CONV_PACKED
                call("pair data")
                io.flags = 0
                defer_dw(name, d, io, lws, base | _lib.CONV_PACKED)
                continue
            # s = dy * lrelu'(y) / deg on owned rows and the d-logits of owned edges, in one call (the deep d-logits kernel
            # computes s in its prologue; packs the operands of stages 2 and 4 when the network did not)
            io.stages = 1 | 2
            call("logits")
            # ONE grouped exchange per layer: the halo rows of s (their owners') and the d-logits of incoming cross-shard
            # edges, both first read by the data kernel
            ds_rows = (B["ds"].view(torch.bfloat16) if self.dtype == "bf16" else B["ds"])[:nloc * cout].view(nloc, cout)
            items = [("rows", lay.level, ds_rows, False), ("edges", lay.level)]
            if self.overlap and g.tiles["ttiles_int"][1] >= self.split_min_tiles:
                # ... it travels under the data kernel of the interior tiles (all in-edges from owned rows); boundary
                # tiles and the weight gradients follow
                yield ("xchg", items, "bwd")
                flush_dw()
                self._tag("bwd:" + name)
                io.stages, io.flags = 4, base | _lib.CONV_PACKED
                io.data_tile_list, io.n_data_tiles = g.tiles["ttiles_int"][0].data_ptr(), g.tiles["ttiles_int"][1]
                call("data/interior")
                yield ("wait", "bwd")
                io.data_tile_list, io.n_data_tiles = g.tiles["ttiles_bnd"][0].data_ptr(), g.tiles["ttiles_bnd"][1]
                call("data/boundary")
                io.data_tile_list, io.n_data_tiles, io.flags = None, 0, 0
            else:
                key = "bwd" if pending_dw[0] is not None else None
                yield ("xchg", items, key)
                flush_dw()
                if key:
                    yield ("wait", key)
                self._tag("bwd:" + name)
                io.stages, io.flags = 4, base | _lib.CONV_PACKED
                call("data")
                io.flags = 0
            defer_dw(name, d, io, lws, base | _lib.CONV_PACKED)
        flush_dw()
        if self.batched:
            A = M["arrays"]
            for lname in self.grouped_dw_layers:      # (the staged calls of a sharded step leave other flags behind)
                M["ios"][lname].flags = _lib.CONV_PACKED | _lib.CONV_DEFER_REDUCE | _lib.CONV_DEFER_DW
            self._tag("bwd:reduce")
            _lib.check(L.fgc_conv_bwd_reduce(A["descs"], A["ios"], A["wsb"], A["count"], st), "reduce")
        if self.sharded:
            # every rank summed its own facets: one flat all-reduce (gradient + the loss sum in the tail)
            yield ("sum", self.params.grad_ext)
            if not fused:
                B["loss"][0:1] = self.params.grad_ext[-4:-3] / B["loss"][1:2]

    # ---- exchange items -> (send buffer, send counts, receive view, receive counts) ------------------------
    def _block(self, item):
        """One block of a grouped exchange (shard.PackedExchange): (src rows, send index, send counts, halo tail, recv
        counts)."""
        M = self._mesh
        g = M["graphs"][item[1]]
        if item[0] == "rows":
            t, parent = item[2], item[3]
            if t.dtype == torch.bfloat16:
                # a row of C bf16 channels travels as C / 2 dwords (every exchanged width is even): the copies and the
                # collective only move bytes
                t = t.view(torch.float32)
            if parent:
                # a coarse tensor read 4x-upsampled by this level: its tail holds the unique parents of the level's halo nodes
                pg = g.pair
                return (t, pg.send_rows, pg.send_counts, t[t.shape[0] - pg.n_halo:], pg.recv_counts)
            tail = t.shape[0] - g.n_halo
            return (t, g.send_rows, g.send_counts, t[tail:], g.recv_counts)
        if item[0] == "pedges":
            # per-pair rows (dt, d-logits) of the pairs whose parent another rank owns: they land behind the owned pairs
            t, pg = item[2], g.pair
            if t.dtype == torch.bfloat16:
                t = t.view(torch.float32)
            return (t, pg.send_edges, pg.cross_send_counts, t[pg.n_pairs:], pg.cross_recv_counts)
        if item[0] == "edges":
            dl = M["B"]["dl"][:(g.nnz + g.n_cross_in) * DL_LD].view(-1, DL_LD)
            return (dl, g.send_edges, g.cross_send_counts, dl[g.nnz:], g.cross_recv_counts)
        raise ValueError(item[0])

    def _packed(self, req):
        """The PackedExchange of an ("xchg", items, key) request; built at its first use, the same every step."""
        from .shard import PackedExchange
        cache = self._mesh.setdefault("packed", {})
        key = tuple((it[0], it[1]) + ((it[2].data_ptr(), tuple(it[2].shape), bool(it[3])) if it[0] == "rows" else
                                      ((it[2].data_ptr(), tuple(it[2].shape)) if it[0] == "pedges" else ()))
                    for it in req[1])
        px = cache.get(key)
        if px is None:
            world = len(self._mesh["graphs"][0].send_counts)
            px = cache[key] = PackedExchange([self._block(it) for it in req[1]], world)
        return px

    def _serve(self, req, pending):
        """One request of a schedule through self.comm (exchange now / begin / await, or an all-reduce)."""
        if req[0] == "wait":
            self.comm.finish(pending.pop(req[1]))
        elif req[0] == "call":
            req[1]()
        elif req[0] == "sum":
            self.comm.all_reduce_sum(req[1])
        else:
            px = self._packed(req)
            if req[2] is None:
                self.comm.exchange(px)
            else:
                pending[req[2]] = self.comm.exchange_begin(px)

    def _drain(self, gen):
        """Run a schedule on this rank: no-op exchanges when unsharded, collectives through self.comm otherwise."""
        pending = {}
        for req in gen:
            self._serve(req, pending)

    def _capture_segments(self, make_gen):
        """A facet-sharded schedule as a list of (hipGraph, request): the launches between two exchanges are captured
        into one graph each; the exchanges themselves (row gather + grouped point-to-point, all-reduces) stay eager
        calls between the replays - a collective is not a graph node here.  The schedule is static (same buffers, same
        tile lists, same message sizes every step), so the requests recorded with the graphs are replayed as they are."""
        # (one collection in front of the first capture, the collector off until the last one has ended: a full collection
        #  per segment - thirty a step and shard - made the capture of an 8-shard step take seconds)
        with _no_gc_while_capturing():
            gen = make_gen()
            segs = []
            self.segment_nodes = getattr(self, "segment_nodes", [])
            while True:
                g = torch.cuda.CUDAGraph(keep_graph=True)
                # (thread_local: the collective back end's watchdog thread may query its events while this thread captures)
                # Whether a stretch has launches is only known once it has been captured - library launches and torch
                # operations alike become nodes, and nothing else sees both - so a stretch of two requests back to back IS
                # captured; torch's "The CUDA Graph is empty" warning about it is expected here and silenced for exactly
                # these captures (the graph is dropped below: never instantiated, never replayed).
                with warnings.catch_warnings():
                    warnings.filterwarnings("ignore", message="The CUDA Graph is empty")
                    with torch.cuda.graph(g, capture_error_mode="thread_local"):
                        try:
                            req = next(gen)
                        except StopIteration:
                            req = None
                # a stretch WITHOUT launches (two requests back to back, or nothing behind the last one) is no graph at all:
                # an empty hipGraph is never instantiated or replayed (segment_nodes keeps the counts for the tests)
                nodes = _graph_node_count(g)
                if nodes == 0:
                    g = None
                else:
                    g.instantiate()
                self.segment_nodes.append(nodes)
                segs.append((g, req))
                if req is None:
                    return segs

    def _replay_segments(self, segs):
        pending = {}
        for g, req in segs:
            if g is not None:
                g.replay()
            if req is not None:
                self._serve(req, pending)

    def _fused_loss_now(self):
        return self.fused_loss

    def _enqueue_forward(self, rotate, training=False):
        self._drain(self._forward_gen(rotate, defer_normalize=training and self._fused_loss_now()))

    def _enqueue_loss_backward(self, rotate):
        self._drain(self._loss_backward_gen(rotate))

    # ------------------------------------------------------------------------------------------
    # public API
    # ------------------------------------------------------------------------------------------
    def _upload(self, dst, host_array):
        """Stream-ordered host -> device update of a small per-step input.  The source is a fresh PINNED tensor and
        the copy is non_blocking: it is a DMA enqueued on the current stream (so it cannot overtake kernels of the
        previous step that still read `dst`), and torch's caching host allocator keeps the pinned block alive until
        the copy has run.  A plain pageable copy is not ordered against in-flight work on ROCm: with several steps
        queued it changed the inputs of steps that had not executed yet."""
        t = torch.from_numpy(np.ascontiguousarray(host_array)).pin_memory()
        dst.copy_(t, non_blocking=True)

    def set_rotation(self, R):
        """R [3,3] (train.py:563-565); identity = no augmentation."""
        self._own_step_inputs()
        self._upload(self._mesh["B"]["R"], np.asarray(R, dtype=np.float32).reshape(9))

    def set_samples(self, sample_ind):
        """Indices of the rows the loss is evaluated on (train.py:561), any of the N0 padded rows."""
        t = np.asarray(sample_ind).astype(np.int32)
        self._own_step_inputs()
        B = self._mesh["B"]
        if t.size != B["sample_ind"].numel():
            R = B["R"].clone()
            self._alloc_step_inputs(B, t.size)
            B["R"].copy_(R)
            self._graph_fb = None
        self._upload(B["sample_ind"], t)
        if self.sharded:
            lo = self._mesh["own_lo"]
            loc = t[(t >= lo) & (t < lo + self._mesh["ns"][0])] - lo
            buf = torch.empty(loc.size, dtype=torch.int32, device=self.device)
            if loc.size:
                self._upload(buf, loc.astype(np.int32))
            B["sample_ind_local"] = buf

    def set_step_inputs_device(self, sample_ind_dev, R_dev, sample_local_dev=None):
        """Per-step inputs that are ALREADY on the device (e.g. a window of steps uploaded in one go): device-to-device
        copies are kernels on the compute queue, ordered with hipGraph replays; host -> device DMAs between replays of a
        captured graph were observed to race on this stack when many steps are queued.  sample_local_dev: a facet-sharded
        rank's own part of the samples (local_samples_device), used in place as this step's list."""
        self._own_step_inputs()
        B = self._mesh["B"]
        B["sample_ind"].copy_(sample_ind_dev)
        B["R"].copy_(R_dev.reshape(9))
        if self.sharded:
            if sample_local_dev is None:
                raise ValueError("a sharded network needs the rank's local sample list (local_samples_device)")
            B["sample_ind_local"] = sample_local_dev

    @staticmethod
    def pack_step_inputs(sample_inds, rotations, device):
        """[steps, ns + 12] int32 on the device: row k = the samples and the rotation (bit pattern of 9 floats) of step k,
        in the layout of the network's step-input buffer."""
        S = np.stack([np.asarray(s).astype(np.int32) for s in sample_inds])
        R = np.stack([np.asarray(r, dtype=np.float32).reshape(9) for r in rotations]).view(np.int32)
        out = np.zeros((S.shape[0], S.shape[1] + 12), dtype=np.int32)
        out[:, :S.shape[1]] = S
        out[:, S.shape[1]:S.shape[1] + 9] = R
        return torch.from_numpy(out).to(device)

    def set_step_inputs_packed(self, packed_row, sample_local_dev=None, in_place=False):
        """Samples and rotation of the next step(s) = a row of pack_step_inputs, COPIED into the network's own buffer (one
        device-to-device copy; the caller may reuse the row at once).  in_place=True: with eager launches the step reads the
        row where it is - no copy launch, and the caller must keep the row alive and unchanged until every queued step that
        uses it has run (bench.py, whose rows are a window bound once); a network whose step is held by a hipGraph copies
        anyway, because the graph holds the address of the own buffer."""
        B = self._mesh["B"]
        if (in_place and self._graph_fb is None and packed_row.dtype == torch.int32 and packed_row.is_contiguous()
                and packed_row.numel() == B["step_in_own"].numel() and packed_row.device == B["step_in_own"].device
                and packed_row.data_ptr() % 4 == 0):
            self._bind_step_inputs(B, packed_row)
        else:
            if B["step_in"] is not B["step_in_own"]:
                self._bind_step_inputs(B, B["step_in_own"])
            B["step_in_own"].copy_(packed_row)
        if self.sharded:
            if sample_local_dev is None:
                raise ValueError("a sharded network needs the rank's local sample list (local_samples_device)")
            B["sample_ind_local"] = sample_local_dev

    def local_samples_device(self, sample_ind):
        """The samples (row ids of the WHOLE mesh, train.py:561) that fall into this rank's owned rows, as local row
        ids on the device; the unsharded network keeps them all."""
        t = np.asarray(sample_ind).astype(np.int32)
        if self.sharded:
            lo = self._mesh["own_lo"]
            t = t[(t >= lo) & (t < lo + self._mesh["ns"][0])] - lo
        return torch.from_numpy(np.ascontiguousarray(t.astype(np.int32))).to(self.device)

    def forward(self, rotate=False):
        """Normalised normals of the bound mesh: [N0,3] (padded, permuted order). Un-normalised output in buffers['y0']."""
        self._enqueue_forward(rotate)
        return self._mesh["B"]["nconv"]

    def _forward_ms_gen(self, rotate):
        """The multi-scale forward as a schedule: the network, then normalizeTensor on the two coarse heads (on a
        facet-sharded run one scalar all-reduce each: utils.py:1705 takes the mean over the whole tensor)."""
        yield from self._forward_gen(rotate)
        M, L, st = self._mesh, self.L, self._st()
        B = M["B"]
        for k, level in (("1", 1), ("2", 2)):
            nk = M["ns"][level]
            y, out, part, sc = B["y" + k], B["nconv" + k], B["abs_part" + k], B["norm_scratch" + k]
            if not self.sharded:
                _lib.check(L.fgc_normalize_fwd(_p(y), nk, _p(part), part.numel(), _p(out), _p(sc), st), "normalize")
            else:
                tot = part.sum().reshape(1)
                yield ("sum", tot)
                sc[0:1] = tot / (3.0 * M["n_total"][level]) + 1e-5
                _lib.check(L.fgc_normalize_apply(_p(y), nk, _p(sc), _p(out), st), "normalize")

    def forward_multi_scale(self, rotate=False):
        """The multi-scale denoising forward of inferNet (train.py:188-193): the three heads, each through
        normalizeTensor.  Returns (n_conv0 [N0,3], n_conv1 [N0/4,3], n_conv2 [N0/16,3]) in node order."""
        if not self.multi_scale:
            raise RuntimeError("the network was built without the multi-scale heads (multi_scale=True)")
        self._drain(self._forward_ms_gen(rotate))
        B = self._mesh["B"]
        return B["nconv"], B["nconv1"], B["nconv2"]

    def measure_exchanges(self, step_fn, steps=3):
        """Facet-sharded runs: what a step's exchanges cost when nothing overlaps them.  Runs `steps` steps with
        every collective made blocking and bracketed by device synchronises (host clock); returns the collectives per
        step, their summed time and bytes sent by this rank.  Diagnostic only: never inside a timed region."""
        import time
        comm = self.comm
        stats = {"n": 0, "s": 0.0, "bytes": 0, "each": []}

        class Timed:
            world, rank, host_staged = comm.world, comm.rank, comm.host_staged

            def _t(self, fn, nbytes, kind="all_to_all"):
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                fn()
                torch.cuda.synchronize()
                dt = time.perf_counter() - t0
                stats["s"] += dt
                stats["n"] += 1
                stats["bytes"] += nbytes
                stats["each"].append((kind, nbytes, dt))

            def exchange(self, px):
                self._t(lambda: comm.exchange(px), px.send_buf.numel() * 4)

            def exchange_begin(self, px):
                self.exchange(px)
                return None

            def finish(self, h):
                pass

            def all_reduce_sum(self, t):
                self._t(lambda: comm.all_reduce_sum(t), t.numel() * t.element_size(), "all_reduce")

        self.comm = Timed()
        try:
            for _ in range(steps):
                step_fn()
            torch.cuda.synchronize()
        finally:
            self.comm = comm
        per = stats["n"] // steps
        # the step's collectives in schedule order (pack + collective + unpack, blocking), averaged over the steps
        each = [{"kind": stats["each"][i][0], "bytes_sent": stats["each"][i][1],
                 "ms": round(sum(stats["each"][i + k * per][2] for k in range(steps)) / steps * 1e3, 4)}
                for i in range(per)] if per * steps == stats["n"] else None
        return {"collectives_per_step": per, "blocking_ms_per_step": stats["s"] / steps * 1e3,
                "bytes_sent_per_step": stats["bytes"] // steps, "per_collective": each}

    def forward_backward(self, rotate=True, capture=False):
        """One forward + backward (train.py:492-520 without the optimiser); loss in buffers['loss'][0]."""
        if not self._mesh["has_gt"]:
            raise RuntimeError("bind_mesh(..., gt=...) is required for training")
        if capture:
            from . import require_graph_replay_safe
            require_graph_replay_safe()
            if self.sharded:
                # one hipGraph per stretch of launches between two exchanges (17 collectives -> ~25 graphs per step
                # instead of ~110 launches); the first call runs one step eagerly (lazy one-time set-up inside the
                # library must not happen under capture), every later one replays
                if self._graph_fb is None:
                    self._own_step_inputs()
                    self._enqueue_forward(rotate, training=True)
                    self._enqueue_loss_backward(rotate)
                    torch.cuda.synchronize()
                    self._graph_fb = ((self._capture_segments(lambda: self._forward_gen(rotate, self._fused_loss_now())),
                                       self._capture_segments(lambda: self._loss_backward_gen(rotate))), rotate)
                    return self._mesh["B"]["loss"]
                if self._graph_fb[1] != rotate:
                    raise RuntimeError("the captured schedule was recorded with rotate=%s" % self._graph_fb[1])
                for segs in self._graph_fb[0]:
                    self._replay_segments(segs)
                return self._mesh["B"]["loss"]
            if self._graph_fb is None:
                # warm up on a side stream, then capture the whole enqueue sequence into one hipGraph
                self._own_step_inputs()
                s = torch.cuda.Stream()
                s.wait_stream(torch.cuda.current_stream())
                with torch.cuda.stream(s):
                    self._enqueue_forward(rotate, training=True)
                    self._enqueue_loss_backward(rotate)
                torch.cuda.current_stream().wait_stream(s)
                g = torch.cuda.CUDAGraph()
                with _no_gc_while_capturing(), torch.cuda.graph(g):
                    self._enqueue_forward(rotate, training=True)
                    self._enqueue_loss_backward(rotate)
                self._graph_fb = (g, rotate)
            self._graph_fb[0].replay()
        else:
            self._enqueue_forward(rotate, training=True)
            self._enqueue_loss_backward(rotate)
        return self._mesh["B"]["loss"]

    def eval_loss(self, rotate=True):
        """Forward + sampled angular loss, no gradients: the validation pass of trainNet (train.py:588-617 runs
        customLoss alone on the validation feed).  Uses the bound rotation and samples; loss in buffers['loss'][0]."""
        if not self._mesh["has_gt"]:
            raise RuntimeError("bind_mesh(..., gt=...) is required for a loss")
        if self.sharded:
            raise NotImplementedError("validation runs on an unsharded network")
        self._enqueue_forward(rotate)
        B, L, st = self._mesh["B"], self.L, self._st()
        gt = B["gt"]
        if rotate:
            _lib.check(L.fgc_rotate_rows(_p(B["gt"]), _p(B["gtr"]), self._mesh["ns"][0], 1, _p(B["R"]), st), "rotate gt")
            gt = B["gtr"]
        samp = B["sample_ind"]
        _lib.check(L.fgc_angular_loss_fwd(_p(B["nconv"]), _p(gt), _p(samp), samp.numel(), _p(B["loss"]), st), "loss")
        return B["loss"]

    def adam_step(self, lr=1e-3, b1=0.9, b2=0.999, eps=1e-8):
        P = self.params
        P.step += 1
        self._tag("adam")
        _lib.check(self.L.fgc_adam_step(_p(P.theta), _p(P.grad), _p(P.m), _p(P.v), P.total, P.step, lr, b1, b2, eps,
                                        self._st()), "adam")

    def train_step(self, sample_ind=None, R=None, capture=False):
        """One iteration of trainNet's loop body (train.py:558-575,619): returns the loss tensor (device, [2])."""
        n0 = self._mesh["ns"][0]
        if sample_ind is None:
            sample_ind = np.random.randint(n0, size=COST_SAMPLES)
        if R is None:
            from .utils import rand_rotation_matrix
            R = rand_rotation_matrix()
        self.set_samples(sample_ind)
        self.set_rotation(R)
        loss = self.forward_backward(rotate=True, capture=capture)
        self.adam_step()
        return loss

    def infer_normals(self, permutations, num_faces):
        """Denoised unit normals in the ORIGINAL face order, [F,3] (train.py:115-121,136)."""
        n_conv = self.forward(rotate=False)
        perm = torch.as_tensor(np.asarray(permutations).astype(np.int32)).to(self.device)
        out = torch.empty(num_faces, 3, dtype=torch.float32, device=self.device)
        _lib.check(self.L.fgc_infer_epilogue(_p(n_conv), _p(perm), int(num_faces), _p(out), self._st()), "epilogue")
        return out

    @property
    def buffers(self):
        return self._mesh["B"]

    # ------------------------------------------------------------------------------------------
    # per-kernel timing through the library's hipEvent hooks
    # ------------------------------------------------------------------------------------------
    def profile_start(self):
        self.profile = True
        self.L.fgc_profile_enable(1)

    def profile_stop(self):
        """Returns {"tag/kernel": (launches, total_ms)}."""
        buf = C.create_string_buffer(1 << 16)
        n = self.L.fgc_profile_collect(buf, len(buf))
        self.L.fgc_profile_enable(0)
        self.profile = False
        out = {}
        if n > 0:
            for line in buf.value.decode().splitlines():
                name, cnt, ms = line.rsplit(" ", 2)
                out[name] = (int(cnt), float(ms))
        return out

    def layer_dims(self):
        """[(layer, n, nnz, cin, cout)] of the bound mesh, in forward order (for the roofline accounting)."""
        M = self._mesh
        out = []
        for lay in self.layers:
            g = M["graphs"][lay.level]
            d = M["descs"][lay.name]
            out.append((lay.name, g.n, g.nnz, d.c0 + d.c1, d.cout))
        return out

    def pair_dims(self):
        """{layer: (coarse rows, pairs)} of the layers that run in the pair form (include/fgc.h: fgc_conv_uses_pairs)."""
        M = self._mesh
        return {name: (d.n >> 2, d.n_pairs) for name, d in M["descs"].items() if self.L.fgc_conv_uses_pairs(C.byref(d))}
