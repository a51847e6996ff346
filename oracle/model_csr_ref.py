"""ORACLE (test infrastructure, never shipped, never measured as the product).

Second, memory-lean restatement of the reference's graph-conv network: the CLOSED FORM of SURVEY.md Appendix A
(aggregate-first), float64, torch CPU with autograd.  Where oracle/model_ref.py keeps the reference's op sequence - and
with it the [n, 23, 9 cout] patch tensor of model.py:470,482-488, whose autograd tape does not fit in host memory beyond
~40k facets - this form gathers `cin`-wide rows into a degree-padded [n, Dmax, cin] tensor (Dmax = the graph's largest
degree, 13-15 on a regular mesh) and contracts it with the soft assignment by one batched matmul:

    a = x u^T + c,  g = x v^T                               (model.py:79-80)
    q_ik = softmax_m(a_i + g_j(i,k)),  0 on empty slots     (model.py:74-95; empty slots multiply the zero row there)
    z_i[m, :] = sum_k q_ikm x_j(i,k)                        (= sum_k q_ikm W_m x_j with W pulled out of the sum)
    y_i = (1/deg_i) sum_m W0[m] z_i[m, :] + b [deg_i > 0]   (model.py:463-500)

O(n Dmax (cin + 9)) memory per layer instead of O(n 23 9 cout): forward + backward of the whole net at 100 000 and
200 000 facets fit.  It is the gradient oracle at the benchmark's own size (tests/test_gpu_scale.py).

Pinned by tests/test_oracle_golden.py: against the reference fixtures (conv_* in float64, net_*) and against
oracle/model_ref.py on a 39 200-facet mesh.  Only tests/ may import this module.
"""
import numpy as np
import torch

from . import model_ref as R

DT = torch.float64


def pad_klist(adj):
    """K-list int [1, n, K] / [n, K] (one-indexed, 0 = empty; utils.py:243-295, utils.py:1799-1827) -> (idx [n, D] long,
    zero-based, empty slots = n (the zero row), mask [n, D] float64, deg [n] long).  Slot order is kept; D = max degree."""
    a = np.asarray(adj)
    a = a[0] if a.ndim == 3 else a
    n = a.shape[0]
    nz = a != 0
    deg = nz.sum(1)
    D = max(int(deg.max()) if n else 0, 1)
    order = np.argsort(~nz, axis=1, kind="stable")[:, :D]          # non-empty slots first, in slot order
    idx = np.take_along_axis(a, order, axis=1).astype(np.int64) - 1
    live = np.take_along_axis(nz, order, axis=1)
    idx[~live] = n
    return torch.from_numpy(idx), torch.from_numpy(live.astype(np.float64)), torch.from_numpy(deg.astype(np.int64))


def custom_conv2d(x, graph, params, biasMask=True):
    """ref: model.py:427-504 in the aggregate-first closed form.  x [n, cin] float64."""
    idx, mask, deg = graph
    W0, b, u, c, v = params
    M, cout, cin = W0.shape
    n = x.shape[0]
    a = x @ u.t() + c                                               # model.py:79 (+ c, :93)
    g = x @ v.t()                                                   # model.py:80
    xp = torch.cat([x, torch.zeros(1, cin, dtype=x.dtype)], 0)      # the zero row of get_patches, model.py:383-384
    gp = torch.cat([g, torch.zeros(1, M, dtype=x.dtype)], 0)
    q = torch.softmax(a[:, None, :] + gp[idx], dim=-1) * mask[:, :, None].to(x.dtype)      # [n, D, M]
    z = torch.bmm(q.transpose(1, 2), xp[idx])                       # [n, M, cin]
    Wr = W0.permute(0, 2, 1).reshape(M * cin, cout)                 # [(m, c), o]
    degf = deg.to(x.dtype)
    inv = torch.where(deg > 0, 1.0 / degf.clamp_min(1.0), torch.zeros_like(degf))   # model.py:436-443
    y = (z.reshape(n, M * cin) @ Wr) * inv[:, None]                 # model.py:491-493
    if biasMask:
        return torch.where((deg > 0)[:, None], y + b, y)            # model.py:496-500
    return y + b


def _pool(x):
    return x.reshape(-1, 4, x.shape[1]).amax(dim=1)                 # model.py:779-788, steps = 2


def _up(x):
    return x.repeat_interleave(4, dim=0)                            # model.py:817-825, steps = 2


def get_model(x, graphs, params, multi_scale=False):
    """ref: model.py:837-946.  x [n0, 6]; graphs = three pad_klist tuples; multi_scale: also the two coarse heads
    (model.py:894-899, 915-920), returned as (y0, y1, y2) like oracle/model_ref.get_model_reg_multi_scale."""
    p = list(params)
    take = lambda k: [p.pop(0) for _ in range(k)]
    a = R.LRELU_ALPHA
    head = lambda t, W1, b1, W2, b2: R.lrelu(t @ W1 + b1, a) @ W2 + b2          # model.py:763-769
    h1 = R.lrelu(custom_conv2d(x, graphs[0], take(5)), a)
    h2 = R.lrelu(custom_conv2d(_pool(h1), graphs[1], take(5)), a)
    h3 = R.lrelu(custom_conv2d(_pool(h2), graphs[2], take(5)), a)
    d3 = R.lrelu(custom_conv2d(h3, graphs[2], take(5)), a)
    y2 = head(d3, *take(4)) if multi_scale else None
    u2 = custom_conv2d(_up(d3), graphs[1], take(5))                 # no activation, model.py:905
    d2 = R.lrelu(custom_conv2d(torch.cat([u2, h2], 1), graphs[1], take(5)), a)
    y1 = head(d2, *take(4)) if multi_scale else None
    u1 = custom_conv2d(_up(d2), graphs[0], take(5))                 # model.py:926
    d1 = R.lrelu(custom_conv2d(torch.cat([u1, h1], 1), graphs[0], take(5)), a)
    y0 = head(d1, *take(4))                                         # model.py:937-941
    assert not p
    return (y0, y1, y2) if multi_scale else y0


def train_loss(x, adjs, gt, params, sample_ind, Rm):
    """ref: train.py:439-517 in float64.  x [1, n0, 6], adjs three K-lists, gt [1, n0, 3] (numpy or tensors of any float
    type), params float64 leaves; returns (loss, n_conv [1, n0, 3])."""
    xt = torch.as_tensor(np.asarray(x), dtype=DT).reshape(1, -1, 6)
    gtt = torch.as_tensor(np.asarray(gt), dtype=DT).reshape(1, -1, 3)
    x_r, gt_r = R.rotate_inputs(xt, gtt, torch.as_tensor(np.asarray(Rm), dtype=DT))
    graphs = [pad_klist(a) for a in adjs]
    y = get_model(x_r[0], graphs, params)
    n_conv = R.normalizeTensor(y[None])
    idx = torch.as_tensor(np.asarray(sample_ind), dtype=torch.long)
    return R.faceNormalsLoss(n_conv[:, idx], gt_r[:, idx]), n_conv


def train_loss_ms(x, adjs, gt, params, sample_ind, Rm):
    """oracle/model_ref.train_loss_ms (the build's multi-scale training objective: one sampled angular loss per head against
    the pooled ground truth) in float64 on the closed form.  Returns (loss0 + loss1 + loss2, [loss_k], [n_conv_k])."""
    xt = torch.as_tensor(np.asarray(x), dtype=DT).reshape(1, -1, 6)
    gtt = torch.as_tensor(np.asarray(gt), dtype=DT).reshape(1, -1, 3)
    Rt = torch.as_tensor(np.asarray(Rm), dtype=DT)
    x_r, gt_r = R.rotate_inputs(xt, gtt, Rt)
    ys = get_model(x_r[0], [pad_klist(a) for a in adjs], params, multi_scale=True)
    gts = [gt_r, None, None]
    g = gtt
    for k in (1, 2):
        g = R.pooled_gt(g)
        gts[k] = torch.matmul(g, Rt.t())
    idx = torch.as_tensor(np.asarray(sample_ind), dtype=torch.long)
    losses, nconvs = [], []
    for y, gk in zip(ys, gts):
        n_conv = R.normalizeTensor(y[None])
        ik = idx % y.shape[0]
        losses.append(R.faceNormalsLoss(n_conv[:, ik], gk[:, ik]))
        nconvs.append(n_conv)
    return losses[0] + losses[1] + losses[2], losses, nconvs


def init_params(seed=0, multi_scale=False):
    """The fixture generator's seeded parameters (model_ref.init_params) as float64 leaves."""
    return [p.to(DT).requires_grad_(True) for p in R.init_params(seed, multi_scale=multi_scale)]
