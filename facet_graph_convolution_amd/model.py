"""Drop-in counterparts of the reference's ``Code/model.py`` operator functions, on torch tensors.

Same names, argument order, defaults and tensor layouts as the reference (x float32 ``[1, n, C]`` channels
innermost; adj int ``[1, n, K]`` one-indexed, 0 = no neighbour, slot 0 = self), so code written against
``model.py`` reads the same:

    custom_conv2d(x, adj, out_channels, M, biasMask=True, ...) -> (y, [W0, u, c])      model.py:427-504
    custom_lin(input, out_channels)                                                     model.py:763-769
    custom_binary_tree_pooling(x, steps=1, pooltype='max')                              model.py:779-788
    custom_upsampling(x, steps=1)                                                       model.py:817-825
    lrelu(x, alpha)                                                                     model.py:828-830
    get_model_reg_multi_scale(x, adjs, keep_prob, coarsening_steps=2, multiScale=False) model.py:837-946

Every op is a ``torch.autograd.Function`` over the HIP kernels of libfgc (ops.py); there is no eager fallback.
TensorFlow creates variables inside these calls (``tf.Variable``); here they come from the ambient
``VariableStore`` in the same creation order (44 variables, 52 with multiScale), created on first use and reused
after ``store.rewind()`` - the moral equivalent of building the graph once and running it many times.

This module is the compatibility surface.  The throughput path is ``net.FacetDenoiser``, which schedules the same
kernels by hand with every pooling / upsampling / concat / activation folded into the conv kernels.
"""
import contextlib

import numpy as np
import torch

from . import ops
from .graph import as_graph

std_dev = 0.05        # model.py:17
std_dev_bias = 0.01   # model.py:18


# ---------------------------------------------------------------------------------------------------
# variables
# ---------------------------------------------------------------------------------------------------
class VariableStore:
    """Creation-order list of parameters (leaf tensors with requires_grad)."""

    def __init__(self, device="cuda", seed=0):
        self.device = torch.device(device)
        self.rs = np.random.RandomState(seed)
        self.vars = []
        self.kinds = []
        self._cursor = 0

    def rewind(self):
        self._cursor = 0

    def get(self, kind, shape):
        if self._cursor < len(self.vars):
            v = self.vars[self._cursor]
            if tuple(v.shape) != tuple(shape):
                raise RuntimeError("variable %d was created with shape %s, now requested as %s" %
                                   (self._cursor, tuple(v.shape), tuple(shape)))
        else:
            std = std_dev_bias if kind == "bias" else std_dev
            v = torch.tensor(self.rs.normal(0.0, std, size=shape).astype(np.float32), device=self.device,
                             requires_grad=True)
            self.vars.append(v)
            self.kinds.append(kind)
        self._cursor += 1
        return v

    def load(self, tensors):
        self.vars = [torch.as_tensor(t, dtype=torch.float32).to(self.device).requires_grad_(True) for t in tensors]
        self._cursor = 0


_STORE = [None]


def default_store():
    if _STORE[0] is None:
        _STORE[0] = VariableStore()
    return _STORE[0]


@contextlib.contextmanager
def variable_store(store):
    prev = _STORE[0]
    _STORE[0] = store
    store.rewind()
    try:
        yield store
    finally:
        _STORE[0] = prev


def weight_variable(shape):
    return default_store().get("weight", shape)


def bias_variable(shape):
    return default_store().get("bias", shape)


def assignment_variable(shape):
    return default_store().get("assignment", shape)


# ---------------------------------------------------------------------------------------------------
# autograd functions over libfgc
# ---------------------------------------------------------------------------------------------------
def _rows(x):
    if x.dim() != 3 or x.shape[0] != 1:
        raise ValueError("expected a [1, n, C] tensor (batch size is 1, train.py:405)")
    return x[0]


class _ConvFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, W0, b, u, c, v, graph, bias_mask):
        y, _, ag = ops.conv_fwd(graph, x, None, 0, [W0, b, u, c, v], bias_mask=bias_mask)
        ctx.save_for_backward(x, W0, b, u, c, v, ag, y)
        ctx.graph, ctx.bias_mask = graph, bias_mask
        return y

    @staticmethod
    def backward(ctx, dy):
        x, W0, b, u, c, v, ag, y = ctx.saved_tensors
        dx, _, grads = ops.conv_bwd(ctx.graph, x, None, 0, [W0, b, u, c, v], ag, y, dy.contiguous(),
                                    bias_mask=ctx.bias_mask, need_dx=ctx.needs_input_grad[0])
        return (dx, grads[0], grads[1], grads[2], grads[3], grads[4], None, None)


class _LreluFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, alpha):
        y = ops.lrelu_fwd(x, alpha)
        ctx.save_for_backward(y)
        ctx.alpha = alpha
        return y

    @staticmethod
    def backward(ctx, dy):
        (y,) = ctx.saved_tensors
        return ops.lrelu_bwd(y, dy.contiguous(), ctx.alpha), None


class _Pool4Fn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        y = ops.pool4_fwd(x)
        ctx.save_for_backward(x, y)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, y = ctx.saved_tensors
        return ops.pool4_bwd(x, y, dy.contiguous())


class _Up4Fn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        return ops.upsample4_fwd(x)

    @staticmethod
    def backward(ctx, dy):
        return ops.upsample4_bwd(dy.contiguous())


class _PoolFn(torch.autograd.Function):
    """max over 2^steps consecutive rows for any steps (the 4:1 form has its own kernels above)."""

    @staticmethod
    def forward(ctx, x, group):
        y = ops.pool_fwd(x, group)
        ctx.save_for_backward(x, y)
        ctx.group = group
        return y

    @staticmethod
    def backward(ctx, dy):
        x, y = ctx.saved_tensors
        return ops.pool_bwd(x, y, dy.contiguous(), ctx.group), None


class _UpFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, group):
        ctx.group = group
        return ops.upsample_fwd(x, group)

    @staticmethod
    def backward(ctx, dy):
        return ops.upsample_bwd(dy.contiguous(), ctx.group), None


class _LinFn(torch.autograd.Function):
    """x W + b through libfgc (fgc_lin_fwd / fgc_lin_bwd: fp32 MFMA, fixed-order sum over the rows)."""

    @staticmethod
    def forward(ctx, x, W, b):
        ctx.save_for_backward(x, W)
        return ops.lin_fwd(x, W, b)

    @staticmethod
    def backward(ctx, dy):
        x, W = ctx.saved_tensors
        dx, dW, db = ops.lin_bwd(x, dy.contiguous(), W, need_dx=ctx.needs_input_grad[0])
        return dx, dW, db


class _MlpFn(torch.autograd.Function):
    """lrelu(x W1 + b1) W2 + b2 with the hidden layer kept on chip."""

    @staticmethod
    def forward(ctx, x, W1, b1, W2, b2, alpha):
        y = ops.mlp_fwd(x, W1, b1, W2, b2, alpha)
        ctx.save_for_backward(x, W1, b1, W2)
        ctx.alpha = alpha
        return y

    @staticmethod
    def backward(ctx, dy):
        x, W1, b1, W2 = ctx.saved_tensors
        dx, dW1, db1, dW2, db2 = ops.mlp_bwd(x, dy.contiguous(), W1, b1, W2, ctx.alpha)
        return dx, dW1, db1, dW2, db2, None


class _NormalizeFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        y, scratch = ops.normalize_fwd(x)
        ctx.save_for_backward(x, scratch)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, scratch = ctx.saved_tensors
        return ops.normalize_bwd(x, dy.contiguous(), scratch)


# ---------------------------------------------------------------------------------------------------
# the reference's operator functions
# ---------------------------------------------------------------------------------------------------
def custom_conv2d(x, adj, out_channels, M, biasMask=True, translation_invariance=False, rotation_invariance=False):
    """model.py:427-504 (the invariance-off branch, the only one the network uses: model.py:841-842)."""
    if translation_invariance or rotation_invariance:
        raise NotImplementedError("invariant assignments are dead code in the reference network (model.py:841-842)")
    if M != 9:
        raise NotImplementedError("libfgc is built for M = 9 (model.py:855)")
    xr = _rows(x)
    in_channels = xr.shape[1]
    W0 = weight_variable([M, out_channels, in_channels])
    b = bias_variable([out_channels])
    u = assignment_variable([M, in_channels])
    c = assignment_variable([M])
    v = assignment_variable([M, in_channels])
    graph = as_graph(adj, xr.device)
    y = _ConvFn.apply(xr.contiguous(), W0, b, u, c, v, graph, bool(biasMask))
    return y.unsqueeze(0), [W0, u, c]


def custom_lin(input, out_channels):
    """model.py:763-769, through libfgc (fgc_lin_fwd / fgc_lin_bwd).  A caller that composes custom_lin -> lrelu ->
    custom_lin itself (model.py:937-941) materialises the hidden tensor, as the reference does; get_model_reg_multi_scale
    below uses the fused head (fgc_mlp_*), which keeps it on chip."""
    xr = _rows(input)
    W = weight_variable([xr.shape[1], out_channels])
    b = bias_variable([out_channels])
    return _LinFn.apply(xr.contiguous(), W, b).unsqueeze(0)


def custom_binary_tree_pooling(x, steps=1, pooltype='max'):
    """model.py:779-788.  steps=2 ('max') is what the network uses; any other step count is one 2^steps : 1 launch."""
    if pooltype == 'avg_ignore_zeros':      # model.py:792-814: inference-time pooling of positions / normals
        if steps % 2 or x.requires_grad:
            raise NotImplementedError("avg_ignore_zeros is built 4:1 per call and without a gradient")
        xr = _rows(x)
        for _ in range(steps // 2):
            xr = ops.pool4_avg_iz(xr.contiguous())
        return xr.unsqueeze(0)
    if pooltype != 'max':
        raise NotImplementedError("'max' (model.py:863,875) and 'avg_ignore_zeros' are built")
    xr = _rows(x)
    if steps == 2:
        return _Pool4Fn.apply(xr.contiguous()).unsqueeze(0)
    if steps < 0 or xr.shape[0] % (2 ** steps):
        raise ValueError("pooling %d rows by 2^%d" % (xr.shape[0], steps))
    return _PoolFn.apply(xr.contiguous(), 2 ** steps).unsqueeze(0)      # any other step count: one 2^steps : 1 launch


def custom_upsampling(x, steps=1):
    """model.py:817-825."""
    xr = _rows(x)
    if steps == 2:
        return _Up4Fn.apply(xr.contiguous()).unsqueeze(0)
    if steps < 0:
        raise ValueError("steps must be >= 0")
    return _UpFn.apply(xr.contiguous(), 2 ** steps).unsqueeze(0)


def lrelu(x, alpha):
    """model.py:828-830."""
    return _LreluFn.apply(x.contiguous(), float(alpha))


def normalizeTensor(x):
    """utils.py:1700-1715."""
    return _NormalizeFn.apply(_rows(x).contiguous()).unsqueeze(0)


def _head(x, hidden, out_channels, alpha):
    """lrelu(custom_lin(x, hidden)) -> custom_lin(., out): variables in the reference order, fused kernel."""
    xr = _rows(x)
    W1 = weight_variable([xr.shape[1], hidden])
    b1 = bias_variable([hidden])
    W2 = weight_variable([hidden, out_channels])
    b2 = bias_variable([out_channels])
    return _MlpFn.apply(xr.contiguous(), W1, b1, W2, b2, float(alpha)).unsqueeze(0)


def get_model_reg_multi_scale(x, adjs, keep_prob, coarsening_steps=2, multiScale=False):
    """model.py:837-946.  keep_prob is accepted and unused, coarsening_steps is forced to 2 and alpha to 0.1,
    exactly as in the reference (model.py:846-847)."""
    alpha = 0.1
    coarsening_steps = 2
    out_channels_reg = 3
    h_conv1, _ = custom_conv2d(x, adjs[0], 32, 9)
    h_conv1_act = lrelu(h_conv1, alpha)
    pool1 = custom_binary_tree_pooling(h_conv1_act, steps=coarsening_steps)
    h_conv2, _ = custom_conv2d(pool1, adjs[1], 64, 9)
    h_conv2_act = lrelu(h_conv2, alpha)
    pool2 = custom_binary_tree_pooling(h_conv2_act, steps=coarsening_steps)
    h_conv3, _ = custom_conv2d(pool2, adjs[2], 128, 9)
    h_conv3_act = lrelu(h_conv3, alpha)
    dconv3, _ = custom_conv2d(h_conv3_act, adjs[2], 128, 9)
    dconv3_act = lrelu(dconv3, alpha)
    if multiScale:
        y_conv2 = _head(dconv3_act, 1024, out_channels_reg, alpha)
    upsamp2 = custom_upsampling(dconv3_act, steps=coarsening_steps)
    upconv2, _ = custom_conv2d(upsamp2, adjs[1], 64, 9)
    concat2 = torch.cat([upconv2, h_conv2_act], dim=-1)
    dconv2, _ = custom_conv2d(concat2, adjs[1], 64, 9)
    dconv2_act = lrelu(dconv2, alpha)
    if multiScale:
        y_conv1 = _head(dconv2_act, 1024, out_channels_reg, alpha)
    upsamp1 = custom_upsampling(dconv2_act, steps=coarsening_steps)
    upconv1, _ = custom_conv2d(upsamp1, adjs[0], 32, 9)
    concat1 = torch.cat([upconv1, h_conv1_act], dim=-1)
    dconv1, _ = custom_conv2d(concat1, adjs[0], 32, 9)
    dconv1_act = lrelu(dconv1, alpha)
    y_conv0 = _head(dconv1_act, 1024, out_channels_reg, alpha)
    if multiScale:
        return y_conv0, y_conv1, y_conv2
    return y_conv0
