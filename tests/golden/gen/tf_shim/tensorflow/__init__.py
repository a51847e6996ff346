"""Eager stand-in for the ~55 ``tf.*`` symbols the reference's live path touches.

TEST TOOLING ONLY.  TensorFlow is not installed in the build container, so the
golden vectors under ``tests/golden/`` are produced by executing the reference
*source* (``/root/reference/Code/{model,utils,train}.py``, unmodified, read
where it lies) with each ``tf`` op mapped one-to-one onto a torch CPU op of the
same semantics.  Nothing here is reference code and nothing here is imported by
the product (``facet_graph_convolution_amd``), the tests, or the GPU box.

float64 mode: set ``TF_SHIM_DTYPE=float64`` to run the same graph in double
precision (used to measure the fp32 error budget of the fixtures).
"""
import builtins as _bi
import contextlib
import os

import numpy as _np
import torch as _t

__version__ = "1.15.0"

_FDT = _t.float64 if os.environ.get("TF_SHIM_DTYPE", "float32") == "float64" else _t.float32
float32 = _FDT
int32 = _t.int32
int64 = _t.int64

# variables created by the model, in creation order (read back by the harness)
VARIABLES = []
# when set to an iterator, tf.Variable pops fixed initial values from it
VARIABLE_FEED = None


def _as_t(x, dtype=None):
    if isinstance(x, _t.Tensor):
        return x if dtype is None else x.to(dtype)
    a = _np.asarray(x)
    if dtype is None:
        dtype = _FDT if a.dtype.kind == "f" else None
    return _t.tensor(a, dtype=dtype)


class _Shape(list):
    def as_list(self):
        return list(self)


def _get_shape(self):
    return _Shape(self.shape)


_t.Tensor.get_shape = _get_shape


def constant(value, dtype=None, shape=None, name=None):
    t = _as_t(value, dtype)
    if shape is not None:
        t = t.reshape(shape) if t.numel() > 1 or len(shape) == 0 else t.expand(shape).clone()
    return t


def random_normal(shape, mean=0.0, stddev=1.0, dtype=None, seed=None, name=None):
    return _t.randn(*shape, dtype=_FDT) * stddev + mean


def truncated_normal(shape, mean=0.0, stddev=1.0, dtype=None, seed=None, name=None):
    raise NotImplementedError("dead path in the reference")


def Variable(initial_value, name=None, trainable=True, dtype=None):
    if VARIABLE_FEED is not None and trainable:
        v = next(VARIABLE_FEED)
        v = _as_t(v, _FDT)
        assert tuple(v.shape) == tuple(initial_value.shape), (v.shape, initial_value.shape, name)
    else:
        v = _as_t(initial_value)
    v = v.detach().clone()
    if trainable and v.dtype.is_floating_point:
        v.requires_grad_(True)
        VARIABLES.append((name, v))
    return v


@contextlib.contextmanager
def variable_scope(name=None, *a, **k):
    yield


name_scope = variable_scope


@contextlib.contextmanager
def device(name=None):
    yield


def count_nonzero(x, axis=None, keepdims=False, dtype=None):
    return (x != 0).sum(dim=axis, keepdim=keepdims)


def not_equal(a, b):
    return a != b


def equal(a, b):
    return a == b


def greater(a, b):
    return a > b


def less_equal(a, b):
    return a <= b


def cast(x, dtype):
    return x.to(dtype)


def where(c, a, b):
    return _t.where(c, a, b)


def reciprocal(x, name=None):
    return 1.0 / x


def zeros_like(x, dtype=None, name=None):
    return _t.zeros_like(x, dtype=dtype)


def ones_like(x, dtype=None, name=None):
    return _t.ones_like(x, dtype=dtype)


def zeros(shape, dtype=None):
    return _t.zeros(*shape, dtype=dtype or _FDT)


def reshape(x, shape, name=None):
    return x.reshape(shape)


def transpose(x, perm=None):
    return x.permute(*perm)


def map_fn(fn, elems):
    return _t.stack([fn(e) for e in elems])


def matmul(a, b):
    return _t.matmul(a, b)


def concat(values, axis, name=None):
    return _t.cat(list(values), dim=axis)


def stack(values, axis=0):
    return _t.stack(list(values), dim=axis)


def gather(params, indices, axis=0):
    idx = indices.long()
    sl = (_bi.slice(None),) * axis + (idx,)
    return params[sl]


def add(a, b):
    return a + b


def subtract(a, b):
    return a - b


def multiply(a, b, name=None):
    return a * b


def divide(a, b):
    return a / b


def div(a, b):
    # tf.div: Python-2 division semantics, i.e. floor division for integer tensors
    if not a.is_floating_point():
        return _t.div(a, b, rounding_mode="floor")
    return a / b


def reduce_sum(x, axis=None, keepdims=False, name=None):
    return x.sum() if axis is None else x.sum(dim=axis, keepdim=keepdims)


def reduce_mean(x, axis=None, keepdims=False, name=None):
    return x.mean() if axis is None else x.mean(dim=axis, keepdim=keepdims)


def reduce_max(x, axis=None, keepdims=False, name=None):
    # amax spreads the gradient evenly over ties, as tf.reduce_max does
    return x.amax() if axis is None else x.amax(dim=axis, keepdim=keepdims)


def reduce_any(x, axis=None, name=None):
    return x.any() if axis is None else x.any(dim=axis)


def reduce_all(x, axis=None, keepdims=False, name=None):
    return x.all() if axis is None else x.all(dim=axis, keepdim=keepdims)


def is_nan(x):
    return _t.isnan(x)


def tile(x, multiples):
    return x.repeat(*multiples)


def expand_dims(x, axis, name=None):
    return x.unsqueeze(axis)


def squeeze(x, axis=None):
    return x.squeeze() if axis is None else x.squeeze(axis)


def abs(x, name=None):  # noqa: A001
    return x.abs()


def square(x, name=None):
    return x * x


def sqrt(x, name=None):
    return x.sqrt()


def minimum(a, b):
    return _t.minimum(_as_t(a, _FDT) if not isinstance(a, _t.Tensor) else a,
                      _as_t(b, _FDT) if not isinstance(b, _t.Tensor) else b)


def maximum(a, b):
    return _t.maximum(_as_t(a, _FDT) if not isinstance(a, _t.Tensor) else a,
                      _as_t(b, _FDT) if not isinstance(b, _t.Tensor) else b)


def acos(x):
    return _t.acos(x)


def slice(x, begin, size):  # noqa: A001
    idx = tuple(_bi.slice(b, None if s == -1 else b + s) for b, s in zip(begin, size))
    return x[idx]


class nn:  # noqa: N801
    @staticmethod
    def softmax(x, axis=-1):
        return _t.softmax(x, dim=axis)

    @staticmethod
    def relu(x):
        return _t.relu(x)


class _Missing:
    def __getattr__(self, k):
        raise NotImplementedError("tf shim: symbol not mapped: " + k)


train = _Missing()
