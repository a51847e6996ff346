"""The fixture generator's TensorFlow stand-in (tests/golden/gen/tf_shim) pinned against the documented semantics of
the tf ops whose meaning is not obvious from their name: the golden vectors are only as good as this mapping."""
import importlib.util
import os

import numpy as np
import pytest
import torch

HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.fixture(scope="module")
def tf():
    path = os.path.join(HERE, "golden", "gen", "tf_shim", "tensorflow", "__init__.py")
    spec = importlib.util.spec_from_file_location("fgc_tf_shim_under_test", path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_gather_is_numpy_take_along_the_given_axis(tf):
    p = torch.arange(2 * 5 * 3, dtype=torch.float32).reshape(2, 5, 3)
    idx = torch.tensor([[4, 0], [1, 1]], dtype=torch.int32)
    for axis in (0, 1, 2):
        src = p if axis else p[:, :2]
        i = idx % src.shape[axis]
        np.testing.assert_array_equal(tf.gather(src, i, axis=axis).numpy(), np.take(src.numpy(), i.numpy(), axis=axis))
    # default axis = 0; result shape = params.shape[:axis] + indices.shape + params.shape[axis+1:]
    assert tuple(tf.gather(p, torch.tensor([1, 0, 1])).shape) == (3, 5, 3)
    assert tuple(tf.gather(p, idx, axis=1).shape) == (2, 2, 2, 3)


def test_reduce_max_gradient_is_split_evenly_over_ties(tf):
    """tf's _MinOrMaxGrad divides the incoming gradient by the number of maxima; the pooling fixture relies on it."""
    x = torch.tensor([[1.0, 3.0, 3.0, 2.0], [5.0, 5.0, 5.0, 5.0]], requires_grad=True)
    tf.reduce_max(x, axis=1).sum().backward()
    np.testing.assert_allclose(x.grad.numpy(), [[0, 0.5, 0.5, 0], [0.25, 0.25, 0.25, 0.25]])
    assert tf.reduce_max(x, axis=1, keepdims=True).shape == (2, 1)
    assert float(tf.reduce_max(x)) == 5.0


def test_div_floors_integers_like_python2_and_divides_floats(tf):
    a = torch.tensor([-1, 0, 3, 7, -5], dtype=torch.int32)
    np.testing.assert_array_equal(tf.div(a, 4).numpy(), [-1, 0, 0, 1, -2])   # -1 stays -1: fake-node ids survive
    np.testing.assert_allclose(tf.div(torch.tensor([1.0, -3.0]), 4.0).numpy(), [0.25, -0.75])
    np.testing.assert_allclose(tf.divide(torch.tensor([1.0]), torch.tensor([4.0])).numpy(), [0.25])


def test_softmax_axis_count_nonzero_tile_and_slice(tf):
    x = torch.tensor([[[1.0, 2.0, 3.0], [0.0, 0.0, 0.0]]])
    s = tf.nn.softmax(x)
    np.testing.assert_allclose(s.sum(-1).numpy(), [[1.0, 1.0]], rtol=1e-6)
    np.testing.assert_allclose(s[0, 1].numpy(), [1 / 3] * 3, rtol=1e-6)
    adj = torch.tensor([[[1, 5, 0], [0, 0, 0]]], dtype=torch.int32)
    np.testing.assert_array_equal(tf.count_nonzero(adj, axis=2).numpy(), [[2, 0]])
    t = tf.tile(torch.tensor([[1, 2]]), [2, 3])
    np.testing.assert_array_equal(t.numpy(), np.tile(np.array([[1, 2]]), (2, 3)))
    y = torch.arange(24).reshape(2, 3, 4)
    np.testing.assert_array_equal(tf.slice(y, [0, 1, 0], [-1, 1, -1]).numpy(), y.numpy()[:, 1:2, :])


def test_transpose_map_fn_where_and_constant(tf):
    x = torch.arange(24, dtype=torch.float32).reshape(2, 3, 4)
    np.testing.assert_array_equal(tf.transpose(x, [2, 0, 1]).numpy(), np.transpose(x.numpy(), (2, 0, 1)))
    np.testing.assert_array_equal(tf.map_fn(lambda e: e * 2, x).numpy(), 2 * x.numpy())
    c = torch.tensor([True, False])
    np.testing.assert_array_equal(tf.where(c, torch.tensor([1.0, 1.0]), torch.tensor([0.0, 0.0])).numpy(), [1.0, 0.0])
    k = tf.constant(0.25, shape=[3])
    assert tuple(k.shape) == (3,) and float(k[2]) == 0.25
    assert tf.constant([1, 2], dtype=tf.int32).dtype == torch.int32


def test_variable_feed_order_and_registry(tf):
    tf.VARIABLES.clear()
    tf.VARIABLE_FEED = iter([np.full((2, 2), 7.0, dtype=np.float32), np.full((3,), 9.0, dtype=np.float32)])
    a = tf.Variable(tf.zeros([2, 2]), name="a")
    b = tf.Variable(tf.zeros([3]), name="b")
    c = tf.Variable(tf.zeros([1]), name="frozen", trainable=False)
    tf.VARIABLE_FEED = None
    assert float(a[0, 0]) == 7.0 and float(b[0]) == 9.0 and not c.requires_grad
    assert [n for n, _ in tf.VARIABLES] == ["a", "b"]
    tf.VARIABLES.clear()
