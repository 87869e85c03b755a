"""Third-party cross-checks of the stand-in primitives behind the golden vectors.

The fixtures under tests/golden/ are what the reference's own files (gan_utils.py, data_utils.py)
return when they run on ``oracle/refshim/tensorflow`` (TensorFlow is not installed).  The stand-in's
statements of TF semantics are checked here against implementations nobody in this repo wrote
(scipy, torch, NumPy's own std) and against worked examples from the TensorFlow documentation, so
that a common-mode misreading (ddof, REFLECT vs SYMMETRIC, correlation vs convolution, channel
order) cannot hide behind a green parity suite.
"""
import os
import sys

import numpy as np
import pytest
import scipy.ndimage
import scipy.special
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "oracle", "refshim"))
import tensorflow as tf  # noqa: E402  (the stand-in; test infrastructure only)

sys.path.pop(0)
sys.modules.pop("tensorflow", None)      # nobody else in the test session should find a "tensorflow"


def test_reduce_logsumexp_matches_scipy():
    # gan_utils.py:153,156 -- reduce_logsumexp(..., axis=1 / axis=0, keepdims=True)
    rng = np.random.default_rng(0)
    x = (rng.standard_normal((17, 23)) * 300).astype(np.float64)
    for axis in (0, 1):
        for keep in (True, False):
            got = tf.reduce_logsumexp(x, axis=axis, keepdims=keep)
            ref = scipy.special.logsumexp(x, axis=axis, keepdims=keep)
            np.testing.assert_allclose(got, ref, rtol=1e-13)
            assert got.shape == ref.shape
    # the example of the TensorFlow docstring: x = [[0,0,0],[0,0,0]]
    z = np.zeros((2, 3))
    assert np.isclose(tf.reduce_logsumexp(z), np.log(6))
    np.testing.assert_allclose(tf.reduce_logsumexp(z, 0), np.log(2) * np.ones(3))
    np.testing.assert_allclose(tf.reduce_logsumexp(z, 1), np.log(3) * np.ones(2))
    np.testing.assert_allclose(tf.reduce_logsumexp(z, 1, keepdims=True), np.log(3) * np.ones((2, 1)))
    # lines that are all -inf stay -inf (TF replaces the non-finite max by 0, no NaN)
    m = np.full((3, 4), -np.inf)
    assert np.all(np.isneginf(tf.reduce_logsumexp(m, axis=1)))
    assert np.all(np.isneginf(scipy.special.logsumexp(m, axis=1)))


def test_reduce_std_is_population_std():
    # gan_utils.py:193 -- tf.math.reduce_std(M, axis=(0, 1)); TF documents ddof = 0
    rng = np.random.default_rng(1)
    x = rng.standard_normal((6, 9, 4))
    np.testing.assert_allclose(tf.math.reduce_std(x, axis=(0, 1)), np.std(x, axis=(0, 1), ddof=0), rtol=1e-13)
    np.testing.assert_allclose(tf.math.reduce_std(x), torch.from_numpy(x).std(unbiased=False).item(), rtol=1e-13)
    # TensorFlow's docstring example: reduce_std([[1., 2.], [3., 4.]]) = 1.118034, axis 0 -> [1, 1], axis 1 -> [.5, .5]
    e = np.array([[1.0, 2.0], [3.0, 4.0]])
    assert abs(float(tf.math.reduce_std(e)) - 1.118034) < 5e-7
    np.testing.assert_allclose(tf.math.reduce_std(e, 0), [1.0, 1.0])
    np.testing.assert_allclose(tf.math.reduce_std(e, 1), [0.5, 0.5])
    # and it is NOT the sample std (ddof = 1)
    assert abs(float(tf.math.reduce_std(e)) - np.std(e, ddof=1)) > 0.1


def test_pad_reflect_and_symmetric_match_torch_and_scipy():
    # data_utils.py:512-513,562-565 -- tf.pad(..., "REFLECT")
    # TensorFlow's docstring example: t = [[1,2,3],[4,5,6]], paddings [[1,1],[2,2]]
    t = np.array([[1, 2, 3], [4, 5, 6]])
    p = [[1, 1], [2, 2]]
    np.testing.assert_array_equal(tf.pad(t, p, "REFLECT"),
                                  [[6, 5, 4, 5, 6, 5, 4], [3, 2, 1, 2, 3, 2, 1], [6, 5, 4, 5, 6, 5, 4], [3, 2, 1, 2, 3, 2, 1]])
    np.testing.assert_array_equal(tf.pad(t, p, "SYMMETRIC"),
                                  [[2, 1, 1, 2, 3, 3, 2], [2, 1, 1, 2, 3, 3, 2], [5, 4, 4, 5, 6, 6, 5], [5, 4, 4, 5, 6, 6, 5]])
    np.testing.assert_array_equal(tf.pad(t, p, "CONSTANT"),
                                  [[0] * 7, [0, 0, 1, 2, 3, 0, 0], [0, 0, 4, 5, 6, 0, 0], [0] * 7])
    rng = np.random.default_rng(2)
    x = rng.standard_normal((3, 11, 2)).astype(np.float32)
    got = tf.pad(x, [[0, 0], [3, 3], [0, 0]], "REFLECT")
    ref = torch.nn.functional.pad(torch.from_numpy(x).permute(0, 2, 1), (3, 3), mode="reflect").permute(0, 2, 1).numpy()
    np.testing.assert_array_equal(got, ref)
    # a unit-tap correlation under scipy's boundary modes reads the padded sample: 'mirror' = REFLECT, 'reflect' = SYMMETRIC
    line = rng.standard_normal(9)
    for tfmode, spmode in (("REFLECT", "mirror"), ("SYMMETRIC", "reflect")):
        padded = tf.pad(line, [[3, 3]], tfmode)
        for shift in range(-3, 4):
            w = np.zeros(7)
            w[3 + shift] = 1.0
            np.testing.assert_array_equal(scipy.ndimage.correlate1d(line, w, mode=spmode), padded[3 + shift: 3 + shift + 9])


@pytest.mark.parametrize("nd", [1, 2, 3])
def test_conv_is_valid_cross_correlation_in_channels_last_layout(nd):
    # data_utils.py:515,537,567 -- tf.nn.conv{1,2,3}d(x [N,*S,Cin], w [*k,Cin,Cout], stride 1, "VALID")
    rng = np.random.default_rng(3 + nd)
    spatial = (9, 8, 7)[:nd]
    ksz = (3, 4, 2)[:nd]
    cin, cout = 2, 3
    x = rng.standard_normal((2,) + spatial + (cin,))
    w = rng.standard_normal(ksz + (cin, cout))            # asymmetric: a flipped kernel or swapped channel axes would show
    got = (tf.nn.conv1d, tf.nn.conv2d, tf.nn.conv3d)[nd - 1](x, w, 1, "VALID")
    out_sp = tuple(s - k + 1 for s, k in zip(spatial, ksz))
    assert got.shape == (2,) + out_sp + (cout,)
    ref = np.zeros_like(got)
    for n in range(2):
        for co in range(cout):
            for ci in range(cin):
                full = scipy.ndimage.correlate(x[n, ..., ci], w[..., ci, co], mode="constant", origin=0)
                # scipy centres the kernel at k // 2; the VALID window starts there
                sl = tuple(slice(k // 2, k // 2 + o) for k, o in zip(ksz, out_sp))
                ref[n, ..., co] += full[sl]
    np.testing.assert_allclose(got, ref, rtol=1e-12, atol=1e-12)


def test_small_primitives_against_numpy_documented_behaviour():
    a = np.arange(6.0).reshape(2, 3)
    assert tf.expand_dims(a, 1).shape == (2, 1, 3) and tf.expand_dims(a, 0).shape == (1, 2, 3)
    assert tf.reduce_sum(a, -1).tolist() == [3.0, 12.0]
    assert tf.transpose(a).shape == (3, 2) and tf.transpose(np.zeros((2, 3, 4)), (0, 2, 1)).shape == (2, 4, 3)
    assert tf.math.greater(np.float32(2.0), np.float32(1.0)) is True and tf.math.greater(np.float32(0.5), np.float32(1.0)) is False
    assert tf.range(5).tolist() == [0, 1, 2, 3, 4] and tf.range(-3, 4, dtype=np.float32).dtype == np.float32
    gx, gy = tf.meshgrid(np.arange(3), np.arange(2))             # TF default indexing 'xy' like NumPy's
    assert gx.shape == (2, 3) and gy.shape == (2, 3)
