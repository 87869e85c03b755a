"""Known-answer tests pinning the KernelSmoothing oracle (oracle/smoothing_np.py) -- the
reference's data_utils.py cannot be imported here and has no fixtures, so
these KATs are derived from the cited reference lines.  CPU only."""
import numpy as np
import pytest
from scipy import ndimage

from oracle import smoothing_np as sm


def test_taps_sum_to_one_and_are_symmetric():     # data_utils.py:488-491
    for r, s in ((3, 5.0), (3, 0.7), (4, 2.0)):
        k = sm.gaussian_kernel1d(r, s)
        assert k.dtype == np.float32 and len(k) == 2 * r + 1
        assert abs(float(k.sum()) - 1) < 1e-6 and np.allclose(k, k[::-1])
    k3 = sm.gaussian_kernel3d(3, 5.0)
    assert k3.shape == (7, 7, 7) and abs(float(k3.sum()) - 1) < 1e-5
    k1 = sm.gaussian_kernel1d(3, 5.0)
    np.testing.assert_allclose(k3, np.einsum("a,b,c->abc", k1, k1, k1), rtol=2e-6)   # separable


def test_constant_video_gives_ones():              # smoothed == const, / max -> 1 (data_utils.py:520)
    v = np.full((2, 8, 9, 8, 1), 0.37, np.float32)
    np.testing.assert_allclose(sm.temporal_convolution(v, 5.0), 1.0, rtol=1e-6)
    np.testing.assert_allclose(sm.gaussian_convolution3D_separable(v, 5.0), 1.0, rtol=1e-6)


def test_reflect_border_does_not_repeat_the_edge():   # data_utils.py:512-513: tf.pad REFLECT
    v = np.zeros((1, 1, 8, 1, 1), np.float32)
    v[0, 0, :, 0, 0] = np.arange(8)
    w = sm.gaussian_kernel1d(3, 5.0)
    s = sm._conv_axis(v, w, 2)[0, 0, :, 0, 0]
    padded = np.array([3, 2, 1, 0, 1, 2, 3, 4, 5, 6, 7, 6, 5, 4], np.float32)
    np.testing.assert_allclose(s, np.correlate(padded, w, mode="valid"), rtol=1e-6)
    np.testing.assert_allclose(s, ndimage.correlate1d(v[0, 0, :, 0, 0], w, mode="mirror"), rtol=1e-6)


@pytest.mark.parametrize("C", [1, 3])
def test_dense_3d_equals_separable_and_scipy(C):      # data_utils.py:552-582
    rng = np.random.default_rng(C)
    v = rng.random((2, 9, 8, 10, C), dtype=np.float32)
    dense = sm.gaussian_convolution3D(v, 2.0, normalise=False)
    sep = sm.gaussian_convolution3D_separable(v, 2.0, normalise=False)
    np.testing.assert_allclose(dense, sep, rtol=2e-5, atol=2e-6)
    k = sm.gaussian_kernel3d(3, 2.0)
    for b in range(2):
        for c in range(C):
            ref = ndimage.correlate(v[b, :, :, :, c].astype(np.float64), k.astype(np.float64), mode="mirror")
            np.testing.assert_allclose(dense[b, :, :, :, c], ref, rtol=2e-5, atol=2e-6)
    out = sm.gaussian_convolution3D(v, 2.0)
    assert abs(float(out.max()) - 1) < 1e-6


def test_sigma_schedule():                          # data_utils.py:584-586
    assert sm.annealing_sigma(5.0, 0) == 5.0
    assert abs(sm.annealing_sigma(5.0, 500) - 5.0 * 0.975) < 1e-12
    assert abs(sm.annealing_sigma(5.0, 250) - 5.0 * 0.975 ** 0.5) < 1e-12


def test_torch_flavour_matches_numpy():
    import torch
    from oracle import smoothing_torch as st
    v = np.random.default_rng(5).random((2, 9, 8, 10, 3), dtype=np.float32)
    np.testing.assert_allclose(st.smooth(torch.from_numpy(v), 2.0, 3, (2,)).numpy(),
                               sm.temporal_convolution(v, 2.0), rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(st.smooth(torch.from_numpy(v), 5.0, 3, (2, 1, 3)).numpy(),
                               sm.gaussian_convolution3D_separable(v, 5.0), rtol=1e-5, atol=1e-6)
