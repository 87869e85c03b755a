"""CPU oracle for KernelSmoothing (reference data_utils.py:478-586), NumPy.

TEST INFRASTRUCTURE ONLY.  PINNED (round 2): tests/golden/smooth_*.npz hold the outputs of the
reference's own ``data_utils.KernelSmoothing`` -- the file imported verbatim from /root/reference by
tests/golden/make_golden_smoothing.py over ``oracle/refshim`` (TensorFlow is not installed; the
stand-in supplies tf.pad/transpose/reshape/... in NumPy and tf.nn.conv1d/conv3d as torch CPU
convolutions) -- for C = 1 and C = 3 (both code branches), sigma 5.0 / 2.0 / 1.3, radius 3 and 4,
small shapes plus the configs[0] frame size and T = 30.  tests/test_smoothing_golden.py checks
this restatement against them (fp32: 1e-6 temporal / 4e-6 conv3d absolute on outputs in [0,1];
fp64: 1e-13), the known-answer tests in tests/test_oracle_smoothing.py stay as a second net.
What is not pinned: TensorFlow/Eigen's own summation order (as for the loss path).

The dense functions follow the reference literally (pad, then a VALID correlation with the full
kernel); the ``*_separable`` ones are the algebraically identical form the HIP kernels use.
"""
import numpy as np


def gaussian_kernel1d(radius, sigma, dtype=np.float32):
    """data_utils.py:483-491."""
    x = np.arange(-radius, radius + 1).astype(dtype)
    k = np.exp(dtype(-0.5 / (sigma * sigma)) * x ** 2)
    return (k / np.sum(k)).astype(dtype)


def gaussian_kernel3d(radius, sigma, dtype=np.float32):
    """data_utils.py:493-501 (without the two trailing unit axes)."""
    x = np.arange(-radius, radius + 1).astype(dtype)
    xx, yy, zz = np.meshgrid(x, x, x)
    k = np.exp(dtype(-0.5 / (sigma * sigma)) * (xx ** 2 + yy ** 2 + zz ** 2))
    return (k / np.sum(k)).astype(dtype)


def _conv_axis(v, w, axis):
    """REFLECT-pad r on `axis` then VALID correlation with the symmetric taps w."""
    r = (len(w) - 1) // 2
    pad = [(0, 0)] * v.ndim
    pad[axis] = (r, r)
    vp = np.pad(v, pad, mode="reflect")
    out = np.zeros_like(v)
    n = v.shape[axis]
    for d in range(2 * r + 1):
        sl = [slice(None)] * v.ndim
        sl[axis] = slice(d, d + n)
        out = out + w[d] * vp[tuple(sl)]
    return out


def temporal_convolution(inputs, sigma, temporal_radius=3, dtype=np.float32):
    """data_utils.py:503-521 on [B,H,T,W,C]."""
    v = np.asarray(inputs, dtype=dtype)
    s = _conv_axis(v, gaussian_kernel1d(temporal_radius, sigma, dtype), 2)
    return s / np.max(s)


def gaussian_convolution3D_separable(inputs, sigma, spatial_radius=3, dtype=np.float32, normalise=True):
    v = np.asarray(inputs, dtype=dtype)
    w = gaussian_kernel1d(spatial_radius, sigma, dtype)
    s = _conv_axis(_conv_axis(_conv_axis(v, w, 2), w, 1), w, 3)
    return s / np.max(s) if normalise else s


def gaussian_convolution3D(inputs, sigma, spatial_radius=3, dtype=np.float32, normalise=True):
    """data_utils.py:552-582, dense: REFLECT-pad (T,H,W) by r, VALID 3-D correlation with the
    (2r+1)^3 kernel; channels are independent (the reference folds C into the batch)."""
    v = np.asarray(inputs, dtype=dtype)           # [B,H,T,W,C]
    r = spatial_radius
    k = gaussian_kernel3d(r, sigma, dtype)        # indexed [T?,H?,W?]: symmetric in all three, so order is moot
    vp = np.pad(v, [(0, 0), (r, r), (r, r), (r, r), (0, 0)], mode="reflect")
    B, H, T, W, C = v.shape
    s = np.zeros_like(v)
    for a in range(2 * r + 1):
        for b in range(2 * r + 1):
            for c in range(2 * r + 1):
                s = s + k[a, b, c] * vp[:, a:a + H, b:b + T, c:c + W, :]
    return s / np.max(s) if normalise else s


def spatial_convolution_reflect(inputs, sigma, spatial_radius=3, dtype=np.float32):
    """The package's documented EXTENSION (no reference behaviour: data_utils.py:523-550 raises)."""
    v = np.asarray(inputs, dtype=dtype)
    w = gaussian_kernel1d(spatial_radius, sigma, dtype)
    s = _conv_axis(_conv_axis(v, w, 1), w, 3)
    return s / np.max(s)


def annealing_sigma(init_sigma, step, decay_steps=500, decay_rate=0.975):
    """data_utils.py:584-586."""
    return init_sigma * decay_rate ** (step / decay_steps)
