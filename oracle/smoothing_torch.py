"""Torch flavour of the KernelSmoothing oracle (autograd gives the gradient oracle, including
the arg-max path of the global-max normalisation).  TEST INFRASTRUCTURE ONLY; pinned against
oracle/smoothing_np.py in tests/test_oracle_smoothing.py."""
import torch


def gaussian_kernel1d(radius, sigma, dtype=torch.float32):
    x = torch.arange(-radius, radius + 1, dtype=dtype)
    k = torch.exp(torch.tensor(-0.5 / (sigma * sigma), dtype=dtype) * x ** 2)
    return k / k.sum()


def _reflect_index(n, r):
    idx = torch.arange(-r, n + r)
    idx = torch.where(idx < 0, -idx, idx)
    return torch.where(idx >= n, 2 * (n - 1) - idx, idx)


def conv_axis(v, w, axis):
    r = (len(w) - 1) // 2
    n = v.shape[axis]
    vp = v.index_select(axis, _reflect_index(n, r))
    out = 0
    for d in range(2 * r + 1):
        out = out + w[d] * vp.narrow(axis, d, n)
    return out


def smooth(v, sigma, radius, axes, normalise=True):
    """v: [B,H,T,W,C]; axes: subset of (2, 1, 3) = (T, H, W)."""
    w = gaussian_kernel1d(radius, sigma, v.dtype)
    s = v
    for a in axes:
        s = conv_axis(s, w, a)
    return s / s.max() if normalise else s
