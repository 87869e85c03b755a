"""Drop-in for the ``KernelSmoothing`` class of the reference's ``data_utils.py`` (:478-586):
Gaussian smoothing of [B,H,T,W,C] videos by the HIP kernels behind ``include/kccot.h``.

Only the part of ``data_utils.py`` that sits on the loss path is mirrored; dataset loaders,
plotting and learning-rate schedules are out of scope.
"""
import torch

from . import _lib
from ._lib import lib, check, ptr, stream_of, workspace

__all__ = ["KernelSmoothing"]


class _Smooth(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, sigma, radius, axes):
        B, H, T, W, C = x.shape
        out = _lib.empty_like(x)
        mx = _lib.empty((1,), torch.float32, x.device)
        ws, wsb = workspace(lib.kccot_smooth_workspace_bytes(B, H, T, W, C), x)
        check(lib.kccot_smooth_fwd_f32(ptr(x), B, H, T, W, C, sigma, radius, axes, ptr(out), ptr(mx), ws, wsb,
                                       stream_of(x)), "smooth_fwd")
        ctx.save_for_backward(out, mx)
        ctx.cfg = (sigma, radius, axes)
        return out

    @staticmethod
    def backward(ctx, g):
        out, mx = ctx.saved_tensors
        sigma, radius, axes = ctx.cfg
        B, H, T, W, C = out.shape
        g = g.contiguous()
        din = _lib.empty_like(out)
        ws, wsb = workspace(lib.kccot_smooth_workspace_bytes(B, H, T, W, C), out)
        check(lib.kccot_smooth_bwd_f32(ptr(g), ptr(out), ptr(mx), B, H, T, W, C, sigma, radius, axes, ptr(din),
                                       ws, wsb, stream_of(out)), "smooth_bwd")
        return din, None, None, None


class _SmoothSharded(torch.autograd.Function):
    """The same smoothing for a batch that is sharded over the ranks of ``group`` (each rank holds B/G samples): the
    reference divides by the maximum of the WHOLE smoothed batch (data_utils.py:520,573,581), so the local maxima are
    all-reduced(MAX) before the division, and the adjoint of that division needs two sums over the whole batch
    (sum(g * out) and the number of arg-max ties), all-reduced(SUM) -- SURVEY.md section 8(e), message (4).  Results
    equal the unsharded call on the concatenated batch (tests/test_dist_gloo.py)."""

    @staticmethod
    def forward(ctx, x, sigma, radius, axes, group):
        import torch.distributed as dist
        B, H, T, W, C = x.shape
        out = _lib.empty_like(x)
        mx = _lib.empty((1,), torch.float32, x.device)
        ws, wsb = workspace(lib.kccot_smooth_workspace_bytes(B, H, T, W, C), x)
        check(lib.kccot_smooth_fwd_f32(ptr(x), B, H, T, W, C, sigma, radius, axes | _lib.SMOOTH_NO_DIVIDE, ptr(out), ptr(mx),
                                       ws, wsb, stream_of(x)), "smooth_fwd")
        _all_reduce(mx, dist.ReduceOp.MAX, group)
        check(lib.kccot_smooth_fwd_f32(ptr(x), B, H, T, W, C, sigma, radius, axes | _lib.SMOOTH_EXTERNAL_MAX, ptr(out), ptr(mx),
                                       ws, wsb, stream_of(x)), "smooth_fwd")
        ctx.save_for_backward(out, mx)
        ctx.cfg = (sigma, radius, axes, group)
        return out

    @staticmethod
    def backward(ctx, g):
        import torch.distributed as dist
        out, mx = ctx.saved_tensors
        sigma, radius, axes, group = ctx.cfg
        B, H, T, W, C = out.shape
        g = g.contiguous()
        din = _lib.empty_like(out)
        stats = _lib.empty((2,), torch.float32, out.device)
        ws, wsb = workspace(lib.kccot_smooth_workspace_bytes(B, H, T, W, C), out)
        check(lib.kccot_smooth_bwd_sharded_f32(ptr(g), ptr(out), ptr(mx), ptr(stats), B, H, T, W, C, sigma, radius,
                                               axes | _lib.SMOOTH_STATS_ONLY, None, ws, wsb, stream_of(out)), "smooth_bwd")
        _all_reduce(stats, dist.ReduceOp.SUM, group)
        check(lib.kccot_smooth_bwd_sharded_f32(ptr(g), ptr(out), ptr(mx), ptr(stats), B, H, T, W, C, sigma, radius,
                                               axes | _lib.SMOOTH_EXTERNAL_STATS, ptr(din), ws, wsb, stream_of(out)),
              "smooth_bwd")
        return din, None, None, None, None


def _all_reduce(t, op, group):
    import torch.distributed as dist
    if dist.get_backend(group) == "gloo" and t.is_cuda:       # CPU rehearsal backend: staged through the host
        h = t.cpu()
        dist.all_reduce(h, op=op, group=group)
        t.copy_(h)
    else:
        dist.all_reduce(t, op=op, group=group)


def _video(x):
    if x.dim() != 5:
        raise ValueError("expected a [B,H,T,W,C] video tensor, got shape %s" % (tuple(x.shape),))
    _lib.require_gpu(x)
    x = x if x.dtype == torch.float32 else x.float()
    return x.contiguous()


class KernelSmoothing:
    """data_utils.py:478-586.  Same constructor and method names as the reference."""

    def __init__(self, temporal_kernel_size=6, spatial_kernel_size=8, group=None, sharded=False):
        """``sharded=True`` (or a ``group``): the inputs are this rank's shard of a batch spread over the ranks of
        ``group`` (default group if None); the global maximum and its adjoint are all-reduced (_SmoothSharded)."""
        self.temporal_radius = temporal_kernel_size // 2   # data_utils.py:480
        self.spatial_radius = spatial_kernel_size // 2     # data_utils.py:481
        self.group, self.sharded = group, bool(sharded or group is not None)

    def _apply(self, inputs, sigma, radius, axes):
        if self.sharded:
            return _SmoothSharded.apply(_video(inputs), float(sigma), radius, axes, self.group)
        return _Smooth.apply(_video(inputs), float(sigma), radius, axes)

    def gaussian_kernel1d(self, radius, sigma):
        """data_utils.py:483-491 (host-side helper; the kernels compute the same taps)."""
        x = torch.arange(-radius, radius + 1, dtype=torch.float32)
        k = torch.exp(torch.tensor(-0.5 / (sigma * sigma), dtype=torch.float32) * x ** 2)
        return k / k.sum()

    def gaussian_kernel3d(self, radius, sigma):
        """data_utils.py:493-501: shape [2r+1, 2r+1, 2r+1, 1, 1], normalised."""
        x = torch.arange(-radius, radius + 1, dtype=torch.float32)
        xx, yy, zz = torch.meshgrid(x, x, x, indexing="xy")
        k = torch.exp(torch.tensor(-0.5 / (sigma * sigma), dtype=torch.float32) * (xx ** 2 + yy ** 2 + zz ** 2))
        return (k / k.sum())[:, :, :, None, None]

    def temporal_convolution(self, inputs, sigma):
        """data_utils.py:503-521: 1-D Gaussian along T (REFLECT), then / global max."""
        return self._apply(inputs, sigma, self.temporal_radius, _lib.SMOOTH_T)

    def spatial_convolution(self, inputs, sigma):
        """NOT reference behaviour.  The reference's 2-D path (data_utils.py:523-550) convolves
        VALID without padding and then reshapes the shrunken result to the input shape, which
        raises for every input; there is nothing to be compatible with.  Provided as the
        consistent extension: 2-D Gaussian over (H, W) with REFLECT borders, then / global max."""
        return self._apply(inputs, sigma, self.spatial_radius, _lib.SMOOTH_H | _lib.SMOOTH_W)

    def gaussian_convolution3D(self, inputs, sigma):
        """data_utils.py:552-582: 3-D Gaussian over (T, H, W), all with the SPATIAL radius
        (data_utils.py:553,562-564), REFLECT borders, then / global max."""
        return self._apply(inputs, sigma, self.spatial_radius, _lib.SMOOTH_T | _lib.SMOOTH_H | _lib.SMOOTH_W)

    def annealing_sigma(self, init_sigma, step, decay_steps=500, decay_rate=0.975):
        """data_utils.py:584-586."""
        return init_sigma * decay_rate ** (step / decay_steps)
