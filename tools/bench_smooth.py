#!/usr/bin/env python3
"""Time KernelSmoothing (fwd and fwd+bwd) at the configs[1] video shape; report HBM fraction.
Algorithmic bytes per call: one read + one write of the tensor (2*B*K*4)."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from kccotgan_amd.data_utils import KernelSmoothing

shape = (64, 64, 30, 64, 1)
x = torch.rand(shape, device="cuda")
ks = KernelSmoothing(6, 6)
nbytes = x.numel() * 4


def timeit(fn, reps=50):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


res = {}
for name, fn in (("temporal", ks.temporal_convolution), ("3d", ks.gaussian_convolution3D)):
    us = timeit(lambda: fn(x, 5.0))
    xg = x.clone().requires_grad_(True)
    w = torch.rand_like(x)

    def fb():
        y = fn(xg, 5.0)
        (g,) = torch.autograd.grad(y, xg, w)
        return g
    us_fb = timeit(fb)
    res[name] = dict(fwd_us=us, fwd_bwd_us=us_fb, fwd_alg_GBs=2 * nbytes / us / 1e3, fwd_hbm_frac=2 * nbytes / us / 1e3 / 8000)
print(json.dumps(res))
