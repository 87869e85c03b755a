#!/usr/bin/env python3
"""Kernel-only timing of KernelSmoothing backward (temporal and 3-D), HIP events around `reps` calls.
usage: bench_smooth_bwd.py [B H T W C [temporal|conv3d]]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from kccotgan_amd import _lib
from kccotgan_amd._lib import lib, ptr, check
B, H, T, W, C = (int(a) for a in sys.argv[1:6]) if len(sys.argv) > 5 else (64, 64, 30, 64, 1)
ONLY = sys.argv[6] if len(sys.argv) > 6 else None            # "temporal" or "conv3d": time that call only (PMC passes)
x = torch.rand(B, H, T, W, C, device="cuda"); g = torch.randn_like(x)
o = torch.empty_like(x); d = torch.empty_like(x); m = torch.empty(1, device="cuda")
wsb = int(lib.kccot_smooth_workspace_bytes(B, H, T, W, C))
wst = torch.empty(wsb, dtype=torch.uint8, device="cuda")
n = x.numel()
for name, axes in (("temporal", _lib.SMOOTH_T), ("conv3d", _lib.SMOOTH_T | _lib.SMOOTH_H | _lib.SMOOTH_W)):
    if ONLY and name != ONLY:
        continue
    check(lib.kccot_smooth_fwd_f32(ptr(x), B, H, T, W, C, 5.0, 3, axes, ptr(o), ptr(m), wst.data_ptr(), wsb, None), "fwd")
    def run(): check(lib.kccot_smooth_bwd_f32(ptr(g), ptr(o), ptr(m), B, H, T, W, C, 5.0, 3, axes, ptr(d), wst.data_ptr(), wsb, None), "bwd")
    for _ in range(5): run()
    torch.cuda.synchronize()
    reps = 100
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): run()
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / reps * 1e3
    print("bwd %-8s B=%d %dx%dx%d T=%d: %.1f us  (%.2f TB/s of 3 tensors)" % (name, B, H, W, C, T, us, 12.0 * n / us / 1e6))
