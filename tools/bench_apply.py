#!/usr/bin/env python3
"""Kernel-only timing of the video-gradient stage (kccot_pairwise_cost3_bwd_f32) at a large batch.
usage: bench_apply.py [B H T W C]; options through KCCOT_OPTIONS (apply_m256=0, apply_f32=1)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from kccotgan_amd._lib import lib, ptr, workspace, check
B, H, T, W, C = (int(a) for a in sys.argv[1:6]) if len(sys.argv) > 5 else (256, 64, 30, 64, 3)
K = H * T * W * C
dev = "cuda"
real = torch.rand(B, K, device=dev); fake = torch.rand(B, K, device=dev)
g3 = torch.randn(3, B, B, device=dev) * 1e-3
dfake = torch.empty_like(fake)
ws, wsb = workspace(lib.kccot_pairwise_cost3_bwd_workspace_bytes(B, K), real)
def run(): check(lib.kccot_pairwise_cost3_bwd_f32(ptr(g3), ptr(real), ptr(fake), B, K, 1 / 15.0, None, None, None, None, 1, 1, ptr(dfake), None, None, None, None, ws, wsb, None), "bwd")
for _ in range(2): run()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
reps = 5 if B <= 256 else 2
e0.record()
for _ in range(reps): run()
e1.record(); torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / reps
fl = 2.0 * B * 2 * B * K * 6
print("B=%d K=%d: %.3f ms  (%.2f PFLOP/s bf16 executed)  KCCOT_OPTIONS=%s" % (B, K, ms, fl / ms / 1e12, os.environ.get("KCCOT_OPTIONS", "")))
