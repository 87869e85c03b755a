#!/usr/bin/env python3
"""gram partial kernel time vs K (fixed B=64): separates fixed cost from per-stage cost."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from kccotgan_amd import _lib
from kccotgan_amd._lib import lib, ptr, stream_of, workspace, check
B, T, J = 64, 30, 8
h = [torch.rand(B, T, J, device="cuda") for _ in range(4)]
for K in (15360, 30720, 61440, 122880, 245760, 491520, 983040):
    real = torch.rand(B, K, device="cuda"); fake = torch.rand(B, K, device="cuda")
    C3 = torch.empty(3, B, B, device="cuda")
    ws, wsb = workspace(lib.kccot_pairwise_cost3_workspace_bytes(B, K), real)
    def launch():
        check(lib.kccot_pairwise_cost3_f32(ptr(real), ptr(fake), B, K, 1/15, ptr(h[0]), ptr(h[1]), ptr(h[2]), ptr(h[3]), T, J,
                                           _lib.COST_PARTIAL_ONLY, ptr(C3), ws, wsb, stream_of(real)), "c3")
    for _ in range(5): launch()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(100): launch()
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 100 * 1e3
    print("K=%7d  %.1f us   %.2f TB/s algorithmic" % (K, us, 2 * B * K * 4 / us / 1e6), flush=True)
