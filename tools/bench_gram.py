#!/usr/bin/env python3
"""Kernel-only timing of the cost-assembly stage (kccot_pairwise_cost3_f32) at a large batch.
usage: bench_gram.py [B H T W C]; options through KCCOT_OPTIONS (e.g. cost_tiled=0)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from kccotgan_amd._lib import lib, ptr, workspace, check
B, H, T, W, C = (int(a) for a in sys.argv[1:6]) if len(sys.argv) > 5 else (256, 64, 30, 64, 3)
K = H * T * W * C
dev = "cuda"
real = torch.rand(B, K, device=dev); fake = (real + 0.05 * torch.randn(B, K, device=dev)).clamp_(0, 1)
f = [torch.rand(B, T, 8, device=dev) for _ in range(4)]
C3 = torch.empty(3, B, B, device=dev)
ws, wsb = workspace(lib.kccot_pairwise_cost3_workspace_bytes(B, K), real)
def run(): check(lib.kccot_pairwise_cost3_f32(ptr(real), ptr(fake), B, K, 1 / 15.0, ptr(f[0]), ptr(f[1]), ptr(f[2]), ptr(f[3]), T, 8, 0, ptr(C3), ws, wsb, None), "cost3")
if os.environ.get("KCCOT_PRIME_LIB"):
    # timing ablations that skip a store (tools/micro/q256_estore_ablate.sh): run ANOTHER build of the library once on the same
    # workspace first, so that what the ablated build reads back is real data, not the zero pages of a fresh allocation
    # (zero operands switch less, the chip clocks higher: MI355X_MICROARCH.md, DVFS give-back)
    import ctypes
    from kccotgan_amd import _lib as _L
    prime = ctypes.CDLL(os.environ["KCCOT_PRIME_LIB"])
    fn = prime.kccot_pairwise_cost3_f32
    fn.argtypes, fn.restype = _L.lib.kccot_pairwise_cost3_f32.argtypes, ctypes.c_int
    assert fn(ptr(real), ptr(fake), B, K, 1 / 15.0, ptr(f[0]), ptr(f[1]), ptr(f[2]), ptr(f[3]), T, 8, 0, ptr(C3), ws, wsb, None) == 0
    torch.cuda.synchronize()
for _ in range(2): run()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
reps = 5 if B <= 256 else 2
e0.record()
for _ in range(reps): run()
e1.record(); torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / reps
fl = 2.0 * B * B * K * 2.0 * 6          # useful: X X^T and E E^T upper triangles + X E^T, six bf16 products per fp32 product
print("B=%d K=%d: cost stage %.3f ms  (%.2f PFLOP/s bf16 useful, workspace %.2f GB)  KCCOT_OPTIONS=%s" % (
    B, K, ms, fl / ms / 1e12, wsb / 1e9, os.environ.get("KCCOT_OPTIONS", "")))
