#!/usr/bin/env python3
"""Kernel-only timing of the Sinkhorn forward / reverse sweep at configs[1] size (three 64x64 problems).
usage: bench_sinkhorn.py [near|far] ; options through KCCOT_OPTIONS (sinkhorn_shortcut=0, sinkhorn_lanes_per_line=8, ...)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from kccotgan_amd._lib import lib, ptr
regime = sys.argv[1] if len(sys.argv) > 1 else "near"
g = np.load(os.path.join(ROOT, "tests", "golden", "cfg2_s0_near.npz" if regime == "near" else "cfg2_s1_far.npz"))
C = torch.from_numpy(np.stack([g["C_xy"], g["C_xx"], g["C_yy"]])).cuda().contiguous()
n, L = C.shape[1], 100
uh = torch.empty(3, L, n, device="cuda"); vh = torch.empty(3, L, n, device="cuda")
cost = torch.empty(3, device="cuda"); nits = torch.zeros(6, dtype=torch.int32, device="cuda")
gc = torch.tensor([2.0, -1.0, -1.0], device="cuda"); dC = torch.empty_like(C)
def fwd(): assert lib.kccot_sinkhorn_fwd_f32(ptr(C), 3, n, 1.0, L, 100, 1e-2, 0, ptr(uh), ptr(vh), ptr(cost), ptr(nits), None, None, 0, None) == 0
def bwd(): assert lib.kccot_sinkhorn_bwd_f32(ptr(C), ptr(uh), ptr(vh), ptr(nits), 3, n, 1.0, L, ptr(gc), ptr(dC), None, 0, None) == 0
def timeit(f, reps=200):
    for _ in range(10): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3
tf = timeit(fwd); tb = timeit(bwd)
print("%s: fwd %.1f us  bwd %.1f us  [counts | fwd executed] %s  cost %s" % (regime, tf, tb, nits.tolist(), [round(c, 4) for c in cost.tolist()]))
if lib.kccot_sinkhorn_fused_eligible(n, L):
    loss = torch.empty(1, device="cuda"); ticket = torch.zeros(1, dtype=torch.int32, device="cuda"); dCu = torch.empty_like(C)
    def fused(): assert lib.kccot_sinkhorn_divergence_fused_f32(ptr(C), n, 1.0, L, 100, 1e-2, ptr(cost), ptr(nits), ptr(loss), ptr(ticket), ptr(dCu), None) == 0
    tfu = timeit(fused)
    bwd(); torch.cuda.synchronize()
    print("%s: fused solve+sweep %.1f us (%.3f us per half-step over %d)  LPR=%s  max|dC_fused - dC_two_kernel| %.3g of %.3g" % (
        regime, tfu, tfu / (4 * max(nits[3:].tolist())), 4 * max(nits[3:].tolist()), os.environ.get("KCCOT_OPTIONS", "defaults"),
        float((dCu - dC).abs().max()), float(dC.abs().max())))
