#!/usr/bin/env python3
"""Kernel-only timing of the Sinkhorn solvers for 128 < n <= 1024: the multi-CU flag-in-data kernels vs the
one-workgroup streaming kernels (option sinkhorn_coop).  Also checks that the variants agree.
usage: bench_sinkhorn_n.py [n ...]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from kccotgan_amd._lib import lib, ptr

from kccotgan_amd import _lib
VARIANTS = {"multi-CU, problem per XCD": dict(sinkhorn_coop=1, sinkhorn_coop_xcd=1), "multi-CU, 2-D grid": dict(sinkhorn_coop=1, sinkhorn_coop_xcd=0),
            "multi-CU, 2-D grid + failing XCD check": dict(sinkhorn_coop=1, sinkhorn_coop_xcd=2),
            "one workgroup (streaming)": dict(sinkhorn_coop=0)}


def run(n, L=100):
    rng = np.random.default_rng(n)
    C = torch.from_numpy((rng.random((3, n, n), dtype=np.float32) * 3)).cuda().contiguous()
    uh = torch.empty(3, L, n, device="cuda"); vh = torch.empty(3, L, n, device="cuda")
    cost = torch.empty(3, device="cuda"); nits = torch.zeros(6, dtype=torch.int32, device="cuda")
    gc = torch.tensor([2.0, -1.0, -1.0], device="cuda"); dC = torch.empty_like(C)
    wsb = lib.kccot_sinkhorn_workspace_bytes(3, n)
    ws = torch.empty(wsb // 4 + 64, dtype=torch.float32, device="cuda")
    def fwd(): assert lib.kccot_sinkhorn_fwd_f32(ptr(C), 3, n, 1.0, L, 100, 1e-2, 0, ptr(uh), ptr(vh), ptr(cost), ptr(nits), None, ptr(ws), wsb, None) == 0
    def bwd(): assert lib.kccot_sinkhorn_bwd_f32(ptr(C), ptr(uh), ptr(vh), ptr(nits), 3, n, 1.0, L, ptr(gc), ptr(dC), ptr(ws), wsb, None) == 0
    def timeit(f, reps=20):
        for _ in range(3): f()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps): f()
        e1.record(); torch.cuda.synchronize()
        return e0.elapsed_time(e1) / reps * 1e3
    res = {}
    for name, kv in VARIANTS.items():
        with _lib.options(**kv):
            tf = timeit(fwd); tb = timeit(bwd)
        res[name] = (cost.cpu().numpy().copy(), nits.cpu().numpy().copy(), uh.cpu().numpy().copy(), dC.cpu().numpy().copy())
        print("n=%d %-26s fwd %8.1f us (%.2f us/iter)  bwd %8.1f us  nits %s cost %s" % (
            n, name, tf, tf / L, tb, nits.tolist()[:3], [round(c, 5) for c in cost.tolist()]), flush=True)
    names = list(VARIANTS)
    print("   the two multi-CU block maps: bit-identical %s" % all(bool((a == b).all()) for k in (1, 2) for a, b in zip(res[names[0]], res[names[k]])), flush=True)
    ref, r = res[names[0]], res[names[3]]
    print("   streaming vs multi-CU: cost rel %.2e  nits equal %s  u_hist max|d| %.2e  dC max|d|/max %.2e  finite %s" % (
        float(np.abs(r[0] - ref[0]).max() / np.abs(ref[0]).max()), bool((r[1] == ref[1]).all()),
        float(np.abs(r[2] - ref[2]).max()), float(np.abs(r[3] - ref[3]).max() / np.abs(ref[3]).max()),
        bool(np.isfinite(r[3]).all())), flush=True)


for n in [int(a) for a in sys.argv[1:]] or [256, 512, 1024, 200]:
    run(n)
