#!/usr/bin/env python3
"""Kernel-only timing of the video gradient of a row block (kccot_pairwise_cost3_bwd_rows_f32): the whole batch on one
GPU at the configs[2] shape, and a rank's rows of the batch-sharded loss at configs[2] / [3] / [4]; one-launch tiles
(option apply_m256 = 1, default) against the 64-row block form (= 0)."""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from kccotgan_amd import _lib
from kccotgan_amd.dist import HipOps as H

dev = torch.device("cuda:0")
CASES = [("configs[2] whole batch", 128, 368640, 128), ("configs[2] on 4 ranks", 128, 368640, 32),
         ("configs[3] on 8 ranks", 256, 368640, 32), ("configs[4] on 8 ranks", 512, 2359296, 64)]
if len(sys.argv) > 1 and sys.argv[1] == "big":      # the whole batch at configs[4] (apply_coeffs_x3_m256n128) and configs[3]
    CASES = [("configs[4] whole batch", 512, 2359296, 512), ("configs[3] whole batch", 256, 368640, 256)]
for name, B, K, rows in CASES:
    real = torch.rand(B, K, device=dev); fake = (real + 0.05 * torch.randn(B, K, device=dev)).clamp_(0, 1)
    f = [torch.rand(B, 30, 8, device=dev) for _ in range(4)]
    g3 = torch.randn(3, B, B, device=dev)
    rec = {"case": name, "B": B, "K": K, "rows": rows}
    for mode in ((1,) if rows >= 256 else (1, 0)):
        with _lib.options(apply_m256=mode):
            run = lambda: H.cost3_bwd_rows(g3, real, fake, f[0], f[1], f[2], f[3], 1 / 15.0, 0, rows)
            out = run(); torch.cuda.synchronize()
            reps = 3 if K > 1000000 else 10
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(reps): run()
            e1.record(); torch.cuda.synchronize()
            rec["one_launch_ms" if mode else "block_form_ms"] = e0.elapsed_time(e1) / reps
            rec["sum_%d" % mode] = float(out[0].double().abs().sum())
    print(json.dumps(rec), flush=True)
    del real, fake
