#!/usr/bin/env python3
"""apply_q256 (256 x 256 output tiles, W panel through LDS) against the 256 x 64 / 128 tile kernels: max deviation of the video
gradient and ms per call.  usage: check_apply_q256.py [B K] ..."""
import os, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from kccotgan_amd import _lib
from kccotgan_amd._lib import lib, ptr, workspace, check
args = [int(a) for a in sys.argv[1:]] or [256, 368640]
for B, K in zip(args[0::2], args[1::2]):
    gen = torch.Generator(device="cuda").manual_seed(B + K)
    real = torch.rand(B, K, device="cuda", generator=gen)
    fake = torch.rand(B, K, device="cuda", generator=gen)
    g3 = torch.randn(3, B, B, device="cuda", generator=gen) * 1e-3
    ws, wsb = workspace(lib.kccot_pairwise_cost3_bwd_workspace_bytes(B, K), real)
    out = {}
    for mode in (0, 1):
        _lib.set_option("apply_q256", mode)
        dfake = torch.full((B, K), float("nan"), device="cuda")
        run = lambda: check(lib.kccot_pairwise_cost3_bwd_f32(ptr(g3), ptr(real), ptr(fake), B, K, 1 / 15.0, None, None, None, None, 1, 1,
                                                             ptr(dfake), None, None, None, None, ws, wsb, None), "bwd")
        run(); run()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        reps = 5 if B * K < 3e8 else 3
        e0.record()
        for _ in range(reps):
            run()
        e1.record(); torch.cuda.synchronize()
        out[mode] = (dfake.clone() if B * K < 3e8 else dfake[:, ::97].clone(), e0.elapsed_time(e1) / reps)
    scale = float(out[0][0].abs().max())
    diff = float((out[1][0] - out[0][0]).abs().max())
    print(json.dumps({"B": B, "K": K, "ms_tiles_64_128": out[0][1], "ms_q256": out[1][1], "max_abs_diff_over_max": diff / scale,
                      "finite": bool(torch.isfinite(out[1][0]).all())}), flush=True)
    del real, fake, out
    torch.cuda.empty_cache()
