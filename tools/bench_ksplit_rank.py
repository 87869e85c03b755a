#!/usr/bin/env python3
"""Per-rank device time of the contraction-sharded protocol's two sliced kernels on ONE GPU (no collectives): the fp64
Gram sums + finalize of a [B, K/G] slice and the video gradient of all B samples on that slice.
usage: bench_ksplit_rank.py  (configs[1] and configs[4] shapes, G = 1, 2, 4, 8)"""
import ctypes, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from kccotgan_amd import _lib
from kccotgan_amd._lib import lib, ptr, check

dev = torch.device("cuda:0")


def timeit(fn, reps):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


for name, B, K, T, J, reps in (("configs[1]", 64, 122880, 30, 8, 50), ("configs[4]", 512, 2359296, 48, 8, 3)):
    for G in (1, 2, 4, 8):
        Ks = K // G
        real = torch.rand(B, Ks, device=dev)
        fake = (real + 0.05 * torch.randn(B, Ks, device=dev)).clamp_(0, 1)
        f = [torch.rand(B, T, J, device=dev) for _ in range(4)]
        C3 = torch.empty(3, B, B, device=dev); g3 = torch.randn(3, B, B, device=dev) * 1e-3
        dfake = torch.empty(B, Ks, device=dev)
        wsb = int(lib.kccot_pairwise_cost3_workspace_bytes(B, Ks)); ws = torch.empty(wsb, dtype=torch.uint8, device=dev)
        wbb = int(lib.kccot_pairwise_cost3_bwd_workspace_bytes(B, Ks)); wb = torch.empty(wbb, dtype=torch.uint8, device=dev)

        def sums():
            check(lib.kccot_pairwise_cost3_f32(ptr(real), ptr(fake), B, Ks, 1 / 15, ptr(f[0]), ptr(f[1]), ptr(f[2]), ptr(f[3]), T, J,
                                               _lib.COST_GRAM_SUMS_ONLY, ptr(C3), ws.data_ptr(), wsb, None), "sums")

        def fin():
            check(lib.kccot_pairwise_cost3_f32(ptr(real), ptr(fake), B, Ks, 1 / 15, ptr(f[0]), ptr(f[1]), ptr(f[2]), ptr(f[3]), T, J,
                                               _lib.COST_FROM_GRAM_SUMS, ptr(C3), ws.data_ptr(), wsb, None), "fin")

        def bwd():
            check(lib.kccot_pairwise_cost3_bwd_f32(ptr(g3), ptr(real), ptr(fake), B, Ks, 1 / 15, None, None, None, None, 1, 1,
                                                   ptr(dfake), None, None, None, None, wb.data_ptr(), wbb, None), "bwd")

        print(json.dumps(dict(shape=name, B=B, G=G, Ks=Ks, gram_sums_us=timeit(sums, reps), finalize_us=timeit(fin, reps),
                              video_grad_us=timeit(bwd, reps))), flush=True)
        del real, fake, dfake, ws, wb
        torch.cuda.empty_cache()
