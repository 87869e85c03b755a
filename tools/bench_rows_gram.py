#!/usr/bin/env python3
"""Per-rank device time of the row-block cost stage of the batch-sharded loss at a large BASELINE config on ONE GPU (no
collectives): the Gram row block on the matrix pipe (kccot_pairwise_cost3_rows_gram_f32 + the rank's share of the norm
pass) against the direct-difference kernel (kccot_pairwise_cost3_rows_f32), and their agreement.
usage: bench_rows_gram.py [B H T W C G]   (default configs[4]: 512 128 48 128 3 on 8 ranks)"""
import os, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from kccotgan_amd.dist import HipOps as H

B, Hh, T, W, C, G = (int(a) for a in sys.argv[1:7]) if len(sys.argv) > 6 else (512, 128, 48, 128, 3, 8)
K = Hh * T * W * C
dev = torch.device("cuda:0")
gen = torch.Generator(device=dev).manual_seed(B)
real = torch.rand((B, K), device=dev, generator=gen)
fake = (real + 0.05 * torch.randn((B, K), device=dev, generator=gen)).clamp_(0, 1)
f = [torch.rand((B, T, 8), device=dev, generator=gen) for _ in range(4)]
sc = 1.0 / 15.0
m = B // G


def timeit(fn, reps):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


assert H.rows_gram_supported(m, B, K), (m, B, K)
norms = H.row_norms(real, fake)                                      # all ranks' rows (what the all-gather delivers)
reps = 3 if B >= 512 else 10
t_norm = timeit(lambda: H.row_norms(real[:m], fake[:m]), reps)        # a rank's own share
t_gram = timeit(lambda: H.cost3_rows(real, fake, *f, sc, m, m, norms), reps)
t_direct = timeit(lambda: H.cost3_rows(real, fake, *f, sc, m, m), reps)
a = H.cost3_rows(real, fake, *f, sc, m, m, norms)
b = H.cost3_rows(real, fake, *f, sc, m, m)
err = float((a - b).abs().max() / b.abs().max())
diag = torch.arange(m, device=dev)
derr = float(((a[0][diag, m + diag] - b[0][diag, m + diag]).abs() / b[0][diag, m + diag].abs()).max())
print(json.dumps(dict(B=B, K=K, G=G, rows=m, rows_gram_ms=t_gram, own_row_norms_ms=t_norm, rows_direct_ms=t_direct,
                      max_abs_diff_over_max=err, xy_diagonal_rel_diff=derr)), flush=True)
