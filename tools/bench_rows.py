#!/usr/bin/env python3
"""Per-rank device time of the batch-sharded loss path at configs[1] on ONE GPU (no collectives): what rank 0 of a
G-rank job launches between its all-gathers -- cost3_rows, the replicated divergence forward / backward, cost3_bwd_rows.
usage: bench_rows.py [G ...]   (default 1 2 4 8)"""
import os, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["KCCOT_OPTIONS"] = "sinkhorn_shortcut=0"
import torch
import bench
from kccotgan_amd.dist import HipOps as H

dev = torch.device("cuda:0")
inp, t = bench.make_inputs(bench.SHAPE["B"], 0, dev)
B = t["real"].shape[0]
real, fake = t["real"].reshape(B, -1).contiguous(), t["fake"].reshape(B, -1).contiguous()
f = [t[k].contiguous() for k in ("h_fake", "h_real", "m_real", "m_fake")]
g = torch.ones((), device=dev)


def timeit(fn, reps=50):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


for G in [int(a) for a in sys.argv[1:]] or [1, 2, 4, 8]:
    rc = B // G
    blk = H.cost3_rows(real, fake, *f, bench.SC, 0, rc)
    C3 = torch.cat([H.cost3_rows(real, fake, *f, bench.SC, r * rc, rc) for r in range(G)], dim=1).contiguous()
    loss, saved = H.divergence_fwd(C3, 1.0, 100)
    dC3 = H.divergence_bwd(saved, g)
    out = dict(G=G, rows=rc,
               cost3_rows_us=timeit(lambda: H.cost3_rows(real, fake, *f, bench.SC, 0, rc)),
               divergence_fwd_us=timeit(lambda: H.divergence_fwd(C3, 1.0, 100)),
               divergence_bwd_us=timeit(lambda: H.divergence_bwd(saved, g)),
               cost3_bwd_rows_us=timeit(lambda: H.cost3_bwd_rows(dC3, real, fake, *f, bench.SC, 0, rc)),
               loss=float(loss))
    out["sum_us"] = sum(v for k, v in out.items() if k.endswith("_us"))
    print(json.dumps(out), flush=True)
