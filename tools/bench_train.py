#!/usr/bin/env python3
"""Time the full training iteration (disc step + gen step, kernel_train.py:313-314) at the
configs[1] shape: B=64, T=30 (5 context + 25 predicted), 64x64x1, filter sizes 8, z 128.

    python tools/bench_train.py [none|1d|3d ...]                 # human-readable, 5 timed iterations each
    python tools/bench_train.py --json --iters 3 --kernel none   # one JSON line (what bench.py's child process runs)

MIOPEN_FIND_MODE=2 in the environment selects MIOpen's fast find (10 s to the first iteration instead of ~290 s)."""
import argparse, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import kccotgan_amd  # noqa: F401  (first: sets the MIOpen solver switch before any convolution can run)
import torch
if os.environ.get("KCCOT_TRAIN_NATIVE") == "1":
    torch.backends.cudnn.enabled = False      # conservative mode: no MIOpen kernel at all
from kccotgan_amd.kernel_train import KCCOTTrainer


def run(kernel, iters=5, B=64):
    tr = KCCOTTrainer(B, total_time_steps=30, int_time_steps=5, x_height=64, x_width=64, channels=1, kernel=kernel,
                      device="cuda:0")
    x = torch.rand(B, 64, 30, 64, 1, device="cuda:0")
    t0 = time.perf_counter()
    tr.train_iteration(x)
    torch.cuda.synchronize()
    first = time.perf_counter() - t0
    print("warm-up iteration (MIOpen solver selection included): %.1f s" % first, file=sys.stderr, flush=True)
    t0 = time.perf_counter()
    for i in range(iters):
        pm, loss = tr.train_iteration(x)
        if i == 0:
            torch.cuda.synchronize()
            print("first timed iteration: %.2f s" % (time.perf_counter() - t0), file=sys.stderr, flush=True)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / iters
    return dict(kernel=kernel, ms_per_train_step=dt * 1e3, train_steps_per_sec=1 / dt, pm=float(pm), loss=float(loss),
                iterations=iters, first_iteration_s=first, find_mode=os.environ.get("MIOPEN_FIND_MODE", "default"))


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("kernels", nargs="*")
    ap.add_argument("--json", action="store_true")
    ap.add_argument("--iters", type=int, default=5)
    ap.add_argument("--kernel", default=None)
    a = ap.parse_args()
    for k in ([a.kernel] if a.kernel else (a.kernels or ["none", "3d"])):
        print(json.dumps(run(k, a.iters)), flush=True)
