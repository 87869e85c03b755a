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


def init_dist():
    """--dist: one process per rank (RANK / WORLD_SIZE / LOCAL_RANK / MASTER_* from the environment, as bench.py's children get
    them); KCCOT_TRAIN_DIST_BACKEND=gloo is the one-GPU rehearsal (all ranks on cuda:0, host-staged collectives)."""
    import torch.distributed as dist
    backend = os.environ.get("KCCOT_TRAIN_DIST_BACKEND", "nccl")
    local = int(os.environ.get("LOCAL_RANK", "0"))
    ndev = torch.cuda.device_count()
    idx = local if (backend == "nccl" or local < ndev) else local % max(ndev, 1)
    torch.cuda.set_device(idx)
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    if backend == "nccl":
        dist.init_process_group("nccl", device_id=torch.device("cuda", idx))
    else:
        dist.init_process_group(backend)
    return dist, "cuda:%d" % idx


def run(kernel, iters=5, B=64, dist=None, device="cuda:0"):
    tr = KCCOTTrainer(B, total_time_steps=30, int_time_steps=5, x_height=64, x_width=64, channels=1, kernel=kernel,
                      device=device)
    gen = torch.Generator(device=device).manual_seed(100 + (dist.get_rank() if dist else 0))
    x = torch.rand(B, 64, 30, 64, 1, device=device, generator=gen)          # this rank's shard of the global batch
    sync = (lambda: (torch.cuda.synchronize(), dist.barrier(), torch.cuda.synchronize())) if dist else torch.cuda.synchronize
    t0 = time.perf_counter()
    tr.train_iteration(x)
    sync()
    first = time.perf_counter() - t0
    print("warm-up iteration (MIOpen solver selection included): %.1f s" % first, file=sys.stderr, flush=True)
    t0 = time.perf_counter()
    for i in range(iters):
        pm, loss = tr.train_iteration(x)
        if i == 0:
            torch.cuda.synchronize()
            print("first timed iteration: %.2f s" % (time.perf_counter() - t0), file=sys.stderr, flush=True)
    sync()
    dt = (time.perf_counter() - t0) / iters
    if dist:                                   # the slowest rank's clock
        tt = torch.tensor([dt], dtype=torch.float64, device=device if dist.get_backend() == "nccl" else "cpu")
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt)
    return dict(kernel=kernel, world=(dist.get_world_size() if dist else 1), per_rank_batch=B, ms_per_train_step=dt * 1e3, train_steps_per_sec=1 / dt, pm=float(pm), loss=float(loss),
                iterations=iters, first_iteration_s=first, find_mode=os.environ.get("MIOPEN_FIND_MODE", "default"))


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("kernels", nargs="*")
    ap.add_argument("--json", action="store_true")
    ap.add_argument("--iters", type=int, default=5)
    ap.add_argument("--kernel", default=None)
    ap.add_argument("--batch", type=int, default=64, help="per-rank batch")
    ap.add_argument("--dist", action="store_true", help="data-parallel: one process per rank, global-batch loss (kccotgan_amd.dist)")
    a = ap.parse_args()
    d, device = init_dist() if a.dist else (None, "cuda:0")
    for k in ([a.kernel] if a.kernel else (a.kernels or ["none", "3d"])):
        r = run(k, a.iters, a.batch, d, device)
        if d is None or d.get_rank() == 0:
            print(json.dumps(r), flush=True)
    if d is not None:
        d.barrier()
        d.destroy_process_group()
