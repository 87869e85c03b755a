#!/usr/bin/env python3
"""Wall clock per sharded loss step at world_size 1 (nccl initialised, collectives short-circuited): device-bound
or host-bound?  Compare with tools/bench_rows.py's device sum for G = 1."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["KCCOT_OPTIONS"] = "sinkhorn_shortcut=0"
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29512")
os.environ.setdefault("RANK", "0"); os.environ.setdefault("WORLD_SIZE", "1")
import torch, torch.distributed as dist
dev = torch.device("cuda:0"); torch.cuda.set_device(dev)
dist.init_process_group("nccl", device_id=dev)
import bench
from kccotgan_amd import dist as kd
inp, t = bench.make_inputs(bench.SHAPE["B"], 0, dev)
shard = kd.shard_batch(t, 0, 1)
for _ in range(10): kd.sharded_loss_step(shard, bench.SC)
torch.cuda.synchronize()
for reps in (50, 50):
    t0 = time.perf_counter()
    for _ in range(reps): loss, g = kd.sharded_loss_step(shard, bench.SC)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print("issue %.1f us/step   complete %.1f us/step" % ((t1 - t0) / reps * 1e6, (t2 - t0) / reps * 1e6), flush=True)
# host cost of one small all_gather_into_tensor call (world 1 still goes through the RCCL call path)
x = torch.zeros(8, 1024, device=dev); out = torch.empty(8, 1024, device=dev)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(200): dist.all_gather_into_tensor(out, x)
t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
print("all_gather_into_tensor (world 1): issue %.1f us  complete %.1f us" % ((t1 - t0) / 200 * 1e6, (t2 - t0) / 200 * 1e6))
dist.destroy_process_group()
