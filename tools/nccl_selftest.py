#!/usr/bin/env python3
"""RCCL plumbing check on a one-GPU box: world_size 1 over the nccl backend (two ranks cannot share a card under
RCCL).  Exercises exactly the calls bench.py / kccotgan_amd.dist make at N > 1 -- init_process_group with a
device id, all_gather_into_tensor / all_reduce(MAX) / barrier on device tensors -- and one sharded loss step.
Launch: python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29511 tools/nccl_selftest.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import torch.distributed as dist

os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29511")
os.environ.setdefault("RANK", "0")
os.environ.setdefault("WORLD_SIZE", "1")
dev = torch.device("cuda", int(os.environ.get("LOCAL_RANK", "0")))
torch.cuda.set_device(dev)
dist.init_process_group("nccl", device_id=dev)
x = torch.arange(12, dtype=torch.float32, device=dev).reshape(3, 4)
out = torch.empty((dist.get_world_size() * 3, 4), device=dev)
dist.all_gather_into_tensor(out, x)
assert torch.equal(out, x)
m = torch.tensor([3.5], dtype=torch.float64, device=dev)
dist.all_reduce(m, op=dist.ReduceOp.MAX)
dist.barrier()
torch.cuda.synchronize()
from kccotgan_amd import dist as kd
import bench
inp, t = bench.make_inputs(bench.SHAPE["B"], 0, dev)
for k in ("fake", "h_fake", "h_real", "m_real", "m_fake"):
    t[k].requires_grad_(True)
shard = kd.shard_batch(t, 0, 1)
loss, grads = kd.sharded_loss_step(shard, bench.SC)
torch.cuda.synchronize()
from kccotgan_amd.graph import GraphedShardedStep
gs = GraphedShardedStep(shard, bench.SC)
for _ in range(3):
    gl, gg = gs()
torch.cuda.synchronize()
assert torch.equal(gl.reshape(()), loss.detach().reshape(())), (float(gl), float(loss))
assert all(torch.equal(gg[k], g) for k, g in zip(("fake", "h_fake", "h_real", "m_real", "m_fake"), grads))
import time
t0 = time.perf_counter()
for _ in range(100):
    gs()
torch.cuda.synchronize()
print("graphed sharded step (world 1, nccl calls in place): %.1f us/step; coalesced input gathers: %s" % (
    (time.perf_counter() - t0) / 100 * 1e6, getattr(gs, "_coalesce", None)))
# the contraction-sharded protocol: all_to_all_single / all_reduce(SUM, fp64) in place of the all-gathers
from kccotgan_amd.graph import GraphedKSplitStep
lk = kd.sharded_sinkhorn_loss(shard["real"], shard["fake"], bench.SC, shard["h_fake"], shard["m_real"], shard["h_real"],
                              shard["m_fake"], protocol="ksplit")
gk = torch.autograd.grad(lk, [shard[k] for k in ("fake", "h_fake", "h_real", "m_real", "m_fake")])
ks = GraphedKSplitStep(shard, bench.SC)
for _ in range(3):
    kl, kg = ks()
torch.cuda.synchronize()
assert torch.equal(kl.reshape(()), lk.detach().reshape(())), (float(kl), float(lk))
assert all(torch.equal(kg[k], g) for k, g in zip(("fake", "h_fake", "h_real", "m_real", "m_fake"), gk))
assert abs(float(lk) - float(loss)) <= 2e-6 * abs(float(loss))
t0 = time.perf_counter()
for _ in range(100):
    ks()
torch.cuda.synchronize()
print("graphed ksplit step (world 1, nccl calls in place): %.1f us/step" % ((time.perf_counter() - t0) / 100 * 1e6))
# the chunked video gather's RCCL branch: one async all_gather_into_tensor per column range, waited for range by range
vid = torch.rand(8, 1024, device=dev)
bounds = kd.gather_chunk_bounds(1024, 3)
pieces = kd._gather_columns_async(vid, bounds, None, force_collective=True)
assert len(pieces) == len(bounds) >= 2 and all(w is not None for w, _ in pieces)
for (w, full), (a, b) in zip(pieces, bounds):
    w.wait()
    assert torch.equal(full, vid[:, a:b])
torch.cuda.synchronize()
# raw collectives the contraction-sharded protocol uses, on device tensors
a2a_in = torch.arange(8, dtype=torch.float32, device=dev)
a2a_out = torch.empty_like(a2a_in)
dist.all_to_all_single(a2a_out, a2a_in)
s64 = torch.tensor([1.25, 2.5], dtype=torch.float64, device=dev)
dist.all_reduce(s64, op=dist.ReduceOp.SUM)
torch.cuda.synchronize()
assert torch.equal(a2a_out, a2a_in) and s64.tolist() == [1.25, 2.5]
# sharded kernel smoothing: all_reduce(MAX) of the maxima, all_reduce(SUM) of the adjoint's two sums (RCCL, fp32)
from kccotgan_amd.data_utils import KernelSmoothing
v = torch.rand(2, 16, 12, 16, 1, device=dev)
outs = {}
for sharded in (False, True):
    xv = v.clone().requires_grad_(True)
    o3 = KernelSmoothing(6, 6, sharded=sharded).gaussian_convolution3D(xv, 2.0)
    o3.backward(torch.ones_like(o3) * 0.5 + v)
    outs[sharded] = (o3.detach(), xv.grad)
torch.cuda.synchronize()
# (the sharded call runs the two-phase kernels, the plain one the streamed walk kernels: same taps, another fma order)
assert float((outs[True][0] - outs[False][0]).abs().max()) <= 1e-6
_d = (outs[True][1] - outs[False][1]).abs()
_i = int(_d.argmax())
print("sharded smoothing adjoint: max|diff| %.3g at flat index %d (values %.6g vs %.6g), max|grad| %.3g, #elements differing > 1e-6 max: %d" % (
    float(_d.max()), _i, float(outs[True][1].reshape(-1)[_i]), float(outs[False][1].reshape(-1)[_i]), float(outs[False][1].abs().max()),
    int((_d > 1e-6 * outs[False][1].abs().max()).sum())))
assert float(_d.max()) <= 2e-5 * float(outs[False][1].abs().max())
print("nccl selftest ok: backend=%s loss=%.6f" % (dist.get_backend(), float(loss)))
dist.destroy_process_group()
