#!/usr/bin/env python3
"""Same-box A/B of the configs[1] cost stage (kccot_pairwise_cost3_f32, B = 64, K = 122 880) between library builds.

usage: ab_cost_stage.py [reps]     -- the library is the one KCCOT_LIB_PATH names (default: the package's)
Prints one JSON line: us per launch of the Gram partial kernel alone (KCCOT_COST_PARTIAL_ONLY), of the whole stage as eager
back-to-back launches, of the whole stage replayed from a hipGraph holding 20 of them, and of the one-call loss forward +
backward (GraphedLossStep, the bench's step).  tools/ab_cost_stage.sh alternates builds on one box (boxes of the pool differ
by +-8 % on these kernels, so only same-box pairs are comparable)."""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from kccotgan_amd import _lib
from kccotgan_amd._lib import lib, ptr, workspace, check, stream_of

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 200
B, H, T, W, C, J = 64, 64, 30, 64, 1, 8
K = H * T * W * C
dev = torch.device("cuda", 0)
gen = torch.Generator(device=dev).manual_seed(0)
real = torch.rand(B, K, device=dev, generator=gen)
fake = (real + 0.05 * torch.randn(B, K, device=dev, generator=gen)).clamp_(0, 1)
f = [torch.rand(B, T, J, device=dev, generator=gen) for _ in range(4)]
C3 = torch.empty(3, B, B, device=dev)
ws, wsb = workspace(lib.kccot_pairwise_cost3_workspace_bytes(B, K), real)


def launch(flags):
    check(lib.kccot_pairwise_cost3_f32(ptr(real), ptr(fake), B, K, 1 / 15.0, ptr(f[0]), ptr(f[1]), ptr(f[2]), ptr(f[3]), T, J, flags,
                                       ptr(C3), ws, wsb, stream_of(real)), "cost3")


def timed(fn, n, warm=10):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n


out = {"lib": os.path.basename(_lib.LIB_PATH), "options": os.environ.get("KCCOT_OPTIONS", ""), "version": int(lib.kccot_version())}
out["partial_us"] = timed(lambda: launch(_lib.COST_PARTIAL_ONLY), reps)
out["stage_eager_us"] = timed(lambda: launch(0), reps)
side = torch.cuda.Stream(device=dev)
side.wait_stream(torch.cuda.current_stream(dev))
with torch.cuda.stream(side):
    launch(0)
torch.cuda.current_stream(dev).wait_stream(side)
torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    for _ in range(20):
        launch(0)
out["stage_graph_us"] = timed(g.replay, max(reps // 20, 5), 3) / 20
ref = C3.clone()
launch(0)
torch.cuda.synchronize()
out["replay_equals_eager"] = bool(torch.equal(ref, C3))

# the bench's step: one-call loss forward + backward as a hipGraph
from kccotgan_amd.graph import GraphedLossStep
t = {"real": real.reshape(B, H, T, W, C), "fake": fake.reshape(B, H, T, W, C), "h_fake": f[0], "h_real": f[1], "m_real": f[2], "m_fake": f[3]}
with _lib.options(sinkhorn_shortcut=0):
    step = GraphedLossStep(t, 1 / 15.0)
    out["loss_step_graph_us"] = timed(step.graph.replay, reps, 10)
    out["loss"] = float(step.loss)
print(json.dumps(out))
