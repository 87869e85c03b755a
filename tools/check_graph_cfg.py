#!/usr/bin/env python3
"""Eager launches against hipGraph replay of one loss forward + backward at a BASELINE config (bench.py's `configs` block).
usage: check_graph_cfg.py configs[3] eager|graph   (run under rocprofv3 --kernel-trace --stats to see what a replay executes)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench
from kccotgan_amd import gan_utils as G, _lib
from kccotgan_amd.graph import GraphedLossStep
name, mode = sys.argv[1], sys.argv[2]
B, H, T, W, C, L, _ = bench.OTHER_CONFIGS[name]
dev = torch.device("cuda:0")
_lib.set_option("sinkhorn_shortcut", 0)
t = bench.config_inputs(B, H, T, W, C, dev)
for k in bench.WRT:
    t[k].requires_grad_(True)
def step():
    loss = G.compute_sinkhorn_loss(t["real"], t["fake"], bench.SC, 1.0, L, t["h_fake"], t["m_real"], t["h_real"], t["m_fake"], honor_eps_l=True)
    return loss, torch.autograd.grad(loss, [t[k] for k in bench.WRT])
loss, grads = step()
torch.cuda.synchronize()
if mode == "graph":
    gs = GraphedLossStep(t, bench.SC, 1.0, L, warmup=1, honor_eps_l=True, clone=False)
    run = lambda: gs()
    gl, gg = gs()
    torch.cuda.synchronize()
    print("graph loss", float(gl), "eager loss", float(loss), "grad equal:", all(torch.equal(gg[k], g) for k, g in zip(bench.WRT, grads)),
          "nits", gs.nits.tolist())
else:
    run = step
print(mode, bench.event_stats(run, 10))
if mode == "graph":
    for i in range(4):
        gl, gg = gs()
        torch.cuda.synchronize()
        print("replay", i, "loss", float(gl), "grad equal:", [bool(torch.equal(gg[k], g)) for k, g in zip(bench.WRT, grads)],
              "finite:", bool(torch.isfinite(gg["fake"]).all()), "nits", gs.nits.tolist(), gs.nits_executed.tolist())
