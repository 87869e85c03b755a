#!/usr/bin/env python3
"""Graph capture of the multi-CU Sinkhorn solve (128 < n <= 1024): replays with and without host syncs in between."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from kccotgan_amd import _lib
from kccotgan_amd.dist import HipOps as H
dev = torch.device("cuda:0")
n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
_lib.set_option("sinkhorn_shortcut", 0)
g = torch.Generator(device=dev).manual_seed(1)
C3 = torch.rand((3, n, n), device=dev, generator=g) * 50
one = torch.ones((), device=dev)
def step():
    loss, saved = H.divergence_fwd(C3, 1.0, 100)
    dC3 = H.divergence_bwd(saved, one)
    return loss, dC3, saved[3]
l0, d0, n0 = step(); torch.cuda.synchronize()
print("eager", float(l0), n0.tolist())
side = torch.cuda.Stream(device=dev)
side.wait_stream(torch.cuda.current_stream(dev))
with torch.cuda.stream(side):
    step()
torch.cuda.current_stream(dev).wait_stream(side)
torch.cuda.synchronize()
gr = torch.cuda.CUDAGraph()
with torch.cuda.graph(gr):
    gl, gd, gn = step()
for i in range(3):
    gr.replay(); torch.cuda.synchronize()
    print("synced replay", i, float(gl), gn.tolist(), bool(torch.equal(gd, d0)))
for i in range(10):
    gr.replay()
torch.cuda.synchronize()
print("after 10 back-to-back", float(gl), gn.tolist(), bool(torch.equal(gd, d0)))
for i in range(3):
    gr.replay(); torch.cuda.synchronize()
    print("synced replay again", i, float(gl), gn.tolist(), bool(torch.equal(gd, d0)))
l1, d1, n1 = step(); torch.cuda.synchronize()
print("eager again", float(l1), n1.tolist(), bool(torch.equal(d1, d0)))
