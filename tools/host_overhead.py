#!/usr/bin/env python3
"""Where does the host spend its time per loss evaluation, and what does a captured graph replay cost?"""
import cProfile, os, pstats, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import torch
import bench
from kccotgan_amd import gan_utils as G

dev = torch.device("cuda", 0)
inp, t = bench.make_inputs(64, 0, dev)
for k in ("fake", "h_fake", "h_real", "m_real", "m_fake"):
    t[k].requires_grad_(True)
step = lambda: bench.loss_step(G, t)
for _ in range(20): step()
torch.cuda.synchronize()
N = 500
t0 = time.perf_counter()
for _ in range(N): step()
t_issue = time.perf_counter() - t0
torch.cuda.synchronize()
t_all = time.perf_counter() - t0
print("eager: issue %.1f us/step, complete %.1f us/step" % (t_issue / N * 1e6, t_all / N * 1e6))
pr = cProfile.Profile(); pr.enable()
for _ in range(N): step()
pr.disable(); torch.cuda.synchronize()
pstats.Stats(pr).sort_stats("tottime").print_stats(18)

# graph capture of forward + backward
g = torch.cuda.CUDAGraph()
s = torch.cuda.Stream()
s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    for _ in range(3): step()
torch.cuda.current_stream().wait_stream(s)
torch.cuda.synchronize()
with torch.cuda.graph(g):
    loss, grads = step()
torch.cuda.synchronize()
for _ in range(20): g.replay()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(N): g.replay()
torch.cuda.synchronize()
print("graph replay: %.1f us/step  loss %.6f" % ((time.perf_counter() - t0) / N * 1e6, float(loss)))
l2, g2 = step()
print("eager loss %.6f  grad diff %g" % (float(l2), float((g2[0] - grads[0]).abs().max())))
