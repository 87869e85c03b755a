"""Calibration of the gradient tolerance: HIP gradient error against the fp64 autograd oracle, next to the
oracle's own fp32-vs-fp64 gap (tests/golden/grad_gap.json).  GPU box: python tools/grad_err.py"""
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import cases  # noqa: E402
from kccotgan_amd import _lib as L, gan_utils as G  # noqa: E402
from oracle import gan_utils_torch as ot  # noqa: E402

WRT = ("fake", "h_fake", "h_real", "m_real", "m_fake")
gaps = json.load(open(os.path.join(ROOT, "tests", "golden", "grad_gap.json")))["gaps"]
out = {}
for shape, seed, regime in cases.CASES:
    name = cases.case_name(shape, seed, regime)
    inp = cases.gen_inputs(shape, seed, regime)
    if shape == "cfg2":
        g = np.load(os.path.join(ROOT, "tests", "golden", "grad_%s.npz" % name))
        ref = {k: g[k] for k in WRT if k != "fake"}
    else:
        d = {k: torch.from_numpy(v).double() for k, v in inp.items()}
        for k in WRT:
            d[k].requires_grad_(True)
        val = ot.compute_sinkhorn_loss(d["real"], d["fake"], cases.SC, 0.8, 100, d["h_fake"], d["m_real"], d["h_real"], d["m_fake"])
        ref = dict(zip(WRT, [a.numpy() for a in torch.autograd.grad(val, [d[k] for k in WRT])]))
    K = inp["real"][0].size
    for path in ("auto", "direct", "mfma", "mfma_f32"):
        if path.startswith("mfma") and (K % 4 or K < 32):
            continue
        L.set_option("gram_f32", 1 if path == "mfma_f32" else 0)
        G.cost_flags = {"auto": 0, "direct": L.COST_FORCE_DIRECT, "mfma": L.COST_FORCE_MFMA, "mfma_f32": L.COST_FORCE_MFMA}[path]
        t = {k: torch.from_numpy(v).cuda() for k, v in inp.items()}
        for k in WRT:
            t[k].requires_grad_(True)
        loss = G.compute_sinkhorn_loss(t["real"], t["fake"], cases.SC, 0.8, 100, t["h_fake"], t["m_real"], t["h_real"], t["m_fake"])
        grads = dict(zip(WRT, [a.cpu().numpy().astype(np.float64) for a in torch.autograd.grad(loss, [t[k] for k in WRT])]))
        errs = {}
        for k in WRT:
            if k == "fake" and shape == "cfg2":
                df = grads[k].reshape(grads[k].shape[0], -1)
                errs[k] = float(np.abs(df[:, ::97] - g["dfake_strided"]).max() / float(g["dfake_absmax"]))
            else:
                errs[k] = float(np.abs(grads[k] - ref[k]).max() / np.abs(ref[k]).max())
        out["%s/%s" % (name, path)] = errs
        print("%-16s %-9s " % (name, path) + "  ".join("%s %.1e (x%.0f)" % (k, errs[k], errs[k] / max(gaps[name][k], 1e-12)) for k in WRT), flush=True)
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
json.dump(out, open(os.path.join(ROOT, "gpurun_out", "grad_err.json"), "w"), indent=1)
