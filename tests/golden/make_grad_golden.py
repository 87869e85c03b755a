#!/usr/bin/env python3
"""Gradient fixtures and the measured fp32-vs-fp64 gradient gap of the ORACLE.

    python tests/golden/make_grad_golden.py            # build container, ~10 min, CPU only

The reference differentiates ``compute_sinkhorn_loss`` with tf.GradientTape through the unrolled
Sinkhorn loop (kernel_train.py:252,287-289; gan_utils.py:149-160).  TensorFlow cannot run here,
so the gradient oracle is torch autograd through ``oracle/gan_utils_torch.py`` (the restatement
that tests/test_oracle_golden.py pins to the reference-generated loss fixtures).  This script

 1. writes ``grad_<case>.npz`` for the BASELINE configs[1] full-size cases (B = 64, K = 122 880):
    fp64 autograd of the as-called loss w.r.t. fake, h_fake, h_real, m_real, m_fake -- the four
    feature gradients in full, and for the 31 MB video gradient its per-sample L2 norms and sums,
    NPROJ seeded random projections per sample and a strided sample (every STRIDE-th entry).
    The [B,B,T,D] broadcast would retain 3 x 4 GB for the tape; the same chain rule is applied in
    two autograd stages instead: (i) d loss / d C through the unrolled loop, (ii) column chunks of
    the cost matrices re-built with grad and back-propagated with the matching dC columns.
 2. measures, on every golden case, how far the oracle's own **fp32** autograd (the arithmetic
    the reference's tape runs in) sits from the fp64 one: ``grad_gap.json`` holds
    max|g32 - g64| / max|g64| per tensor.  tests/test_gpu_parity.py derives its gradient
    tolerance from these numbers (GRAD_TOL_FACTOR x gap, floor 1e-5) instead of asserting one.
"""
import json
import os
import sys
import time

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, HERE)
sys.path.insert(0, ROOT)
import cases  # noqa: E402
from oracle import gan_utils_torch as ot  # noqa: E402

WRT = ("fake", "h_fake", "h_real", "m_real", "m_fake")
NPROJ, STRIDE, CHUNK = 8, 97, 8


def projections(K, seed=1234):
    return np.random.default_rng(seed).standard_normal((NPROJ, K)).astype(np.float64)


def staged_grads(inp, dtype):
    """d loss / d (fake, h_fake, h_real, m_real, m_fake) by two-stage autograd (see module docstring)."""
    t = {k: torch.from_numpy(v).to(dtype) for k, v in inp.items()}
    x, y = ot.flatten_video(t["real"]), ot.flatten_video(t["fake"])
    probs = dict(xy=(x, y, "h_fake", "m_real"), xx=(x, x, "h_real", "m_real"), yy=(y, y, "h_fake", "m_fake"))
    sign = dict(xy=2.0, xx=-1.0, yy=-1.0)
    with torch.no_grad():
        C = {k: ot.modified_cost(a, b, t[h], t[m], cases.SC, chunk=CHUNK) for k, (a, b, h, m) in probs.items()}
    for k in C:
        C[k].requires_grad_(True)
    loss = sum(sign[k] * ot.sinkhorn_from_cost(C[k])[0] for k in C)
    dC = dict(zip(C, torch.autograd.grad(loss, list(C.values()))))
    for k in WRT:
        t[k].requires_grad_(True)
    B = x.shape[0]
    for k, (_, _, h, m) in probs.items():
        for j0 in range(0, B, CHUNK):
            xf, yf = ot.flatten_video(t["real"]), ot.flatten_video(t["fake"])
            a = xf
            b = yf if k in ("xy", "yy") else xf
            a = yf if k == "yy" else a
            Cc = ot.cost_xy(a, b[j0:j0 + CHUNK], cases.SC) + ot.causal_term(t[h], t[m][j0:j0 + CHUNK], cases.SC)
            Cc.backward(dC[k][:, j0:j0 + CHUNK])
    return float(loss), {k: t[k].grad.numpy() for k in WRT}


def full_grads(inp, dtype):
    t = {k: torch.from_numpy(v).to(dtype) for k, v in inp.items()}
    for k in WRT:
        t[k].requires_grad_(True)
    loss = ot.compute_sinkhorn_loss(t["real"], t["fake"], cases.SC, 0.8, 100, t["h_fake"], t["m_real"], t["h_real"],
                                    t["m_fake"])
    g = torch.autograd.grad(loss, [t[k] for k in WRT])
    return float(loss), {k: v.numpy() for k, v in zip(WRT, g)}


def gap(g32, g64):
    return {k: float(np.abs(g32[k].astype(np.float64) - g64[k]).max() / np.abs(g64[k]).max()) for k in WRT}


def main():
    torch.set_num_threads(8)
    gaps = {}
    for shape, seed, regime in cases.CASES:
        name = cases.case_name(shape, seed, regime)
        inp = cases.gen_inputs(shape, seed, regime)
        t0 = time.time()
        fn = staged_grads if shape == "cfg2" else full_grads
        l64, g64 = fn(inp, torch.float64)
        l32, g32 = fn(inp, torch.float32)
        gaps[name] = gap(g32, g64)
        if shape in ("small", "tiny"):          # the two-stage route is the same chain rule: check it where both run
            _, gs = staged_grads(inp, torch.float64)
            for k in WRT:
                assert np.abs(gs[k] - g64[k]).max() <= 1e-10 * np.abs(g64[k]).max(), (name, k)
        gold = np.load(os.path.join(HERE, name + ".npz"))
        assert abs(l64 - float(gold["loss_f64"])) <= 1e-9 * abs(float(gold["loss_f64"])), (l64, float(gold["loss_f64"]))
        print("%-16s loss %.6f  gap32 %s  %.0fs" % (name, l64, {k: "%.1e" % v for k, v in gaps[name].items()},
                                                   time.time() - t0), flush=True)
        if shape == "cfg2":
            B = inp["fake"].shape[0]
            df = g64["fake"].reshape(B, -1)
            res = {k: g64[k] for k in WRT if k != "fake"}
            res.update(dfake_norm=np.sqrt((df ** 2).sum(1)), dfake_sum=df.sum(1), dfake_absmax=np.abs(df).max(),
                       dfake_proj=df @ projections(df.shape[1]).T, dfake_strided=df[:, ::STRIDE].astype(np.float32),
                       loss_f64=np.asarray(l64), loss_f32_oracle=np.asarray(l32))
            np.savez_compressed(os.path.join(HERE, "grad_%s.npz" % name), **res)
    with open(os.path.join(HERE, "grad_gap.json"), "w") as f:
        json.dump({"what": "max|g_fp32 - g_fp64| / max|g_fp64| of torch autograd through oracle/gan_utils_torch.py "
                           "(unrolled loop, eps = 1, L = 100), per golden case and gradient",
                   "gaps": gaps}, f, indent=1, sort_keys=True)


if __name__ == "__main__":
    main()
