"""Gradients at the FULL BASELINE sizes configs[2..4] against the fp64 oracle ("what's weak" 2).

The kernels that only run at these sizes -- the 256-row video-gradient tiles, the large-batch Gram, the cooperative
multi-CU reverse sweep at n = 256 / 512 -- reached the oracle only through equalities at K <= 3200.  Here, at the real
K (368 640 / 2 359 296) and the real n:
  (a) the reverse sweep: d loss / d C3 of the GPU against fp64 autograd through the unrolled loop of
      oracle.sinkhorn_from_cost on the GPU's own cost matrices (kernel_train.py:287-289 differentiates that loop);
  (b) the feature gradients dh_*, dM_* IN FULL against the fp64 chain rule of the causal term (gan_utils.py:34-38)
      applied to the GPU's own dC3;
  (c) 32 sampled rows x ~2000 strided columns of dfake against the fp64 formula
          2 sc sum_i dCxy[i,j] (y_j - x_i) + 2 sc sum_i (dCyy[i,j] + dCyy[j,i]) (y_j - y_i)
      evaluated on the rows involved (gan_utils.py:14-17 differentiated);
  (d) the one-call loss path (what compute_sinkhorn_loss runs) against the staged path that exposes dC3.
"""
import numpy as np
import pytest
import torch

import cases
from oracle import gan_utils_torch as ot

pytestmark = pytest.mark.gpu
DEV = "cuda:0"

FULL_SIZE = [((128, 64, 30, 64, 3), 100),                                     # BASELINE configs[2]
             ((256, 64, 30, 64, 3), 200), ((512, 128, 48, 128, 3), 300)]      # configs[3] (L = 200), configs[4] (L = 300)
FEATS = ("h_fake", "m_real", "h_real", "m_fake")
GRAD_TOL_FLOOR = 2.5e-5          # as tests/test_gpu_parity.py
GRAD_TOL_FACTOR = 4.0


def _inputs(shape):
    B, H, T, W, C = shape
    gen = torch.Generator(device=DEV).manual_seed(B)
    real = torch.rand(shape, device=DEV, generator=gen)
    fake = (real + 0.05 * torch.randn(shape, device=DEV, generator=gen)).clamp_(0, 1)
    f = {k: torch.rand((B, T, 8), device=DEV, generator=gen) for k in FEATS}
    return real, fake, f


def _oracle_dC(Cp, nits, dtype):
    """d cost / d C by autograd through exactly `nits` iterations of the reference loop (gan_utils.py:151-160)."""
    C = torch.from_numpy(Cp).to(dtype).requires_grad_(True)
    cost, n = ot.sinkhorn_from_cost(C, 1.0, nits, Lmin=nits)
    assert n == nits
    (g,) = torch.autograd.grad(cost, C)
    return float(cost), g.double().numpy()


@pytest.mark.parametrize("shape,Lc", FULL_SIZE)
def test_full_size_gradients_against_the_fp64_oracle(shape, Lc):
    from kccotgan_amd import gan_utils as G
    B, H, T, W, C = shape
    sc = cases.SC
    real, fake, f = _inputs(shape)
    x = real.reshape(B, -1)
    K = x.shape[1]

    # ---- (d) the production path: one library call each way
    fk = fake.clone().requires_grad_(True)
    fr = {k: f[k].clone().requires_grad_(True) for k in FEATS}
    loss = G.compute_sinkhorn_loss(real, fk, sc, 1.0, Lc, fr["h_fake"], fr["m_real"], fr["h_real"], fr["m_fake"],
                                   honor_eps_l=True)
    nits = [int(v) for v in G.last_info["compute_sinkhorn_loss"].tolist()[:3]]
    assert all(n > 0 for n in nits), nits
    prod = torch.autograd.grad(loss, [fk] + [fr[k] for k in FEATS])

    # ---- the staged path (same kernels, dC3 visible)
    y2 = fake.reshape(B, -1).clone().requires_grad_(True)
    f2 = {k: f[k].clone().requires_grad_(True) for k in FEATS}
    C3 = G._Cost3.apply(x, y2, f2["h_fake"], f2["h_real"], f2["m_real"], f2["m_fake"], sc)
    loss_s = G._SinkhornDivergence.apply(C3, 1.0, Lc, 100, "staged")
    assert [int(v) for v in G.last_info["staged"].tolist()] == nits
    (dC3,) = torch.autograd.grad(loss_s, C3, retain_graph=True)
    staged = torch.autograd.grad(loss_s, [y2] + [f2[k] for k in FEATS])
    assert abs(float(loss) - float(loss_s)) <= 1e-6 * abs(float(loss_s))
    for name, a, b in zip(("fake",) + FEATS, prod, staged):
        scale = float(b.abs().max())
        assert float((a.reshape(b.shape) - b).abs().max()) <= 2e-6 * scale, name

    # ---- (a) reverse sweep at the real n against fp64 autograd through the unrolled loop
    C3n = C3.detach().double().cpu().numpy()
    dC3n = dC3.double().cpu().numpy()
    weights = (2.0, -1.0, -1.0)                                   # gan_utils.py:225
    costs = []
    for p in range(3):
        c64, g64 = _oracle_dC(C3n[p], nits[p], torch.float64)
        _, g32 = _oracle_dC(C3n[p], nits[p], torch.float32)
        costs.append(c64)
        scale = np.abs(g64).max()
        gap = np.abs(g32 - g64).max() / scale                     # the oracle's own fp32-vs-fp64 distance on THIS problem
        tol = max(GRAD_TOL_FLOOR, GRAD_TOL_FACTOR * gap)
        err = np.abs(dC3n[p] / weights[p] - g64).max() / scale
        print("dC3[%d] n=%d nits=%d: err %.2e (oracle fp32 gap %.2e, tol %.2e)" % (p, B, nits[p], err, gap, tol))
        assert err <= tol, (p, err, tol)
    ref_loss = 2.0 * costs[0] - costs[1] - costs[2]
    assert abs(float(loss) - ref_loss) <= 1e-4 * abs(ref_loss)

    # ---- (b) feature gradients in full: fp64 chain rule of the causal term on the GPU's dC3
    d = {k: f[k].double().cpu().requires_grad_(True) for k in FEATS}
    w3 = torch.from_numpy(dC3n)

    def causal(h, M):                                              # gan_utils.py:34-38, rows <- h, columns <- M
        return torch.einsum("itk,jtk->ij", h[:, :-1, :], M[:, 1:, :] - M[:, :-1, :]) * sc

    lin = (w3[0] * causal(d["h_fake"], d["m_real"])).sum() + (w3[1] * causal(d["h_real"], d["m_real"])).sum() \
        + (w3[2] * causal(d["h_fake"], d["m_fake"])).sum()
    gref = torch.autograd.grad(lin, [d[k] for k in FEATS])
    for k, a, b in zip(FEATS, staged[1:], gref):
        scale = float(b.abs().max())
        err = float((a.double().cpu() - b).abs().max()) / scale
        print("d%s: err %.2e" % (k, err))
        assert err <= GRAD_TOL_FLOOR, (k, err)

    # ---- (c) sampled rows x strided columns of dfake at the full K
    rng = np.random.default_rng(B)
    rows = np.sort(rng.choice(B, size=32, replace=False))
    stride = 181 if K < 1_000_000 else 1151
    cols = torch.arange(int(rng.integers(0, stride)), K, stride, device=DEV)
    Xc = x.index_select(1, cols).double().cpu().numpy()            # [B, ncol]
    Yc = fake.reshape(B, -1).index_select(1, cols).double().cpu().numpy()
    got = staged[0].index_select(1, cols)[torch.from_numpy(rows).to(DEV)].double().cpu().numpy()
    S = dC3n[2] + dC3n[2].T
    ref = np.empty_like(got)
    for r, j in enumerate(rows):
        wxy, wyy = dC3n[0][:, j], S[:, j]
        ref[r] = 2.0 * sc * ((wxy.sum() + wyy.sum()) * Yc[j] - wxy @ Xc - wyy @ Yc)
    scale = np.abs(ref).max()
    err = np.abs(got - ref).max() / scale
    print("dfake sample (%d rows x %d columns of K = %d): err %.2e" % (len(rows), len(cols), K, err))
    assert err <= GRAD_TOL_FLOOR, err
    # per-row too: a row whose gradient is small must not hide behind the largest one
    row_scale = np.abs(ref).max(axis=1)
    row_err = np.abs(got - ref).max(axis=1) / np.maximum(row_scale, 1e-30)
    assert row_err.max() <= 4 * GRAD_TOL_FLOOR, row_err.max()
