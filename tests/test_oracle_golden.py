"""The CPU oracle (oracle/gan_utils_np.py) against the golden vectors produced by
the reference's own gan_utils.py (tests/golden/make_golden.py).  CPU only."""
import glob
import os

import numpy as np
import pytest

import cases
from oracle import gan_utils_np as o

GOLD = os.path.join(os.path.dirname(__file__), "golden")
SMALL_CASES = [c for c in cases.CASES if c[0] != "cfg2"]


def load(shape, seed, regime):
    g = np.load(os.path.join(GOLD, cases.case_name(shape, seed, regime) + ".npz"))
    inp = cases.gen_inputs(shape, seed, regime)
    np.testing.assert_allclose(cases.checksum(inp), g["checksum"], rtol=0, atol=0)
    return g, inp


def rel(a, b):
    return abs(float(a) - float(b)) / max(abs(float(b)), 1e-30)


def test_all_golden_files_present():
    names = {os.path.basename(p) for p in glob.glob(os.path.join(GOLD, "*.npz"))}
    want = {cases.case_name(*c) + ".npz" for c in cases.CASES} | {"line32.npz"}
    assert want <= names


@pytest.mark.parametrize("shape,seed,regime", SMALL_CASES)
@pytest.mark.parametrize("dtype", [np.float32, np.float64])
def test_loss_path_matches_reference(shape, seed, regime, dtype):
    g, inp = load(shape, seed, regime)
    sfx = "" if dtype == np.float32 else "_f64"
    tol = 2e-6 if dtype == np.float32 else 1e-12
    loss, ex = o.compute_sinkhorn_loss_ex(inp["real"], inp["fake"], cases.SC, 0.8, 100, inp["h_fake"],
                                          inp["m_real"], inp["h_real"], inp["m_fake"], dtype=dtype)
    # the three cost matrices (reference: modified_cost, gan_utils.py:21-43)
    for t in ("xy", "xx", "yy"):
        C = g["C_" + t + sfx]
        np.testing.assert_allclose(ex["C" + t], C, rtol=tol, atol=tol * np.abs(C).max())
        assert rel(ex[t], g["w_" + t + sfx]) < 50 * tol or abs(ex[t] - g["w_" + t + sfx]) < 50 * tol
    assert ex["nits"] == (int(g["nits_xy" + sfx]), int(g["nits_xx" + sfx]), int(g["nits_yy" + sfx]))
    assert rel(loss, g["loss" + sfx]) < (2e-5 if dtype == np.float32 else 1e-10)
    # quirk 1 (gan_utils.py:221-223): sinkhorn_eps / sinkhorn_l are ignored
    assert float(g["loss_eps0p1_L5" + sfx]) == float(g["loss" + sfx])
    loss2 = o.compute_sinkhorn_loss(inp["real"], inp["fake"], cases.SC, 0.1, 5, inp["h_fake"],
                                    inp["m_real"], inp["h_real"], inp["m_fake"], dtype=dtype)
    assert float(loss2) == float(loss)
    pm = o.scale_invariante_martingale_regularization(inp["m_real"], cases.LAM, cases.SC, dtype)
    assert rel(pm, g["pM" + sfx]) < 10 * tol


@pytest.mark.parametrize("shape,seed,regime", SMALL_CASES)
def test_keyword_eps_L_bicausal_benchmark(shape, seed, regime):
    g, inp = load(shape, seed, regime)
    x, y = o.flatten_video(inp["real"]), o.flatten_video(inp["fake"])
    np.testing.assert_allclose(o.cost_xy(x, y, cases.SC), g["C_plain"], rtol=2e-6)
    np.testing.assert_array_equal(o.compute_N(inp["m_real"]), g["N_m_real"])
    for eps, L in cases.EPS_L:
        key = "e%g_L%d" % (eps, L)
        w, n, _ = o.compute_sinkhorn_ex(x, y, inp["h_fake"], inp["m_real"], cases.SC, epsilon=eps, L=L)
        assert n == int(g["nits_" + key]), key
        assert rel(w, g["w_" + key]) < 2e-5, key
        assert rel(w, g["w_" + key + "_f64"]) < 1e-4, key   # fp32 vs the fp64 value of the same algorithm
    w, n, C = o.compute_sinkhorn_ex(x, y, inp["h_fake"], inp["m_real"], cases.SC, hx=inp["h_real"],
                                    My=inp["m_fake"], bi_causal=True)
    np.testing.assert_allclose(C, g["C_bicausal"], rtol=2e-6, atol=2e-6 * np.abs(C).max())
    assert n == int(g["nits_bicausal"]) and rel(w, g["w_bicausal"]) < 2e-5
    assert rel(o.benchmark_sinkhorn(x, y, cases.SC), g["w_bench_default"]) < 2e-5
    assert rel(o.benchmark_sinkhorn(x, y, cases.SC, epsilon=0.8, L=50, Lmin=20),
               g["w_bench_e0.8_L50_Lmin20"]) < 2e-5
    C = o.cost_xy(x, y, cases.SC)
    n_idx = o.sinkhorn_from_cost(C, 0.8, 50, 20, stop_on_index=True)[1]
    assert n_idx == int(g["nits_bench_e0.8_L50_Lmin20"])  # index-based stop: Lmin + 1 iterations


def test_cfg2_cost_matrices_chunked():
    """BASELINE configs[1] (B=64,T=30,64x64x1): the oracle evaluated in column chunks
    (same arithmetic, no 2 GB temporary) against the reference's full broadcast."""
    g, inp = load("cfg2", 0, "near")
    x, y = o.flatten_video(inp["real"]), o.flatten_video(inp["fake"])
    C = o.modified_cost(x[:8], y, inp["h_fake"][:8], inp["m_real"], cases.SC, chunk=8)
    np.testing.assert_allclose(C, g["C_xy"][:8], rtol=3e-6)
    w = o.sinkhorn_from_cost(g["C_xy"])[0]
    assert rel(w, g["w_xy"]) < 2e-5


@pytest.mark.parametrize("seed,regime", [(0, "near"), (1, "far")])
def test_cfg2_sinkhorn_variants_on_reference_cost_matrices(seed, regime):
    """configs[1] full size: the (eps, L) / bi-causal / benchmark_sinkhorn variants, with the reference's own
    cost matrices from the fixture as input (re-building them on the CPU takes 10 s each); cfg2_s1_far at
    eps = 0.25 runs PAST Lmin (161 iterations in the reference's fp32, 168 in its fp64 run)."""
    g, inp = load("cfg2", seed, regime)
    for eps, L in cases.EPS_L:
        key = "e%g_L%d" % (eps, L)
        w, n = o.sinkhorn_from_cost(g["C_xy"], eps, L)[:2]
        assert n == int(g["nits_" + key]) and rel(w, g["w_" + key]) < 2e-6, key
        w64, n64 = o.sinkhorn_from_cost(g["C_xy_f64"], eps, L, dtype=np.float64)[:2]
        assert n64 == int(g["nits_" + key + "_f64"]) and rel(w64, g["w_" + key + "_f64"]) < 1e-12, key
    w, n = o.sinkhorn_from_cost(g["C_bicausal"])[:2]
    assert n == int(g["nits_bicausal"]) and rel(w, g["w_bicausal"]) < 2e-6
    w, n = o.sinkhorn_from_cost(g["C_plain"], 1.0, 10, 10, stop_on_index=True)[:2]
    assert n == int(g["nits_bench_default"]) and rel(w, g["w_bench_default"]) < 2e-6
    w, n = o.sinkhorn_from_cost(g["C_plain"], 0.8, 50, 20, stop_on_index=True)[:2]
    assert n == int(g["nits_bench_e0.8_L50_Lmin20"]) and rel(w, g["w_bench_e0.8_L50_Lmin20"]) < 2e-6
    # the causal terms on top of the plain matrix (gan_utils.py:34-38,60-68), from the oracle
    caus = o.causal_term(inp["h_fake"], inp["m_real"], cases.SC)
    np.testing.assert_allclose(g["C_plain"] + caus, g["C_xy"], rtol=2e-6)
    np.testing.assert_allclose(g["C_xy"] + o.causal_term(inp["h_real"], inp["m_fake"], cases.SC), g["C_bicausal"], rtol=2e-6)


def test_quirk2_lmin_and_late_stop():
    """gan_utils.py:149-160: exactly L iterations for L <= 100; for L > 100 the loop
    stops at the first iteration >= 100 whose sum|u - u_prev| < 1e-2."""
    g = np.load(os.path.join(GOLD, "line32.npz"))
    x, y, h, M = cases.gen_line_inputs()
    for sc, L in cases.LINE_RUNS:
        key = "sc%g_L%d" % (sc, L)
        w, n, _ = o.compute_sinkhorn_ex(x, y, h, M, sc, L=L)
        assert n == int(g["nits_" + key]), key
        assert rel(w, g["w_" + key]) < 5e-5, key
    assert int(g["nits_sc100_L300"]) == 198 and int(g["nits_sc300_L300"]) == 300


# ---- known-answer tests derived from the cited reference lines ----

def test_kat_cost_diag_zero_symmetric():
    x = np.random.default_rng(3).random((6, 4, 9), dtype=np.float32)
    C = o.cost_xy(x, x, 0.3)
    assert np.all(np.diag(C) == 0) and np.array_equal(C, C.T)


def test_kat_constant_cost_uniform_plan():
    C = np.full((7, 7), 2.5, np.float32)
    cost, nits, u, v, pi = o.sinkhorn_from_cost(C, 1.0, 100)
    np.testing.assert_allclose(pi, 1 / 49, rtol=1e-5)
    assert abs(cost - 2.5) < 1e-5 and nits == 100
    assert abs(o.sinkhorn_from_cost(np.array([[3.25]], np.float32))[0] - 3.25) < 1e-6   # n = 1


def test_kat_column_marginals_after_v_update():
    C = np.random.default_rng(0).random((9, 9), dtype=np.float32) * 3
    pi = o.sinkhorn_from_cost(C, 0.7, 13)[4]
    np.testing.assert_allclose(pi.sum(0), 1 / 9, rtol=2e-6)


def test_kat_causal_orientation():
    """gan_utils.py:37: h indexes rows, M indexes columns."""
    rng = np.random.default_rng(1)
    h = np.zeros((5, 6, 3), np.float32)
    h[2] = rng.random((6, 3), dtype=np.float32)
    M = rng.random((5, 6, 3), dtype=np.float32)
    ch = o.causal_term(h, M, 1.0)
    assert np.all(ch[[0, 1, 3, 4]] == 0) and np.any(ch[2] != 0)
    M2 = np.zeros((5, 6, 3), np.float32)
    M2[3] = rng.random((6, 3), dtype=np.float32)
    cm = o.causal_term(rng.random((5, 6, 3), dtype=np.float32), M2, 1.0)
    assert np.all(cm[:, [0, 1, 2, 4]] == 0) and np.any(cm[:, 3] != 0)


def test_kat_martingale_constant_in_time_is_zero():
    M = np.repeat(np.random.default_rng(2).random((4, 1, 3), dtype=np.float32), 6, axis=1)
    assert o.scale_invariante_martingale_regularization(M, 1.0, 0.5) == 0


# ---- the torch flavour of the oracle (gradient oracle + CPU-baseline code) against the same vectors ----

@pytest.mark.parametrize("shape,seed,regime", [c for c in SMALL_CASES if c[0] in ("tiny", "small", "deci64")])
def test_torch_oracle_matches_reference(shape, seed, regime):
    import torch
    from oracle import gan_utils_torch as ot
    g, inp = load(shape, seed, regime)
    t = {k: torch.from_numpy(v) for k, v in inp.items()}
    loss = ot.compute_sinkhorn_loss(t["real"], t["fake"], cases.SC, 0.8, 100, t["h_fake"], t["m_real"],
                                    t["h_real"], t["m_fake"])
    assert rel(loss, g["loss"]) < 2e-5
    t64 = {k: v.double() for k, v in t.items()}
    loss64 = ot.compute_sinkhorn_loss(t64["real"], t64["fake"], cases.SC, 0.8, 100, t64["h_fake"], t64["m_real"],
                                      t64["h_real"], t64["m_fake"])
    assert rel(loss64, g["loss_f64"]) < 1e-10
    pm = ot.scale_invariante_martingale_regularization(t["m_real"], cases.LAM, cases.SC)
    assert rel(pm, g["pM"]) < 1e-5
    x, y = ot.flatten_video(t["real"]), ot.flatten_video(t["fake"])
    w = ot.compute_sinkhorn(x, y, t["h_fake"], t["m_real"], cases.SC, hx=t["h_real"], My=t["m_fake"], bi_causal=True)
    assert rel(w, g["w_bicausal"]) < 2e-5
    assert rel(ot.benchmark_sinkhorn(x, y, cases.SC), g["w_bench_default"]) < 2e-5
