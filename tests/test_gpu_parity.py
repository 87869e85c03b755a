"""GPU parity tests: the HIP path, called through the C-ABI (ctypes -> libkccot.so), against
  (1) the golden vectors produced by the reference's own gan_utils.py (tests/golden),
  (2) the CPU oracle on the same seeded inputs,
  (3) size-independent properties at BASELINE configs[1] full size.

Tolerances (fp32 path; BASELINE.json north_star: loss within 1e-4 relative of the reference):
  cost matrices     : 1e-5 * max|C|            (reference fp32 vs our fp32/fp64-combine)
  Sinkhorn costs    : 5e-5 relative to the fp32 golden value, 1e-4 to the fp64 one
  final loss        : 1e-4 relative (the stated target), against BOTH golden values
  iteration counts  : identical
  gradients         : GRAD_TOL_FACTOR x the MEASURED distance between the oracle's own fp32 and fp64 autograd
                      (tests/golden/grad_gap.json, written by tests/golden/make_grad_golden.py), floor
                      GRAD_TOL_FLOOR, relative to max|grad| -- see grad_tol() below
"""
import os

import numpy as np
import pytest
import torch

import cases
from oracle import gan_utils_np as o
from oracle import gan_utils_torch as ot

import json

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(__file__), "golden")
GRAD_GAP = json.load(open(os.path.join(GOLD, "grad_gap.json")))["gaps"]
# The reference differentiates in fp32 (tf.GradientTape).  How far fp32 autograd of the SAME unrolled loop sits
# from the fp64 one is a property of the problem's conditioning: 6e-8 .. 2e-6 on the O(1)-cost cases and in the
# near regime, 1e-4 .. 2e-4 on the sharp far-regime problems (C = O(1e3) against eps = 1: one ulp of C moves the
# plan by 1e-4).  A correct fp32 implementation with another operation order (log2 units, fp64 Gram
# combination, DPP trees) lands within a small multiple of that distance; measured on the MI355X
# (tools/grad_err.py -> profiles/r02_grad_err.txt) every HIP gradient of every golden case on every cost path is
# either <= 1.7e-5 of max|grad| (all well-conditioned cases) or <= 2.3 x the oracle's own gap (far regime:
# 2e-4 .. 5e-4).  Tolerance = max(floor, factor x gap): 80 x tighter than round 1's flat 2e-3 where the problem
# allows it, 2-3 x tighter where fp32 itself is the limit.
GRAD_TOL_FACTOR = 4.0
GRAD_TOL_FLOOR = 2.5e-5


def grad_tol(case, key):
    """Relative (to max|grad|) tolerance for gradient `key` of golden case `case`."""
    return max(GRAD_TOL_FLOOR, GRAD_TOL_FACTOR * GRAD_GAP[case][key])

DEV = "cuda:0"
ALL = cases.CASES
SMALL = [c for c in cases.CASES if c[0] != "cfg2"]


@pytest.fixture(scope="module")
def G():
    from kccotgan_amd import gan_utils
    return gan_utils


@pytest.fixture(scope="module")
def L():
    from kccotgan_amd import _lib
    return _lib


def load(shape, seed, regime):
    g = np.load(os.path.join(GOLD, cases.case_name(shape, seed, regime) + ".npz"))
    inp = cases.gen_inputs(shape, seed, regime)
    np.testing.assert_array_equal(cases.checksum(inp), g["checksum"])
    t = {k: torch.from_numpy(v).to(DEV) for k, v in inp.items()}
    return g, inp, t


def rel(a, b):
    return abs(float(a) - float(b)) / max(abs(float(b)), 1e-30)


def flat(v):
    return v.permute(0, 2, 1, 3, 4).reshape(v.shape[0], v.shape[2], -1).contiguous()


# "mfma" = the default MFMA kernels (bf16 exact-split for full stacks, f32-input otherwise);
# "mfma_f32" forces the f32-input MFMA kernel everywhere (option "gram_f32" = 1)
PATHS = ["auto", "direct", "mfma", "mfma_f32"]


def set_path(G, L, path, K=None):
    """Select the cost kernel.  The MFMA path needs K % 4 == 0 and K >= 32 (one k-tile); the
    'tiny' golden shape (K = 24) is below that and only runs the direct kernel."""
    if path.startswith("mfma") and K is not None and (K % 4 != 0 or K < 32):
        pytest.skip("MFMA path not eligible for K=%d" % K)
    L.set_option("gram_f32", 1 if path == "mfma_f32" else 0)
    G.cost_flags = {"auto": 0, "direct": L.COST_FORCE_DIRECT, "mfma": L.COST_FORCE_MFMA,
                    "mfma_f32": L.COST_FORCE_MFMA}[path]


def kdim(shape):
    B, H, T, W, C, J = cases.SHAPES[shape]
    return H * T * W * C


@pytest.fixture(autouse=True)
def _reset_flags(G, L):
    """Every test starts and ends on the library's default options (kccot_set_option is process-wide)."""
    defaults = {k: L.get_option(k) for k in L.option_names()}
    yield
    G.cost_flags = 0
    for k, v in defaults.items():
        L.set_option(k, v)


# ---------------------------------------------------------------- cost matrices
@pytest.mark.parametrize("path", PATHS)
@pytest.mark.parametrize("shape,seed,regime", ALL)
def test_cost3_matches_reference(G, L, shape, seed, regime, path):
    g, inp, t = load(shape, seed, regime)
    set_path(G, L, path, kdim(shape))
    B = t["real"].shape[0]
    real, fake = t["real"].reshape(B, -1), t["fake"].reshape(B, -1)
    C3 = G._Cost3.apply(real, fake, t["h_fake"], t["h_real"], t["m_real"], t["m_fake"], cases.SC).cpu().numpy()
    for k, tag in enumerate(("xy", "xx", "yy")):
        ref = g["C_" + tag]
        np.testing.assert_allclose(C3[k], ref, rtol=0, atol=1e-5 * np.abs(ref).max(), err_msg=tag)
        ref64 = g["C_" + tag + "_f64"]
        np.testing.assert_allclose(C3[k], ref64, rtol=0, atol=1e-5 * np.abs(ref64).max(), err_msg=tag + " f64")
    # the diagonal of C_xy (||x_i - y_i||^2: ~1e-2 of max|C| in the near regime, where the absolute tolerance above
    # would only check it to 1e-3) RELATIVELY, against the reference's fp64 run, on every kernel path
    np.testing.assert_allclose(np.diag(C3[0]), np.diag(g["C_xy_f64"]), rtol=2e-5, atol=0, err_msg="C_xy diagonal")
    # x == y problems: the l2 part of the diagonal is exactly 0, as (x-x)^2 in the reference
    caus = o.causal_term(inp["h_real"], inp["m_real"], cases.SC)
    np.testing.assert_allclose(np.diag(C3[1]), np.diag(caus), rtol=1e-5, atol=1e-6)


@pytest.mark.parametrize("path", PATHS)
@pytest.mark.parametrize("shape,seed,regime", ALL)
def test_public_cost_functions(G, L, shape, seed, regime, path):
    g, inp, t = load(shape, seed, regime)
    set_path(G, L, path, kdim(shape))
    x, y = flat(t["real"]), flat(t["fake"])
    tol = lambda ref: dict(rtol=0, atol=1e-5 * np.abs(ref).max())
    np.testing.assert_allclose(G.cost_xy(x, y, cases.SC).cpu().numpy(), g["C_plain"], **tol(g["C_plain"]))
    np.testing.assert_allclose(G.modified_cost(x, y, t["h_fake"], t["m_real"], cases.SC).cpu().numpy(),
                               g["C_xy"], **tol(g["C_xy"]))
    np.testing.assert_allclose(
        G.bi_causal_modified_cost(x, y, t["h_fake"], t["m_real"], t["h_real"], t["m_fake"], cases.SC).cpu().numpy(),
        g["C_bicausal"], **tol(g["C_bicausal"]))
    Cxx = G.cost_xy(x, x, cases.SC).cpu().numpy()
    assert np.all(np.diag(Cxx) == 0) and np.array_equal(Cxx, Cxx.T)
    np.testing.assert_array_equal(G.compute_N(t["m_real"]).cpu().numpy(), g["N_m_real"])


def test_cost_ragged_shapes_and_unaligned_k(G, L):
    """Bx != By, K not a multiple of 4 (scalar tail path), more than one 64-row tile."""
    rng = np.random.default_rng(11)
    for Bx, By, K in ((3, 7, 13), (70, 5, 37), (130, 67, 96), (1, 1, 1)):
        x = rng.random((Bx, K), dtype=np.float32)
        y = rng.random((By, K), dtype=np.float32)
        ref = o.cost_xy(x[:, None, :], y[:, None, :], 0.5, dtype=np.float64)
        got = G.cost_xy(torch.from_numpy(x).to(DEV), torch.from_numpy(y).to(DEV), 0.5).cpu().numpy()
        np.testing.assert_allclose(got, ref, rtol=2e-6, atol=1e-6)
    x = rng.random((130, 52), dtype=np.float32)   # x == y across tiles: mirrored lower triangle
    xt = torch.from_numpy(x).to(DEV)
    got = G.cost_xy(xt, xt, 1.0).cpu().numpy()
    np.testing.assert_allclose(got, o.cost_xy(x[:, None], x[:, None], 1.0, dtype=np.float64), rtol=2e-6, atol=1e-6)
    assert np.all(np.diag(got) == 0) and np.array_equal(got, got.T)


# ---------------------------------------------------------------- Sinkhorn
def nits_ok(got, g, key):
    """Iteration counts are identical to the reference's -- except where the reference's own fp32 and fp64 runs
    stop at different iterations (sum|u - u_prev| crosses 1e-2 on a slowly decaying tail, so rounding decides):
    there any count in the span of the two, widened by its width, is a stop decision the reference's arithmetic
    could have taken (cfg1_s1_far benchmark: 21 / 22; cfg2_s1_far eps = 0.25: 161 / 168)."""
    a, b = int(g["nits_" + key]), int(g["nits_" + key + "_f64"])
    lo, hi = min(a, b), max(a, b)
    return lo - (hi - lo) <= int(got) <= hi + (hi - lo)


def test_pointer_guard_refuses_float64_for_float_parameters(G, L):
    """`_lib.ptr` is the only dtype guard between torch tensors and the `float*` ABI: a float64 tensor must raise instead of
    being reinterpreted; `ptr_f64` serves the few `double*` parameters (Gram sums, row norms) and refuses fp32."""
    x64 = torch.zeros(8, dtype=torch.float64, device=DEV)
    x32 = torch.zeros(8, dtype=torch.float32, device=DEV)
    with pytest.raises(TypeError):
        L.ptr(x64)
    with pytest.raises(TypeError):
        L.ptr_f64(x32)
    assert L.ptr(x32) == x32.data_ptr() and L.ptr_f64(x64) == x64.data_ptr() and L.ptr(None) is None
    # the public entry points CONVERT other float types (as the reference's tf.cast does): never a reinterpretation
    a = torch.rand(4, 3, 8, dtype=torch.float64, device=DEV)
    assert torch.equal(G.cost_xy(a, a, 1.0), G.cost_xy(a.float(), a.float(), 1.0))


@pytest.mark.parametrize("shape,seed,regime", ALL)
def test_sinkhorn_variants_match_reference(G, shape, seed, regime):
    g, inp, t = load(shape, seed, regime)
    x, y = flat(t["real"]), flat(t["fake"])
    for eps, Lc in cases.EPS_L:
        key = "e%g_L%d" % (eps, Lc)
        w = G.compute_sinkhorn(x, y, t["h_fake"], t["m_real"], cases.SC, epsilon=eps, L=Lc)
        assert nits_ok(G.last_info["compute_sinkhorn"][0], g, key), (key, G.last_info["compute_sinkhorn"])
        assert rel(w, g["w_" + key]) < 5e-5, key
        assert rel(w, g["w_" + key + "_f64"]) < 1e-4, key
    w = G.compute_sinkhorn(x, y, t["h_fake"], t["m_real"], cases.SC, hx=t["h_real"], My=t["m_fake"], bi_causal=True)
    assert rel(w, g["w_bicausal"]) < 5e-5 and int(G.last_info["compute_sinkhorn"][0]) == int(g["nits_bicausal"])
    w = G.benchmark_sinkhorn(x, y, cases.SC)
    assert rel(w, g["w_bench_default"]) < 5e-5 and int(G.last_info["benchmark_sinkhorn"][0]) == 10
    w = G.benchmark_sinkhorn(x, y, cases.SC, epsilon=0.8, L=50, Lmin=20)
    assert rel(w, g["w_bench_e0.8_L50_Lmin20"]) < 5e-5
    # the stop test compares sum|u-u_prev| with 1e-2; where that sum passes the threshold within fp32
    # rounding the reference's own fp32 and fp64 runs disagree by one iteration (cfg1_s1_far: 21 vs 22)
    assert nits_ok(G.last_info["benchmark_sinkhorn"][0], g, "bench_e0.8_L50_Lmin20")


def test_sinkhorn_stop_rule_past_lmin(G):
    """Quirk 2 (gan_utils.py:149-160): L <= 100 runs exactly L; beyond, the loop stops at the
    first iteration >= 100 with sum|u-u_prev| < 1e-2 -- 198 on this crafted problem."""
    g = np.load(os.path.join(GOLD, "line32.npz"))
    x, y, h, M = (torch.from_numpy(a).to(DEV) for a in cases.gen_line_inputs())
    for sc, Lc in cases.LINE_RUNS:
        key = "sc%g_L%d" % (sc, Lc)
        w = G.compute_sinkhorn(x, y, h, M, sc, L=Lc)
        assert int(G.last_info["compute_sinkhorn"][0]) == int(g["nits_" + key]), key
        assert rel(w, g["w_" + key]) < 1e-4, key


def test_sinkhorn_kats(G, L):
    from kccotgan_amd._lib import lib, ptr
    for n in (1, 2, 7, 33, 64, 100, 128):
        C = torch.full((1, n, n), 2.5, device=DEV)
        cost = G._Sinkhorn.apply(C, 1.0, 100, 100, L.STOP_COUNT, "kat")
        assert abs(float(cost[0]) - 2.5) < 2e-5, n          # constant C -> uniform plan -> cost = C
    # column marginals equal 1/n after the v-update (update order: u, then v with the new u)
    n = 9
    Cn = np.random.default_rng(0).random((n, n), dtype=np.float32) * 3
    C = torch.from_numpy(Cn).to(DEV).reshape(1, n, n)
    cost = torch.empty(1, device=DEV)
    nits = torch.empty(2, dtype=torch.int32, device=DEV)
    pi = torch.empty(1, n, n, device=DEV)
    rc = lib.kccot_sinkhorn_fwd_f32(ptr(C), 1, n, 0.7, 13, 100, 1e-2, 0, None, None, ptr(cost), ptr(nits), ptr(pi),
                                    None, 0, None)
    assert rc == 0
    torch.cuda.synchronize()
    np.testing.assert_allclose(pi[0].sum(0).cpu().numpy(), 1 / n, rtol=3e-6)
    ref = o.sinkhorn_from_cost(Cn, 0.7, 13)
    assert int(nits[0]) == 13 and rel(cost[0], ref[0]) < 1e-5
    np.testing.assert_allclose(pi[0].cpu().numpy(), ref[4], rtol=1e-4, atol=1e-7)
    # several problems per launch, sizes that do not fill the thread grid
    for n in (5, 48, 100):
        Cn = np.random.default_rng(n).random((3, n, n), dtype=np.float32) * 4
        got = G._Sinkhorn.apply(torch.from_numpy(Cn).to(DEV), 0.5, 37, 100, L.STOP_COUNT, "kat").cpu().numpy()
        for p in range(3):
            assert rel(got[p], o.sinkhorn_from_cost(Cn[p], 0.5, 37)[0]) < 2e-5, (n, p)


def _raw_sinkhorn(L, C, eps, Lc, Lmin, mode, shortcut):
    """kccot_sinkhorn_fwd_f32 + _bwd_f32 through the C ABI; every output, for bitwise comparison."""
    lib, ptr = L.lib, L.ptr
    nprob, n, _ = C.shape
    with L.options(sinkhorn_shortcut=1 if shortcut else 0):
        uh = torch.full((nprob, Lc, n), float("nan"), device=DEV)
        vh = torch.full((nprob, Lc, n), float("nan"), device=DEV)
        cost = torch.empty(nprob, device=DEV)
        nits = torch.zeros(2 * nprob, dtype=torch.int32, device=DEV)
        pi = torch.empty_like(C)
        dC = torch.empty_like(C)
        gc = torch.tensor([2.0, -1.0, -1.0][:nprob], device=DEV)
        wsb = lib.kccot_sinkhorn_workspace_bytes(nprob, n)
        ws = torch.empty(max(wsb // 4, 4), device=DEV)
        assert lib.kccot_sinkhorn_fwd_f32(ptr(C), nprob, n, eps, Lc, Lmin, 1e-2, mode, ptr(uh), ptr(vh), ptr(cost),
                                          ptr(nits), ptr(pi), ptr(ws), wsb, None) == 0
        assert lib.kccot_sinkhorn_bwd_f32(ptr(C), ptr(uh), ptr(vh), ptr(nits), nprob, n, eps, Lc, ptr(gc), ptr(dC),
                                          ptr(ws), wsb, None) == 0
        torch.cuda.synchronize()
    nit = nits.cpu().numpy()
    out = dict(cost=cost.cpu().numpy(), pi=pi.cpu().numpy(), dC=dC.cpu().numpy(), nits=nit[:nprob], executed=nit[nprob:])
    # only the executed-or-filled part of the history is defined: rows [0, nits)
    out["u"] = [uh[p, :nit[p]].cpu().numpy() for p in range(nprob)]
    out["v"] = [vh[p, :nit[p]].cpu().numpy() for p in range(nprob)]
    return out


def _same_bits(a, b):
    a, b = np.ascontiguousarray(a), np.ascontiguousarray(b)
    return a.shape == b.shape and np.array_equal(a.view(np.uint32), b.view(np.uint32))


def test_sinkhorn_periodic_state_shortcut_is_bit_exact(G, L):
    """The forward skips iterations once the fp32 state (u, v) repeats bit for bit with period <= 4
    (sinkhorn.hip).  It claims to be EXACT: every output -- cost, plan, iteration count, the full dual history, dC -- must be
    bit-identical to a run that executes every iteration (option "sinkhorn_shortcut" = 0)."""
    probs = {}
    for name, (shape, seed, regime) in dict(near=("cfg2", 0, "near"), far=("cfg2", 1, "far"), small=SMALL[0]).items():
        g, inp, t = load(shape, seed, regime)
        probs[name] = G._Cost3.apply(G._flat2(t["real"]), G._flat2(t["fake"]), G._feat(t["h_fake"]),
                                     G._feat(t["h_real"]), G._feat(t["m_real"]), G._feat(t["m_fake"]),
                                     float(cases.SC)).detach().contiguous()
    rng = np.random.default_rng(7)
    for n, scale in ((64, 2000.0), (64, 30.0), (33, 500.0), (100, 3000.0), (128, 800.0), (7, 100.0)):
        probs["rand_n%d_s%g" % (n, scale)] = torch.from_numpy(
            (rng.random((3, n, n), dtype=np.float32) * np.float32(scale))).to(DEV)
    x, y, h, M = (torch.from_numpy(a).to(DEV) for a in cases.gen_line_inputs())     # stops at 198 of 300 (quirk 2)
    probs["line32"] = G.modified_cost(x, y, h, M, 100.0).detach().reshape(1, x.shape[0], y.shape[0]).contiguous()
    skipped_somewhere = False
    for name, C in probs.items():
        for eps, Lc, Lmin, mode in ((1.0, 100, 100, L.STOP_COUNT), (0.8, 50, 20, L.STOP_INDEX),
                                    (1.0, 300, 100, L.STOP_COUNT), (1.0, 7, 100, L.STOP_COUNT),
                                    (0.5, 40, 3, L.STOP_COUNT)):
            a = _raw_sinkhorn(L, C, eps, Lc, Lmin, mode, shortcut=True)
            b = _raw_sinkhorn(L, C, eps, Lc, Lmin, mode, shortcut=False)
            tag = (name, eps, Lc, Lmin, mode)
            assert np.array_equal(a["nits"], b["nits"]), tag
            assert np.array_equal(b["executed"], b["nits"]), tag
            assert (a["executed"] <= a["nits"]).all(), tag
            skipped_somewhere |= bool((a["executed"] < a["nits"]).any())
            for k in ("cost", "pi", "dC"):
                assert _same_bits(a[k], b[k]), (tag, k)
            for p in range(C.shape[0]):
                assert _same_bits(a["u"][p], b["u"][p]) and _same_bits(a["v"][p], b["v"][p]), (tag, p)
    assert skipped_somewhere      # the test must exercise the jump, not only the fall-through


@pytest.mark.parametrize("solver", ["coop", "stream"])
def test_sinkhorn_large_n_streaming_path(G, L, solver):
    """n > 128 (BASELINE configs 3-5 batch sizes): the multi-CU solver (flag-in-data exchange) and the single-workgroup
    streaming kernels it falls back to (option "sinkhorn_coop" = 0), forward and reverse sweep, against the oracle /
    fp64 autograd on random cost matrices."""
    L.set_option("sinkhorn_coop", 1 if solver == "coop" else 0)
    for n, Lc, eps in ((130, 25, 0.7), (256, 40, 1.0), (512, 12, 0.5), (1000, 5, 1.0)):
        Cn = (np.random.default_rng(n).random((2, n, n), dtype=np.float32) * 6).astype(np.float32)
        C = torch.from_numpy(Cn).to(DEV).requires_grad_(True)
        cost = G._Sinkhorn.apply(C, eps, Lc, 100, L.STOP_COUNT, "large")
        w = torch.tensor([1.0, -0.5], device=DEV)
        (cost * w).sum().backward()
        assert G.last_info["large"].tolist() == [Lc, Lc]
        for p in range(2):
            Cd = torch.from_numpy(Cn[p]).double().requires_grad_(True)
            ref, nits = ot.sinkhorn_from_cost(Cd, eps, Lc)
            ref.backward()
            assert rel(cost[p], ref) < 2e-5, (n, p)
            gref = Cd.grad.numpy() * float(w[p])
            np.testing.assert_allclose(C.grad[p].cpu().numpy(), gref, rtol=0, atol=2e-4 * np.abs(gref).max())


def test_sinkhorn_cooperative_stop_rule_and_many_problems(G, L):
    """The cooperative solver's stop rule (every workgroup sums the same per-workgroup partials) against the
    streaming kernels: same executed iteration count and costs on a problem that stops early (L > Lmin); and
    a launch with more problems than fit co-resident falls back to the streaming kernels on its own."""
    n = 160
    rng = np.random.default_rng(11)
    Cn = (rng.random((3, n, n), dtype=np.float32) * 3).astype(np.float32)
    C = torch.from_numpy(Cn).to(DEV)
    res = {}
    for solver in ("coop", "stream"):
        with L.options(sinkhorn_coop=1 if solver == "coop" else 0):
            cost = G._Sinkhorn.apply(C, 1.0, 400, 100, L.STOP_COUNT, "stop")
            res[solver] = (cost.cpu().numpy(), G.last_info["stop"].tolist())
    assert res["coop"][1] == res["stream"][1] and res["coop"][1][0] < 400
    np.testing.assert_allclose(res["coop"][0], res["stream"][0], rtol=2e-5)
    many = torch.from_numpy((rng.random((24, n, n), dtype=np.float32) * 3).astype(np.float32)).to(DEV)   # 24 x 10 workgroups > 192
    cost = G._Sinkhorn.apply(many, 1.0, 30, 100, L.STOP_COUNT, "many")
    ref = [o.sinkhorn_from_cost(many[p].cpu().numpy(), 1.0, 30)[0] for p in (0, 23)]
    assert rel(cost[0], ref[0]) < 2e-5 and rel(cost[23], ref[1]) < 2e-5


def test_sinkhorn_falls_back_when_the_device_cannot_hold_the_cooperative_grid(G, L):
    """The multi-CU solver is launched only if CU count x occupancy (queried at run time) covers its grid.
    Option "sinkhorn_coop_max_wg" stands in for a small / partitioned / CU-masked device: the same call then runs the
    one-workgroup streaming solver and returns the same costs, iteration counts and gradients."""
    n = 256
    Cn = (np.random.default_rng(5).random((3, n, n), dtype=np.float32) * 4).astype(np.float32)
    out = {}
    for cap in (None, "8"):
        L.set_option("sinkhorn_coop_max_wg", int(cap) if cap else 0)   # 3 problems x 16 workgroups = 48 > 8
        C = torch.from_numpy(Cn).to(DEV).requires_grad_(True)
        cost = G._Sinkhorn.apply(C, 1.0, 30, 100, L.STOP_COUNT, "cap")
        cost.sum().backward()
        out[cap] = (cost.detach().cpu().numpy(), G.last_info["cap"].tolist(), C.grad.cpu().numpy())
    L.set_option("sinkhorn_coop_max_wg", 0)
    assert out[None][1] == out["8"][1] == [30, 30, 30]
    np.testing.assert_allclose(out["8"][0], out[None][0], rtol=2e-5)
    np.testing.assert_allclose(out["8"][2], out[None][2], rtol=0, atol=2e-4 * np.abs(out[None][2]).max())
    ref = o.sinkhorn_from_cost(Cn[1], 1.0, 30)[0]
    assert rel(out["8"][0][1], ref) < 2e-5


def test_sinkhorn_abort_is_nan_plus_status_never_a_plausible_number():
    """Fault injection (one workgroup of problem 0 never takes part -- what a non-resident workgroup looks like to its
    siblings).  The hook exists in the diagnostic twin of the library only (make libkccot_diag.so, -DKCCOT_DIAG), so the
    scenario runs in a child process that loads that build: tests/sk_abort_child.py holds the assertions (cost NaN, a
    NEGATIVE iteration count, kccot_sinkhorn_status = KCCOT_EABORTED, NaN gradients, KccotError from the wrapper; the
    problem next to it in the same launch and the next launch unaffected)."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    diag = os.path.join(root, "kccotgan_amd", "csrc", "libkccot_diag.so")
    if not os.path.exists(diag):
        pytest.skip("libkccot_diag.so is not built (python -c 'import __graft_entry__ as g; g.build()' builds it)")
    env = dict(os.environ, KCCOT_LIB_PATH=diag, KCCOT_SK_FAULT_INJECT="1")
    r = subprocess.run([sys.executable, os.path.join(root, "tests", "sk_abort_child.py")], env=env, capture_output=True,
                       text=True, timeout=300)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert "abort path ok" in r.stdout


@pytest.mark.parametrize("B,regime", [(128, "near"), (192, "far"), (256, "near")])
def test_cost3_blocked_mfma_path_above_64(G, L, B, regime):
    """Batches above 64 (multiples of 64, K >= 256): the three matrices are assembled from 64 x 64 blocks on the
    MFMA kernels -- pair-difference form on the diagonal blocks, plain Gram form off the diagonal, mirrored blocks
    for the x == y problems.  Checked against the f64 oracle and against the direct-difference kernel."""
    rng = np.random.default_rng(1000 + B)
    H, T, W, Cc, J = 8, 10, 8, 4, 8
    real = rng.random((B, H, T, W, Cc), dtype=np.float32)
    if regime == "near":
        fake = np.clip(real + np.float32(0.01) * rng.standard_normal(real.shape, dtype=np.float32), 0, 1).astype(np.float32)
    else:
        fake = rng.random(real.shape, dtype=np.float32)
    f = {k: rng.random((B, T, J), dtype=np.float32) for k in ("h_fake", "m_real", "h_real", "m_fake")}
    t = {k: torch.from_numpy(v).to(DEV) for k, v in dict(real=real, fake=fake, **f).items()}
    args = (t["real"].reshape(B, -1), t["fake"].reshape(B, -1), t["h_fake"], t["h_real"], t["m_real"], t["m_fake"], cases.SC)
    G.cost_flags = 0
    blocked = G._Cost3.apply(*args).cpu().numpy()
    G.cost_flags = L.COST_FORCE_DIRECT
    direct = G._Cost3.apply(*args).cpu().numpy()
    x64, y64 = real.reshape(B, -1).astype(np.float64), fake.reshape(B, -1).astype(np.float64)
    pairs = {"xy": (x64, y64, "h_fake", "m_real"), "xx": (x64, x64, "h_real", "m_real"), "yy": (y64, y64, "h_fake", "m_fake")}
    for k, tag in enumerate(("xy", "xx", "yy")):
        a, b, hk, mk = pairs[tag]
        l2 = ((a * a).sum(1)[:, None] + (b * b).sum(1)[None, :] - 2 * a @ b.T)
        if tag == "xy":
            l2[np.arange(B), np.arange(B)] = ((a - b) ** 2).sum(1)
        else:
            l2[np.arange(B), np.arange(B)] = 0.0
        ref = l2 * cases.SC + o.causal_term(f[hk], f[mk], cases.SC, dtype=np.float64)
        np.testing.assert_allclose(blocked[k], ref, rtol=0, atol=1e-5 * np.abs(ref).max(), err_msg=tag)
        np.testing.assert_allclose(blocked[k], direct[k], rtol=0, atol=1e-5 * np.abs(ref).max(), err_msg=tag + " vs direct")
        # the sample-against-its-own-fake entries are the small ones: relative accuracy there
        np.testing.assert_allclose(np.diag(blocked[k]), np.diag(ref), rtol=2e-5, atol=1e-7, err_msg=tag + " diagonal")


def _tile_inputs(B, K, seed, T=10, J=8):
    rng = np.random.default_rng(seed)
    real = torch.from_numpy(rng.random((B, K), dtype=np.float32)).to(DEV)
    fake = torch.from_numpy(np.clip(real.cpu().numpy() + 0.05 * rng.standard_normal((B, K)).astype(np.float32), 0, 1)).to(DEV)
    f = [torch.from_numpy(rng.random((B, T, J), dtype=np.float32)).to(DEV) for _ in range(4)]    # h_fake, h_real, m_real, m_fake
    return real, fake, f


def _cost3_oracle(real, fake, f):
    x, y = real.cpu().numpy().astype(np.float64), fake.cpu().numpy().astype(np.float64)
    fn = [t.cpu().numpy() for t in f]
    out = []
    for a, b, h, M, pairdiff in ((x, y, fn[0], fn[2], True), (x, x, fn[1], fn[2], False), (y, y, fn[0], fn[3], False)):
        l2 = (a * a).sum(1)[:, None] + (b * b).sum(1)[None, :] - 2 * a @ b.T
        n = np.arange(a.shape[0])
        l2[n, n] = ((a - b) ** 2).sum(1) if pairdiff else 0.0
        out.append(l2 * cases.SC + o.causal_term(h, M, cases.SC, dtype=np.float64))
    return out


@pytest.mark.parametrize("B,K", [(256, 2560), (256, 2560 + 36), (512, 3072 + 4), (768, 1536), (128, 2560 + 36), (128, 4096)])
def test_tile256_gram_against_the_128_row_tiles_the_direct_kernel_and_the_oracle(G, L, B, K):
    """B % 256 == 0 and B == 128 (the stack [X ; E] is ONE 256-row panel, E formed while staging): the 256-row pair tiles
    (cost_tile256.hip: E written by the (X_i, E_i) pairs, triangular tile map on diagonal pairs, one K-chunk per partial
    tile) against the 128-row tiles they replace (option "cost_tile256" = 0), the
    direct-difference kernel and the fp64 oracle -- K a multiple of 32, K % 32 != 0 (the RAGGED instantiation: last granule
    partial), a single chunk, six panels."""
    real, fake, f = _tile_inputs(B, K, 2000 + B + K)
    out = {}
    out["t256"] = G._Cost3.apply(real, fake, f[0], f[1], f[2], f[3], cases.SC).cpu().numpy()
    with L.options(cost_tile256=0):
        out["t128"] = G._Cost3.apply(real, fake, f[0], f[1], f[2], f[3], cases.SC).cpu().numpy()
    G.cost_flags = L.COST_FORCE_DIRECT
    out["direct"] = G._Cost3.apply(real, fake, f[0], f[1], f[2], f[3], cases.SC).cpu().numpy()
    G.cost_flags = 0
    ref = _cost3_oracle(real, fake, f)
    for k, tag in enumerate(("xy", "xx", "yy")):
        tol = 1e-5 * np.abs(ref[k]).max()
        for name in ("t256", "t128", "direct"):
            np.testing.assert_allclose(out[name][k], ref[k], rtol=0, atol=tol, err_msg="%s %s" % (name, tag))
        np.testing.assert_allclose(np.diag(out["t256"][k]), np.diag(ref[k]), rtol=2e-5, atol=1e-7, err_msg=tag + " diagonal")
    assert np.all(np.diag(out["t256"][1]) == np.diag(out["direct"][1]))          # x == y: exactly the causal term
    # symmetric distances of the x == y problems: bitwise mirror images (one Gram entry serves both)
    caus = [o.causal_term(f[1].cpu().numpy(), f[2].cpu().numpy(), cases.SC, dtype=np.float64),
            o.causal_term(f[0].cpu().numpy(), f[3].cpu().numpy(), cases.SC, dtype=np.float64)]
    for k in (1, 2):
        d = out["t256"][k].astype(np.float64) - caus[k - 1]
        np.testing.assert_allclose(d, d.T, rtol=0, atol=2e-6 * np.abs(d).max())


def test_rows_gram_in_column_ranges_equals_the_one_call_form(G, L):
    """kccot_pairwise_cost3_rows_gram_sums_f64 / _from_sums_f32 (the chunked all-gather of the batch-sharded caller: the
    Gram sums of a rank's row block accumulated over column ranges) against the one-call form on the whole K, at the cut
    kccotgan_amd.dist.gather_chunk_bounds makes -- fp64 sums of disjoint column ranges in a fixed order: equal to fp64
    rounding of sums whose fp32 partial tiles are cut at other columns (1e-6 of max|C|)."""
    from kccotgan_amd import _lib
    from kccotgan_amd.dist import HipOps as H, gather_chunk_bounds
    B, K, rows = 256, 4096 + 36, 64
    real, fake, f = _tile_inputs(B, K, 8123)
    norms = H.row_norms(real, fake)
    bounds = gather_chunk_bounds(K, 4)
    assert len(bounds) == 4 and bounds[0][0] == 0 and bounds[-1][1] == K and all(a % 32 == 0 and b - a >= 256 for a, b in bounds)
    for r0 in (0, 128):
        ref = H.cost3_rows(real, fake, f[0], f[1], f[2], f[3], cases.SC, r0, rows, norms).cpu().numpy()
        gsum = torch.full((int(_lib.lib.kccot_pairwise_cost3_rows_gram_sums_count(rows, B)),), float("nan"), dtype=torch.float64, device=DEV)
        for c, (a, b) in enumerate(bounds):
            H.rows_gram_sums(real[:, a:b].contiguous(), fake[:, a:b].contiguous(), r0, rows, gsum, c > 0)
        got = H.rows_gram_from_sums(gsum, B, f[0], f[1], f[2], f[3], cases.SC, r0, rows, norms).cpu().numpy()
        np.testing.assert_allclose(got, ref, rtol=0, atol=1e-6 * np.abs(ref).max())


def _capture(fn):
    """Warm up on a side stream, capture `fn` into a hipGraph, return (graph, static outputs)."""
    side = torch.cuda.Stream(device=DEV)
    side.wait_stream(torch.cuda.current_stream(DEV))
    with torch.cuda.stream(side):
        fn()
    torch.cuda.current_stream(DEV).wait_stream(side)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        out = fn()
    return g, out


@pytest.mark.parametrize("n", [200, 256, 512])
def test_multi_cu_sinkhorn_replays_as_a_graph(G, L, n):
    """128 < n <= 1024: the multi-CU solve and reverse sweep captured into a hipGraph and replayed -- with a host sync after
    every replay and back to back -- give the eager result bit for bit EVERY time.  (Round 3: the exchange area was zeroed
    with hipMemsetAsync; as a graph node its zeros were not what the next node's agent-scope loads saw, so every replay
    after the first polled to its bound and aborted to NaN.  It is zeroed by a kernel with agent-scope stores now.)"""
    from kccotgan_amd.dist import HipOps as H
    gen = torch.Generator(device=DEV).manual_seed(n)
    C3 = torch.rand((3, n, n), device=DEV, generator=gen) * 40
    one = torch.ones((), device=DEV)

    def step():
        loss, saved = H.divergence_fwd(C3, 1.0, 100)
        return loss, H.divergence_bwd(saved, one), saved[3]

    with L.options(sinkhorn_shortcut=0):
        l0, d0, n0 = step()
        torch.cuda.synchronize()
        assert n0[:3].tolist() == [100, 100, 100] and bool(torch.isfinite(d0).all())
        g, (gl, gd, gn) = _capture(step)
        for i in range(3):
            g.replay()
            torch.cuda.synchronize()
            assert gn.tolist() == n0.tolist() and torch.equal(gl, l0) and torch.equal(gd, d0), ("synced replay", i)
        for i in range(6):
            g.replay()
        torch.cuda.synchronize()
        assert gn.tolist() == n0.tolist() and torch.equal(gl, l0) and torch.equal(gd, d0), "back-to-back replays"


@pytest.mark.parametrize("n", [200, 256])
def test_multi_cu_sinkhorn_replays_with_changing_iteration_counts(G, L, n):
    """The case a stale exchange word would corrupt SILENTLY: half-step tags restart at 1 in every launch, so if a word of
    the previous launch survived the zeroing, a solve that runs LONGER than its predecessor would find valid-looking tags
    with the predecessor's values.  One captured solve + reverse sweep over a static cost buffer, replayed with contents
    that stop at 100 (three random problems), at 198 and at 300 iterations (points on a line, gan_utils.py:149-160's
    Lmin quirk) in the order short, long, short, longest, long: every replay equals the eager run on the same contents bit for
    bit.  (Round 4: the tags also carry a per-launch epoch that the zeroing kernel increments -- replays included -- so a
    surviving word could only make a poll run to its bound, never be consumed.)"""
    from kccotgan_amd.dist import HipOps as H
    gen = torch.Generator(device=DEV).manual_seed(77 + n)
    i = torch.arange(n, device=DEV, dtype=torch.float32) / n
    line = lambda sc: (sc * 2.0 * (i[:, None] - (i[None, :] + 0.5 / n)) ** 2).expand(3, n, n).contiguous()
    contents = {"short": torch.rand((3, n, n), device=DEV, generator=gen) * 3, "long": line(100.0), "longest": line(300.0)}
    C3 = torch.empty((3, n, n), device=DEV)
    one = torch.ones((), device=DEV)

    def step():
        loss, saved = H.divergence_fwd(C3, 1.0, 300)
        return loss, H.divergence_bwd(saved, one), saved[3]

    with L.options(sinkhorn_shortcut=0):
        eager = {}
        for k, v in contents.items():
            C3.copy_(v)
            l, d, nn = step()
            torch.cuda.synchronize()
            eager[k] = (l.clone(), d.clone(), nn.tolist())
        assert eager["short"][2][:3] == [100] * 3 and eager["long"][2][:3] == [198] * 3 and eager["longest"][2][:3] == [300] * 3
        g, (gl, gd, gn) = _capture(step)
        for k in ("short", "long", "short", "longest", "long", "longest", "short"):
            C3.copy_(contents[k])
            g.replay()
            torch.cuda.synchronize()
            assert gn.tolist() == eager[k][2] and torch.equal(gl, eager[k][0]) and torch.equal(gd, eager[k][1]), k


@pytest.mark.parametrize("n", [130, 200, 256, 384, 512, 640])
def test_multi_cu_sinkhorn_exchange_through_the_xcd_l2_is_bit_identical(G, L, n):
    """Option "sinkhorn_coop_xcd": 1 = one problem per XCD by layout, verified in the kernel, duals exchanged through the L2
    the problem's workgroups share (the default); 0 = the agent-scope exchange on the 2-D grid; 2 = the check on the 2-D
    grid, where it must FAIL and fall back (a wrongly passing check would leave the workgroups blind to each other: abort,
    NaN).  Same arithmetic on the same values: costs, iteration counts, dual histories and dC agree bit for bit."""
    from kccotgan_amd.dist import HipOps as H
    gen = torch.Generator(device=DEV).manual_seed(1000 + n)
    C3 = torch.rand((3, n, n), device=DEV, generator=gen) * 25
    g3 = torch.tensor([2.0, -1.0, -1.0], device=DEV)
    out = {}
    with L.options(sinkhorn_shortcut=0):
        for mode in (1, 0, 2):
            with L.options(sinkhorn_coop_xcd=mode):
                cost, saved = H.sinkhorn3_fwd(C3, 1.0, 100)
                dC = H.sinkhorn3_bwd(saved, g3)
                torch.cuda.synchronize()
                out[mode] = (cost.clone(), saved[1].clone(), saved[2].clone(), saved[3].clone(), dC.clone())
    assert out[1][3][:3].tolist() == [100, 100, 100] and bool(torch.isfinite(out[1][4]).all())
    for mode in (0, 2):
        for a_, b_ in zip(out[1], out[mode]):
            assert torch.equal(a_, b_), (n, mode)


@pytest.mark.parametrize("nprob", [1, 8, 9])
def test_multi_cu_sinkhorn_problem_counts_around_the_xcd_layout(L, nprob):
    """kccot_sinkhorn_fwd_f32 / bwd at n = 160 with 1, 8 (one problem per XCD, all eight used) and 9 problems (more than XCDs:
    the 2-D grid with the agent-scope exchange): every problem against the same problem solved alone with the layout off."""
    from kccotgan_amd._lib import lib, ptr, check, workspace
    n, Lit = 160, 100
    gen = torch.Generator(device=DEV).manual_seed(50 + nprob)
    C = (torch.rand((nprob, n, n), device=DEV, generator=gen) * 20).contiguous()
    gc = torch.randn((nprob,), device=DEV, generator=gen)

    def solve(Cs, g, xcd):
        k = Cs.shape[0]
        uh, vh = torch.empty(k, Lit, n, device=DEV), torch.empty(k, Lit, n, device=DEV)
        cost, nits = torch.empty(k, device=DEV), torch.zeros(2 * k, dtype=torch.int32, device=DEV)
        dC = torch.empty_like(Cs)
        ws, wsb = workspace(lib.kccot_sinkhorn_workspace_bytes(k, n), Cs)
        with L.options(sinkhorn_shortcut=0, sinkhorn_coop_xcd=xcd):
            check(lib.kccot_sinkhorn_fwd_f32(ptr(Cs), k, n, 1.0, Lit, 100, 1e-2, 0, ptr(uh), ptr(vh), ptr(cost), ptr(nits), None, ws, wsb,
                                             None), "fwd")
            check(lib.kccot_sinkhorn_bwd_f32(ptr(Cs), ptr(uh), ptr(vh), ptr(nits), k, n, 1.0, Lit, ptr(g), ptr(dC), ws, wsb, None), "bwd")
        torch.cuda.synchronize()
        return cost, nits[:k], dC

    cost, nits, dC = solve(C, gc, 1)
    assert nits.tolist() == [100] * nprob and bool(torch.isfinite(dC).all())
    for p in range(nprob):
        c1, n1, d1 = solve(C[p:p + 1].contiguous(), gc[p:p + 1].contiguous(), 0)
        assert torch.equal(c1[0], cost[p]) and int(n1[0]) == int(nits[p]) and torch.equal(d1[0], dC[p]), p


def test_graphed_loss_step_at_a_large_batch(G, L):
    """GraphedLossStep at B = 256 (256-row Gram tiles, multi-CU Sinkhorn, one-launch video gradient): replays equal the
    eager step bit for bit, repeatedly, and see new inputs."""
    from kccotgan_amd.graph import GraphedLossStep
    B, H_, T, W, C = 256, 8, 10, 8, 5
    gen = torch.Generator(device=DEV).manual_seed(99)
    t = {"real": torch.rand((B, H_, T, W, C), device=DEV, generator=gen)}
    t["fake"] = (t["real"] + 0.05 * torch.randn(t["real"].shape, device=DEV, generator=gen)).clamp_(0, 1)
    for k in ("h_fake", "m_real", "h_real", "m_fake"):
        t[k] = torch.rand((B, T, 8), device=DEV, generator=gen)
    names = ("fake", "h_fake", "h_real", "m_real", "m_fake")
    for k in names:
        t[k].requires_grad_(True)
    loss = G.compute_sinkhorn_loss(t["real"], t["fake"], cases.SC, 0.8, 100, t["h_fake"], t["m_real"], t["h_real"], t["m_fake"])
    grads = torch.autograd.grad(loss, [t[k] for k in names])
    gs = GraphedLossStep(t, cases.SC)
    for i in range(4):
        gl, gg = gs()
        torch.cuda.synchronize()
        assert torch.equal(gl.reshape(()), loss.detach().reshape(())), i
        assert all(torch.equal(gg[k], g) for k, g in zip(names, grads)), i
        assert gs.nits.tolist() == [100, 100, 100]
    gl2, _ = gs(fake=t["fake"].detach() * 0.5)
    torch.cuda.synchronize()
    assert not torch.equal(gl2.reshape(()), loss.detach().reshape(())) and bool(torch.isfinite(gl2))


@pytest.mark.parametrize("shape,fused", [((2, 8, 9, 10, 2), 1), ((2, 9, 30, 64, 1), 1), ((2, 6, 20, 16, 3), 1),
                                         ((3, 20, 30, 64, 1), 2), ((2, 24, 30, 64, 3), 2)])
def test_smoothing_replays_as_a_graph(L, shape, fused):
    """KernelSmoothing forward + backward (temporal and 3-D; the shapes take the LDS plane kernel with its memset scalars,
    the lane-exchange walk, the any-channel kernels; statistics folded and not; fused = 2: the 3-D calls as the fused walks
    of round 4, H cut into segments / three channels) captured and replayed three times."""
    from kccotgan_amd.data_utils import KernelSmoothing
    ks = KernelSmoothing(6, 6)
    gen = torch.Generator(device=DEV).manual_seed(sum(shape))
    x = torch.rand(shape, device=DEV, generator=gen)
    w = torch.randn(shape, device=DEV, generator=gen)
    for fold in (0, 2):
        with L.options(smooth_bwd_fold=fold, smooth_fused3=fused):
            for fn in (ks.temporal_convolution, ks.gaussian_convolution3D):
                def step():
                    xi = x.detach().requires_grad_(True)
                    y = fn(xi, 2.0)
                    (gx,) = torch.autograd.grad(y, xi, w)
                    return y.detach(), gx
                y0, g0 = step()
                torch.cuda.synchronize()
                g, (gy, gg) = _capture(step)
                for i in range(3):
                    g.replay()
                    torch.cuda.synchronize()
                    assert torch.equal(gy, y0) and torch.equal(gg, g0), (shape, fold, fn.__name__, i)


def test_c_abi_from_a_native_program(tmp_path):
    """The drop-in boundary without Python or torch: tests/abi_gpu_smoke.cpp (hipMalloc, its own stream, nothing linked but
    the HIP runtime and libkccot.so) calls kccot_sinkhorn_loss_{fwd,bwd}_f32 and checks the three cost matrices against a
    double-precision host evaluation of gan_utils.py:14-17,34-38, the iteration counts, and the video gradient against a
    central difference of the library's own loss along the gradient direction."""
    import shutil
    import subprocess
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.skip("no hipcc on this box")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    libdir = os.path.join(root, "kccotgan_amd", "csrc")
    exe = str(tmp_path / "abi_gpu_smoke")
    r = subprocess.run([hipcc, "-O1", "--offload-arch=gfx950", "-I", os.path.join(root, "include"), "-o", exe,
                        os.path.join(root, "tests", "abi_gpu_smoke.cpp"), "-L", libdir, "-lkccot", "-Wl,-rpath," + libdir],
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-3000:]
    r = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0 and "abi_gpu_smoke ok" in r.stdout, r.stdout[-2000:] + r.stderr[-2000:]


def test_tiled_gram_128_row_tiles_with_materialised_difference_rows(G, L):
    """B = 640 (a multiple of 128 that is not one of 256, >= 512): the 128-row tiles with E = fake - real formed once
    (ediff_rows), ten panels, against the fp64 oracle."""
    B, K = 640, 2560 + 36
    real, fake, f = _tile_inputs(B, K, 3640)
    got = G._Cost3.apply(real, fake, f[0], f[1], f[2], f[3], cases.SC).cpu().numpy()
    ref = _cost3_oracle(real, fake, f)
    for k, tag in enumerate(("xy", "xx", "yy")):
        np.testing.assert_allclose(got[k], ref[k], rtol=0, atol=1e-5 * np.abs(ref[k]).max(), err_msg=tag)
        np.testing.assert_allclose(np.diag(got[k]), np.diag(ref[k]), rtol=2e-5, atol=1e-7, err_msg=tag + " diagonal")


@pytest.mark.parametrize("B,G_,K", [(128, 4, 2560), (256, 8, 2560 + 36), (512, 8, 3072), (384, 6, 1536 + 4), (128, 2, 1024)])
def test_rows_gram_row_blocks_against_the_direct_kernel_and_the_oracle(G, L, B, G_, K):
    """The batch-sharded rank's row block on the matrix pipe (cost_rows.hip: Gram row block [X_I ; E_I][X ; E]^T + gathered
    row norms) for 32 and 64 rows per rank, column panels of X rows, E rows and the mixed panel of B % 256 == 128, K a multiple
    of 32 and not: every rank's block against the direct-difference kernel and the fp64 oracle, the C_xy diagonal entries
    (sample against its own fake) relatively, and with norms = NULL (the call computes them itself)."""
    from kccotgan_amd.dist import HipOps as H
    real, fake, f = _tile_inputs(B, K, 5000 + B + K)
    m = B // G_
    assert H.rows_gram_supported(m, B, K)
    norms = torch.cat([H.row_norms(real[r * m:(r + 1) * m], fake[r * m:(r + 1) * m]) for r in range(G_)])   # as the all-gather delivers
    ref = _cost3_oracle(real, fake, f)
    x64, y64 = real.double().cpu().numpy(), fake.double().cpu().numpy()
    e64 = y64 - x64
    np.testing.assert_allclose(norms.cpu().numpy(), np.stack([(x64 * x64).sum(1), (e64 * e64).sum(1), (x64 * e64).sum(1)], 1),
                               rtol=2e-6, atol=1e-4)
    for r in sorted({0, G_ // 2, G_ - 1}):
        got = H.cost3_rows(real, fake, f[0], f[1], f[2], f[3], cases.SC, r * m, m, norms).cpu().numpy()
        direct = H.cost3_rows(real, fake, f[0], f[1], f[2], f[3], cases.SC, r * m, m).cpu().numpy()
        for k, tag in enumerate(("xy", "xx", "yy")):
            want = ref[k][r * m:(r + 1) * m]
            tol = 1e-5 * np.abs(ref[k]).max()
            np.testing.assert_allclose(got[k], want, rtol=0, atol=tol, err_msg="gram %s rank %d" % (tag, r))
            np.testing.assert_allclose(direct[k], want, rtol=0, atol=tol, err_msg="direct %s rank %d" % (tag, r))
        i = np.arange(m)
        np.testing.assert_allclose(got[0][i, r * m + i], ref[0][r * m + i, r * m + i], rtol=2e-5, atol=1e-7)
        assert np.all(got[1][i, r * m + i] == direct[1][i, r * m + i])          # x == y: exactly the causal term
    own = torch.empty((3, m, B), device=DEV)
    ws = torch.empty(int(L.lib.kccot_pairwise_cost3_rows_gram_workspace_bytes(m, B, K)), dtype=torch.uint8, device=DEV)
    L.check(L.lib.kccot_pairwise_cost3_rows_gram_f32(L.ptr(real), L.ptr(fake), B, K, cases.SC, L.ptr(f[0]), L.ptr(f[1]), L.ptr(f[2]),
                                                     L.ptr(f[3]), f[0].shape[1], f[0].shape[2], m, m, None, L.ptr(own),
                                                     ws.data_ptr(), ws.numel(), None), "rows_gram")
    np.testing.assert_allclose(own.cpu().numpy(), H.cost3_rows(real, fake, f[0], f[1], f[2], f[3], cases.SC, m, m, norms).cpu().numpy(),
                               rtol=0, atol=2e-6 * np.abs(ref[0]).max())
    assert not H.rows_gram_supported(16, B, K) and not H.rows_gram_supported(m, B + 64, K)


@pytest.mark.parametrize("B,K", [(64, 4100), (37, 3076), (7, 256), (64, 64 * 300 + 36), (33, 128)])
def test_gram_producers_on_ragged_shapes(G, L, B, K):
    """The producers of gram128_partial_x3ws keep two stages of UNCONDITIONAL (clamped) loads in flight and zero the
    out-of-range values with selects: ragged row blocks (B < 64), K-chunks that end inside a stage, chunks of one or two
    stages and K that is no multiple of the stage, against the direct-difference kernel and the fp64 oracle."""
    real, fake, f = _tile_inputs(B, K, 77 + B + K, T=6, J=4)
    G.cost_flags = L.COST_FORCE_MFMA
    got = G._Cost3.apply(real, fake, f[0], f[1], f[2], f[3], cases.SC).cpu().numpy()
    G.cost_flags = L.COST_FORCE_DIRECT
    direct = G._Cost3.apply(real, fake, f[0], f[1], f[2], f[3], cases.SC).cpu().numpy()
    G.cost_flags = 0
    ref = _cost3_oracle(real, fake, f)
    assert np.isfinite(got).all()
    for k in range(3):
        tol = 1e-5 * np.abs(ref[k]).max()
        np.testing.assert_allclose(got[k], ref[k], rtol=0, atol=tol)
        np.testing.assert_allclose(direct[k], ref[k], rtol=0, atol=tol)


@pytest.mark.parametrize("B", [48, 64, 128, 256])
def test_cost3_gram_sums_split_equals_one_call(L, B):
    """KCCOT_COST_GRAM_SUMS_ONLY + KCCOT_COST_FROM_GRAM_SUMS (the contraction-sharded caller's two calls, here without
    the all-reduce in between) give the bits of the one-shot call; and summing the Gram sums of two K-halves before
    the finalize step gives the full-K result to rounding (what two ranks do)."""
    import ctypes
    from kccotgan_amd import _lib
    from kccotgan_amd._lib import lib, ptr, check
    rng = np.random.default_rng(500 + B)
    K, T, J = 2048, 6, 4
    real = torch.from_numpy(rng.random((B, K), dtype=np.float32)).to(DEV)
    fake = (real + 0.02 * torch.randn(B, K, device=DEV)).clamp_(0, 1).contiguous()
    f = [torch.from_numpy(rng.random((B, T, J), dtype=np.float32)).to(DEV) for _ in range(4)]

    def call(x, y, k, flags, C3, ws):
        check(lib.kccot_pairwise_cost3_f32(ptr(x), ptr(y), B, k, cases.SC, ptr(f[0]), ptr(f[1]), ptr(f[2]), ptr(f[3]), T, J,
                                           flags, ptr(C3), ws.data_ptr(), ws.numel(), None), "pairwise_cost3")

    def span(k):
        off, cnt = ctypes.c_size_t(0), ctypes.c_size_t(0)
        check(lib.kccot_pairwise_cost3_gram_sums_span(B, k, ctypes.byref(off), ctypes.byref(cnt)), "span")
        return off.value, cnt.value

    ws = torch.empty(int(lib.kccot_pairwise_cost3_workspace_bytes(B, K)), dtype=torch.uint8, device=DEV)
    one, two = torch.empty(3, B, B, device=DEV), torch.empty(3, B, B, device=DEV)
    call(real, fake, K, 0, one, ws)
    call(real, fake, K, _lib.COST_GRAM_SUMS_ONLY, two, ws)
    call(real, fake, K, _lib.COST_FROM_GRAM_SUMS, two, ws)
    torch.cuda.synchronize()
    assert torch.equal(one, two)
    # two K-halves, summed like an all-reduce would
    Kh = K // 2
    off, cnt = span(Kh)
    assert cnt > 0
    halves = [real[:, :Kh].contiguous(), real[:, Kh:].contiguous()], [fake[:, :Kh].contiguous(), fake[:, Kh:].contiguous()]
    wsh = [torch.empty(int(lib.kccot_pairwise_cost3_workspace_bytes(B, Kh)), dtype=torch.uint8, device=DEV) for _ in range(2)]
    out = torch.empty(3, B, B, device=DEV)
    for r in range(2):
        call(halves[0][r], halves[1][r], Kh, _lib.COST_GRAM_SUMS_ONLY, out, wsh[r])
    g0 = wsh[0][off:off + 8 * cnt].view(torch.float64)
    g0 += wsh[1][off:off + 8 * cnt].view(torch.float64)
    call(halves[0][0], halves[1][0], Kh, _lib.COST_FROM_GRAM_SUMS, out, wsh[0])
    torch.cuda.synchronize()
    ref = one.cpu().numpy()
    np.testing.assert_allclose(out.cpu().numpy(), ref, rtol=0, atol=2e-6 * np.abs(ref).max())


@pytest.mark.parametrize("B", [128, 256])
def test_loss_at_larger_batches(G, B):
    """configs 3/4 batch sizes (decimated frames so that the CPU oracle finishes in seconds): direct
    cost kernel over several 64-row tiles + Sinkhorn at n = 128 (registers) / 256 (streaming)."""
    rng = np.random.default_rng(B)
    H, T, W, Cc, J = 8, 10, 8, 3, 8
    real = rng.random((B, H, T, W, Cc), dtype=np.float32)
    fake = np.clip(real + np.float32(0.05) * rng.standard_normal(real.shape, dtype=np.float32), 0, 1).astype(np.float32)
    f = {k: rng.random((B, T, J), dtype=np.float32) for k in ("h_fake", "m_real", "h_real", "m_fake")}
    t = {k: torch.from_numpy(v).to(DEV) for k, v in dict(real=real, fake=fake, **f).items()}
    wrt = ["fake", "h_fake", "m_real"]
    for k in wrt:
        t[k].requires_grad_(True)
    loss = G.compute_sinkhorn_loss(t["real"], t["fake"], cases.SC, 0.8, 100, t["h_fake"], t["m_real"], t["h_real"],
                                   t["m_fake"])
    grads = torch.autograd.grad(loss, [t[k] for k in wrt])
    d = {k: torch.from_numpy(v).double() for k, v in dict(real=real, fake=fake, **f).items()}
    for k in wrt:
        d[k].requires_grad_(True)
    ref = ot.compute_sinkhorn_loss(d["real"], d["fake"], cases.SC, 0.8, 100, d["h_fake"], d["m_real"], d["h_real"],
                                   d["m_fake"])
    gref = torch.autograd.grad(ref, [d[k] for k in wrt])
    assert rel(loss, ref) < 1e-4
    # near regime at O(10)-sized costs: same conditioning class as deci64_s0_near
    for k, a, b in zip(wrt, grads, gref):
        np.testing.assert_allclose(a.cpu().numpy(), b.numpy(), rtol=0, atol=grad_tol("deci64_s0_near", k) * float(b.abs().max()),
                                   err_msg=k)


# ---------------------------------------------------------------- the loss
@pytest.mark.parametrize("path", PATHS)
@pytest.mark.parametrize("shape,seed,regime", ALL)
def test_loss_matches_reference(G, L, shape, seed, regime, path):
    g, inp, t = load(shape, seed, regime)
    set_path(G, L, path, kdim(shape))
    loss = G.compute_sinkhorn_loss(t["real"], t["fake"], cases.SC, 0.8, 100, t["h_fake"], t["m_real"],
                                   t["h_real"], t["m_fake"], video=True)
    nits = G.last_info["compute_sinkhorn_loss"].cpu().numpy().tolist()
    assert nits == [int(g["nits_xy"]), int(g["nits_xx"]), int(g["nits_yy"])]
    assert rel(loss, g["loss"]) < 1e-4 and rel(loss, g["loss_f64"]) < 1e-4
    # quirk 1 (gan_utils.py:221-223): the eps / L arguments are ignored
    loss2 = G.compute_sinkhorn_loss(t["real"], t["fake"], cases.SC, 0.1, 5, t["h_fake"], t["m_real"],
                                    t["h_real"], t["m_fake"])
    assert float(loss2) == float(loss)
    pm = G.scale_invariante_martingale_regularization(t["m_real"], cases.LAM, cases.SC)
    assert rel(pm, g["pM"]) < 2e-5


def test_loss_honor_eps_l_opt_in(G):
    g, inp, t = load("small", 1, "far")
    x, y = flat(t["real"]), flat(t["fake"])
    got = G.compute_sinkhorn_loss(t["real"], t["fake"], cases.SC, 0.8, 20, t["h_fake"], t["m_real"], t["h_real"],
                                  t["m_fake"], honor_eps_l=True)
    xy = G.compute_sinkhorn(x, y, t["h_fake"], t["m_real"], cases.SC, epsilon=0.8, L=20)
    xx = G.compute_sinkhorn(x, x, t["h_real"], t["m_real"], cases.SC, epsilon=0.8, L=20)
    yy = G.compute_sinkhorn(y, y, t["h_fake"], t["m_fake"], cases.SC, epsilon=0.8, L=20)
    assert rel(got, 2 * xy - xx - yy) < 1e-5
    assert rel(xy, g["w_e0.8_L20"]) < 5e-5


def test_very_near_regime_keeps_relative_accuracy(G, L):
    """fake within 0.5% of real (late training / after smoothing): distances are 1e-4 of the
    norms, the regime where a plain Gram form loses everything.  Both paths must hold 1e-4."""
    B, H, T, W, C, J = cases.SHAPES["cfg1"]
    rng = np.random.default_rng(7)
    real = rng.random((B, H, T, W, C), dtype=np.float32)
    fake = np.clip(real + np.float32(0.005) * rng.standard_normal(real.shape, dtype=np.float32), 0, 1).astype(np.float32)
    f = {k: rng.random((B, T, J), dtype=np.float32) * np.float32(0.01) for k in ("hf", "mr", "hr", "mf")}
    ref = o.compute_sinkhorn_loss(real, fake, cases.SC, 0, 0, f["hf"], f["mr"], f["hr"], f["mf"], dtype=np.float64)
    tt = lambda a: torch.from_numpy(a).to(DEV)
    for path in PATHS:
        set_path(G, L, path)
        got = G.compute_sinkhorn_loss(tt(real), tt(fake), cases.SC, 0, 0, tt(f["hf"]), tt(f["mr"]), tt(f["hr"]), tt(f["mf"]))
        assert rel(got, ref) < 1e-4, (path, float(got), float(ref))


# ---------------------------------------------------------------- properties at full size (configs[1])
def test_cfg2_fixture_properties(G, L):
    g, inp, t = load("cfg2", 0, "near")
    B = t["real"].shape[0]
    args = lambda r, f, hf, mr, hr, mf: (r, f, cases.SC, 0.8, 100, hf, mr, hr, mf)
    base = G.compute_sinkhorn_loss(*args(t["real"], t["fake"], t["h_fake"], t["m_real"], t["h_real"], t["m_fake"]))
    # batch-permutation invariance (the property that makes batch sharding legal)
    perm = torch.randperm(B, generator=torch.Generator().manual_seed(3)).to(DEV)
    p = lambda a: a[perm].contiguous()
    permuted = G.compute_sinkhorn_loss(*args(p(t["real"]), p(t["fake"]), p(t["h_fake"]), p(t["m_real"]),
                                             p(t["h_real"]), p(t["m_fake"])))
    assert rel(permuted, base) < 2e-5
    # translation invariance of the l2 cost, linearity in scaling_coef, zero diagonal / symmetry
    x, y = t["real"].reshape(B, -1), t["fake"].reshape(B, -1)
    for path in ("direct", "mfma", "mfma_f32"):
        set_path(G, L, path)
        C = G.cost_xy(x, y, cases.SC)
        C2 = G.cost_xy(x, y, 2 * cases.SC)
        assert float((C2 - 2 * C).abs().max()) <= 2e-6 * float(C.abs().max())
        Cs = G.cost_xy(x + 0.25, y + 0.25, cases.SC)
        assert float((Cs - C).abs().max()) <= 2e-5 * float(C.abs().max())
        Cxx = G.cost_xy(x, x, cases.SC)
        assert float(Cxx.diagonal().abs().max()) == 0 and torch.equal(Cxx, Cxx.t())
    set_path(G, L, "auto")
    # plan is a coupling: feed the golden cost matrix, check both marginals and the cost
    from kccotgan_amd._lib import lib, ptr
    C = torch.from_numpy(g["C_xy"]).to(DEV).reshape(1, B, B).contiguous()
    cost = torch.empty(1, device=DEV); nits = torch.empty(2, dtype=torch.int32, device=DEV); pi = torch.empty(1, B, B, device=DEV)
    assert lib.kccot_sinkhorn_fwd_f32(ptr(C), 1, B, 1.0, 100, 100, 1e-2, 0, None, None, ptr(cost), ptr(nits), ptr(pi),
                                      None, 0, None) == 0
    torch.cuda.synchronize()
    np.testing.assert_allclose(pi[0].sum(0).cpu().numpy(), 1 / B, rtol=1e-5)
    np.testing.assert_allclose(pi[0].sum(1).cpu().numpy(), 1 / B, rtol=1e-3)
    assert rel(cost[0], g["w_xy"]) < 5e-5


# ---------------------------------------------------------------- gradients
def _grad_oracle(inp, wrt, fn):
    t = {k: torch.from_numpy(v).double() for k, v in inp.items()}
    for k in wrt:
        t[k].requires_grad_(True)
    val = fn(t)
    grads = torch.autograd.grad(val, [t[k] for k in wrt])
    return float(val), {k: gk.numpy() for k, gk in zip(wrt, grads)}


@pytest.mark.parametrize("path", PATHS)
@pytest.mark.parametrize("shape,seed,regime", [("tiny", 0, "near"), ("tiny", 1, "far"), ("small", 0, "near"),
                                               ("small", 1, "far"), ("deci64", 0, "near"), ("deci64", 1, "far")])
def test_loss_gradients_match_autograd_through_the_unrolled_loop(G, L, shape, seed, regime, path):
    g, inp, t = load(shape, seed, regime)
    set_path(G, L, path, kdim(shape))
    wrt = ["fake", "h_fake", "h_real", "m_real", "m_fake"]     # kernel_train.py:252,289 (never real)
    ref_val, ref = _grad_oracle(inp, wrt, lambda d: ot.compute_sinkhorn_loss(
        d["real"], d["fake"], cases.SC, 0.8, 100, d["h_fake"], d["m_real"], d["h_real"], d["m_fake"]))
    for k in wrt:
        t[k].requires_grad_(True)
    loss = G.compute_sinkhorn_loss(t["real"], t["fake"], cases.SC, 0.8, 100, t["h_fake"], t["m_real"], t["h_real"],
                                   t["m_fake"])
    grads = torch.autograd.grad(loss, [t[k] for k in wrt])
    assert rel(loss, ref_val) < 1e-4
    name = cases.case_name(shape, seed, regime)
    for k, gk in zip(wrt, grads):
        gk = gk.cpu().numpy()
        scale = np.abs(ref[k]).max()
        np.testing.assert_allclose(gk, ref[k], rtol=0, atol=grad_tol(name, k) * scale, err_msg=k)


@pytest.mark.parametrize("path", ["auto", "mfma_f32"])
@pytest.mark.parametrize("seed,regime", [(0, "near"), (1, "far")])
def test_cfg2_full_size_gradients_match_fixture(G, L, seed, regime, path):
    """BASELINE configs[1] (B = 64, K = 122 880), the shape kernel_train.py:287-289 differentiates: dh / dM in
    full, and the 31 MB video gradient through its per-sample norms and sums, 8 seeded random projections per
    sample and every 97th entry, against fp64 autograd through the unrolled loop of the oracle
    (tests/golden/grad_cfg2_*.npz, written by tests/golden/make_grad_golden.py in the build container)."""
    g, inp, t = load("cfg2", seed, regime)
    name = cases.case_name("cfg2", seed, regime)
    fx = np.load(os.path.join(GOLD, "grad_%s.npz" % name))
    set_path(G, L, path)
    wrt = ["fake", "h_fake", "h_real", "m_real", "m_fake"]
    for k in wrt:
        t[k].requires_grad_(True)
    loss = G.compute_sinkhorn_loss(t["real"], t["fake"], cases.SC, 0.8, 100, t["h_fake"], t["m_real"], t["h_real"],
                                   t["m_fake"])
    grads = dict(zip(wrt, (a.cpu().numpy().astype(np.float64) for a in torch.autograd.grad(loss, [t[k] for k in wrt]))))
    assert rel(loss, fx["loss_f64"]) < 1e-4
    for k in wrt[1:]:
        np.testing.assert_allclose(grads[k], fx[k], rtol=0, atol=grad_tol(name, k) * np.abs(fx[k]).max(), err_msg=k)
    B = grads["fake"].shape[0]
    df = grads["fake"].reshape(B, -1)
    tol = grad_tol(name, "fake")
    amax = float(fx["dfake_absmax"])
    np.testing.assert_allclose(df[:, ::97], fx["dfake_strided"], rtol=0, atol=tol * amax)
    assert abs(np.abs(df).max() - amax) <= tol * amax
    np.testing.assert_allclose(np.sqrt((df ** 2).sum(1)), fx["dfake_norm"], rtol=tol * 4)
    K = df.shape[1]
    proj = np.random.default_rng(1234).standard_normal((8, K))
    # a projection sums K entries with random signs: its error grows like sqrt(K) * entry error at worst K * ...
    np.testing.assert_allclose(df @ proj.T, fx["dfake_proj"], rtol=0, atol=tol * amax * np.sqrt(K) * 4)
    np.testing.assert_allclose(df.sum(1), fx["dfake_sum"], rtol=0, atol=tol * amax * np.sqrt(K) * 4)


def test_loss_and_gradients_on_random_ragged_shapes(G):
    """30 seeded random configurations the fixtures do not hold -- B from 1 to 70 (odd, prime, one sample, just above 64),
    T from 2, frames down to 1 x 1, C = 1..3, J = 1..5, K not a multiple of 4 (scalar tail of the direct kernel), near and
    far regimes, O(1) to O(100) costs -- against the fp64 torch oracle (value at 1e-4, gradients at the floor / 4 x the
    far-regime gap).  Whatever kernel path the dispatcher picks for the shape is the one tested."""
    rng = np.random.default_rng(20262)
    for trial in range(30):
        B = int(rng.choice([1, 2, 3, 5, 7, 11, 16, 23, 33, 48, 63, 65, 70]))
        T, H, W = int(rng.integers(2, 7)), int(rng.integers(1, 7)), int(rng.integers(1, 7))
        Cc, J = int(rng.integers(1, 4)), int(rng.integers(1, 6))
        far = bool(rng.integers(0, 2))
        real = rng.random((B, H, T, W, Cc), dtype=np.float32)
        fake = rng.random(real.shape, dtype=np.float32) if far else \
            np.clip(real + np.float32(0.05) * rng.standard_normal(real.shape, dtype=np.float32), 0, 1).astype(np.float32)
        f = {k: rng.random((B, T, J), dtype=np.float32) for k in ("h_fake", "m_real", "h_real", "m_fake")}
        sc = float(rng.choice([cases.SC, 1.0, 0.01]))
        inp = dict(real=real, fake=fake, **f)
        wrt = ["fake", "h_fake", "h_real", "m_real", "m_fake"]
        t = {k: torch.from_numpy(v).to(DEV) for k, v in inp.items()}
        d = {k: torch.from_numpy(v).double() for k, v in inp.items()}
        for k in wrt:
            t[k].requires_grad_(True)
            d[k].requires_grad_(True)
        loss = G.compute_sinkhorn_loss(t["real"], t["fake"], sc, 0.8, 100, t["h_fake"], t["m_real"], t["h_real"], t["m_fake"])
        ref = ot.compute_sinkhorn_loss(d["real"], d["fake"], sc, 0.8, 100, d["h_fake"], d["m_real"], d["h_real"], d["m_fake"])
        tag = (trial, B, T, H, W, Cc, J, far, sc)
        assert abs(float(loss) - float(ref)) <= 1e-4 * abs(float(ref)) + 2e-6, (tag, float(loss), float(ref))
        grads = torch.autograd.grad(loss, [t[k] for k in wrt])
        gref = torch.autograd.grad(ref, [d[k] for k in wrt])
        for k, a, b in zip(wrt, grads, gref):
            scale = float(b.abs().max())
            if scale == 0:
                assert float(a.abs().max()) == 0, (tag, k)
                continue
            tol = max(GRAD_TOL_FLOOR, GRAD_TOL_FACTOR * 2.5e-4 if far else 0.0) * 4
            np.testing.assert_allclose(a.cpu().numpy(), b.numpy(), rtol=0, atol=tol * scale, err_msg=str((tag, k)))


def test_general_cost_and_sinkhorn_gradients(G):
    """compute_sinkhorn (both operands differentiable, bi-causal) and the martingale penalty."""
    g, inp, t = load("small", 1, "far")
    wrt = ["real", "fake", "h_fake", "m_real", "h_real", "m_fake"]

    def fn(d):
        x, y = ot.flatten_video(d["real"]), ot.flatten_video(d["fake"])
        return (ot.compute_sinkhorn(x, y, d["h_fake"], d["m_real"], cases.SC, hx=d["h_real"], My=d["m_fake"],
                                    epsilon=0.8, L=30, bi_causal=True)
                + ot.compute_sinkhorn(x, x, d["h_real"], d["m_real"], cases.SC, epsilon=0.8, L=30)
                + 3.0 * ot.scale_invariante_martingale_regularization(d["m_real"], 1.5, cases.SC))

    ref_val, ref = _grad_oracle(inp, wrt, fn)
    for k in wrt:
        t[k].requires_grad_(True)
    x, y = flat(t["real"]), flat(t["fake"])
    val = (G.compute_sinkhorn(x, y, t["h_fake"], t["m_real"], cases.SC, hx=t["h_real"], My=t["m_fake"], epsilon=0.8,
                              L=30, bi_causal=True)
           + G.compute_sinkhorn(x, x, t["h_real"], t["m_real"], cases.SC, epsilon=0.8, L=30)
           + 3.0 * G.scale_invariante_martingale_regularization(t["m_real"], 1.5, cases.SC))
    grads = torch.autograd.grad(val, [t[k] for k in wrt])
    assert rel(val, ref_val) < 1e-4
    for k, gk in zip(wrt, grads):
        tol = max(grad_tol("small_s1_far", kk) for kk in GRAD_GAP["small_s1_far"])       # same inputs, eps 0.8 / L 30
        np.testing.assert_allclose(gk.cpu().numpy(), ref[k], rtol=0, atol=tol * np.abs(ref[k]).max(), err_msg=k)


def test_martingale_kats(G):
    M = torch.rand(4, 1, 3, generator=torch.Generator().manual_seed(2)).repeat(1, 6, 1).to(DEV)
    assert float(G.scale_invariante_martingale_regularization(M, 1.0, 0.5)) == 0      # constant in time
    Mn = np.random.default_rng(9).random((6, 5, 4), dtype=np.float32)
    got = G.scale_invariante_martingale_regularization(torch.from_numpy(Mn).to(DEV), 0.7, 0.3)
    assert rel(got, o.scale_invariante_martingale_regularization(Mn, 0.7, 0.3, np.float64)) < 1e-5


# ---------------------------------------------------------------- kernel smoothing
@pytest.mark.parametrize("shape", [(2, 8, 9, 10, 1), (3, 16, 7, 12, 3), (2, 64, 20, 64, 1)])
def test_smoothing_matches_oracle(shape):
    from kccotgan_amd.data_utils import KernelSmoothing
    from oracle import smoothing_np as sm
    ks = KernelSmoothing(temporal_kernel_size=6, spatial_kernel_size=6)      # kernel_train.py:216
    v = np.random.default_rng(shape[0]).random(shape, dtype=np.float32)
    vt = torch.from_numpy(v).to(DEV)
    for sigma in (5.0, 1.3):
        np.testing.assert_allclose(ks.temporal_convolution(vt, sigma).cpu().numpy(),
                                   sm.temporal_convolution(v, sigma), rtol=1e-5, atol=1e-6)
        got = ks.gaussian_convolution3D(vt, sigma).cpu().numpy()
        np.testing.assert_allclose(got, sm.gaussian_convolution3D_separable(v, sigma), rtol=1e-5, atol=1e-6)
        assert float(got.max()) == 1.0
        np.testing.assert_allclose(ks.spatial_convolution(vt, sigma).cpu().numpy(),
                                   sm.spatial_convolution_reflect(v, sigma), rtol=1e-5, atol=1e-6)
    if v.size < 20000:   # the literal dense 7x7x7 form of data_utils.py:552-582
        np.testing.assert_allclose(ks.gaussian_convolution3D(vt, 2.0).cpu().numpy(),
                                   sm.gaussian_convolution3D(v, 2.0), rtol=2e-5, atol=2e-6)
    const = torch.full(shape, 0.37, device=DEV)
    np.testing.assert_allclose(ks.gaussian_convolution3D(const, 5.0).cpu().numpy(), 1.0, rtol=1e-6)


@pytest.mark.parametrize("shape,ksize", [((2, 8, 9, 10, 2), 6),      # LDS plane kernel for W (W*C not a multiple of 4 pieces)
                                         ((2, 8, 9, 16, 1), 6),      # register W kernel (C == 1), short axes
                                         ((2, 40, 12, 8, 1), 6),     # H > 32: the 64-step walk; W = 8: two pieces per row
                                         ((1, 12, 33, 12, 1), 8),    # radius 4, T > 32
                                         ((2, 9, 10, 12, 3), 8)])    # radius 4, C = 3
def test_smoothing_gradient_matches_autograd(shape, ksize):
    from kccotgan_amd.data_utils import KernelSmoothing
    from oracle import smoothing_torch as st
    ks = KernelSmoothing(ksize, ksize)
    v = np.random.default_rng(4).random(shape, dtype=np.float32)
    wgt = np.random.default_rng(5).standard_normal(v.shape).astype(np.float32)
    for axes, fn in (((2,), ks.temporal_convolution), ((2, 1, 3), ks.gaussian_convolution3D)):
        a = torch.from_numpy(v).double().requires_grad_(True)
        (st.smooth(a, 2.0, ksize // 2, axes) * torch.from_numpy(wgt).double()).sum().backward()
        b = torch.from_numpy(v).to(DEV).requires_grad_(True)
        (fn(b, 2.0) * torch.from_numpy(wgt).to(DEV)).sum().backward()
        np.testing.assert_allclose(b.grad.cpu().numpy(), a.grad.numpy(), rtol=0, atol=2e-4 * float(a.grad.abs().max()))


def test_smoothing_stream_and_legacy_paths_agree(L):
    """The streamed kernels (register walks along T / H, register W kernel) against the per-axis chain
    they replace (option "smooth_stream" = 0): forward and gradient, at a shape that takes every fast path."""
    from kccotgan_amd.data_utils import KernelSmoothing
    ks = KernelSmoothing(6, 6)
    v = torch.rand((3, 40, 30, 64, 1), device=DEV, generator=torch.Generator(device=DEV).manual_seed(3))
    wgt = torch.randn(v.shape, device=DEV, generator=torch.Generator(device=DEV).manual_seed(4))
    res = {}
    for mode in ("stream", "legacy"):
        with L.options(smooth_stream=0 if mode == "legacy" else 1):
            for name, fn in (("t", ks.temporal_convolution), ("3d", ks.gaussian_convolution3D)):
                x = v.clone().requires_grad_(True)
                y = fn(x, 3.0)
                (g,) = torch.autograd.grad(y, x, wgt)
                res[mode, name] = (y.detach().cpu().numpy(), g.cpu().numpy())
    for name in ("t", "3d"):
        np.testing.assert_allclose(res["stream", name][0], res["legacy", name][0], rtol=2e-6, atol=1e-7)
        gs, gl = res["stream", name][1], res["legacy", name][1]
        np.testing.assert_allclose(gs, gl, rtol=0, atol=2e-6 * float(np.abs(gl).max()))


@pytest.mark.parametrize("shape,ksize", [((3, 40, 30, 64, 1), 6),      # H cut into segments (halo planes), two column tiles
                                         ((2, 64, 30, 64, 1), 6),      # configs[1]'s frames
                                         ((2, 24, 30, 64, 3), 6),      # configs[3]'s frames, C = 3
                                         ((1, 20, 48, 128, 3), 6),     # configs[4]'s plane: three items per thread, four tiles
                                         ((2, 12, 12, 16, 1), 8),      # radius 4
                                         ((2, 11, 14, 24, 3), 8),      # radius 4, C = 3, odd H
                                         ((70, 9, 8, 8, 1), 6),        # T = 2 R + 2: every row but two is mirrored; one tile
                                         ((4200, 8, 8, 8, 1), 6)])     # > 4096 workgroups: the maxima are reduced by a launch of their own
def test_fused_3d_smoothing_is_bit_identical_to_the_chain(L, shape, ksize):
    """Round 4: gaussian_convolution3D as ONE pass per phase (csrc/smooth.hip, smooth_fused3: T and W stencils through LDS, H
    stencil over a register window while the workgroup walks along H; maxima pass + writing pass = three tensor moves) against
    the chain of per-axis stages (option "smooth_fused3" = 0: five to seven moves).  Same fma order on every axis: the same
    bits, normalised and raw (the batch-sharded protocol's two phases), and a maximum of exactly 1.  That the fused kernel is
    the one that ran is read off the workspace: it never touches the tensor-sized intermediate buffer the chain writes."""
    from kccotgan_amd._lib import lib, ptr, check
    from kccotgan_amd import _lib
    B, H, T, W, C = shape
    r = ksize // 2
    axes = _lib.SMOOTH_T | _lib.SMOOTH_H | _lib.SMOOTH_W
    v = torch.rand(shape, device=DEV, generator=torch.Generator(device=DEV).manual_seed(sum(shape)))
    n = v.numel()
    res = {}
    for fused in (2, 0):         # 2: wherever the kernel can run (1 = only where it is faster: C = 3 from 20 M elements on)
        with L.options(smooth_fused3=fused):
            wsb = int(lib.kccot_smooth_workspace_bytes(B, H, T, W, C))
            outs = []
            for flags in (axes, axes | _lib.SMOOTH_NO_DIVIDE, axes | _lib.SMOOTH_EXTERNAL_MAX):
                out = torch.full(shape, float("nan"), device=DEV)
                mx = outs[1].clone() if flags & _lib.SMOOTH_EXTERNAL_MAX else torch.zeros(1, device=DEV)   # phase 2 of the sharded call
                buf = torch.empty(wsb // 4 + 64, device=DEV)
                buf[:n] = -7.0                              # the tensor-sized intermediate buffer is the head of the workspace
                check(lib.kccot_smooth_fwd_f32(ptr(v), B, H, T, W, C, 2.0, r, flags, ptr(out), ptr(mx), buf.data_ptr(), wsb, None),
                      "smooth_fwd")
                torch.cuda.synchronize()
                outs += [out, mx.clone()]
                touched = bool((buf[:n] != -7.0).any())
                assert touched == (fused == 0), (shape, fused, "intermediate buffer touched" if touched else "fused kernel did not run")
            res[fused] = outs
    for a, b in zip(res[2], res[0]):
        assert torch.equal(a, b), shape
    assert float(res[2][0].max()) == 1.0
    assert torch.equal(res[2][4], res[2][0])        # division by the handed-in maximum = the one-call result


def test_fused_3d_smoothing_is_the_default_where_it_was_measured_faster(L):
    """Option "smooth_fused3" = 1 (default): three channels from 20 M elements on (profiles/r4_ab_smooth_fused3.txt)."""
    from kccotgan_amd._lib import lib, ptr, check
    from kccotgan_amd import _lib
    axes = _lib.SMOOTH_T | _lib.SMOOTH_H | _lib.SMOOTH_W
    for shape, fused in (((64, 64, 30, 64, 3), True), ((32, 64, 30, 64, 3), False), ((64, 64, 30, 64, 1), False)):
        B, H, T, W, C = shape
        v = torch.rand(shape, device=DEV, generator=torch.Generator(device=DEV).manual_seed(B + C))
        wsb = int(lib.kccot_smooth_workspace_bytes(B, H, T, W, C))
        buf = torch.empty(wsb // 4 + 64, device=DEV)
        buf[:v.numel()] = -7.0
        out, mx = torch.empty_like(v), torch.zeros(1, device=DEV)
        with L.options(smooth_fused3=1):        # (the default, also under a KCCOT_OPTIONS seed that forces the kernels on or off)
            check(lib.kccot_smooth_fwd_f32(ptr(v), B, H, T, W, C, 2.0, 3, axes, ptr(out), ptr(mx), buf.data_ptr(), wsb, None), "smooth_fwd")
        torch.cuda.synchronize()
        assert bool((buf[:v.numel()] != -7.0).any()) == (not fused), shape
        assert float(out.max()) == 1.0


# ---------------------------------------------------------------- extension: RBF kernel / MMD (no reference behaviour)
def test_rbf_mmd_matches_sklearn_definition():
    from sklearn.metrics.pairwise import rbf_kernel
    from kccotgan_amd import mmd
    rng = np.random.default_rng(21)
    for B, K in ((16, 320), (64, 2048)):
        x = rng.random((B, K), dtype=np.float32)
        y = np.clip(x + 0.2 * rng.standard_normal((B, K), dtype=np.float32), 0, 1).astype(np.float32)
        K3, m = mmd.rbf_kernels(torch.from_numpy(x).to(DEV), torch.from_numpy(y).to(DEV))
        ref = [rbf_kernel(x.astype(np.float64), y.astype(np.float64)), rbf_kernel(x.astype(np.float64)),
               rbf_kernel(y.astype(np.float64))]
        for p in range(3):
            np.testing.assert_allclose(K3[p].cpu().numpy(), ref[p], rtol=2e-5, atol=1e-7)
        want = ref[1].mean() + ref[2].mean() - 2 * ref[0].mean()
        assert abs(float(m) - want) < 1e-5 * max(abs(want), 1e-3)
        g = 0.01
        m2 = mmd.rbf_mmd2(torch.from_numpy(x).to(DEV), torch.from_numpy(y).to(DEV), gamma=g)
        want2 = rbf_kernel(x, x, g).mean() + rbf_kernel(y, y, g).mean() - 2 * rbf_kernel(x, y, g).mean()
        assert abs(float(m2) - want2) < 1e-5 * max(abs(want2), 1e-3)


def test_rbf_mmd_gradient_wrt_fake():
    """d mmd^2 / d fake against fp64 autograd of the same definition (B = 16 small K; B = 128 through the blocked costs)."""
    from kccotgan_amd import mmd
    rng = np.random.default_rng(22)
    for B, K, gamma in ((16, 320, None), (64, 2048, 0.002), (128, 512, 0.01)):
        x = rng.random((B, K), dtype=np.float32)
        y = np.clip(x + 0.2 * rng.standard_normal((B, K), dtype=np.float32), 0, 1).astype(np.float32)
        yt = torch.from_numpy(y).to(DEV).requires_grad_(True)
        m = mmd.rbf_mmd2(torch.from_numpy(x).to(DEV), yt, gamma)
        (3.0 * m).backward()
        xd, yd = torch.from_numpy(x).double(), torch.from_numpy(y).double().requires_grad_(True)
        gm = gamma if gamma is not None else 1.0 / K
        kern = lambda a, b: torch.exp(-gm * torch.cdist(a, b) ** 2)
        ref = kern(xd, xd).mean() + kern(yd, yd).mean() - 2 * kern(xd, yd).mean()
        (3.0 * ref).backward()
        assert abs(float(m) - float(ref)) < 1e-5 * max(abs(float(ref)), 1e-3)
        gref = yd.grad.numpy()
        np.testing.assert_allclose(yt.grad.cpu().numpy(), gref, rtol=0, atol=2e-5 * np.abs(gref).max())


# ---------------------------------------------------------------- one-call loss and graph capture
def test_one_call_loss_equals_staged_path(G, L):
    """The one-call loss entry points (cost assembly + fused solve/sweep, or the two-kernel sequence where the history
    does not fit LDS) against _Cost3 followed by _SinkhornDivergence: bit-identical values and gradients.  (The
    staged reverse sweep is pinned to the fused kernel's 8 lanes per line for 32 < n <= 64, where it defaults to 16:
    another summation order of the same terms otherwise.)  Batches that take the slab layout of the Gram partials (B = 8,
    16: at most 32 rows per operand) and the compact record of round 4 (B = 64, decimated and full K): loss, gradients
    AND the three cost matrices the call leaves behind."""
    L.set_option("sinkhorn_lanes_per_line", 8)
    for shape, seed, regime in (SMALL[0], ("small", 1, "far"), ("deci64", 0, "near"), ("cfg1", 1, "far"), ("cfg2", 1, "far")):
        g, inp, t = load(shape, seed, regime)
        wrt = ["fake", "h_fake", "h_real", "m_real", "m_fake"]
        outs = []
        for fused in (True, False):
            tt = {k: v.clone().requires_grad_(k in wrt) for k, v in t.items()}
            if fused:
                loss = G.compute_sinkhorn_loss(tt["real"], tt["fake"], cases.SC, 0.8, 100, tt["h_fake"], tt["m_real"],
                                               tt["h_real"], tt["m_fake"], video=True)
                C3 = G.last_info["compute_sinkhorn_loss_C3"].clone()
            else:
                C3 = G._Cost3.apply(G._flat2(tt["real"]), G._flat2(tt["fake"]), G._feat(tt["h_fake"]), G._feat(tt["h_real"]),
                                    G._feat(tt["m_real"]), G._feat(tt["m_fake"]), float(cases.SC))
                loss = G._SinkhornDivergence.apply(C3, 1.0, 100, 100, "staged")
            grads = torch.autograd.grad(loss, [tt[k] for k in wrt])
            outs.append([loss.detach().reshape(1), C3.detach().reshape(-1)] + [x.reshape(-1) for x in grads])
        for a, b in zip(*outs):
            assert _same_bits(a.cpu().numpy(), b.cpu().numpy()), (shape, seed, regime)


@pytest.mark.parametrize("B,K", [(64, 4096), (40, 2052), (16, 512), (8, 260), (64, 64 * 33 + 4)])
def test_backward_in_one_launch_equals_the_two_launch_form(G, L, B, K):
    """Round 4: at B <= 64 (B % 8 == 0) the loss's backward is ONE launch -- the apply kernel's consumer waves form their
    coefficient fragments from dC, its producer waves compute the four feature gradients (csrc/cost_bwd.hip,
    apply_coeffs_x3_loss3) -- option "apply_one_launch" = 0 keeps coefficient build + apply.  Same products in the same
    order except the diagonal sum of W (a pair of 32-term sums instead of a 256-thread tree; it multiplies the sample's own
    row and then cancels against the other terms): the video gradient agrees to 4e-6 of max|grad| (measured <= 2.2e-6; the
    oracle tolerance is 2.5e-5), the feature gradients (same terms, same order) bit for bit.  Shapes: full and partial row blocks,
    K with a ragged last tile, fewer column tiles than CUs (a workgroup then owns several feature-gradient tasks)."""
    T, J = 6, 8
    gen = torch.Generator(device=DEV).manual_seed(B * 1000 + K)
    real = torch.rand((B, K), device=DEV, generator=gen)
    fake = (real + 0.05 * torch.randn((B, K), device=DEV, generator=gen)).clamp_(0, 1)
    f = {k: torch.rand((B, T, J), device=DEV, generator=gen) for k in ("h_fake", "m_real", "h_real", "m_fake")}
    wrt = ["fake", "h_fake", "h_real", "m_real", "m_fake"]
    res = {}
    for mode in (1, 0):
        with L.options(apply_one_launch=mode):
            tt = dict(f, fake=fake.clone())
            for k in wrt:
                tt[k] = tt[k].clone().requires_grad_(True)
            loss = G.compute_sinkhorn_loss(real, tt["fake"], cases.SC, 0.8, 100, tt["h_fake"], tt["m_real"], tt["h_real"],
                                           tt["m_fake"], video=False)
            res[mode] = [loss.detach()] + list(torch.autograd.grad(loss, [tt[k] for k in wrt]))
    assert torch.equal(res[0][0], res[1][0])
    scale = float(res[0][1].abs().max())
    assert float((res[1][1] - res[0][1]).abs().max()) <= 4e-6 * scale
    for a, b in zip(res[1][2:], res[0][2:]):
        assert torch.equal(a, b)


def test_graphed_loss_is_bit_identical(G, L):
    """A captured hipGraph replays the eager kernels with the eager arguments: same bits; and a replay
    after new inputs were copied into the static buffers follows them."""
    from kccotgan_amd.graph import GraphedLossStep, graphed_loss
    shape, seed, regime = SMALL[0]
    g, inp, t = load(shape, seed, regime)
    wrt = ["fake", "h_fake", "h_real", "m_real", "m_fake"]

    def eager(tt):
        tt = {k: v.clone().requires_grad_(k in wrt) for k, v in tt.items()}
        loss = G.compute_sinkhorn_loss(tt["real"], tt["fake"], cases.SC, 0.8, 100, tt["h_fake"], tt["m_real"],
                                       tt["h_real"], tt["m_fake"], video=True)
        return loss.detach(), dict(zip(wrt, torch.autograd.grad(loss, [tt[k] for k in wrt])))

    step = GraphedLossStep(t, cases.SC)
    t2 = {k: (v * 0.5 + 0.25 * torch.rand_like(v)) for k, v in t.items()}
    for cur in (t, t2, t):
        loss, grads = step(**cur)
        eloss, egrads = eager(cur)
        torch.cuda.synchronize()
        assert _same_bits(loss.reshape(1).cpu().numpy(), eloss.reshape(1).cpu().numpy())
        for k in wrt:
            assert _same_bits(grads[k].cpu().numpy(), egrads[k].cpu().numpy()), k
    # forward / backward captured separately, inside a larger autograd graph
    sample = {k: v.clone().requires_grad_(k in wrt) for k, v in t.items()}
    f = graphed_loss(sample, cases.SC)
    tt = {k: v.clone().requires_grad_(k in wrt) for k, v in t2.items()}
    scale = torch.tensor(-1.0, device=DEV)
    out = f(tt["real"], tt["fake"] * 1.0, tt["h_fake"], tt["m_real"], tt["h_real"], tt["m_fake"]) * scale
    out.backward()
    eloss, egrads = eager(t2)
    assert _same_bits(out.detach().reshape(1).cpu().numpy(), (-eloss).reshape(1).cpu().numpy())
    for k in wrt:
        np.testing.assert_array_equal(tt[k].grad.cpu().numpy(), -egrads[k].cpu().numpy())


@pytest.mark.parametrize("shortcut", ["on", "off"])
@pytest.mark.parametrize("shape,seed,regime", [("tiny", 0, "near"), ("small", 1, "far"), ("cfg1", 1, "far"), ("deci64", 0, "near"),
                                               ("cfg2", 0, "near"), ("cfg2", 1, "far")])
def test_fused_solve_and_sweep_equals_the_two_kernel_path(G, L, shape, seed, regime, shortcut):
    """compute_sinkhorn_loss with a gradient runs the three solves AND the reverse sweep as one persistent launch
    with the dual history in LDS (sinkhorn_fused_reg) wherever that history fits; option "sinkhorn_fused" = 0 restores the
    forward kernel + global history + sweep kernel.  Same arithmetic instruction for instruction: loss, costs and
    iteration counts are bit-identical, and so are all gradients when the two-kernel sweep runs at the same
    lanes-per-line (option "sinkhorn_lanes_per_line" = 8 for 32 < n <= 64, where it otherwise uses 16).  A non-unit upstream gradient rides
    on scaling_coef in the fused form (one extra rounding)."""
    g, inp, _ = load(shape, seed, regime)
    if shortcut == "off":
        L.set_option("sinkhorn_shortcut", 0)
    wrt = ["fake", "h_fake", "h_real", "m_real", "m_fake"]

    def run(upstream):
        t = {k: torch.from_numpy(v).to(DEV) for k, v in inp.items()}
        for k in wrt:
            t[k].requires_grad_(True)
        loss = G.compute_sinkhorn_loss(t["real"], t["fake"], cases.SC, 0.8, 100, t["h_fake"], t["m_real"], t["h_real"], t["m_fake"])
        grads = torch.autograd.grad(loss * upstream, [t[k] for k in wrt])
        return (loss.detach().cpu().numpy(), G.last_info["compute_sinkhorn_loss"].tolist(),
                G.last_info["compute_sinkhorn_loss_costs"].cpu().numpy(), bool(G.last_info["compute_sinkhorn_loss_fused_sweep"]),
                [a.cpu().numpy() for a in grads])

    fused = {u: run(u) for u in (1.0, -0.37)}
    L.set_option("sinkhorn_fused", 0)
    L.set_option("sinkhorn_lanes_per_line", 8)
    plain = {u: run(u) for u in (1.0, -0.37)}
    assert fused[1.0][3] and not plain[1.0][3]
    assert _same_bits(fused[1.0][0].reshape(1), plain[1.0][0].reshape(1)) and fused[1.0][1] == plain[1.0][1]
    assert _same_bits(fused[1.0][2], plain[1.0][2])
    assert rel(fused[1.0][0], g["loss"]) < 1e-4
    for a, b in zip(fused[1.0][4], plain[1.0][4]):
        np.testing.assert_array_equal(a, b)
    # upstream != 1: the two-kernel sweep carries the factor from its first term on, the fused form multiplies at the
    # end -- another rounding pattern of a cancelling / ill-conditioned sum, so the comparison is at twice the
    # gradient tolerance of the case (two fp32 evaluations, each within it of the fp64 value)
    name = cases.case_name(shape, seed, regime)
    for k, a, b in zip(wrt, fused[-0.37][4], plain[-0.37][4]):
        np.testing.assert_allclose(a, b, rtol=0, atol=2 * grad_tol(name, k) * np.abs(b).max())


@pytest.mark.parametrize("Lc", [0, 1, 3, 101, 130])
def test_fused_path_at_degenerate_and_late_stopping_iteration_counts(G, L, Lc):
    """L = 0 (no iteration: plan exp(-C/eps), gan_utils.py:151 never enters the loop), 1, 3, and L > Lmin = 100 where the
    stop rule may fire (quirk 2): fused launch == two-kernel path bit for bit, both == the fp64 oracle."""
    g, inp, _ = load("small", 1, "far")
    wrt = ["fake", "h_fake", "m_real"]

    def run():
        t = {k: torch.from_numpy(v).to(DEV) for k, v in inp.items()}
        for k in wrt:
            t[k].requires_grad_(True)
        loss = G.compute_sinkhorn_loss(t["real"], t["fake"], cases.SC, 0.5, Lc, t["h_fake"], t["m_real"], t["h_real"], t["m_fake"],
                                       honor_eps_l=True)
        grads = torch.autograd.grad(loss, [t[k] for k in wrt])
        return loss.detach().cpu().numpy().reshape(1), G.last_info["compute_sinkhorn_loss"].tolist(), [a.cpu().numpy() for a in grads]

    fused = run()
    assert bool(G.last_info["compute_sinkhorn_loss_fused_sweep"])
    L.set_option("sinkhorn_fused", 0)
    plain = run()
    assert _same_bits(fused[0], plain[0]) and fused[1] == plain[1]
    for a, b in zip(fused[2], plain[2]):
        np.testing.assert_array_equal(a, b)
    d = {k: torch.from_numpy(v).double() for k, v in inp.items()}
    for k in wrt:
        d[k].requires_grad_(True)
    x, y = ot.flatten_video(d["real"]), ot.flatten_video(d["fake"])
    costs = [ot.sinkhorn_from_cost(ot.modified_cost(a, b, d[h], d[m], cases.SC), 0.5, Lc) for a, b, h, m in
             ((x, y, "h_fake", "m_real"), (x, x, "h_real", "m_real"), (y, y, "h_fake", "m_fake"))]
    ref = 2.0 * costs[0][0] - costs[1][0] - costs[2][0]
    assert rel(fused[0][0], ref) < 1e-4 and fused[1][:3] == [c[1] for c in costs]
    gref = torch.autograd.grad(ref, [d[k] for k in wrt])
    for a, b in zip(fused[2], gref):
        np.testing.assert_allclose(a, b.numpy(), rtol=0, atol=4 * GRAD_TOL_FLOOR * float(b.abs().max()))


# ---------------------------------------------------------------- size-independent properties at BASELINE full sizes
FULL_SIZE = [((64, 64, 30, 64, 1), 100), ((128, 64, 30, 64, 3), 100),       # BASELINE configs[1], configs[2]
             ((256, 64, 30, 64, 3), 200), ((512, 128, 48, 128, 3), 300)]      # configs[3] (L = 200), configs[4] (L = 300)


def _full_size_inputs(shape):
    B, H, T, W, C = shape
    gen = torch.Generator(device=DEV).manual_seed(B)
    real = torch.rand(shape, device=DEV, generator=gen)
    fake = (real + 0.05 * torch.randn(shape, device=DEV, generator=gen)).clamp_(0, 1)
    f = {k: torch.rand((B, T, 8), device=DEV, generator=gen) for k in ("h_fake", "m_real", "h_real", "m_fake")}
    return real, fake, f, gen


@pytest.mark.parametrize("shape,Lc", FULL_SIZE)
def test_full_size_properties(G, shape, Lc):
    """The domain's exact / near-exact invariants at the full BASELINE sizes (configs[3]/[4] on ONE GPU; L > 100
    through the keyword-only `honor_eps_l`, eps = 1 as the reference effectively runs):
      * fake == real with matching features -> the three problems coincide and the divergence is EXACTLY 0;
      * relabelling the batch (same permutation of every tensor) leaves the loss unchanged up to the
        rounding of a different summation order, and permutes the gradient accordingly."""
    B, H, T, W, C = shape
    real, fake, f, gen = _full_size_inputs(shape)
    kw = dict(honor_eps_l=True) if Lc != 100 else {}
    same = G.compute_sinkhorn_loss(real, real.clone(), cases.SC, 1.0, Lc, f["h_real"], f["m_real"], f["h_real"], f["m_real"], **kw)
    assert float(same) == 0.0
    fk = fake.clone().requires_grad_(True)
    loss = G.compute_sinkhorn_loss(real, fk, cases.SC, 1.0, Lc, f["h_fake"], f["m_real"], f["h_real"], f["m_fake"], **kw)
    nits = G.last_info["compute_sinkhorn_loss"].tolist()
    assert all(min(100, Lc) <= n <= Lc for n in nits[:3]), nits         # quirk 2: never fewer than Lmin = 100
    (g,) = torch.autograd.grad(loss, fk)
    assert bool(torch.isfinite(loss)) and bool(torch.isfinite(g).all())
    perm = torch.randperm(B, device=DEV, generator=gen)
    fk2 = fake[perm].clone().requires_grad_(True)
    loss2 = G.compute_sinkhorn_loss(real[perm].contiguous(), fk2, cases.SC, 1.0, Lc, f["h_fake"][perm].contiguous(),
                                    f["m_real"][perm].contiguous(), f["h_real"][perm].contiguous(),
                                    f["m_fake"][perm].contiguous(), **kw)
    (g2,) = torch.autograd.grad(loss2, fk2)
    assert rel(loss2, loss) < 2e-5
    scale = float(g.abs().max())
    # two fp32 evaluations of the same near-regime problem in different summation orders: a few x the fp32 gap
    assert float((g2 - g[perm]).abs().max()) < 2e-4 * scale


@pytest.mark.parametrize("shape,Lc", FULL_SIZE[1:])
def test_full_size_configs_2_3_and_4_against_the_oracle(G, shape, Lc):
    """BASELINE configs[2] (B = 128, 64x64x3, T = 30, L = 100: the single-panel 256-row Gram tile `Q256_HALF` at its real
    K = 368 640, 240 partial tiles), configs[3] (B = 256, 64x64x3, T = 30, L = 200, + the RBF-MMD extension) and configs[4]
    (B = 512, 128x128x3, T = 48, L = 300) at FULL size on one GPU, oracle-checked where the oracle finishes in seconds:
      * cost matrices: 48 sampled entries + 16 diagonal ones of each of C_xy, C_xx, C_yy against the fp64 oracle
        formula on the two rows involved (K = 368 640 / 2 359 296 terms each) -- the tiled Gram at its real K;
      * Sinkhorn at n = 256 / 512 with L = 200 / 300: the fp64 oracle loop run on the GPU's own cost matrices, cost
        and executed iteration count of each of the three problems, and the mixed divergence at 1e-4."""
    B, H, T, W, C = shape
    real, fake, f, gen = _full_size_inputs(shape)
    x, y = real.reshape(B, -1), fake.reshape(B, -1)
    C3 = G._Cost3.apply(x, y, f["h_fake"], f["h_real"], f["m_real"], f["m_fake"], cases.SC)
    C3n = C3.cpu().numpy().astype(np.float64)
    fn = {k: v.cpu().numpy().astype(np.float64) for k, v in f.items()}
    rng = np.random.default_rng(B)
    pairs = [(int(i), int(i)) for i in rng.integers(0, B, 16)] + [(int(i), int(j)) for i, j in rng.integers(0, B, (48, 2))]
    rows = {}
    def row(t, i):
        if (id(t), i) not in rows:
            rows[(id(t), i)] = t[i].cpu().numpy().astype(np.float64)
        return rows[(id(t), i)]
    # (rows, cols, h (rows), M (cols)) of the three problems: gan_utils.py:221-223
    probs = ((x, y, "h_fake", "m_real"), (x, x, "h_real", "m_real"), (y, y, "h_fake", "m_fake"))
    cmax = np.abs(C3n).max()
    for k, (a, b, hk, mk) in enumerate(probs):
        for i, j in pairs:
            d = row(a, i) - row(b, j)
            dM = fn[mk][j, 1:, :] - fn[mk][j, :-1, :]
            ref = cases.SC * float(d @ d) + cases.SC * float((fn[hk][i, :-1, :] * dM).sum())
            assert abs(C3n[k, i, j] - ref) <= 1e-5 * cmax, (k, i, j, C3n[k, i, j], ref)
            if i == j and k == 0:        # the near-regime diagonal of C_xy is ~1e-2 of max|C|: check it RELATIVELY too
                assert abs(C3n[k, i, j] - ref) <= 2e-5 * abs(ref), (i, C3n[k, i, j], ref)
    fk = fake.clone().requires_grad_(True)
    kw = dict(honor_eps_l=True) if Lc != 100 else {}
    loss = G.compute_sinkhorn_loss(real, fk, cases.SC, 1.0, Lc, f["h_fake"], f["m_real"], f["h_real"], f["m_fake"], **kw)
    nits = G.last_info["compute_sinkhorn_loss"].tolist()[:3]
    w, n_ref = zip(*(o.sinkhorn_from_cost(C3n[k], 1.0, Lc, dtype=np.float64)[:2] for k in range(3)))
    ref_loss = 2.0 * w[0] - w[1] - w[2]
    assert rel(loss, ref_loss) < 1e-4, (float(loss), ref_loss)
    for k in range(3):       # stop decisions: identical, or within the fp32-vs-fp64 slack on a slowly decaying tail
        assert abs(nits[k] - n_ref[k]) <= max(0, (n_ref[k] - 100) // 10), (nits, n_ref)
    if B == 256:             # configs[3] names the kernel-MMD too (extension, sklearn semantics: kccotgan_amd/mmd.py)
        from kccotgan_amd.mmd import rbf_mmd2, rbf_kernels
        K3, m = rbf_kernels(real, fake)
        gamma = 1.0 / x.shape[1]
        D3 = (C3n - np.stack([o.causal_term(fn[hk], fn[mk], cases.SC, np.float64) for _, _, hk, mk in probs])) / cases.SC
        ref_m = np.exp(-gamma * D3[1]).mean() + np.exp(-gamma * D3[2]).mean() - 2 * np.exp(-gamma * D3[0]).mean()
        assert abs(float(m) - ref_m) <= 1e-4 * abs(ref_m) + 1e-7
        assert abs(float(rbf_mmd2(real, real.clone()))) <= 1e-6


@pytest.mark.parametrize("B", [256, 128, 192, 384])
def test_large_batch_video_gradient_forms_agree(G, L, B):
    """The one-launch video gradient (B = 256: apply_coeffs_x3_m256, 256-row tiles; its 128-column form takes over at
    K >= 655 360 and is checked against the fp64 formula at configs[4] in tests/test_gpu_fullsize_grads.py; B = 128 / 384:
    apply_coeffs_x3_rows with 128-row tiles, B = 192: 64-row tiles) against the 64-row block form (option "apply_m256" = 0):
    same exact split, same products, another tiling of the stack -- equal to fp32 summation order (1e-6 of max|grad|)."""
    H, T, W, C = 8, 10, 8, 5                  # K = 3200: 50 column tiles of 64
    gen = torch.Generator(device=DEV).manual_seed(4242 + B)
    real = torch.rand((B, H, T, W, C), device=DEV, generator=gen)
    fake = (real + 0.05 * torch.randn(real.shape, device=DEV, generator=gen)).clamp_(0, 1)
    f = {k: torch.rand((B, T, 8), device=DEV, generator=gen) for k in ("h_fake", "m_real", "h_real", "m_fake")}
    grads = {}
    for mode in ("tile", "block"):
        with L.options(apply_m256=1 if mode == "tile" else 0):
            fk = fake.clone().requires_grad_(True)
            loss = G.compute_sinkhorn_loss(real, fk, cases.SC, 0.8, 100, f["h_fake"], f["m_real"], f["h_real"], f["m_fake"])
            (grads[mode],) = torch.autograd.grad(loss, fk)
    scale = float(grads["block"].abs().max())
    assert float((grads["tile"] - grads["block"]).abs().max()) <= 1e-6 * scale


@pytest.mark.parametrize("B,K", [(256, 131072), (256, 131072 + 36), (512, 65536 + 4)])
def test_video_gradient_on_256x256_tiles_is_bit_identical(L, B, K):
    """Round 4, csrc/cost_bwd_q256.hip (option "apply_q256", default 1 from 512 tiles on): 256 x 256 output tiles, the
    coefficient panel staged through LDS, every wave staging and consuming -- against the 256 x 64 / 128 tile kernels it
    replaces (option 0).  Same products in the same order per output element: the video gradient agrees BIT FOR BIT, also
    with a ragged last column tile (K % 256 != 0) and two row tiles (B = 512)."""
    from kccotgan_amd._lib import lib, ptr, workspace, check
    gen = torch.Generator(device=DEV).manual_seed(B + K)
    real = torch.rand((B, K), device=DEV, generator=gen)
    fake = torch.rand((B, K), device=DEV, generator=gen)
    g3 = torch.randn((3, B, B), device=DEV, generator=gen) * 1e-3
    ws, wsb = workspace(lib.kccot_pairwise_cost3_bwd_workspace_bytes(B, K), real)
    out = {}
    for mode in (0, 1):
        with L.options(apply_q256=mode):
            d = torch.full((B, K), float("nan"), device=DEV)
            check(lib.kccot_pairwise_cost3_bwd_f32(ptr(g3), ptr(real), ptr(fake), B, K, cases.SC, None, None, None, None, 1, 1,
                                                   ptr(d), None, None, None, None, ws, wsb, None), "bwd")
            torch.cuda.synchronize()
            out[mode] = d
    assert bool(torch.isfinite(out[1]).all()) and torch.equal(out[0], out[1])
    # and against the fp64 formula on a few rows (the kernel the option replaces is itself oracle-checked at full size)
    rows = [0, B // 2 - 1, B - 1]
    W = torch.zeros((B, 2 * B), dtype=torch.float64, device=DEV)
    gxy, gyy = g3[0].double(), g3[2].double()
    W[:, :B] = -2 * cases.SC * gxy.t()
    W[:, B:] = -2 * cases.SC * (gyy + gyy.t())
    dsum = gxy.sum(0) + gyy.sum(1) + gyy.sum(0)
    W[torch.arange(B), B + torch.arange(B)] += 2 * cases.SC * dsum
    cols = torch.arange(0, K, 257, device=DEV)
    Z = torch.cat([real[:, cols], fake[:, cols]], 0).double()
    ref = W[rows] @ Z
    got = out[1][rows][:, cols].double()
    assert float((got - ref).abs().max()) <= 2.5e-5 * float(ref.abs().max())


@pytest.mark.parametrize("B,rows", [(128, 32), (128, 64), (256, 32), (256, 64), (384, 128), (256, 96)])
def test_video_gradient_of_a_row_block_in_one_launch(G, L, B, rows):
    """kccot_pairwise_cost3_bwd_rows_f32 (the batch-sharded caller's gradient of ITS samples from the replicated dC):
    row blocks of 32 / 64 / 128 rows run apply_coeffs_x3_rows in ONE launch over the whole stack (96 rows: no tile height
    divides them, the block form serves) -- every block of every rank against the same rows of the full-batch gradient
    and against the block form."""
    from kccotgan_amd.dist import HipOps as H
    K = 2560 + 36
    gen = torch.Generator(device=DEV).manual_seed(777 + B + rows)
    real = torch.rand((B, K), device=DEV, generator=gen)
    fake = (real + 0.05 * torch.randn(real.shape, device=DEV, generator=gen)).clamp_(0, 1)
    f = [torch.rand((B, 6, 8), device=DEV, generator=gen) for _ in range(4)]      # h_fake, h_real, m_real, m_fake
    g3 = torch.randn((3, B, B), device=DEV, generator=gen)
    full = H.cost3_bwd_rows(g3, real, fake, f[0], f[1], f[2], f[3], cases.SC, 0, B)
    scale = float(full[0].abs().max())
    for r0 in range(0, B - rows + 1, rows):
        got = H.cost3_bwd_rows(g3, real, fake, f[0], f[1], f[2], f[3], cases.SC, r0, rows)
        with L.options(apply_m256=0):
            blk = H.cost3_bwd_rows(g3, real, fake, f[0], f[1], f[2], f[3], cases.SC, r0, rows)
        assert float((got[0] - full[0][r0:r0 + rows]).abs().max()) <= 1e-6 * scale, (r0, "vs full batch")
        assert float((got[0] - blk[0]).abs().max()) <= 1e-6 * scale, (r0, "vs block form")
        for a, b in zip(got[1:], blk[1:]):
            assert torch.equal(a, b)


def test_video_gradient_bf16_split_matches_f32_mfma(G, L):
    """dfake = W [X;Y] on the bf16 matrix pipe (exact three-way split of W and of the videos, the default) against
    the f32-input MFMA kernel (option "apply_f32" = 1) at configs[1] full size and at a blocked batch (B = 128)."""
    for shape in ((64, 64, 30, 64, 1), (128, 16, 10, 16, 3)):
        B, H, T, W, C = shape
        gen = torch.Generator(device=DEV).manual_seed(17 + B)
        real = torch.rand(shape, device=DEV, generator=gen)
        fake = (real + 0.05 * torch.randn(shape, device=DEV, generator=gen)).clamp_(0, 1)
        f = {k: torch.rand((B, T, 8), device=DEV, generator=gen) for k in ("h_fake", "m_real", "h_real", "m_fake")}
        grads = {}
        for mode in ("x3", "f32"):
            with L.options(apply_f32=1 if mode == "f32" else 0):
                fk = fake.clone().requires_grad_(True)
                loss = G.compute_sinkhorn_loss(real, fk, cases.SC, 0.8, 100, f["h_fake"], f["m_real"], f["h_real"], f["m_fake"])
                (grads[mode],) = torch.autograd.grad(loss, fk)
        scale = float(grads["f32"].abs().max())
        assert float((grads["x3"] - grads["f32"]).abs().max()) < 2e-6 * scale
