"""KernelSmoothing pinned to the reference: tests/golden/smooth_*.npz hold what the reference's own
``data_utils.KernelSmoothing`` (data_utils.py:478-586, run verbatim by
tests/golden/make_golden_smoothing.py) returns.  CPU tier: the oracle restatement against those
fixtures.  GPU tier (-m gpu): the HIP kernels against the same fixtures, through the package's
``KernelSmoothing`` mirror -> ctypes -> ``kccot_smooth_fwd_f32``.

Tolerances (fp32 path, values in [0, 1] after the division by the maximum): the reference's own fp32
run sits 1.5e-7 (temporal) / 2.3e-6 (dense 343-tap conv3d) from its fp64 run (printed by the
generator); a separable evaluation has a different rounding pattern of the same size, so outputs
are compared at atol 4e-6 (temporal 1e-6) -- 2x the reference's own fp32 noise."""
import os

import numpy as np
import pytest

import smooth_cases

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
ATOL_T, ATOL_3D = 1e-6, 4e-6


def load(name):
    g = np.load(os.path.join(GOLD, "smooth_%s.npz" % name))
    shape, seed, tk, sk, sigma = smooth_cases.CASES[name]
    v = smooth_cases.gen_video(shape, seed)
    assert abs(np.sum(v, dtype=np.float64) - float(g["checksum"])) < 1e-9 * v.size, "RNG drift"
    return g, v, tk, sk, sigma


@pytest.mark.parametrize("name", list(smooth_cases.CASES))
def test_oracle_matches_reference_fixtures(name):
    from oracle import smoothing_np as sm
    g, v, tk, sk, sigma = load(name)
    rt, rs = tk // 2, sk // 2
    assert list(g["radii"]) == [rt, rs]                                # data_utils.py:480-481
    # a8: taps
    np.testing.assert_allclose(sm.gaussian_kernel1d(rt, sigma), g["taps1d_t"], rtol=3e-7, atol=0)
    np.testing.assert_allclose(sm.gaussian_kernel1d(rs, sigma), g["taps1d_s"], rtol=3e-7, atol=0)
    np.testing.assert_allclose(sm.gaussian_kernel3d(rs, sigma), g["taps3d"], rtol=1e-6, atol=0)
    np.testing.assert_allclose(sm.gaussian_kernel1d(rs, sigma, np.float64), g["taps1d_s_f64"], rtol=1e-14)
    # a9 / a10: fp32 oracle against the reference's fp32 run
    t32 = sm.temporal_convolution(v, sigma, rt)
    np.testing.assert_allclose(t32, g["temporal"], rtol=0, atol=ATOL_T)
    d32 = sm.gaussian_convolution3D(v, sigma, rs)                      # dense, as the reference
    s32 = sm.gaussian_convolution3D_separable(v, sigma, rs)            # the form the HIP kernels use
    np.testing.assert_allclose(d32, g["conv3d"], rtol=0, atol=ATOL_3D)
    np.testing.assert_allclose(s32, g["conv3d"], rtol=0, atol=ATOL_3D)
    assert np.argmax(t32) == np.argmax(g["temporal"]) and np.argmax(s32) == np.argmax(g["conv3d"])
    assert float(g["temporal"].max()) == 1.0 and float(g["conv3d"].max()) == 1.0
    # fp64 oracle against the reference's fp64 run: the algorithm itself, free of rounding
    t64 = sm.temporal_convolution(v, sigma, rt, np.float64)
    s64 = sm.gaussian_convolution3D_separable(v, sigma, rs, np.float64)
    if name in smooth_cases.BIG:
        assert abs(t64.sum() - float(g["temporal_f64"])) < 1e-9 * v.size
        assert abs(s64.sum() - float(g["conv3d_f64"])) < 1e-9 * v.size
    else:
        np.testing.assert_allclose(t64, g["temporal_f64"], rtol=0, atol=1e-13)
        np.testing.assert_allclose(s64, g["conv3d_f64"], rtol=0, atol=1e-13)
    # a12
    np.testing.assert_allclose([sm.annealing_sigma(5.0, s) for s in smooth_cases.ANNEAL_STEPS], g["annealing_sigma"],
                               rtol=1e-15)
    assert int(g["spatial_raises"]) == 1                               # a11: data_utils.py:537-538 raises for every input


def test_lr_schedule_matches_reference_warmup():
    """data_utils.py:589-621 (WarmUp, run verbatim) around the staircase decay of kernel_train.py:57-58."""
    from kccotgan_amd.kernel_train import warmup_exponential_decay
    g = np.load(os.path.join(GOLD, "lr_schedule.npz"))
    for tag, (lr, warmup, decay_steps, rate) in smooth_cases.LR_RUNS.items():
        ours = [warmup_exponential_decay(int(s), lr, warmup, decay_steps, rate) for s in g["steps"]]
        np.testing.assert_allclose(ours, g["lr_" + tag], rtol=2e-6, atol=0)


# ------------------------------------------------------------------------------------------------ GPU tier
@pytest.mark.gpu
@pytest.mark.parametrize("name", list(smooth_cases.CASES))
def test_hip_smoothing_matches_reference_fixtures(name):
    import torch
    from kccotgan_amd.data_utils import KernelSmoothing
    g, v, tk, sk, sigma = load(name)
    ks = KernelSmoothing(temporal_kernel_size=tk, spatial_kernel_size=sk)
    x = torch.from_numpy(v).cuda()
    t = ks.temporal_convolution(x, sigma).cpu().numpy()
    c = ks.gaussian_convolution3D(x, sigma).cpu().numpy()
    np.testing.assert_allclose(t, g["temporal"], rtol=0, atol=ATOL_T)
    np.testing.assert_allclose(c, g["conv3d"], rtol=0, atol=ATOL_3D)
    assert float(t.max()) == 1.0 and float(c.max()) == 1.0
    assert np.argmax(t) == np.argmax(g["temporal"]) and np.argmax(c) == np.argmax(g["conv3d"])
    # host-side tap helpers of the mirror (a8)
    np.testing.assert_allclose(ks.gaussian_kernel1d(ks.temporal_radius, sigma).numpy(), g["taps1d_t"], rtol=3e-7)
    np.testing.assert_allclose(ks.gaussian_kernel3d(ks.spatial_radius, sigma).numpy()[..., 0, 0], g["taps3d"], rtol=1e-6)
    assert ks.annealing_sigma(5.0, 250) == float(g["annealing_sigma"][2])


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["c3_r3_s5", "c1_r4_s5"])
def test_hip_smoothing_legacy_path_matches_reference_fixtures(name, monkeypatch):
    """The LDS-plane / per-axis fallback kernels (KCCOT_SMOOTH_NO_STREAM=1) against the same fixtures."""
    import torch
    from kccotgan_amd.data_utils import KernelSmoothing
    monkeypatch.setenv("KCCOT_SMOOTH_NO_STREAM", "1")
    g, v, tk, sk, sigma = load(name)
    ks = KernelSmoothing(temporal_kernel_size=tk, spatial_kernel_size=sk)
    x = torch.from_numpy(v).cuda()
    np.testing.assert_allclose(ks.temporal_convolution(x, sigma).cpu().numpy(), g["temporal"], rtol=0, atol=ATOL_T)
    np.testing.assert_allclose(ks.gaussian_convolution3D(x, sigma).cpu().numpy(), g["conv3d"], rtol=0, atol=ATOL_3D)
