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
def test_hip_smoothing_legacy_path_matches_reference_fixtures(name):
    """The LDS-plane / per-axis fallback kernels (option "smooth_stream" = 0) against the same fixtures."""
    import torch
    from kccotgan_amd import _lib
    from kccotgan_amd.data_utils import KernelSmoothing
    g, v, tk, sk, sigma = load(name)
    ks = KernelSmoothing(temporal_kernel_size=tk, spatial_kernel_size=sk)
    x = torch.from_numpy(v).cuda()
    with _lib.options(smooth_stream=0):
        np.testing.assert_allclose(ks.temporal_convolution(x, sigma).cpu().numpy(), g["temporal"], rtol=0, atol=ATOL_T)
        np.testing.assert_allclose(ks.gaussian_convolution3D(x, sigma).cpu().numpy(), g["conv3d"], rtol=0, atol=ATOL_3D)


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["c1_r3_s5", "c3_r3_s1p3", "cfg1frame"])
def test_two_phase_protocol_of_the_sharded_smoothing_hits_exactly_one(name):
    """The batch-sharded caller runs the forward in two phases -- KCCOT_SMOOTH_NO_DIVIDE (raw sums + local maximum),
    all-reduce(MAX), KCCOT_SMOOTH_EXTERNAL_MAX (divide) -- and the adjoint finds the arg-max through `out == 1`
    (the reference's reduce_max gradient, data_utils.py:520,573,581).  Both phases must therefore evaluate the sums
    with the same kernels: the maximum of the result is EXACTLY 1 and the result equals the one-call form bit for
    bit.  (Round 2 regression: phase 1 ran the per-axis chain, phase 2 the streamed walks, the maximum came out one
    ulp below 1 and the sharded adjoint silently lost its arg-max term.)"""
    import torch
    from kccotgan_amd import _lib
    from kccotgan_amd._lib import lib, ptr, check, workspace, stream_of
    g, v, tk, sk, sigma = load(name)
    x = torch.from_numpy(v).cuda()
    B, H, T, W, C = x.shape
    ws, wsb = workspace(lib.kccot_smooth_workspace_bytes(B, H, T, W, C), x)
    for axes, radius, key in ((_lib.SMOOTH_T, tk // 2, "temporal"), (_lib.SMOOTH_T | _lib.SMOOTH_H | _lib.SMOOTH_W, sk // 2, "conv3d")):
        one = torch.empty_like(x); two = torch.empty_like(x)
        m1 = torch.empty(1, device=x.device); m2 = torch.empty(1, device=x.device)
        check(lib.kccot_smooth_fwd_f32(ptr(x), B, H, T, W, C, sigma, radius, axes, ptr(one), ptr(m1), ws, wsb, stream_of(x)), "one call")
        check(lib.kccot_smooth_fwd_f32(ptr(x), B, H, T, W, C, sigma, radius, axes | _lib.SMOOTH_NO_DIVIDE, ptr(two), ptr(m2), ws, wsb,
                                       stream_of(x)), "phase 1")
        raw_max = float(two.max())
        assert raw_max == float(m2), (raw_max, float(m2))          # phase 1 leaves the raw sums and their maximum
        check(lib.kccot_smooth_fwd_f32(ptr(x), B, H, T, W, C, sigma, radius, axes | _lib.SMOOTH_EXTERNAL_MAX, ptr(two), ptr(m2), ws, wsb,
                                       stream_of(x)), "phase 2")
        torch.cuda.synchronize()
        assert float(m1) == float(m2) and float(two.max()) == 1.0
        assert torch.equal(one, two)
        np.testing.assert_allclose(two.cpu().numpy(), g[key], rtol=0, atol=ATOL_3D)


@pytest.mark.gpu
def test_hip_smoothing_random_shapes_against_the_pinned_oracle():
    """Ragged and odd shapes the fixtures do not hold (T barely above the radius, W*C not a multiple of 4, C = 2 / 4,
    one-sample batches, long axes that leave the streamed kernels' range): every kernel family the dispatcher can pick
    against the oracle that tests/test_smoothing_golden.py pins to the reference, forward and adjoint."""
    import torch
    from kccotgan_amd.data_utils import KernelSmoothing
    from oracle import smoothing_np as sm
    from oracle import smoothing_torch as st
    rng = np.random.default_rng(77)
    shapes = [(1, 5, 4, 6, 1), (2, 7, 5, 9, 2), (3, 9, 11, 5, 3), (1, 6, 6, 7, 4), (2, 70, 8, 6, 1), (2, 8, 66, 8, 1),
              (1, 12, 7, 130, 1), (2, 16, 9, 16, 3), (4, 8, 8, 8, 1), (1, 33, 5, 17, 2),
              # round 2: shapes that only the any-length / any-channel kernels take
              (2, 10, 9, 7, 3), (1, 8, 70, 5, 2), (2, 130, 6, 6, 1), (1, 9, 9, 300, 1), (1, 5, 5, 5, 7), (3, 6, 48, 12, 3)]
    for shape in shapes:
        for ksize, sigma in ((6, 5.0), (8, 1.3)):
            r = ksize // 2
            if min(shape[1], shape[2], shape[3]) <= r:
                continue                                  # REFLECT needs length > radius (tf.pad raises otherwise)
            v = rng.random(shape, dtype=np.float32)
            ks = KernelSmoothing(ksize, ksize)
            x = torch.from_numpy(v).cuda().requires_grad_(True)
            for fn, ref in ((ks.temporal_convolution, sm.temporal_convolution(v, sigma, r)),
                            (ks.gaussian_convolution3D, sm.gaussian_convolution3D_separable(v, sigma, r))):
                out = fn(x, sigma)
                np.testing.assert_allclose(out.detach().cpu().numpy(), ref, rtol=0, atol=ATOL_3D, err_msg=str((shape, ksize)))
            # adjoint of the 3-D call against torch autograd of the oracle (max-normalisation included)
            g = torch.from_numpy(rng.standard_normal(shape).astype(np.float32)).cuda()
            out = ks.gaussian_convolution3D(x, sigma)
            (gx,) = torch.autograd.grad(out, x, g)
            xd = torch.from_numpy(v).double().requires_grad_(True)
            od = st.smooth(xd, sigma, r, (2, 1, 3))
            (gd,) = torch.autograd.grad(od, xd, g.cpu().double())
            np.testing.assert_allclose(gx.cpu().numpy(), gd.numpy(), rtol=0, atol=2e-4 * float(gd.abs().max()),
                                       err_msg=str((shape, ksize)))


@pytest.mark.gpu
@pytest.mark.parametrize("shape", [(64, 64, 30, 64, 1), (3, 9, 7, 8, 1), (2, 12, 32, 16, 1), (5, 7, 11, 32, 1), (1, 10, 9, 128, 1),
                                   (2, 8, 6, 256, 1), (37, 5, 8, 64, 1)])
def test_fused_t_and_w_stage_equals_the_two_kernels(shape):
    """3-D smoothing, C = 1, W/4 a power of two: the T walk applies the W stencil to every T-smoothed piece before storing it
    (neighbour pieces = neighbouring lanes' registers, DPP wave shifts) instead of a T kernel and a W kernel with a round
    trip of the tensor in between (option "smooth_fused_tw" = 0).  Same fma order: bit-identical outputs, radius 3 and 4."""
    import torch
    from kccotgan_amd import _lib
    from kccotgan_amd._lib import lib, ptr, check, stream_of
    from oracle import smoothing_np as sm
    rng = np.random.default_rng(sum(shape))
    v = rng.random(shape, dtype=np.float32)
    x = torch.from_numpy(v).cuda()
    B, H, T, W, C = shape
    wsb = int(lib.kccot_smooth_workspace_bytes(B, H, T, W, C))
    wst = torch.empty(wsb, dtype=torch.uint8, device=x.device)
    axes = _lib.SMOOTH_T | _lib.SMOOTH_H | _lib.SMOOTH_W
    for radius in (3, 4):
        if min(H, T, W) <= radius:
            continue
        outs = []
        for no_tw in (False, True):
            with _lib.options(smooth_fused_tw=0 if no_tw else 1):
                o = torch.empty_like(x); m = torch.empty(1, device=x.device)
                check(lib.kccot_smooth_fwd_f32(ptr(x), B, H, T, W, C, 1.7, radius, axes, ptr(o), ptr(m), wst.data_ptr(), wsb, stream_of(x)),
                      "smooth")
                torch.cuda.synchronize()
            outs.append(o)
        assert torch.equal(outs[0], outs[1]), (shape, radius)
        np.testing.assert_allclose(outs[0].cpu().numpy(), sm.gaussian_convolution3D_separable(v, 1.7, radius), rtol=0, atol=ATOL_3D)


@pytest.mark.gpu
@pytest.mark.parametrize("shape", [(4, 16, 12, 16, 1), (2, 9, 30, 64, 1), (3, 20, 7, 8, 1), (2, 64, 30, 64, 1)])
def test_generic_kernels_equal_the_specialised_ones(shape):
    """Option "smooth_generic" = 1 routes every stage through the any-length / any-channel kernels (smooth_roll, smooth_wrow)
    that serve the shapes the specialised kernels cannot take (C > 1 on the W axis, axes longer than 64).  On shapes both
    families accept: same loads, same fma order -> the forward is bit-identical; the adjoint builds its border weights in
    a different order (table vs in-line folds) and agrees to rounding."""
    import torch
    from kccotgan_amd import _lib
    from kccotgan_amd.data_utils import KernelSmoothing
    rng = np.random.default_rng(sum(shape) + 5)
    v = rng.random(shape, dtype=np.float32)
    gr = torch.from_numpy(rng.standard_normal(shape).astype(np.float32)).cuda()
    for ksize in (6, 8):
        if min(shape[1:4]) <= ksize // 2:
            continue
        ks = KernelSmoothing(ksize, ksize)
        res = {}
        for mode in ("special", "generic"):
            with _lib.options(smooth_generic=1 if mode == "generic" else 0):
                x = torch.from_numpy(v).cuda().requires_grad_(True)
                t = ks.temporal_convolution(x, 2.1)
                (gt,) = torch.autograd.grad(t, x, gr)
                c = ks.gaussian_convolution3D(x, 2.1)
                (gc,) = torch.autograd.grad(c, x, gr)
            res[mode] = (t.detach(), c.detach(), gt, gc)
        assert torch.equal(res["special"][0], res["generic"][0]), (shape, ksize, "temporal")
        assert torch.equal(res["special"][1], res["generic"][1]), (shape, ksize, "3-D")
        for i in (2, 3):
            a, b = res["special"][i], res["generic"][i]
            assert float((a - b).abs().max()) <= 2e-6 * float(a.abs().max()), (shape, ksize, i)


@pytest.mark.gpu
@pytest.mark.parametrize("shape", [(3, 8, 12, 16, 3), (2, 9, 30, 64, 3), (2, 5, 7, 10, 2), (1, 6, 9, 33, 1), (2, 7, 48, 128, 3),
                                   (2, 4, 5, 6, 5), (2, 64, 30, 64, 1), (2, 5, 40, 96, 3), (3, 6, 9, 12, 4), (2, 6, 8, 20, 1)])
def test_fused_t_w_plane_kernel_equals_the_separate_stages(shape):
    """3-D smoothing with the (b, h) plane staged in LDS (smooth_tw_plane: T and W stencils in one pass, any channel
    count) against the separate per-axis kernels (option "smooth_fused_tw" = 0).  Forward: same fma order, bit-identical.
    The adjoint (always the separate stages) against the fp64 autograd of the pinned oracle."""
    import torch
    from kccotgan_amd import _lib
    from kccotgan_amd.data_utils import KernelSmoothing
    from oracle import smoothing_torch as st
    rng = np.random.default_rng(sum(shape) + 11)
    v = rng.random(shape, dtype=np.float32)
    gr = torch.from_numpy(rng.standard_normal(shape).astype(np.float32)).cuda()
    for ksize in (6, 8):
        r = ksize // 2
        if min(shape[1:4]) <= r:
            continue
        ks = KernelSmoothing(ksize, ksize)
        res = {}
        for mode in ("plane", "separate"):
            with _lib.options(smooth_fused_tw=0 if mode == "separate" else 1):
                x = torch.from_numpy(v).cuda().requires_grad_(True)
                c = ks.gaussian_convolution3D(x, 1.9)
                (gc,) = torch.autograd.grad(c, x, gr)
            res[mode] = (c.detach(), gc)
        assert torch.equal(res["plane"][0], res["separate"][0]), (shape, ksize)
        a, b = res["plane"][1], res["separate"][1]
        assert float((a - b).abs().max()) <= 2e-6 * float(a.abs().max()), (shape, ksize)
        xd = torch.from_numpy(v).double().requires_grad_(True)
        (gd,) = torch.autograd.grad(st.smooth(xd, 1.9, r, (2, 1, 3)), xd, gr.cpu().double())
        np.testing.assert_allclose(a.cpu().numpy(), gd.numpy(), rtol=0, atol=2e-4 * float(gd.abs().max()), err_msg=str((shape, ksize)))


@pytest.mark.gpu
@pytest.mark.parametrize("shape", [(256, 64, 30, 64, 3), (64, 128, 48, 128, 3)])
def test_full_size_smoothing_at_the_large_baseline_shapes(shape):
    """BASELINE configs[3] (whole batch on one GPU) and configs[4] (one rank's shard of 8) through KernelSmoothing, C = 3:
    the any-channel / any-length kernels at full size.  Size-independent properties: the maximum of the output is exactly
    1 and nothing is negative; the stencil does not couple samples, so a 2-sample slab smoothed on its own and divided by
    the full tensor's maximum (KCCOT_SMOOTH_EXTERNAL_MAX) reproduces those samples of the full result bit for bit; and that
    slab agrees with the numpy oracle (pinned to the reference by the fixtures above).  Forward and adjoint stay finite."""
    import torch
    from kccotgan_amd import _lib
    from kccotgan_amd._lib import lib, ptr, check, stream_of
    from oracle import smoothing_np as sm
    B, H, T, W, C = shape
    g = torch.Generator(device="cuda").manual_seed(5)
    x = torch.rand(shape, device="cuda", generator=g)
    wsb = int(lib.kccot_smooth_workspace_bytes(B, H, T, W, C))
    ws = torch.empty(wsb, dtype=torch.uint8, device="cuda")
    slab = x[3:5].contiguous()
    wsb2 = int(lib.kccot_smooth_workspace_bytes(2, H, T, W, C))
    for axes, oracle in ((_lib.SMOOTH_T, sm.temporal_convolution), (_lib.SMOOTH_T | _lib.SMOOTH_H | _lib.SMOOTH_W, sm.gaussian_convolution3D_separable)):
        o = torch.empty_like(x); m = torch.empty(1, device="cuda")
        check(lib.kccot_smooth_fwd_f32(ptr(x), B, H, T, W, C, 5.0, 3, axes, ptr(o), ptr(m), ws.data_ptr(), wsb, stream_of(x)), "fwd")
        torch.cuda.synchronize()
        assert float(o.max()) == 1.0 and float(o.min()) >= 0.0 and np.isfinite(float(m))
        o2 = torch.empty_like(slab); m2 = m.clone()
        check(lib.kccot_smooth_fwd_f32(ptr(slab), 2, H, T, W, C, 5.0, 3, axes | _lib.SMOOTH_EXTERNAL_MAX, ptr(o2), ptr(m2), ws.data_ptr(),
                                       wsb2, stream_of(x)), "slab")
        torch.cuda.synchronize()
        assert torch.equal(o2, o[3:5])
        # oracle on the slab: raw sums (its own maximum divided back out), then the full tensor's maximum
        ref = oracle(slab.cpu().numpy(), 5.0, 3).astype(np.float64)
        raw = o2.double().cpu().numpy() * float(m)
        ref_raw = ref * (raw.max() / ref.max())
        np.testing.assert_allclose(raw, ref_raw, rtol=0, atol=4e-6 * raw.max())
        gr = torch.randn(shape, device="cuda", generator=g)
        d = torch.empty_like(x)
        check(lib.kccot_smooth_bwd_f32(ptr(gr), ptr(o), ptr(m), B, H, T, W, C, 5.0, 3, axes, ptr(d), ws.data_ptr(), wsb, stream_of(x)), "bwd")
        torch.cuda.synchronize()
        assert bool(torch.isfinite(d).all())
        del o, d, gr


def _bwd(g, out, mx, radius, axes, sigma=2.1):
    """kccot_smooth_bwd_f32 on a given (gout, forward output, maximum): the forward output is an INPUT of the backward, so
    a test can plant any number of arg-max elements in it."""
    import torch
    from kccotgan_amd import _lib
    from kccotgan_amd._lib import lib, check, ptr, stream_of, workspace
    B, H, T, W, C = out.shape
    din = torch.empty_like(out)
    ws, wsb = workspace(lib.kccot_smooth_workspace_bytes(B, H, T, W, C), out)
    check(lib.kccot_smooth_bwd_f32(ptr(g), ptr(out), ptr(mx), B, H, T, W, C, sigma, radius, axes, ptr(din), ws, wsb,
                                   stream_of(out)), "smooth_bwd")
    torch.cuda.synchronize()
    return din


@pytest.mark.gpu
@pytest.mark.parametrize("shape", [(2, 8, 9, 10, 2), (2, 40, 12, 8, 1), (1, 12, 33, 12, 1), (2, 9, 70, 12, 3), (4, 64, 30, 64, 1),
                                   (2, 6, 8, 20, 1), (3, 16, 7, 12, 3), (64, 64, 30, 64, 1)])
@pytest.mark.parametrize("ties", ["none", "one", "corners", "cluster", "many"])
def test_folded_backward_equals_the_two_pass_backward(shape, ties):
    """Option "smooth_bwd_fold": the first adjoint stage gathers sum(g * out) and the arg-max positions itself and
    maxnorm_bwd_fixup subtracts corr * A^T [out == 1] sparsely afterwards (= 2: at every size; the default 1 does so
    for large tensors only, where the two tensor reads saved outweigh the extra launches), against the backward that computes
    the two sums in a pass of its own first and folds the correction into the first stage's loads (= 0, the round-2 path,
    pinned to fp64 autograd by test_smoothing_gradient_matches_autograd).  Planted arg-max sets: none; one; the tensor's
    corners and border positions (REFLECT folds several taps onto one neighbour); a cluster of adjacent elements inside
    one workgroup's lines (> TIE_PER_WG in one record -> dense fallback on the device) and a few spread ones (overlapping
    neighbourhoods, sparse path); > 32 spread ones (dense fallback).  A^T is linear: both orders agree to rounding."""
    import torch
    from kccotgan_amd import _lib
    rng = np.random.default_rng(sum(shape) + len(ties))
    n = int(np.prod(shape))
    out = rng.random(shape, dtype=np.float32) * 0.98
    flat = out.reshape(-1)
    if ties == "one":
        flat[rng.integers(n)] = 1.0
    elif ties == "corners":
        B, H, T, W, C = shape
        for idx in ((0, 0, 0, 0, 0), (B - 1, H - 1, T - 1, W - 1, C - 1), (0, 1, T - 2, 0, 0), (B - 1, 0, 1, W - 2, C - 1),
                    (0, H - 1, 0, W - 1, 0), (0, 2, 2, 2, 0), (0, 2, 3, 2, 0)):
            out[idx] = 1.0
    elif ties == "cluster":
        start = int(rng.integers(n - 8))
        flat[start:start + 6] = 1.0
    elif ties == "many":
        flat[rng.choice(n, size=45, replace=False)] = 1.0
    g = torch.from_numpy(rng.standard_normal(shape).astype(np.float32)).cuda()
    o = torch.from_numpy(out).cuda()
    mx = torch.tensor([1.7], device="cuda")
    for radius in (3, 4):
        if min(shape[1:4]) <= radius:
            continue
        for axes in (_lib.SMOOTH_T, _lib.SMOOTH_T | _lib.SMOOTH_H | _lib.SMOOTH_W):
            res = {}
            for fold in (2, 1, 0):
                for generic in (0, 1):
                    with _lib.options(smooth_bwd_fold=fold, smooth_generic=generic):
                        res[fold, generic] = _bwd(g, o, mx, radius, axes)
            with _lib.options(smooth_stream=0):
                legacy = _bwd(g, o, mx, radius, axes)
            scale = float(legacy.abs().max())
            # the two-pass kernels sum g * out in fp32 partials (the folded stage in fp64): at 7.9 M elements that alone
            # moves the correction -- the largest entry of the gradient -- by 3e-6 of itself
            tol = (3e-6 if n < 1 << 20 else 1e-5) * scale
            for key, val in res.items():
                assert float((val - legacy).abs().max()) <= tol, (shape, ties, radius, axes, key)
            # the sparse fix-up is deterministic: a second run gives the same bits
            with _lib.options(smooth_bwd_fold=2):
                assert torch.equal(_bwd(g, o, mx, radius, axes), res[2, 0])


@pytest.mark.gpu
@pytest.mark.parametrize("shape", [(3, 40, 30, 64, 1), (2, 24, 30, 64, 3), (1, 20, 48, 128, 3), (70, 9, 8, 8, 1), (2, 11, 14, 24, 3),
                                   (2, 16, 30, 64, 3)])
@pytest.mark.parametrize("ties", ["none", "one", "corners", "many"])
def test_fused_3d_adjoint_equals_the_chain(shape, ties):
    """Round 4: the backward of gaussian_convolution3D as ONE pass (csrc/smooth.hip, smooth_fused3_adj: x = gout / max into a
    register window while the workgroup walks along H, H^T with the folded border weights, W^T and T^T through LDS with the
    pad folded back, the normalisation's two sums gathered on the way) against the chain of per-axis adjoint stages
    (option "smooth_fused3" = 0).  Same products; at the folded border positions the sums are associated differently
    (the chain adds the folded weights first): agreement to 2e-6 of max|din|, and to the per-element legacy kernels
    (smooth_stream = 0) within the folded backward's own tolerance.  Shapes: H cut into segments, two column tiles, one tile
    with both borders, three items per thread, T = 2 R + 2.  Arg-max sets as in the test above (corners: REFLECT folds several
    taps onto one neighbour; many: dense fallback).  That the fused kernel ran is read off the workspace's tensor-sized
    buffer, which only the chain writes."""
    import torch
    from kccotgan_amd import _lib
    from kccotgan_amd._lib import lib, check, ptr
    B, H, T, W, C = shape
    rng = np.random.default_rng(sum(shape) + len(ties))
    n = int(np.prod(shape))
    out = rng.random(shape, dtype=np.float32) * 0.98
    flat = out.reshape(-1)
    if ties == "one":
        flat[rng.integers(n)] = 1.0
    elif ties == "corners":
        for idx in ((0, 0, 0, 0, 0), (B - 1, H - 1, T - 1, W - 1, C - 1), (0, 1, T - 2, 0, 0), (B - 1, 0, 1, W - 2, C - 1),
                    (0, H - 1, 0, W - 1, 0), (0, 2, 2, 2, 0), (0, 2, 3, 2, 0)):
            out[idx] = 1.0
    elif ties == "many":
        flat[rng.choice(n, size=45, replace=False)] = 1.0
    g = torch.from_numpy(rng.standard_normal(shape).astype(np.float32)).cuda()
    o = torch.from_numpy(out).cuda()
    mx = torch.tensor([1.7], device="cuda")
    axes = _lib.SMOOTH_T | _lib.SMOOTH_H | _lib.SMOOTH_W
    wsb = int(lib.kccot_smooth_workspace_bytes(B, H, T, W, C))

    def bwd():
        din = torch.full(shape, float("nan"), device="cuda")
        buf = torch.empty(wsb // 4 + 64, device="cuda")
        buf[:n] = -7.0
        check(lib.kccot_smooth_bwd_f32(ptr(g), ptr(o), ptr(mx), B, H, T, W, C, 2.1, 3, axes, ptr(din), buf.data_ptr(), wsb, None), "bwd")
        torch.cuda.synchronize()
        return din, bool((buf[:n] != -7.0).any())

    with _lib.options(smooth_bwd_fold=2, smooth_fused3=2):
        fused, touched = bwd()
        again, _ = bwd()
    # (many: > 32 arg-max elements; corners where H is one segment: five of them in ONE workgroup's record, > TIE_PER_WG -- both
    # take the dense fallback behind the fix-up, by design: the same walk once more with the correction applied at the loads)
    assert not touched, "the fused adjoint did not run"
    assert torch.equal(fused, again)                    # deterministic
    with _lib.options(smooth_bwd_fold=2, smooth_fused3=0):
        chain, touched = bwd()
    assert touched
    with _lib.options(smooth_stream=0):
        legacy, _ = bwd()
    scale = float(legacy.abs().max())
    assert bool(torch.isfinite(fused).all())
    assert float((fused - chain).abs().max()) <= 2e-6 * scale, (shape, ties)
    assert float((fused - legacy).abs().max()) <= 3e-6 * scale, (shape, ties)


@pytest.mark.gpu
def test_fused_3d_walks_on_random_shapes():
    """The fused 3-D walks (forward: bit-identical to the chain; adjoint: to 2e-6 of max|din|) over forty random shapes: batch
    sizes that leave the chip part-filled (H cut into segments whose last one is short), H and T that are not multiples of
    anything, W that tiles into 1 / 2 / 4 column tiles, both channel counts, both radii forward (the adjoint is radius 3)."""
    import torch
    from kccotgan_amd import _lib
    from kccotgan_amd._lib import lib, check, ptr
    rng = np.random.default_rng(20261005)
    axes = _lib.SMOOTH_T | _lib.SMOOTH_H | _lib.SMOOTH_W
    done = 0
    while done < 40:
        C = int(rng.choice([1, 3]))
        W = int(rng.choice([8, 12, 16, 24, 32, 40, 64, 96, 128])) if C == 3 else int(rng.choice([8, 16, 32, 64, 128]))
        T = int(rng.integers(10, 50))
        H = int(rng.integers(8, 70))
        B = int(rng.choice([1, 2, 3, 5, 8, 17]))
        r = int(rng.choice([3, 3, 4]))
        if (W * C) % 4 or B * H * T * W * C > 6_000_000:
            continue
        done += 1
        shape = (B, H, T, W, C)
        n = B * H * T * W * C
        v = torch.from_numpy(rng.random(shape, dtype=np.float32)).cuda()
        g = torch.from_numpy(rng.standard_normal(shape).astype(np.float32)).cuda()
        wsb = int(lib.kccot_smooth_workspace_bytes(B, H, T, W, C))
        res = {}
        for fused in (2, 0):
            with _lib.options(smooth_fused3=fused, smooth_bwd_fold=2):
                ws = torch.empty(wsb // 4 + 64, device="cuda")
                out = torch.full(shape, float("nan"), device="cuda")
                mx = torch.zeros(1, device="cuda")
                check(lib.kccot_smooth_fwd_f32(ptr(v), B, H, T, W, C, 1.7, r, axes, ptr(out), ptr(mx), ws.data_ptr(), wsb, None), "fwd")
                din = torch.full(shape, float("nan"), device="cuda")
                check(lib.kccot_smooth_bwd_f32(ptr(g), ptr(out), ptr(mx), B, H, T, W, C, 1.7, r, axes, ptr(din), ws.data_ptr(), wsb, None), "bwd")
                torch.cuda.synchronize()
                res[fused] = (out, mx.clone(), din)
        assert torch.equal(res[2][0], res[0][0]) and torch.equal(res[2][1], res[0][1]), shape + (r,)
        scale = float(res[0][2].abs().max())
        assert bool(torch.isfinite(res[2][2]).all()), shape + (r,)
        assert float((res[2][2] - res[0][2]).abs().max()) <= 2e-6 * scale, shape + (r,)


@pytest.mark.gpu
@pytest.mark.parametrize("shape", [(3, 40, 30, 64, 1), (2, 24, 30, 64, 3), (1, 20, 48, 128, 3)])
@pytest.mark.parametrize("ties", ["none", "corners"])
def test_fused_3d_adjoint_with_handed_in_sums(shape, ties):
    """The batch-sharded backward (kccot_smooth_bwd_sharded_f32): phase 1 returns this rank's two sums, phase 2 takes the all-reduced
    ones (KCCOT_SMOOTH_EXTERNAL_STATS).  With "smooth_fused3" phase 2 is the fused adjoint in its XM form -- the normalisation's
    adjoint x = gout / max - corr [out == 1] applied at the loads, halo columns included, nothing gathered -- against the chain
    (WALK_ADJX first stage): 2e-6 of max|din|; and against the one-call backward of the same tensor (world size 1: the handed-in sums
    ARE the tensor's), whose correction goes the sparse way."""
    import torch
    from kccotgan_amd import _lib
    from kccotgan_amd._lib import lib, check, ptr
    B, H, T, W, C = shape
    rng = np.random.default_rng(sum(shape) + len(ties))
    n = int(np.prod(shape))
    out = rng.random(shape, dtype=np.float32) * 0.98
    if ties == "corners":
        for idx in ((0, 0, 0, 0, 0), (B - 1, H - 1, T - 1, W - 1, C - 1), (0, 1, T - 2, 0, 0), (0, H - 1, 0, W - 1, 0), (0, 2, 3, 2, 0)):
            out[idx] = 1.0
    g = torch.from_numpy(rng.standard_normal(shape).astype(np.float32)).cuda()
    o = torch.from_numpy(out).cuda()
    mx = torch.tensor([1.7], device="cuda")
    axes = _lib.SMOOTH_T | _lib.SMOOTH_H | _lib.SMOOTH_W
    wsb = int(lib.kccot_smooth_workspace_bytes(B, H, T, W, C))
    ws = torch.empty(wsb // 4 + 64, device="cuda")
    stats = torch.zeros(2, device="cuda")
    dummy = torch.empty(shape, device="cuda")
    check(lib.kccot_smooth_bwd_sharded_f32(ptr(g), ptr(o), ptr(mx), ptr(stats), B, H, T, W, C, 2.1, 3, axes | _lib.SMOOTH_STATS_ONLY,
                                           ptr(dummy), ws.data_ptr(), wsb, None), "stats")
    torch.cuda.synchronize()
    assert int(stats[1]) == (5 if ties == "corners" else 0)
    res = {}
    for fused in (2, 0):
        with _lib.options(smooth_fused3=fused):
            din = torch.full(shape, float("nan"), device="cuda")
            ws[:n] = -7.0
            check(lib.kccot_smooth_bwd_sharded_f32(ptr(g), ptr(o), ptr(mx), ptr(stats), B, H, T, W, C, 2.1, 3,
                                                   axes | _lib.SMOOTH_EXTERNAL_STATS, ptr(din), ws.data_ptr(), wsb, None), "bwd")
            torch.cuda.synchronize()
            assert bool((ws[:n] != -7.0).any()) == (fused == 0)      # the chain writes the intermediate buffer, the fused pass does not
            res[fused] = din
    with _lib.options(smooth_fused3=0, smooth_bwd_fold=0):
        whole = torch.empty(shape, device="cuda")
        check(lib.kccot_smooth_bwd_f32(ptr(g), ptr(o), ptr(mx), B, H, T, W, C, 2.1, 3, axes, ptr(whole), ws.data_ptr(), wsb, None), "bwd")
        torch.cuda.synchronize()
    scale = float(whole.abs().max())
    assert float((res[2] - res[0]).abs().max()) <= 2e-6 * scale
    assert float((res[2] - whole).abs().max()) <= 3e-6 * scale
