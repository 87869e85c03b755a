#!/usr/bin/env python3
"""Generate tests/golden/smooth_*.npz and lr_schedule.npz by EXECUTING the reference's own
``data_utils.py`` (``KernelSmoothing`` :478-586, ``WarmUp`` :589-621).

Run in the build container only (needs /root/reference; the GPU box never sees it):

    python tests/golden/make_golden_smoothing.py [--ref /root/reference]

How: ``oracle/refshim`` (NumPy stand-in for the ``tf.*`` primitives, convolutions = torch CPU
``conv1d/2d/3d``) goes first on sys.path; ``cv2``, ``absl``, ``IPython`` and
``tensorflow_probability`` -- imported at the top of data_utils.py (:8,9,21,23) but never touched
by the smoothing class -- are empty placeholder modules in ``sys.modules`` for the duration of
the import.  ``data_utils`` is then imported from the reference directory *as it lies there*; no
reference source is copied.  Every stored array is produced by the reference's own code: its
tap formulas, its five-transposes-and-a-reshape layout handling (both the C > 1 and the C == 1
branch), its REFLECT padding, its dense (2r+1)^3 ``conv3d`` and its division by the global
maximum.  Each case runs in fp32 (the reference's arithmetic) and with ``tf.float32`` re-pointed
at float64 (``_f64`` suffix).

Stored per case: gaussian_kernel1d / gaussian_kernel3d taps, temporal_convolution and
gaussian_convolution3D outputs, and the recorded fact that spatial_convolution raises.  Inputs
are regenerated from the seed by ``smooth_cases.gen_video`` (checksum stored).
"""
import argparse
import os
import sys
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, HERE)
import smooth_cases  # noqa: E402


def load_reference(ref_dir):
    sys.path.insert(0, os.path.join(ROOT, "oracle", "refshim"))
    sys.path.insert(1, ref_dir)
    placeholders = {}
    for name in ("cv2", "absl", "absl.logging", "IPython", "IPython.display", "tensorflow_probability"):
        m = types.ModuleType(name)
        placeholders[name] = m
        sys.modules[name] = m
    placeholders["absl"].logging = placeholders["absl.logging"]
    placeholders["IPython"].display = placeholders["IPython.display"]
    placeholders["IPython.display"].HTML = object
    import tensorflow as tf   # the stand-in
    import data_utils         # /root/reference/data_utils.py, verbatim
    assert os.path.realpath(data_utils.__file__).startswith(os.path.realpath(ref_dir)), data_utils.__file__
    assert tf.__version__.endswith("numpy-standin")
    for name in placeholders:
        del sys.modules[name]
    return tf, data_utils


def run_case(tf, du, name, dtype):
    shape, seed, tk, sk, sigma = smooth_cases.CASES[name]
    sfx = "" if dtype == np.float32 else "_f64"
    tf.set_float(dtype)
    v = smooth_cases.gen_video(shape, seed).astype(dtype)
    ks = du.KernelSmoothing(temporal_kernel_size=tk, spatial_kernel_size=sk)
    out = {}
    out["taps1d_t"] = ks.gaussian_kernel1d(ks.temporal_radius, sigma)
    out["taps1d_s"] = ks.gaussian_kernel1d(ks.spatial_radius, sigma)
    out["taps3d"] = ks.gaussian_kernel3d(ks.spatial_radius, sigma)[:, :, :, 0, 0]
    out["temporal"] = ks.temporal_convolution(v, sigma)
    out["conv3d"] = ks.gaussian_convolution3D(v, sigma)
    assert out["temporal"].shape == tuple(shape) and out["conv3d"].shape == tuple(shape)
    try:
        ks.spatial_convolution(v, sigma)
        raised = 0
    except Exception as e:   # data_utils.py:537-538,547-548: VALID conv2d output reshaped to the input size
        raised = 1
        print("   spatial_convolution raises %s: %s" % (type(e).__name__, str(e)[:80]))
    res = {k + sfx: np.asarray(a) for k, a in out.items()}
    if dtype == np.float32:
        for k, a in res.items():
            assert a.dtype == np.float32, (k, a.dtype)    # the reference path stayed in fp32
        res["spatial_raises"] = np.asarray(raised)
        res["checksum"] = np.asarray(np.sum(v, dtype=np.float64))
        res["radii"] = np.asarray([ks.temporal_radius, ks.spatial_radius])
        res["annealing_sigma"] = np.asarray([ks.annealing_sigma(5.0, s) for s in smooth_cases.ANNEAL_STEPS],
                                            dtype=np.float64)
    return res


def lr_fixture(tf, du):
    """kernel_train.py:52-63: ExponentialDecay(staircase) inside data_utils.WarmUp, evaluated at the
    optimiser's iteration counts.  WarmUp.__call__ is the reference's; ExponentialDecay is the
    stand-in's statement of the documented Keras rule."""
    tf.set_float(np.float32)
    res = {}
    for tag, (lr, warmup, decay_steps, rate) in smooth_cases.LR_RUNS.items():
        sched = tf.keras.optimizers.schedules.ExponentialDecay(initial_learning_rate=lr, decay_steps=decay_steps,
                                                               decay_rate=rate, staircase=True)
        wu = du.WarmUp(initial_learning_rate=lr, decay_schedule_fn=sched, warmup_steps=warmup)
        res["lr_" + tag] = np.asarray([wu(np.int64(s)) for s in smooth_cases.LR_STEPS], dtype=np.float64)
    res["steps"] = np.asarray(smooth_cases.LR_STEPS)
    return res


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--ref", default="/root/reference")
    args = ap.parse_args()
    tf, du = load_reference(args.ref)
    for name in smooth_cases.CASES:
        res = {}
        for dtype in (np.float32, np.float64):
            r = run_case(tf, du, name, dtype)
            if dtype == np.float64 and name in smooth_cases.BIG:     # keep the big fixtures small: fp64 sums only
                r = {k: (np.asarray(a.sum()) if a.ndim == 5 else a) for k, a in r.items()}
            res.update(r)
        tf.set_float(np.float32)
        np.savez_compressed(os.path.join(HERE, "smooth_%s.npz" % name), **res)
        print("%-16s temporal max@%s  conv3d sum %.6f (f64 %.9f)  |f32-f64| %.2e / %.2e" % (
            name, np.unravel_index(np.argmax(res["temporal"]), res["temporal"].shape),
            res["conv3d"].sum(dtype=np.float64), res["conv3d_f64"].sum(),
            np.nan if name in smooth_cases.BIG else np.abs(res["temporal"] - res["temporal_f64"]).max(),
            np.nan if name in smooth_cases.BIG else np.abs(res["conv3d"] - res["conv3d_f64"]).max()),
            flush=True)
    np.savez(os.path.join(HERE, "lr_schedule.npz"), **lr_fixture(tf, du))
    print("lr_schedule written")


if __name__ == "__main__":
    main()
