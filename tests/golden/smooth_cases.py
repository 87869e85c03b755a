"""Shared definition of the KernelSmoothing / LR-schedule golden cases.  Used by
``make_golden_smoothing.py`` (build container, where /root/reference exists) and by the tests
(here and on the GPU box, where it does not).  Inputs are regenerated from the seed; every
golden file stores a float64 checksum of the regenerated tensor."""
import numpy as np

# name -> ([B,H,T,W,C], seed, temporal_kernel_size, spatial_kernel_size, sigma)
# kernel sizes 6,6 (radius 3) are what kernel_train.py:216 constructs; 6,8 are the class defaults
# (data_utils.py:479); sigma 5.0 is --init_sigma (kernel_train.py:408); 1.3 is a narrow kernel that
# a decaying sigma reaches late (data_utils.py:584-586).  C = 1 and C = 3 are the two code branches
# of gaussian_convolution3D (data_utils.py:555-573 vs :575-581).
CASES = {
    "c1_r3_s5": ((3, 12, 10, 14, 1), 0, 6, 6, 5.0),
    "c3_r3_s5": ((2, 12, 10, 14, 3), 1, 6, 6, 5.0),
    "c1_r3_s1p3": ((3, 12, 10, 14, 1), 2, 6, 6, 1.3),
    "c3_r3_s1p3": ((2, 12, 10, 14, 3), 3, 6, 6, 1.3),
    "c1_r4_s5": ((2, 10, 9, 12, 1), 4, 8, 8, 5.0),
    "c3_r4_s1p3": ((2, 10, 9, 12, 3), 5, 8, 8, 1.3),
    "c3_r34_s2": ((2, 9, 8, 10, 3), 6, 6, 8, 2.0),          # class defaults: temporal radius 3, spatial radius 4
    "cfg1frame": ((2, 64, 20, 64, 1), 7, 6, 6, 5.0),        # BASELINE configs[0] frame size and T
    "t30c3": ((2, 16, 30, 16, 3), 8, 6, 6, 5.0),            # T = 30 (configs[1..3]), 3 channels
}

BIG = ("cfg1frame", "t30c3")     # fp64 outputs of these are stored as sums only

ANNEAL_STEPS = [0, 1, 250, 500, 12345]

# kernel_train.py:52-59,385,404: lr 5e-4, warmup 10000, decay_steps 5000, decay_rate 0.975 (+ a short run)
LR_RUNS = {"default": (5e-4, 10000, 5000, 0.975), "short": (1e-3, 6, 4, 0.5)}
LR_STEPS = [0, 1, 2, 3, 4, 5, 6, 7, 9, 10, 13, 14, 18, 100, 9999, 10000, 10001, 14999, 15000, 15001, 20000, 60000]


def gen_video(shape, seed):
    """[B,H,T,W,C] fp32: U[0,1) noise plus a smooth moving blob so that the maximum is not at a border."""
    rng = np.random.default_rng(1000 + seed)
    v = rng.random(shape, dtype=np.float32)
    B, H, T, W, C = shape
    hh = np.arange(H, dtype=np.float32)[None, :, None, None, None]
    tt = np.arange(T, dtype=np.float32)[None, None, :, None, None]
    ww = np.arange(W, dtype=np.float32)[None, None, None, :, None]
    blob = np.exp(-((hh - H / 2 - tt / 4) ** 2 + (ww - W / 3 - tt / 5) ** 2) / np.float32(2 * (min(H, W) / 4) ** 2))
    return (np.float32(0.5) * v + np.float32(0.5) * blob.astype(np.float32)).astype(np.float32)
