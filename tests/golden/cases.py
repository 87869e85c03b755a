"""Shared definition of the golden cases: shapes, seeds and the seeded input
generator.  Used by ``make_golden.py`` (in the build container, where
/root/reference exists) and by the tests (here and on the GPU box, where it does
not).  Inputs are regenerated from the seed; every golden file stores a float64
checksum of the regenerated tensors so RNG drift would be detected.

Distributions follow SURVEY.md section 8c/8d: real ~ U[0,1); "near" fake =
clip(real + 0.05 N(0,1), 0, 1) (GAN-like), "far" fake ~ U[0,1) independent;
h, M ~ U[0,1) (sigmoid range of the discriminators, gan.py:418).
"""
import numpy as np

SC = 1.0 / 15.0   # kernel_train.py:71,392  (scaling_coef = 1/15)
LAM = 1.0         # kernel_train.py:380     (reg_penalty)

# name -> (B, H, T, W, C, J)
SHAPES = {
    "small": (16, 4, 10, 4, 2, 8),      # D = 32, costs O(1): exercises the Sinkhorn loop
    "tiny": (5, 2, 4, 3, 1, 3),         # ragged, odd sizes
    "deci64": (64, 8, 30, 8, 1, 8),     # config-2 batch/time, decimated frame (D = 64)
    "cfg1": (8, 64, 20, 64, 1, 8),      # BASELINE configs[0]
    "cfg2": (64, 64, 30, 64, 1, 8),     # BASELINE configs[1]
    "deci128": (128, 8, 10, 8, 4, 8),   # config-3 batch, decimated frames (K = 2560): multi-rank tests only, no golden file
    "deci256": (256, 8, 10, 8, 4, 8),   # config-4 batch, decimated frames: multi-rank tests only (n = 256: the multi-CU Sinkhorn)
}

# (shape name, seed, regime)
CASES = [
    ("tiny", 0, "near"), ("tiny", 1, "far"),
    ("small", 0, "near"), ("small", 1, "far"), ("small", 2, "near"),
    ("deci64", 0, "near"), ("deci64", 1, "far"),
    ("cfg1", 0, "near"), ("cfg1", 1, "far"), ("cfg1", 2, "near"),
    ("cfg2", 0, "near"), ("cfg2", 1, "far"),
]

# (epsilon, L) pairs exercised through compute_sinkhorn's keywords (the only way
# the reference honours them: gan_utils.py:124)
EPS_L = [(1.0, 100), (0.8, 20), (0.8, 200), (0.8, 300), (0.25, 300)]


def case_name(shape, seed, regime):
    return "%s_s%d_%s" % (shape, seed, regime)


def gen_inputs(shape, seed, regime):
    """Returns dict(real, fake [B,H,T,W,C], h_fake, m_real, h_real, m_fake [B,T,J]) fp32."""
    B, H, T, W, C, J = SHAPES[shape]
    rng = np.random.default_rng(seed)
    real = rng.random((B, H, T, W, C), dtype=np.float32)
    if regime == "near":
        noise = rng.standard_normal((B, H, T, W, C), dtype=np.float32)
        fake = np.clip(real + np.float32(0.05) * noise, 0.0, 1.0).astype(np.float32)
    elif regime == "far":
        fake = rng.random((B, H, T, W, C), dtype=np.float32)
    else:
        raise ValueError(regime)
    feats = {k: rng.random((B, T, J), dtype=np.float32)
             for k in ("h_fake", "m_real", "h_real", "m_fake")}
    out = dict(real=real, fake=fake)
    out.update(feats)
    return out


def checksum(inp):
    return np.array([np.sum(inp[k], dtype=np.float64) for k in
                     ("real", "fake", "h_fake", "m_real", "h_real", "m_fake")])


# Crafted slow-converging problem (quirk 2, gan_utils.py:149-160: the loop may only
# stop from iteration 100 on, and runs on while sum|u-u_prev| >= 1e-2): points on a
# line, interleaved, no causal term.  (scaling_coef, L) pairs to run:
LINE_N = 32
LINE_RUNS = [(100.0, 300), (100.0, 150), (300.0, 300), (100.0, 100)]


def gen_line_inputs(n=LINE_N):
    x = np.zeros((n, 2, 1), np.float32)
    y = np.zeros((n, 2, 1), np.float32)
    x[:, :, 0] = (np.arange(n, dtype=np.float32) / np.float32(n))[:, None]
    y[:, :, 0] = ((np.arange(n, dtype=np.float32) + np.float32(0.5)) / np.float32(n))[:, None]
    h = np.zeros((n, 2, 2), np.float32)
    M = np.zeros((n, 2, 2), np.float32)
    return x, y, h, M
