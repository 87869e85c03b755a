"""CPU oracle: NumPy restatement of the reference's causal-OT loss path.

TEST INFRASTRUCTURE ONLY -- imported by ``tests/``, ``__graft_entry__.smoke()``
and ``bench.py``'s ``cpu_baseline`` leg, never by the product package
``kccotgan_amd``.  It is the checker for the HIP path, not a fallback.

Pinning: every function below is checked in ``tests/test_oracle_golden.py``
against the golden vectors in ``tests/golden/*.npz``, which were produced by
executing ``/root/reference/gan_utils.py`` itself (verbatim, over the NumPy
``tensorflow`` stand-in in ``oracle/refshim``; see ``tests/golden/make_golden.py``).
Real-TensorFlow bitwise behaviour (Eigen summation order) is NOT pinned.

Each function cites the reference lines it follows.  ``dtype`` selects the
arithmetic type (float32 = the reference's; float64 = high-precision value of
the same algorithm).  Extra return values (cost matrices, iteration counts,
transport plans) are exposed through the ``*_ex`` variants for the tests.
"""
import numpy as np

THRESH = 10 ** (-2)  # gan_utils.py:91,144
LMIN_CAUSAL = 100    # gan_utils.py:149 (hard-coded)


def _lse(a, axis):
    """tf.reduce_logsumexp: shift by the (finite) max, log-sum-exp, add back."""
    m = np.max(a, axis=axis, keepdims=True)
    m = np.where(np.isfinite(m), m, np.zeros_like(m))
    return np.log(np.sum(np.exp(a - m), axis=axis, keepdims=True)) + m


def cost_xy(x, y, scaling_coef, dtype=np.float32, chunk=None):
    """gan_utils.py:6-18.  C[i,j] = sc * sum_t sum_d (x[i,t,d]-y[j,t,d])^2,
    features reduced first, then time, then scaled.  ``chunk`` evaluates the
    identical arithmetic a few columns j at a time so that the [B,B,T,D]
    temporary of the reference (2 GB at B=64,T=30,D=4096) need not exist."""
    x = np.asarray(x, dtype=dtype)
    y = np.asarray(y, dtype=dtype)
    sc = dtype(scaling_coef)
    bx, by = x.shape[0], y.shape[0]
    if chunk is None:
        d = x[:, None] - y[None, :]
        return np.sum(np.sum(d * d, axis=-1), axis=-1) * sc
    out = np.empty((bx, by), dtype=dtype)
    for j0 in range(0, by, chunk):
        d = x[:, None] - y[None, j0:j0 + chunk]
        out[:, j0:j0 + chunk] = np.sum(np.sum(d * d, axis=-1), axis=-1) * sc
    return out


def causal_term(h, M, scaling_coef, dtype=np.float32):
    """gan_utils.py:34-38.  sc * sum_{t<T-1} sum_k h[i,t,k]*(M[j,t+1,k]-M[j,t,k]);
    h indexes ROWS, M indexes COLUMNS (gan_utils.py:37)."""
    h = np.asarray(h, dtype=dtype)
    M = np.asarray(M, dtype=dtype)
    dM = M[:, 1:, :] - M[:, :-1, :]
    ht = h[:, :-1, :]
    s = np.sum(ht[:, None, :, :] * dM[None, :, :, :], axis=-1)
    return np.sum(s, axis=-1) * dtype(scaling_coef)


def modified_cost(x, y, h, M, scaling_coef, dtype=np.float32, chunk=None):
    """gan_utils.py:21-43."""
    return cost_xy(x, y, scaling_coef, dtype, chunk) + causal_term(h, M, scaling_coef, dtype)


def bi_causal_modified_cost(x, y, hy, Mx, hx, My, scaling_coef, dtype=np.float32, chunk=None):
    """gan_utils.py:46-72: l2 + C_hM(hy, Mx) + C_Mh(hx, My), summed in that order."""
    return (cost_xy(x, y, scaling_coef, dtype, chunk)
            + causal_term(hy, Mx, scaling_coef, dtype)
            + causal_term(hx, My, scaling_coef, dtype))


def sinkhorn_from_cost(C, epsilon=1.0, L=100, Lmin=LMIN_CAUSAL, stop_on_index=False,
                       dtype=np.float32, history=False):
    """The loop of gan_utils.py:138-165 (stop_on_index=False: count-based stop,
    ``actual_nits >= Lmin``) or gan_utils.py:87-121 (stop_on_index=True: the
    loop index ``i >= Lmin`` is tested, gan_utils.py:116).

    Returns (cost, nits, u, v, pi[, u_hist, v_hist])."""
    C = np.asarray(C, dtype=dtype)
    n = C.shape[0]
    eps = dtype(epsilon)
    one = dtype(1.0)
    mu = (one / dtype(n)) * np.ones((n, 1), dtype=dtype)
    nu = (one / dtype(n)) * np.ones((n, 1), dtype=dtype)
    u = dtype(0.0) * mu
    v = dtype(0.0) * nu
    log_mu, log_nu = np.log(mu), np.log(nu)
    nits = 0
    uh, vh = [], []
    for i in range(int(L)):
        u1 = u
        Muv = (-C + u + v.T) / eps
        u = eps * (log_mu - _lse(Muv, 1)) + u
        Muv = (-C + u + v.T) / eps
        v = eps * (log_nu - _lse(Muv.T, 1)) + v
        err = np.sum(np.abs(u - u1))
        nits += 1
        if history:
            uh.append(u[:, 0].copy())
            vh.append(v[:, 0].copy())
        reached = (i >= Lmin) if stop_on_index else (nits >= Lmin)
        if THRESH > err and reached:
            break
    Muv = (-C + u + v.T) / eps
    pi = np.exp(Muv)
    cost = np.sum(pi * C)
    if history:
        return cost, nits, u[:, 0], v[:, 0], pi, np.array(uh, dtype=dtype), np.array(vh, dtype=dtype)
    return cost, nits, u[:, 0], v[:, 0], pi


def benchmark_sinkhorn(x, y, scaling_coef, epsilon=1.0, L=10, Lmin=10, dtype=np.float32):
    """gan_utils.py:75-121."""
    C = cost_xy(x, y, scaling_coef, dtype)
    return sinkhorn_from_cost(C, epsilon, L, Lmin, stop_on_index=True, dtype=dtype)[0]


def compute_sinkhorn_ex(x, y, hy, Mx, scaling_coef, hx=None, My=None, epsilon=1.0, L=100,
                        bi_causal=False, dtype=np.float32, chunk=None):
    """gan_utils.py:124-165, returning (cost, nits, C)."""
    if bi_causal:
        C = bi_causal_modified_cost(x, y, hy, Mx, hx, My, scaling_coef, dtype, chunk)
    else:
        C = modified_cost(x, y, hy, Mx, scaling_coef, dtype, chunk)
    cost, nits = sinkhorn_from_cost(C, epsilon, L, LMIN_CAUSAL, False, dtype)[:2]
    return cost, nits, C


def compute_sinkhorn(x, y, hy, Mx, scaling_coef, hx=None, My=None, epsilon=1.0, L=100,
                     bi_causal=False, dtype=np.float32, chunk=None):
    return compute_sinkhorn_ex(x, y, hy, Mx, scaling_coef, hx, My, epsilon, L, bi_causal,
                               dtype, chunk)[0]


def compute_N(M):
    """gan_utils.py:168-176."""
    M = np.asarray(M)
    return M[:, 1:] - M[:, :M.shape[1] - 1]


def scale_invariante_martingale_regularization(M, reg_lam, scaling_coef, dtype=np.float32):
    """gan_utils.py:179-201: population std over (batch, time) per feature."""
    M = np.asarray(M, dtype=dtype)
    m = dtype(M.shape[0])
    N = M[:, 1:, :] - M[:, :-1, :]
    mean = np.mean(M, axis=(0, 1), keepdims=True)
    std = np.sqrt(np.mean(np.square(M - mean), axis=(0, 1)))
    N_std = N / (std + dtype(1e-06))
    sum_m_std = np.sum(N_std, axis=0) / m
    sum_across_paths = np.sum(np.abs(sum_m_std)) * dtype(scaling_coef)
    return dtype(reg_lam) * sum_across_paths


def flatten_video(f):
    """gan_utils.py:216-220: [B,H,T,W,C] -> transpose(0,2,1,3,4) -> [B,T,H*W*C]."""
    f = np.asarray(f)
    f = np.transpose(f, (0, 2, 1, 3, 4))
    return np.reshape(f, (f.shape[0], f.shape[1], -1))


def compute_sinkhorn_loss_ex(f_real, f_fake, scaling_coef, sinkhorn_eps, sinkhorn_l, h_fake, m_real,
                             h_real, m_fake, video=True, dtype=np.float32, chunk=None):
    """gan_utils.py:204-227 incl. the positional quirk: ``sinkhorn_eps`` and
    ``sinkhorn_l`` are passed into the ``hx`` / ``My`` slots of
    ``compute_sinkhorn`` (gan_utils.py:221-223 vs :124) and, bi_causal being
    False, are ignored -- epsilon = 1.0 and L = 100 always."""
    if video:
        f_real = flatten_video(f_real)
        f_fake = flatten_video(f_fake)
    xy, n_xy, Cxy = compute_sinkhorn_ex(f_real, f_fake, h_fake, m_real, scaling_coef,
                                        sinkhorn_eps, sinkhorn_l, dtype=dtype, chunk=chunk)
    xx, n_xx, Cxx = compute_sinkhorn_ex(f_real, f_real, h_real, m_real, scaling_coef,
                                        sinkhorn_eps, sinkhorn_l, dtype=dtype, chunk=chunk)
    yy, n_yy, Cyy = compute_sinkhorn_ex(f_fake, f_fake, h_fake, m_fake, scaling_coef,
                                        sinkhorn_eps, sinkhorn_l, dtype=dtype, chunk=chunk)
    loss = dtype(2.0) * xy - xx - yy
    return loss, dict(xy=xy, xx=xx, yy=yy, nits=(n_xy, n_xx, n_yy), Cxy=Cxy, Cxx=Cxx, Cyy=Cyy)


def compute_sinkhorn_loss(f_real, f_fake, scaling_coef, sinkhorn_eps, sinkhorn_l, h_fake, m_real,
                          h_real, m_fake, video=True, dtype=np.float32, chunk=None):
    return compute_sinkhorn_loss_ex(f_real, f_fake, scaling_coef, sinkhorn_eps, sinkhorn_l, h_fake,
                                    m_real, h_real, m_fake, video, dtype, chunk)[0]
