"""CPU oracle, torch flavour: the same restatement as ``oracle/gan_utils_np.py`` written with
torch ops so that (a) autograd differentiates THROUGH THE UNROLLED LOOP exactly as
tf.GradientTape does in the reference (kernel_train.py:221,252,262,289) -- this is the gradient
oracle -- and (b) it can be timed on the host cores as the CPU baseline: it deliberately keeps
the reference's formulation ([B,B,T,D] broadcast per cost matrix, three separate cost builds,
one eager op sequence per Sinkhorn iteration).

TEST INFRASTRUCTURE ONLY (tests/, __graft_entry__.smoke(), bench.py's cpu_baseline leg).
Pinned against the golden vectors in tests/test_oracle_golden.py::test_torch_oracle_*.
"""
import torch

THRESH = 10 ** (-2)
LMIN = 100


def cost_xy(x, y, scaling_coef, chunk=None):
    """gan_utils.py:6-18 (broadcast formulation kept on purpose)."""
    if chunk is None:
        d = x.unsqueeze(1) - y.unsqueeze(0)
        return (d ** 2).sum(-1).sum(-1) * scaling_coef
    cols = []
    for j0 in range(0, y.shape[0], chunk):
        d = x.unsqueeze(1) - y[j0:j0 + chunk].unsqueeze(0)
        cols.append((d ** 2).sum(-1).sum(-1) * scaling_coef)
    return torch.cat(cols, dim=1)


def causal_term(h, M, scaling_coef):
    """gan_utils.py:34-38."""
    dM = M[:, 1:, :] - M[:, :-1, :]
    ht = h[:, :-1, :]
    return (ht[:, None, :, :] * dM[None, :, :, :]).sum(-1).sum(-1) * scaling_coef


def modified_cost(x, y, h, M, scaling_coef, chunk=None):
    """gan_utils.py:21-43."""
    return cost_xy(x, y, scaling_coef, chunk) + causal_term(h, M, scaling_coef)


def bi_causal_modified_cost(x, y, hy, Mx, hx, My, scaling_coef, chunk=None):
    """gan_utils.py:46-72."""
    return cost_xy(x, y, scaling_coef, chunk) + causal_term(hy, Mx, scaling_coef) + causal_term(hx, My, scaling_coef)


def sinkhorn_from_cost(C, epsilon=1.0, L=100, Lmin=LMIN, stop_on_index=False):
    """gan_utils.py:138-165 / :87-121.  Returns (cost, nits)."""
    n = C.shape[0]
    mu = torch.full((n, 1), 1.0 / n, dtype=C.dtype)
    log_mu = torch.log(mu)
    u = torch.zeros_like(mu)
    v = torch.zeros_like(mu)
    nits = 0
    for i in range(int(L)):
        u1 = u
        Muv = (-C + u + v.t()) / epsilon
        u = epsilon * (log_mu - torch.logsumexp(Muv, dim=1, keepdim=True)) + u
        Muv = (-C + u + v.t()) / epsilon
        v = epsilon * (log_mu - torch.logsumexp(Muv.t(), dim=1, keepdim=True)) + v
        err = (u - u1).abs().sum()
        nits += 1
        reached = (i >= Lmin) if stop_on_index else (nits >= Lmin)
        if THRESH > float(err) and reached:
            break
    pi = torch.exp((-C + u + v.t()) / epsilon)
    return (pi * C).sum(), nits


def compute_sinkhorn(x, y, hy, Mx, scaling_coef, hx=None, My=None, epsilon=1.0, L=100, bi_causal=False,
                     chunk=None):
    """gan_utils.py:124-165."""
    if bi_causal:
        C = bi_causal_modified_cost(x, y, hy, Mx, hx, My, scaling_coef, chunk)
    else:
        C = modified_cost(x, y, hy, Mx, scaling_coef, chunk)
    return sinkhorn_from_cost(C, epsilon, L)[0]


def benchmark_sinkhorn(x, y, scaling_coef, epsilon=1.0, L=10, Lmin=10):
    """gan_utils.py:75-121."""
    return sinkhorn_from_cost(cost_xy(x, y, scaling_coef), epsilon, L, Lmin, stop_on_index=True)[0]


def scale_invariante_martingale_regularization(M, reg_lam, scaling_coef):
    """gan_utils.py:179-201."""
    m = M.shape[0]
    N = M[:, 1:, :] - M[:, :-1, :]
    std = torch.sqrt(((M - M.mean(dim=(0, 1), keepdim=True)) ** 2).mean(dim=(0, 1)))
    N_std = N / (std + 1e-06)
    return reg_lam * ((N_std.sum(0) / m).abs().sum() * scaling_coef)


def flatten_video(f):
    """gan_utils.py:216-220."""
    f = f.permute(0, 2, 1, 3, 4)
    return f.reshape(f.shape[0], f.shape[1], -1)


def compute_sinkhorn_loss(f_real, f_fake, scaling_coef, sinkhorn_eps, sinkhorn_l, h_fake, m_real, h_real,
                          m_fake, video=True, chunk=None):
    """gan_utils.py:204-227, including the positional quirk (eps = 1.0, L = 100 always)."""
    if video:
        f_real, f_fake = flatten_video(f_real), flatten_video(f_fake)
    xy = compute_sinkhorn(f_real, f_fake, h_fake, m_real, scaling_coef, sinkhorn_eps, sinkhorn_l, chunk=chunk)
    xx = compute_sinkhorn(f_real, f_real, h_real, m_real, scaling_coef, sinkhorn_eps, sinkhorn_l, chunk=chunk)
    yy = compute_sinkhorn(f_fake, f_fake, h_fake, m_fake, scaling_coef, sinkhorn_eps, sinkhorn_l, chunk=chunk)
    return 2.0 * xy - xx - yy
