#!/usr/bin/env python3
"""Single-GPU time of compute_sinkhorn_loss (forward + backward) at the full sizes of BASELINE configs 2-5
(configs 3-5 are multi-GPU configurations in BASELINE.json; this is their one-GPU equivalent: what every rank
of the sharded path would do without the all-gather, plus all row blocks instead of B/G of them)."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from kccotgan_amd import gan_utils as G

CFG = {  # name: (B, H, T, W, C, L)
    "cfg2": (64, 64, 30, 64, 1, 100),
    "cfg3": (128, 64, 30, 64, 3, 100),
    "cfg4": (256, 64, 30, 64, 3, 200),
    "cfg5": (512, 128, 48, 128, 3, 300),
}
dev = "cuda:0"
for name in (sys.argv[1:] or list(CFG)):
    B, H, T, W, C, L = CFG[name]
    g = torch.Generator(device=dev).manual_seed(0)
    real = torch.rand((B, H, T, W, C), device=dev, generator=g)
    fake = (real + 0.05 * torch.randn(real.shape, device=dev, generator=g)).clamp_(0, 1).requires_grad_(True)
    f = {k: torch.rand((B, T, 8), device=dev, generator=g).requires_grad_(True) for k in ("h_fake", "m_real", "h_real", "m_fake")}

    def step():
        loss = G.compute_sinkhorn_loss(real, fake, 1 / 15.0, 1.0, L, f["h_fake"], f["m_real"], f["h_real"], f["m_fake"],
                                       honor_eps_l=True)
        grads = torch.autograd.grad(loss, [fake, f["h_fake"], f["h_real"], f["m_real"], f["m_fake"]])
        return loss, grads

    loss, grads = step()
    torch.cuda.synchronize()
    reps = 5 if B <= 256 else 2
    t0 = time.perf_counter()
    for _ in range(reps):
        loss, grads = step()
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / reps * 1e3
    K = H * T * W * C
    print(json.dumps(dict(config=name, B=B, K=K, L=L, ms_fwd_bwd=ms, loss=float(loss), finite=bool(torch.isfinite(grads[0]).all()),
                          nits=G.last_info["compute_sinkhorn_loss"].tolist(),
                          alg_MB=2 * B * K * 4 / 1e6, alg_GFLOP=4 * B * B * K / 1e9)), flush=True)
    del real, fake, grads
    torch.cuda.empty_cache()
