#!/usr/bin/env python3
"""Loss + gradients of compute_sinkhorn_loss on seeded inputs, saved as .npy (bit-level A/B of two library builds:
run once per build with KCCOT_LIB_PATH, then tools/dump_loss_grads.py --compare a.npz b.npz).
usage: dump_loss_grads.py out.npz [B H T W C]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
if sys.argv[1] == "--compare":
    a, b = np.load(sys.argv[2]), np.load(sys.argv[3])
    bad = 0
    for k in a.files:
        same = a[k].tobytes() == b[k].tobytes()
        d = float(np.abs(a[k].astype(np.float64) - b[k].astype(np.float64)).max())
        print("%-10s %s  max|diff| %.3e" % (k, "bit-identical" if same else "DIFFERENT", d))
        bad += not same
    sys.exit(1 if bad else 0)
import torch
from kccotgan_amd import gan_utils as G
B, H, T, W, C = (int(x) for x in sys.argv[2:7]) if len(sys.argv) > 6 else (64, 64, 30, 64, 1)
g = torch.Generator(device="cpu").manual_seed(1234)
real = torch.rand(B, H, T, W, C, generator=g).cuda()
fake = (real + 0.05 * torch.randn(B, H, T, W, C, generator=g).cuda()).clamp(0, 1).requires_grad_(True)
hs = [torch.rand(B, T - 1, 8, generator=g).cuda().requires_grad_(True) for _ in range(2)]    # h_fake, h_real
ms = [torch.rand(B, T - 1, 8, generator=g).cuda().requires_grad_(True) for _ in range(2)]        # m_real, m_fake
loss = G.compute_sinkhorn_loss(real, fake, 1 / 15.0, 1.0, 100, hs[0], ms[0], hs[1], ms[1])
loss.backward()
torch.cuda.synchronize()
np.savez(sys.argv[1], loss=loss.detach().cpu().numpy(), dfake=fake.grad.cpu().numpy(),
         **{"dh%d" % i: h.grad.cpu().numpy() for i, h in enumerate(hs)}, **{"dm%d" % i: m.grad.cpu().numpy() for i, m in enumerate(ms)})
print("loss %.6f saved %s" % (float(loss), sys.argv[1]))
