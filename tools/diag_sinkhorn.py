#!/usr/bin/env python3
"""Where does a Sinkhorn half-step spend its cycles?  Loads the DIAGNOSTIC twin library
(libkccot_diag.so, built with -DKCCOT_DIAG: s_memtime stamps around the phases of iteration 50)
and prints per-wave cycle counts between stamps.  Never used by the product path."""
import ctypes, os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
lib = ctypes.CDLL(os.path.join(ROOT, "kccotgan_amd", "csrc", "libkccot_diag.so"))
vp, ci, cf, sz = ctypes.c_void_p, ctypes.c_int, ctypes.c_float, ctypes.c_size_t
lib.kccot_sinkhorn_fwd_f32.argtypes = [vp, ci, ci, cf, ci, ci, cf, ci, vp, vp, vp, vp, vp, vp, sz, vp]
n = int(sys.argv[1]) if len(sys.argv) > 1 else 64
g = np.load(os.path.join(ROOT, "tests", "golden", sys.argv[2] if len(sys.argv) > 2 else "cfg2_s0_near.npz"))
C = torch.from_numpy(np.stack([g["C_xy"], g["C_xx"], g["C_yy"]])).cuda()[:, :n, :n].contiguous()
L = 100
uh = torch.empty(3, L, n, device="cuda"); vh = torch.empty(3, L, n, device="cuda")
cost = torch.empty(3, device="cuda"); nits = torch.empty(6, dtype=torch.int32, device="cuda")
diag = torch.zeros(3 * 16 * 16, dtype=torch.int64, device="cuda")
for rep in range(3):
    rc = lib.kccot_sinkhorn_fwd_f32(C.data_ptr(), 3, n, 1.0, L, 100, 1e-2, 0, uh.data_ptr(), vh.data_ptr(), cost.data_ptr(),
                                    nits.data_ptr(), None, diag.data_ptr(), diag.numel() * 8, None)
    assert rc == 0
torch.cuda.synchronize()
d = diag.cpu().numpy().reshape(3, 16, 16)
names = ["row half-step", "store u", "barrier 1", "col half-step", "store v", "barrier 2", "detect"]
print("nits", nits.tolist())
print("cost", cost.tolist())
for w in range(16):
    st = d[0, w, :8]
    if st[0] == 0: continue
    print("wave %2d: " % w + "  ".join("%s %d" % (names[k], st[k + 1] - st[k]) for k in range(7 if st[7] else 6)) + "  | total %d" % (st[6] - st[0]))
