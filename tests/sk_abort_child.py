"""Child process of tests/test_gpu_parity.py::test_sinkhorn_abort_is_nan_plus_status_never_a_plausible_number.

Runs on csrc/libkccot_diag.so (KCCOT_LIB_PATH), the only build that carries the fault-injection hook
(KCCOT_SK_FAULT_INJECT=1: one workgroup of problem 0 of the multi-CU Sinkhorn never takes part).  The bounded polling
must drain the launch (about a second), and the result must be unmistakable: cost NaN, a NEGATIVE iteration count,
kccot_sinkhorn_status = KCCOT_EABORTED, NaN gradients from the reverse sweep, KccotError from the wrapper's status check.
The problem next to it in the same launch and the next launch (hook off) are unaffected.
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from kccotgan_amd import _lib as L                      # noqa: E402
from kccotgan_amd import gan_utils as G                 # noqa: E402
from oracle import gan_utils_np as o                    # noqa: E402

assert L.LIB_PATH.endswith("libkccot_diag.so"), L.LIB_PATH
DEV = "cuda:0"
n = 256
Cn = (np.random.default_rng(6).random((2, n, n), dtype=np.float32) * 4).astype(np.float32)
C = torch.from_numpy(Cn).to(DEV).requires_grad_(True)
cost = G._Sinkhorn.apply(C, 1.0, 20, 100, L.STOP_COUNT, "compute_sinkhorn")
cost.sum().backward()
torch.cuda.synchronize()
nits = G.last_info["compute_sinkhorn"]
assert bool(torch.isnan(cost[0])) and int(nits[0]) < 0, (cost, nits)
assert bool(torch.isnan(C.grad[0]).all())
assert L.lib.kccot_sinkhorn_status(L.ptr(nits.contiguous()), 2, None) == L.EABORTED
try:
    G.raise_if_solver_aborted(("compute_sinkhorn",))
    raise SystemExit("raise_if_solver_aborted did not raise")
except L.KccotError as e:
    assert "aborted" in str(e), e
os.environ["KCCOT_SK_FAULT_INJECT"] = "0"               # the diagnostic build reads the hook per launch
C2 = torch.from_numpy(Cn).to(DEV)
cost2 = G._Sinkhorn.apply(C2, 1.0, 20, 100, L.STOP_COUNT, "compute_sinkhorn")
ref = [o.sinkhorn_from_cost(Cn[p], 1.0, 20)[0] for p in range(2)]
rel = lambda a, b: abs(float(a) - float(b)) / abs(float(b))
assert rel(cost2[0], ref[0]) < 2e-5 and rel(cost2[1], ref[1]) < 2e-5
assert G.last_info["compute_sinkhorn"].tolist() == [20, 20]
G.raise_if_solver_aborted(("compute_sinkhorn",))        # clean again
print("abort path ok")
