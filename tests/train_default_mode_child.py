"""Child process of tests/test_gpu_train_step.py::test_default_convolution_mode_in_a_child_process: the SHIPPED
configuration of KCCOTTrainer (MIOpen convolutions with the one faulting backward solver switched off by the
package's __init__) on the deterministic reproducer of round 1's "Memory access fault by GPU" aborts (B = 2, T = 6,
moving squares, sample + fit).  Exits 0 and prints "done <iterations> <exploded>" when every step completed."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import kccotgan_amd  # noqa: E402,F401  (first import of the process: sets the MIOpen switch before any convolution)
import torch  # noqa: E402
from kccotgan_amd import datasets as ds, gan  # noqa: E402
from kccotgan_amd.kernel_train import KCCOTTrainer  # noqa: E402

assert kccotgan_amd.MIOPEN_WORKAROUND_GUARANTEED and os.environ[kccotgan_amd.MIOPEN_SWITCH] == "0"
assert gan._NATIVE == set(), gan._NATIVE               # default mode: MIOpen, nothing forced onto the native kernels
B, H, W, C, T, iT = 2, 64, 64, 1, 6, 2
tr = KCCOTTrainer(B, total_time_steps=T, int_time_steps=iT, x_height=H, x_width=W, channels=C, kernel="1d", warmup=10,
                  device="cuda:0")
videos = ds.mmnist_videos(ds.synthetic_moving_squares(7, H, T, W, seed=2), T)
test_x = next(ds.batches(videos, B, H, T, W, C))
logged = []
out = tr.fit(ds.batches(videos, B, H, T, W, C, epochs=2), test_x=test_x, decaying_sigma=True, save_freq=3,
             log=lambda name, value, step: logged.append((name, step)))
torch.cuda.synchronize()
assert [s for n, s in logged if n == "Sinkhorn Loss"] == [1, 2, 3, 4, 5, 6]
assert tr.gen_optimiser.iterations == 12 and tr.dischm_optimiser.iterations == 12      # two apply_gradients per step
print("done", out["iterations"], out["exploded"], flush=True)
