"""kccotgan_amd -- MI355X-native causal-OT (Sinkhorn) + kernel-smoothing loss path of KCCOT-GAN.

Importing the package loads ``csrc/libkccot.so`` (HIP, gfx950) and fails loudly if it is missing.
"""
import os as _os
import sys as _sys

# MIOpen's ``ConvAsmImplicitGemmGTCDynamicBwdXdlopsNHWC`` solver (kernel ``igemm_bwd_gtcx35_nhwc_fp32_*``) reads past the
# end of a buffer in the backward of the G/D models on this ROCm image ("Memory access fault by GPU", DESIGN.md
# section 7).  It is switched off through the environment variable MIOpen reads when it first selects solvers -- which
# is why this happens HERE, at the first import of the package and before anything of it can run a convolution.
# MIOpen caches the switch: if the process already had a live GPU context (a convolution may have run) and the user
# had not set the variable, the workaround cannot be guaranteed and ``kccotgan_amd.gan`` falls back to the native
# ATen convolutions (forward and backward) instead.
MIOPEN_SWITCH = "MIOPEN_DEBUG_CONV_IMPLICIT_GEMM_ASM_BWD_GTC_XDLOPS_NHWC"
_user_value = _os.environ.get(MIOPEN_SWITCH)
_gpu_was_live = "torch" in _sys.modules and _sys.modules["torch"].cuda.is_initialized()
_os.environ.setdefault(MIOPEN_SWITCH, "0")
MIOPEN_WORKAROUND_GUARANTEED = (_user_value == "0") or (_user_value is None and not _gpu_was_live)
if not MIOPEN_WORKAROUND_GUARANTEED:
    import warnings as _warnings
    _warnings.warn(
        "kccotgan_amd was imported after the GPU context was initialised (or %s is set to something other than 0): the "
        "faulting MIOpen solver cannot be switched off reliably any more, so the G/D models (kccotgan_amd.gan) will run "
        "their convolutions on the native ATen kernels -- about 7x slower per training iteration -- and the shipped MIOpen "
        "find-db is not installed.  Import kccotgan_amd before the first CUDA call, or export %s=0 before starting Python.  "
        "(The loss path -- gan_utils, data_utils -- is not affected.)" % (MIOPEN_SWITCH, MIOPEN_SWITCH),
        RuntimeWarning, stacklevel=2)


def _install_miopen_find_db(force=False):
    """Ship MIOpen's solver choices for the G/D convolutions of the BASELINE shapes (``miopen_db/*.ufdb.txt``, 30 KB, written
    by MIOpen itself during one run with its default exhaustive find on an MI355X: tools/bench_train.py).  With it the first
    training iteration takes seconds instead of ~5 minutes of solver benchmarking, and the steady state is the exhaustive
    search's (0.94 s per iteration at configs[1]) rather than fast-find's (2.5 s).  The files are copied to a per-user
    cache directory (MIOpen appends to its user database) and MIOPEN_USER_DB_PATH is pointed there -- unless the user has
    set that variable, already has a tuned database in MIOpen's default place, sets KCCOT_NO_MIOPEN_DB=1, or the host has
    no AMD GPU (/dev/kfd).  A database
    written by another MIOpen version has another file name and is simply not read.  Returns the directory or None."""
    import shutil
    if _os.environ.get("KCCOT_NO_MIOPEN_DB") == "1" or "MIOPEN_USER_DB_PATH" in _os.environ:
        return None
    if not force and not _os.path.exists("/dev/kfd"):
        return None                                       # no AMD GPU on this host: nothing will run a convolution
    try:
        src = _os.path.join(_os.path.dirname(_os.path.abspath(__file__)), "miopen_db")
        files = [f for f in _os.listdir(src) if f.endswith(".ufdb.txt")]
        if not files:
            return None
        default_dir = _os.path.join(_os.path.expanduser("~"), ".config", "miopen")
        if _os.path.isdir(default_dir) and any(f.endswith((".ufdb.txt", ".udb.txt")) for f in _os.listdir(default_dir)):
            return None                                   # the user's own tuning (find-db OR perf-db) wins: MIOPEN_USER_DB_PATH moves both
        dst = _os.path.join(_os.path.expanduser("~"), ".cache", "kccotgan_amd", "miopen_db")
        _os.makedirs(dst, exist_ok=True)
        for f in files:
            if not _os.path.exists(_os.path.join(dst, f)):
                tmp = _os.path.join(dst, ".%s.%d.tmp" % (f, _os.getpid()))     # one process per GPU imports this at once:
                shutil.copyfile(_os.path.join(src, f), tmp)                     # never let a rank see a half-written file
                _os.replace(tmp, _os.path.join(dst, f))
        _os.environ["MIOPEN_USER_DB_PATH"] = dst
        return dst
    except OSError:
        return None


MIOPEN_FIND_DB_DIR = None if _gpu_was_live else _install_miopen_find_db()

from . import _lib            # noqa: F401,E402  (loads the HIP library)
from . import gan_utils       # noqa: F401,E402
from . import data_utils      # noqa: F401,E402

__version__ = "0.3.1"
