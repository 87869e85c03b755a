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

from . import _lib            # noqa: F401,E402  (loads the HIP library)
from . import gan_utils       # noqa: F401,E402
from . import data_utils      # noqa: F401,E402

__version__ = "0.2.0"
