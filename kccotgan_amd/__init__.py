"""kccotgan_amd -- MI355X-native causal-OT (Sinkhorn) + kernel-smoothing loss path of KCCOT-GAN.

Importing the package loads ``csrc/libkccot.so`` (HIP, gfx950) and fails loudly if it is missing.
"""
from . import _lib            # noqa: F401  (loads the HIP library)
from . import gan_utils       # noqa: F401
from . import data_utils      # noqa: F401

__version__ = "0.1.0"
