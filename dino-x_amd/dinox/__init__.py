"""dinox -- MI355X-native engine for the DINO-X hot path (student/teacher ViT + DINO/Gram losses).

Importing this package loads ``libdinox_hip.so`` (hand-written gfx950 HIP kernels behind a C ABI,
``include/dinox.h``).  A missing library is an ImportError: there is no fallback compute path.
"""
from . import _lib  # noqa: F401  (raises if the kernel library is absent)
from ._lib import LIB_PATH  # noqa: F401

__version__ = "0.1.0"
