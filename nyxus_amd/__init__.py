"""nyxus_amd -- MI355X-native per-ROI feature reducer behind the Nyxus API.

Only the hot path of PolusAI/nyxus is implemented here:
``reduce_trivial_rois`` for the intensity / GLCM / GLRLM / GLSZM / NGTDM /
Gabor / Zernike families, as hand-written HIP kernels for gfx950 behind the
C ABI of ``include/nyxhip.h``.
"""
from . import _abi  # noqa: F401
from .nyxus import Nyxus  # noqa: F401

__version__ = "0.1.0"
