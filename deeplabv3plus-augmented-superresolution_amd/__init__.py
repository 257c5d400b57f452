"""asr_amd -- MI355X-native Augmented Super-Resolution segmentation hot path.

Host side: Python mirroring the reference's call surface (``model.DeeplabV3Plus``,
``utils``, ``superresolution_scripts.{augmentation_utils,superresolution,optimizer,superres_utils}``).
Device side: hand-written gfx950 HIP kernels behind the C ABI of ``include/asr_hip.h``
(``libasr_hip.so``).  There is no CPU fallback; ``oracle/`` (test infrastructure) is never
imported from here.
"""
from . import _lib  # noqa: F401

__all__ = ["_lib", "ops", "transforms", "weights", "engine", "model", "utils", "superresolution_scripts",
           "distributed"]
__version__ = "0.1.0"
