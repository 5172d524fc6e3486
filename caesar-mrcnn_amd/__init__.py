"""MI355X-native Mask R-CNN hot path (drop-in for the ``mrcnn.model.MaskRCNN`` boundary of
SKA-INAF/caesar-mrcnn).  Import name: ``caesar_mrcnn_amd`` (see the shim at the repo root).

Only light modules are imported eagerly; the kernel library is loaded on first use (``_hip.lib()``)
and its absence is an error, never a fallback.
"""
__version__ = "0.1.0"

from .config import Config, SDetectorConfig  # noqa: F401
