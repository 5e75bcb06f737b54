"""jdet_amd.models -- hot-path subset of python/jdet/models (same registry type strings)."""
from . import backbones, boxes, losses, necks, roi_extractors  # noqa: F401  isort:skip
from . import networks, roi_heads  # noqa: F401
