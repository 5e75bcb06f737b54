"""jdet_amd.models -- hot-path subset of python/jdet/models (same registry type strings)."""
from . import backbones, boxes, losses, necks, networks, roi_heads  # noqa: F401
