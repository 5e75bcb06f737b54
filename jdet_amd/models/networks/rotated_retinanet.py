"""RetinaNet-OBB detector wrapper.  Mirrors python/jdet/models/networks/rotated_retinanet.py:L8-36."""
from jdet_amd.utils.registry import MODELS

from .s2anet import S2ANet


@MODELS.register_module()
class RotatedRetinaNet(S2ANet):
    """backbone -> neck -> RotatedRetinaHead (same wrapper shape as S2ANet)"""
