from .oriented_head import OrientedHead  # noqa: F401
from .oriented_rpn_head import OrientedRPNHead  # noqa: F401
from .s2anet_head import AlignConv, S2ANetHead, bbox_decode  # noqa: F401
from .rotated_retina_head import RotatedRetinaHead  # noqa: F401
