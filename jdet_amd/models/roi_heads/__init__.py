from .oriented_head import OrientedHead  # noqa: F401
from .oriented_rpn_head import OrientedRPNHead  # noqa: F401
from .s2anet_head import AlignConv, S2ANetHead, bbox_decode  # noqa: F401
from .rotated_retina_head import RotatedRetinaHead  # noqa: F401
from .rbbox_head import BBoxHeadRbbox  # noqa: F401
from .convfc_rbbox_head import ConvFCBBoxHeadRbbox, SharedFCBBoxHeadRbbox  # noqa: F401
from .fasterrcnn_head import AnchorHead, FasterrcnnHead  # noqa: F401
