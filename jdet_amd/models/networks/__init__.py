from .rcnn import RCNN, OrientedRCNN  # noqa: F401
from .s2anet import S2ANet  # noqa: F401
from .rotated_retinanet import RotatedRetinaNet  # noqa: F401
from .roi_transformer import RoITransformer  # noqa: F401
