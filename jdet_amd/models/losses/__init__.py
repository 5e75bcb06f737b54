from .focal_loss import FocalLoss, sigmoid_focal_loss  # noqa: F401
from .smooth_l1_loss import L1Loss, SmoothL1Loss, smooth_l1_loss  # noqa: F401
from .cross_entropy_loss import CrossEntropyLoss, CrossEntropyLossForRcnn  # noqa: F401
