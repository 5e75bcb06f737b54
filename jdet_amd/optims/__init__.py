from .lr_scheduler import StepLR, WarmUpLR  # noqa: F401
from .optimizer import SGD  # noqa: F401
