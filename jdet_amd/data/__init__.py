from .datasets import (CustomDataset, DeviceFeeder, DOTADataset, ImageDataset, collate_batch,  # noqa: F401
                       targets_to_device)
from .transforms import (Compose, Normalize, Pad, RandomFlip, Resize, RotatedRandomFlip,  # noqa: F401
                         RotatedResize)
