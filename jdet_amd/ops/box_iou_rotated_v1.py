"""`box_iou_rotated_v1` lives in its own module in the reference (python/jdet/ops/box_iou_rotated_v1.py)."""
from .box_iou_rotated import box_iou_rotated_v1  # noqa: F401

__all__ = ["box_iou_rotated_v1"]
