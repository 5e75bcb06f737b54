"""jdet_amd.ops -- same module names as python/jdet/ops in the reference (hot-path subset).

As in the reference's ops/__init__.py, the package namespace exports the two IoU *functions*
(`from jdet.ops import box_iou_rotated` yields the function, iou_calculator.py relies on it);
everything else is reached as a submodule (`from jdet.ops import roi_align_rotated`,
`from jdet.ops.nms_rotated import multiclass_nms_rotated`, ...).
"""
from . import (dcn_v1, nms_rotated, orn, riroi_align, roi_align, roi_align_rotated,  # noqa: F401
               roi_align_rotated_v1)
from .box_iou_rotated import box_iou_rotated  # noqa: F401
from .box_iou_rotated_v1 import box_iou_rotated_v1  # noqa: F401
