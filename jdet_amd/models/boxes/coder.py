"""BBox coders of the named configs.  Mirrors python/jdet/models/boxes/coder.py:
`DeltaXYWHABBoxCoder` L76-141."""
from jdet_amd.utils.registry import BOXES

from .box_ops import bbox2delta_rotated, delta2bbox_rotated


@BOXES.register_module()
class DeltaXYWHABBoxCoder:
    """encodes (x,y,w,h,a) into (dx,dy,dw,dh,da) relative to a base box and back."""

    def __init__(self, target_means=(0., 0., 0., 0., 0.), target_stds=(1., 1., 1., 1., 1.), clip_border=True):
        self.means = target_means
        self.stds = target_stds
        self.clip_border = clip_border

    def encode(self, bboxes, gt_bboxes):
        assert bboxes.size(0) == gt_bboxes.size(0)
        assert bboxes.size(-1) == gt_bboxes.size(-1) == 5
        return bbox2delta_rotated(bboxes, gt_bboxes, self.means, self.stds)

    def decode(self, bboxes, pred_bboxes, max_shape=None, wh_ratio_clip=16 / 1000):
        assert pred_bboxes.size(0) == bboxes.size(0)
        return delta2bbox_rotated(bboxes, pred_bboxes, self.means, self.stds, max_shape, wh_ratio_clip,
                                  self.clip_border)
