"""Rotated RoIAlign (RoI-Transformer / Gliding dialect).  Mirrors python/jdet/ops/roi_align_rotated.py:
`ROIAlignRotated` (L312-330), `roi_align` (L310) and the `RiRoIAlign` re-export (L4) that
RboxSingleRoIExtractor relies on (`getattr(roi_align_rotated, 'RiRoIAlign')`, rbox_single_level.py:L44-51).
"""
from torch import nn

from ._roi_common import V_ROT, RoIAlignFunction, _pair
from .riroi_align import RiRoIAlign  # noqa: F401  (same re-export as the reference)

__all__ = ["ROIAlignRotated", "RiRoIAlign", "roi_align"]


def roi_align(input, rois, output_size, spatial_scale, sampling_ratio):
    assert rois.shape[1] == 6  # roi_align_rotated.py:L263
    return RoIAlignFunction.apply(input, rois, V_ROT, _pair(output_size), spatial_scale, sampling_ratio, 1)


class ROIAlignRotated(nn.Module):
    def __init__(self, output_size, spatial_scale, sampling_ratio=0):
        super().__init__()
        self.output_size = _pair(output_size)
        self.spatial_scale = spatial_scale
        self.sampling_ratio = sampling_ratio

    def forward(self, input, rois):
        return roi_align(input, rois, self.output_size, self.spatial_scale, self.sampling_ratio)

    execute = forward  # Jittor spelling

    def __repr__(self):
        return (self.__class__.__name__ + "(output_size=" + str(self.output_size) + ", spatial_scale="
                + str(self.spatial_scale) + ", sampling_ratio=" + str(self.sampling_ratio) + ")")
