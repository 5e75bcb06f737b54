"""Rotated RoIAlign, Oriented R-CNN dialect (-0.5 centre shift, opposite rotation sense, `<` clamps,
forward count >= 1).  Mirrors python/jdet/ops/roi_align_rotated_v1.py:L300-372."""
from torch import nn

from ._roi_common import V_ROT_V1, RoIAlignFunction, _pair

__all__ = ["ROIAlignRotated_v1", "roi_align"]


def roi_align(input, rois, output_size, spatial_scale, sampling_ratio):
    assert rois.shape[1] == 6
    return RoIAlignFunction.apply(input, rois, V_ROT_V1, _pair(output_size), spatial_scale, sampling_ratio, 1)


class ROIAlignRotated_v1(nn.Module):
    def __init__(self, output_size, spatial_scale, sampling_ratio=0):
        super().__init__()
        self.output_size = _pair(output_size)
        self.spatial_scale = spatial_scale
        self.sampling_ratio = sampling_ratio

    def forward(self, input, rois):
        return roi_align(input, rois, self.output_size, self.spatial_scale, self.sampling_ratio)

    execute = forward

    def __repr__(self):
        return (self.__class__.__name__ + "(output_size=" + str(self.output_size) + ", spatial_scale="
                + str(self.spatial_scale) + ", sampling_ratio=" + str(self.sampling_ratio) + ")")
