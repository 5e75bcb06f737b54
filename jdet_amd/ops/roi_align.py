"""Horizontal RoIAlign, rois [b,x1,y1,x2,y2]; version=1 is the legacy "+1 pixel" rule.
Mirrors python/jdet/ops/roi_align.py:L209-290."""
from torch import nn

from ._roi_common import V_HBB0, V_HBB1, RoIAlignFunction, _pair

__all__ = ["ROIAlign", "roi_align"]


def roi_align(input, rois, output_size, spatial_scale, sampling_ratio, version=0):
    assert version in (0, 1)
    return RoIAlignFunction.apply(input, rois, V_HBB1 if version == 1 else V_HBB0, _pair(output_size),
                                  spatial_scale, sampling_ratio, 1)


class ROIAlign(nn.Module):
    def __init__(self, output_size, spatial_scale, sampling_ratio=0, version=0):
        super().__init__()
        self.output_size = _pair(output_size)
        self.spatial_scale = spatial_scale
        self.sampling_ratio = sampling_ratio
        self.version = version

    def forward(self, input, rois):
        return roi_align(input, rois, self.output_size, self.spatial_scale, self.sampling_ratio, self.version)

    execute = forward

    def __repr__(self):
        return (self.__class__.__name__ + "(output_size=" + str(self.output_size) + ", spatial_scale="
                + str(self.spatial_scale) + ", sampling_ratio=" + str(self.sampling_ratio) + ", version="
                + str(self.version) + ")")
