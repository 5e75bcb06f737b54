"""Rotation-invariant RoIAlign (ReDet).  Mirrors python/jdet/ops/riroi_align.py:L383-492: input has
C*nOrientation planes; the orientation axis is rolled by floor(theta*nO/2pi) and linearly blended
between the two neighbouring orientation planes."""
from torch import nn

from ._roi_common import V_RI, RoIAlignFunction, _pair

__all__ = ["RiRoIAlign", "riroi_align"]


def riroi_align(features, rois, out_size, spatial_scale, sample_num=0, nOrientation=8):
    if isinstance(out_size, int):
        out_size = (out_size, out_size)
    elif isinstance(out_size, tuple):
        assert len(out_size) == 2 and all(isinstance(v, int) for v in out_size)
    else:
        raise TypeError('"out_size" must be an integer or tuple of integers')  # riroi_align.py:L393-395
    assert features.shape[1] % nOrientation == 0
    return RoIAlignFunction.apply(features, rois, V_RI, out_size, spatial_scale, sample_num, nOrientation)


class RiRoIAlign(nn.Module):
    def __init__(self, output_size, spatial_scale, sample_num=0, nOrientation=8):
        super().__init__()
        self.output_size = _pair(output_size)
        self.spatial_scale = float(spatial_scale)
        self.sample_num = int(sample_num)
        self.nOrientation = int(nOrientation)

    def forward(self, features, rois):
        return riroi_align(features, rois, self.output_size, self.spatial_scale, self.sample_num,
                           self.nOrientation)

    execute = forward

    def __repr__(self):
        return (self.__class__.__name__ + "(output_size=" + str(self.output_size) + ", spatial_scale="
                + str(self.spatial_scale) + ", sample_num=" + str(self.sample_num) + ", nOrientation="
                + str(self.nOrientation) + ")")
