"""RoI feature extractors: map each RoI to an FPN level by floor(log2(scale / 56 + 1e-6)) and pool it
there.  Mirror python/jdet/models/roi_extractors/: `OrientedSingleRoIExtractor`
(oriented_single_level.py:L8-114, optional (h,w) enlargement, layer looked up on
jdet.ops.roi_align_rotated_v1), `RboxSingleRoIExtractor` (rbox_single_level.py, layer on
jdet.ops.roi_align_rotated incl. the re-exported RiRoIAlign), `SingleRoIExtractor` (single_level.py,
horizontal, +1 scale rule).

The RoI layer class is still found by `getattr(<ops module>, cfg.type)`; the per-level routing runs as
one masked launch per level on a shared output (ops/_roi_common.MultiLevelRoIAlignFunction)."""
import torch
from torch import nn

from jdet_amd.ops import riroi_align as _ri_mod
from jdet_amd.ops import roi_align, roi_align_rotated, roi_align_rotated_v1
from jdet_amd.ops._roi_common import V_HBB0, V_HBB1, V_RI, V_ROT, V_ROT_V1, _pair, multi_level_roi_align
from jdet_amd.utils.registry import ROI_EXTRACTORS


def _variant_of(layer):
    if isinstance(layer, roi_align_rotated_v1.ROIAlignRotated_v1):
        return V_ROT_V1, layer.sampling_ratio, 1
    if isinstance(layer, _ri_mod.RiRoIAlign):
        return V_RI, layer.sample_num, layer.nOrientation
    if isinstance(layer, roi_align_rotated.ROIAlignRotated):
        return V_ROT, layer.sampling_ratio, 1
    if isinstance(layer, roi_align.ROIAlign):
        return (V_HBB1 if layer.version == 1 else V_HBB0), layer.sampling_ratio, 1
    raise TypeError(type(layer))


class _SingleLevelBase(nn.Module):
    ops_module = None

    def __init__(self, roi_layer, out_channels, featmap_strides, finest_scale=56):
        super().__init__()
        self.roi_layers = self.build_roi_layers(roi_layer, featmap_strides)
        self.out_channels = out_channels
        self.featmap_strides = featmap_strides
        self.finest_scale = finest_scale

    @property
    def num_inputs(self):
        return len(self.featmap_strides)

    def init_weights(self):
        pass

    def build_roi_layers(self, layer_cfg, featmap_strides):
        cfg = dict(layer_cfg)
        layer_type = cfg.pop("type")
        assert hasattr(self.ops_module, layer_type)
        layer_cls = getattr(self.ops_module, layer_type)
        return nn.ModuleList([layer_cls(spatial_scale=1 / s, **cfg) for s in featmap_strides])

    def _scale(self, rois):
        return torch.sqrt(rois[:, 3] * rois[:, 4])

    def map_roi_levels(self, rois, num_levels):
        target_lvls = torch.floor(torch.log2(self._scale(rois) / self.finest_scale + 1e-6))
        return target_lvls.clamp(min=0, max=num_levels - 1).long()

    def _pool(self, feats, rois, target_lvls):
        layer = self.roi_layers[0]
        variant, sample_num, n_orient = _variant_of(layer)
        scales = [l.spatial_scale for l in self.roi_layers[:len(feats)]]
        return multi_level_roi_align(variant, list(feats), rois, target_lvls, scales, layer.output_size, sample_num,
                                     n_orient)

    def forward(self, feats, rois):
        if len(feats) == 1:
            return self.roi_layers[0](feats[0], rois)
        return self._pool(feats, rois, self.map_roi_levels(rois, len(feats)))

    execute = forward


@ROI_EXTRACTORS.register_module()
class SingleRoIExtractor(_SingleLevelBase):
    ops_module = roi_align

    def _scale(self, rois):
        return torch.sqrt((rois[:, 3] - rois[:, 1] + 1) * (rois[:, 4] - rois[:, 2] + 1))


@ROI_EXTRACTORS.register_module()
class RboxSingleRoIExtractor(_SingleLevelBase):
    """rbox_single_level.py:L8-34; the enlargement factors are applied by the caller
    (roi_transformer.py:L124-125), the extractor only carries them."""
    ops_module = roi_align_rotated

    def __init__(self, roi_layer, out_channels, featmap_strides, finest_scale=56, w_enlarge=1.2, h_enlarge=1.4):
        super().__init__(roi_layer, out_channels, featmap_strides, finest_scale)
        self.w_enlarge = w_enlarge
        self.h_enlarge = h_enlarge


@ROI_EXTRACTORS.register_module()
class OrientedSingleRoIExtractor(_SingleLevelBase):
    ops_module = roi_align_rotated_v1

    def __init__(self, roi_layer, out_channels, featmap_strides, extend_factor=(1., 1.), finest_scale=56):
        super().__init__(roi_layer, out_channels, featmap_strides, finest_scale)
        self.extend_factor = extend_factor

    def roi_rescale(self, rois, scale_factor):
        if scale_factor is None:
            return rois
        h_scale_factor, w_scale_factor = _pair(scale_factor)
        new_rois = rois.clone()
        new_rois[:, 3] = w_scale_factor * new_rois[:, 3]
        new_rois[:, 4] = h_scale_factor * new_rois[:, 4]
        return new_rois

    def forward(self, feats, rois, roi_scale_factor=None):
        if len(feats) == 1:
            return self.roi_layers[0](feats[0], rois)
        rois = self.roi_rescale(rois, self.extend_factor)
        target_lvls = self.map_roi_levels(rois, len(feats))     # levels of the ENLARGED RoIs (L101-102)
        rois = self.roi_rescale(rois, roi_scale_factor)
        return self._pool(feats, rois, target_lvls)

    execute = forward
