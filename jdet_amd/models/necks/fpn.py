"""Feature pyramid network behind the reference's constructor and parameter names (python/jdet/models/necks/fpn.py:L60-201;
`lateral_convs.N.*` / `fpn_convs.N.*` are the checkpoint keys).  What it computes: a 1x1 lateral per used backbone level;
top-down, every level receives the nearest-upsampled sum of the levels above it; a 3x3 output conv per level; levels beyond
the backbone's by stride-2 3x3 convs fed from the last input / lateral / output (`add_extra_convs`) or, without them, by
stride-2 subsampling of the last output.

Own structure: three stages (laterals, top-down merge, levels beyond the backbone's), each its own method; the top-down
step is ONE fused launch per level on channels-last device maps (`ops/upsample_add.py`: (lateral + upsample(top)) / div,
instead of an upsampled copy plus an add, and one launch instead of two in backward); other layouts / resampling modes
take the framework's interpolate."""
import torch.nn.functional as F
from torch import nn

from jdet_amd.models.utils.modules import ConvModule
from jdet_amd.models.utils.weight_init import xavier_init
from jdet_amd.ops import upsample_add as UA
from jdet_amd.utils.registry import NECKS

_EXTRA_SOURCES = ("on_input", "on_lateral", "on_output")


@NECKS.register_module()
class FPN(nn.Module):
    def __init__(self, in_channels, out_channels, num_outs, start_level=0, end_level=-1, add_extra_convs=False,
                 extra_convs_on_inputs=True, relu_before_extra_convs=False, no_norm_on_lateral=False, conv_cfg=None,
                 norm_cfg=None, act_cfg=None, upsample_cfg=dict(mode="nearest"),
                 init_cfg=dict(type="Xavier", layer="Conv2d", distribution="uniform"), upsample_div_factor=1):
        super().__init__()
        if not isinstance(in_channels, list):
            raise AssertionError("in_channels must be a list")
        if not isinstance(add_extra_convs, (str, bool)) or (isinstance(add_extra_convs, str)
                                                            and add_extra_convs not in _EXTRA_SOURCES):
            raise AssertionError("add_extra_convs: bool or one of %s" % (_EXTRA_SOURCES,))
        stop = len(in_channels) if end_level == -1 else end_level
        n_backbone = stop - start_level                       # pyramid levels that have a backbone map under them
        if end_level == -1:
            assert num_outs >= n_backbone
        else:
            assert end_level <= len(in_channels) and num_outs == n_backbone      # no extra level beyond an explicit end
        # attributes the reference exposes (heads / tools read some of them)
        self.in_channels, self.out_channels = in_channels, out_channels
        self.num_ins, self.num_outs = len(in_channels), num_outs
        self.start_level, self.end_level, self.backbone_end_level = start_level, end_level, stop
        self.add_extra_convs = add_extra_convs
        self.relu_before_extra_convs = relu_before_extra_convs
        self.no_norm_on_lateral = no_norm_on_lateral
        self.upsample_cfg = dict(upsample_cfg)
        self.upsample_div_factor = upsample_div_factor

        common = dict(conv_cfg=conv_cfg, norm_cfg=norm_cfg, act_cfg=act_cfg)
        lateral_norm = dict(common, norm_cfg=None) if no_norm_on_lateral else common
        self.lateral_convs = nn.ModuleList(ConvModule(c, out_channels, 1, **lateral_norm)
                                           for c in in_channels[start_level:stop])
        self.fpn_convs = nn.ModuleList(ConvModule(out_channels, out_channels, 3, padding=1, **common)
                                       for _ in range(n_backbone))
        # levels above the backbone's: fpn_convs[n_backbone + k]; only the first may read a raw backbone map
        self._n_backbone = n_backbone
        self._n_extra_convs = (num_outs - n_backbone) if add_extra_convs else 0
        for k in range(self._n_extra_convs):
            first_from_input = k == 0 and add_extra_convs == "on_input"
            self.fpn_convs.append(ConvModule(in_channels[stop - 1] if first_from_input else out_channels, out_channels, 3,
                                             stride=2, padding=1, **common))
        self.init_weights()

    def init_weights(self):
        for m in self.modules():
            if isinstance(m, nn.Conv2d):
                xavier_init(m, distribution="uniform")

    # -- the three stages of the pyramid --------------------------------------------------------------------------
    def _merge_down(self, fine, coarse):
        """one top-down step: (fine + coarse brought to fine's resolution) / upsample_div_factor"""
        div = self.upsample_div_factor
        cfg = self.upsample_cfg
        if cfg.get("mode", "nearest") == "nearest" and "scale_factor" not in cfg and UA.fusable(fine, coarse):
            return UA.upsample_add(fine, coarse, div)
        target = {} if "scale_factor" in cfg else {"size": fine.shape[2:]}
        merged = fine + F.interpolate(coarse, **target, **cfg)
        return merged if div == 1 else merged / div

    def _top_down(self, laterals):
        merged = [None] * len(laterals)
        merged[-1] = laterals[-1]
        for lvl in range(len(laterals) - 2, -1, -1):          # coarsest to finest: each level sees the merged one above
            merged[lvl] = self._merge_down(laterals[lvl], merged[lvl + 1])
        return merged

    def _extend(self, inputs, merged, outs):
        """levels beyond the backbone's, appended to outs"""
        missing = self.num_outs - len(outs)
        if missing <= 0:
            return
        if not self.add_extra_convs:
            for _ in range(missing):
                outs.append(F.max_pool2d(outs[-1], 1, stride=2))      # kernel 1, stride 2: plain subsampling
            return
        source = {"on_input": inputs[self.backbone_end_level - 1], "on_lateral": merged[-1], "on_output": outs[-1]}
        if self.add_extra_convs not in source:
            raise NotImplementedError(self.add_extra_convs)
        feed = source[self.add_extra_convs]
        for k in range(missing):
            if k > 0:
                feed = F.relu(outs[-1]) if self.relu_before_extra_convs else outs[-1]
            outs.append(self.fpn_convs[self._n_backbone + k](feed))

    def forward(self, inputs):
        assert len(inputs) == len(self.in_channels)
        laterals = [conv(inputs[self.start_level + k]) for k, conv in enumerate(self.lateral_convs)]
        merged = self._top_down(laterals)
        outs = [self.fpn_convs[k](m) for k, m in enumerate(merged)]
        self._extend(inputs, merged, outs)
        return tuple(outs)

    execute = forward
