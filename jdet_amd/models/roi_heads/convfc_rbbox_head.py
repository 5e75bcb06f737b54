"""Conv/FC box heads of RoI-Transformer.  Contract of python/jdet/models/roi_heads/convfc_rbbox_head.py:
`ConvFCBBoxHeadRbbox` L7-170 (shared convs -> shared fcs -> cls / reg branches; constructor arguments, module names
`shared_convs / shared_fcs / cls_convs / cls_fcs / reg_convs / reg_fcs / fc_cls / fc_reg`), `SharedFCBBoxHeadRbbox`
L173-189 (two shared 1024-d FCs; the GEMMs are hipBLASLt through nn.Linear).

The pooled features arrive as (R, C, 7, 7) with channels-last strides (the RoIAlign kernels store a bin's channel
vector contiguously).  The layer that first sees them flattened is a `RoIFeatureLinear` (weight columns in
(ph, pw, c) order, converted from / to the reference's order in the state-dict hooks): no layout copy forward, and
the gradient that comes back is channels-last, which the RoIAlign backward gathers from without a transpose pass.
"""
from torch import nn

from jdet_amd.ops.linear import Linear
from jdet_amd.utils.registry import HEADS

from .rbbox_head import BBoxHeadRbbox


@HEADS.register_module()
class ConvFCBBoxHeadRbbox(BBoxHeadRbbox):
    def __init__(self, num_shared_convs=0, num_shared_fcs=0, num_cls_convs=0, num_cls_fcs=0, num_reg_convs=0,
                 num_reg_fcs=0, conv_out_channels=256, fc_out_channels=1024, conv_cfg=None, norm_cfg=None, *args,
                 **kwargs):
        super().__init__(*args, **kwargs)
        assert num_shared_convs + num_shared_fcs + num_cls_convs + num_cls_fcs + num_reg_convs + num_reg_fcs > 0
        if num_cls_convs > 0 or num_reg_convs > 0:
            assert num_shared_fcs == 0
        if not self.with_cls:
            assert num_cls_convs == 0 and num_cls_fcs == 0
        if not self.with_reg:
            assert num_reg_convs == 0 and num_reg_fcs == 0
        self.num_shared_convs, self.num_shared_fcs = num_shared_convs, num_shared_fcs
        self.num_cls_convs, self.num_cls_fcs = num_cls_convs, num_cls_fcs
        self.num_reg_convs, self.num_reg_fcs = num_reg_convs, num_reg_fcs
        self.conv_out_channels, self.fc_out_channels = conv_out_channels, fc_out_channels
        self.conv_cfg, self.norm_cfg = conv_cfg, norm_cfg
        # every branch: [convs on the 4-D features] -> [fcs]; `flat` says whether its input is already a matrix
        self.shared_convs, self.shared_fcs, width, flat = self._branch(num_shared_convs, num_shared_fcs,
                                                                       self.in_channels, False)
        self.shared_out_channels = width
        self.cls_convs, self.cls_fcs, cls_width, cls_flat = self._branch(num_cls_convs, num_cls_fcs, width, flat)
        self.reg_convs, self.reg_fcs, reg_width, reg_flat = self._branch(num_reg_convs, num_reg_fcs, width, flat)
        self.relu = nn.ReLU()
        if self.with_cls:
            self.fc_cls = self._output_layer(cls_width, cls_flat, self.num_classes)
        if self.with_reg:
            self.fc_reg = self._output_layer(reg_width, reg_flat, 5 if self.reg_class_agnostic else 5 * self.num_classes)

    def _branch(self, n_convs, n_fcs, width, flat):
        convs, fcs = nn.ModuleList(), nn.ModuleList()
        for i in range(n_convs):
            assert not flat, "convolutions cannot follow a fully connected layer"
            convs.append(nn.Conv2d(width, self.conv_out_channels, 3, padding=1))
            width = self.conv_out_channels
        for i in range(n_fcs):
            fcs.append(Linear(width, self.fc_out_channels) if flat or self.with_avg_pool else
                       self._roi_linear(width, self.fc_out_channels))
            width, flat = self.fc_out_channels, True
        return convs, fcs, width, flat

    def _output_layer(self, width, flat, out_features):
        return Linear(width, out_features) if flat else self._roi_linear(width, out_features)

    def init_weights(self):
        super().init_weights()
        for module_list in [self.shared_fcs, self.cls_fcs, self.reg_fcs]:
            for m in module_list.modules():
                if isinstance(m, nn.Linear):
                    nn.init.xavier_uniform_(m.weight)
                    nn.init.constant_(m.bias, 0)

    def _run(self, x, convs, fcs):
        for conv in convs:
            x = conv(x)
        if len(fcs) and x.dim() == 4 and self.with_avg_pool:
            x = self.avg_pool(x).reshape(x.size(0), -1)
        for fc in fcs:
            x = self.relu(fc(x))
        return x

    def forward(self, x):
        x = self._run(x, self.shared_convs, self.shared_fcs)
        outs = []
        for on, convs, fcs, last in ((self.with_cls, self.cls_convs, self.cls_fcs, getattr(self, "fc_cls", None)),
                                     (self.with_reg, self.reg_convs, self.reg_fcs, getattr(self, "fc_reg", None))):
            if not on:
                outs.append(None)
                continue
            y = self._run(x, convs, fcs)
            if y.dim() == 4 and self.with_avg_pool:
                y = self.avg_pool(y).reshape(y.size(0), -1)
            outs.append(last(y))
        return tuple(outs)

    execute = forward


@HEADS.register_module()
class SharedFCBBoxHeadRbbox(ConvFCBBoxHeadRbbox):
    def __init__(self, num_fcs=2, fc_out_channels=1024, *args, **kwargs):
        assert num_fcs >= 1
        super().__init__(num_shared_convs=0, num_shared_fcs=num_fcs, num_cls_convs=0, num_cls_fcs=0, num_reg_convs=0,
                         num_reg_fcs=0, fc_out_channels=fc_out_channels, *args, **kwargs)
