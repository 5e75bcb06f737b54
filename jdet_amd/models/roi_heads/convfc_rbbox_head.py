"""Conv/FC box heads of RoI-Transformer.  Mirrors python/jdet/models/roi_heads/convfc_rbbox_head.py:
`ConvFCBBoxHeadRbbox` L7-170 (shared convs -> shared fcs -> cls / reg branches), `SharedFCBBoxHeadRbbox`
L173-189 (two shared 1024-d FCs; the GEMMs are hipBLASLt through nn.Linear)."""
from torch import nn

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
        self.num_shared_convs = num_shared_convs
        self.num_shared_fcs = num_shared_fcs
        self.num_cls_convs = num_cls_convs
        self.num_cls_fcs = num_cls_fcs
        self.num_reg_convs = num_reg_convs
        self.num_reg_fcs = num_reg_fcs
        self.conv_out_channels = conv_out_channels
        self.fc_out_channels = fc_out_channels
        self.conv_cfg = conv_cfg
        self.norm_cfg = norm_cfg
        self.shared_convs, self.shared_fcs, last_layer_dim = self._add_conv_fc_branch(
            self.num_shared_convs, self.num_shared_fcs, self.in_channels, True)
        self.shared_out_channels = last_layer_dim
        self.cls_convs, self.cls_fcs, self.cls_last_dim = self._add_conv_fc_branch(
            self.num_cls_convs, self.num_cls_fcs, self.shared_out_channels)
        self.reg_convs, self.reg_fcs, self.reg_last_dim = self._add_conv_fc_branch(
            self.num_reg_convs, self.num_reg_fcs, self.shared_out_channels)
        if self.num_shared_fcs == 0 and not self.with_avg_pool:
            if self.num_cls_fcs == 0:
                self.cls_last_dim *= self.roi_feat_size * self.roi_feat_size
            if self.num_reg_fcs == 0:
                self.reg_last_dim *= self.roi_feat_size * self.roi_feat_size
        self.relu = nn.ReLU()
        if self.with_cls:
            self.fc_cls = nn.Linear(self.cls_last_dim, self.num_classes)
        if self.with_reg:
            self.fc_reg = nn.Linear(self.reg_last_dim, 5 if self.reg_class_agnostic else 5 * self.num_classes)

    def _add_conv_fc_branch(self, num_branch_convs, num_branch_fcs, in_channels, is_shared=False):
        last_layer_dim = in_channels
        branch_convs = nn.ModuleList()
        for i in range(num_branch_convs):
            branch_convs.append(nn.Conv2d(last_layer_dim if i == 0 else self.conv_out_channels,
                                          self.conv_out_channels, 3, padding=1))
        if num_branch_convs > 0:
            last_layer_dim = self.conv_out_channels
        branch_fcs = nn.ModuleList()
        if num_branch_fcs > 0:
            if (is_shared or self.num_shared_fcs == 0) and not self.with_avg_pool:
                if isinstance(self.roi_feat_size, int):
                    last_layer_dim *= self.roi_feat_size * self.roi_feat_size
                else:
                    assert len(self.roi_feat_size) == 2
                    last_layer_dim *= self.roi_feat_size[0] * self.roi_feat_size[1]
            for i in range(num_branch_fcs):
                branch_fcs.append(nn.Linear(last_layer_dim if i == 0 else self.fc_out_channels, self.fc_out_channels))
            last_layer_dim = self.fc_out_channels
        return branch_convs, branch_fcs, last_layer_dim

    def init_weights(self):
        super().init_weights()
        for module_list in [self.shared_fcs, self.cls_fcs, self.reg_fcs]:
            for m in module_list.modules():
                if isinstance(m, nn.Linear):
                    nn.init.xavier_uniform_(m.weight)
                    nn.init.constant_(m.bias, 0)

    def _branch(self, x, convs, fcs):
        for conv in convs:
            x = conv(x)
        if x.dim() > 2:
            if self.with_avg_pool:
                x = self.avg_pool(x)
            x = x.reshape(x.size(0), -1)
        for fc in fcs:
            x = self.relu(fc(x))
        return x

    def forward(self, x):
        for conv in self.shared_convs:
            x = conv(x)
        if self.num_shared_fcs > 0:
            if self.with_avg_pool:
                x = self.avg_pool(x)
            x = x.reshape(x.size(0), -1)
            for fc in self.shared_fcs:
                x = self.relu(fc(x))
        x_cls = self._branch(x, self.cls_convs, self.cls_fcs)
        x_reg = self._branch(x, self.reg_convs, self.reg_fcs)
        return (self.fc_cls(x_cls) if self.with_cls else None), (self.fc_reg(x_reg) if self.with_reg else None)

    execute = forward


@HEADS.register_module()
class SharedFCBBoxHeadRbbox(ConvFCBBoxHeadRbbox):
    def __init__(self, num_fcs=2, fc_out_channels=1024, *args, **kwargs):
        assert num_fcs >= 1
        super().__init__(num_shared_convs=0, num_shared_fcs=num_fcs, num_cls_convs=0, num_cls_fcs=0, num_reg_convs=0,
                         num_reg_fcs=0, fc_out_channels=fc_out_channels, *args, **kwargs)
