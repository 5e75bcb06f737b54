"""ConvModule = conv (+bias iff no norm) -> [norm] -> [act].  Mirrors python/jdet/models/utils/modules.py
(L43-175): parameter names `.conv.weight/.conv.bias/.bn.*`, default act ReLU, `act_cfg=None` disables
the activation (FPN)."""
import torch
from torch import nn

from jdet_amd.ops import conv_igemm
from jdet_amd.utils.registry import BRICKS, build_from_cfg

for _n, _m in (("Conv2d", nn.Conv2d), ("Conv", nn.Conv2d), ("BN", nn.BatchNorm2d), ("BN2d", nn.BatchNorm2d),
               ("GN", nn.GroupNorm), ("ReLU", nn.ReLU), ("LeakyReLU", nn.LeakyReLU), ("Sigmoid", nn.Sigmoid),
               ("Tanh", nn.Tanh), ("GELU", nn.GELU)):
    if _n not in BRICKS:
        BRICKS.register_module(_n, module=_m)


class ConvModule(nn.Module):
    def __init__(self, in_channels, out_channels, kernel_size, stride=1, padding=0, dilation=1, groups=1,
                 bias="auto", conv_cfg=None, norm_cfg=None, act_cfg=dict(type="ReLU"), order=("conv", "norm", "act")):
        super().__init__()
        assert conv_cfg is None or isinstance(conv_cfg, dict)
        assert norm_cfg is None or isinstance(norm_cfg, dict)
        assert act_cfg is None or isinstance(act_cfg, dict)
        assert isinstance(order, tuple) and set(order) == {"conv", "norm", "act"}
        self.order = order
        self.with_norm = norm_cfg is not None
        self.with_activation = act_cfg is not None
        if bias == "auto":
            bias = not self.with_norm
        self.with_bias = bias
        conv_type = (conv_cfg or {}).get("type", "Conv2d")
        self.conv = BRICKS.get(conv_type)(in_channels, out_channels, kernel_size, stride=stride, padding=padding,
                                         dilation=dilation, groups=groups, bias=bias)
        if self.with_norm:
            ncfg = dict(norm_cfg)
            ntype = ncfg.pop("type")
            ncfg.pop("requires_grad", None)
            nch = out_channels if order.index("norm") > order.index("conv") else in_channels
            if ntype == "GN":
                self.gn = nn.GroupNorm(num_channels=nch, **ncfg)
                self.norm_name = "gn"
            else:
                self.bn = BRICKS.get(ntype)(nch, **ncfg)
                self.norm_name = "bn"
        if self.with_activation:
            acfg = dict(act_cfg)
            self.activate = build_from_cfg(acfg, BRICKS)
        self.init_weights()

    @property
    def norm(self):
        return getattr(self, self.norm_name)

    def init_weights(self):
        nn.init.kaiming_normal_(self.conv.weight, a=0, mode="fan_out", nonlinearity="relu")
        if self.conv.bias is not None:
            nn.init.constant_(self.conv.bias, 0)
        if self.with_norm:
            nn.init.constant_(self.norm.weight, 1)
            nn.init.constant_(self.norm.bias, 0)

    def _fused_conv_relu(self, x, activate):
        """conv [+ bias] [+ ReLU] with nothing in between (ops/conv_igemm.conv_module): one implicit-GEMM kernel with
        the epilogue on the accumulators for the 3x3 / stride 1 layers where that measured faster than the library, and
        a one-pass bias / ReLU backward for every layer with a bias"""
        conv = self.conv
        if self.with_norm or self.order.index("conv") > self.order.index("act") or type(conv) is not nn.Conv2d:
            return None
        relu = bool(activate and self.with_activation)
        if relu and type(self.activate) is not nn.ReLU:
            return None
        if not (x.is_cuda and x.dtype == conv.weight.dtype) or torch.is_autocast_enabled():
            return None      # (autocast: the framework path casts per op; the fused route is fp32 only)
        return conv_igemm.conv_module(conv, x, relu)

    def masked(self, x, rowmask):
        """relu(conv(x)) * mask as one kernel (mask: 0 / 1 per position, flat) when this is a plain conv + bias + ReLU
        module on the fused path; None otherwise"""
        if (self.with_norm or not self.with_activation or type(self.activate) is not nn.ReLU
                or self.order.index("conv") > self.order.index("act") or type(self.conv) is not nn.Conv2d
                or not (x.is_cuda and x.dtype == self.conv.weight.dtype)):
            return None
        return conv_igemm.masked_conv_module(self.conv, x, rowmask)

    def forward(self, x, activate=True, norm=True):
        y = self._fused_conv_relu(x, activate)
        if y is not None:
            return y
        for layer in self.order:
            if layer == "conv":
                x = self.conv(x)
            elif layer == "norm" and norm and self.with_norm:
                x = self.norm(x)
            elif layer == "act" and activate and self.with_activation:
                x = self.activate(x)
        return x

    execute = forward
