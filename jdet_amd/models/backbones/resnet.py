"""ResNet backbones.  Mirrors python/jdet/models/backbones/resnet.py: Bottleneck L61-93 (stride on
the 3x3), `ResNet` L95-185 (return_stages, frozen_stages, norm_eval), `Resnet50` L205-209,
`Resnet101` L219-230.  Parameter names are the torchvision/Jittor ones (conv1, bn1, layerN.M.convK,
downsample.0/1) so reference checkpoints map 1:1.

The bottleneck blocks run on this repo's own fp32-MFMA convolution family (ops/conv_bn.py, csrc/conv_bn.hip): every
1x1 / 3x3 convolution with its eval-mode BatchNorm, the identity add and the ReLU in the epilogue, forward and backward
(data gradient = the same kernel on transposed weights, weight gradient = csrc/conv_wgrad.hip) -- whenever the input
is a channels-last fp32 device tensor and the norm layers are eval-mode BatchNorm2d (`norm_eval`, the reference's
training mode).  What is left to the library: the 7x7 stem, the stride-2 data gradients (three layers) and every case the
fused path does not take (BasicBlock, training-mode BatchNorm, other dtypes), where the per-layer path below runs the
library convolution + one fused BatchNorm / ReLU pass (ops/frozen_bn.py).
"""
import torch
from torch import nn

from jdet_amd.ops import conv_bn
from jdet_amd.ops.frozen_bn import frozen_bn_act
from jdet_amd.utils.registry import BACKBONES

__all__ = ["ResNet", "Resnet18", "Resnet34", "Resnet50", "Resnet101", "Resnet152"]


def conv3x3(in_planes, out_planes, stride=1, groups=1, dilation=1):
    conv = nn.Conv2d(in_planes, out_planes, kernel_size=3, stride=stride, padding=dilation, groups=groups,
                     bias=False, dilation=dilation)
    nn.init.kaiming_normal_(conv.weight, mode="fan_out", nonlinearity="relu")
    return conv


def conv1x1(in_planes, out_planes, stride=1):
    conv = nn.Conv2d(in_planes, out_planes, kernel_size=1, stride=stride, bias=False)
    nn.init.kaiming_normal_(conv.weight, mode="fan_out", nonlinearity="relu")
    return conv


def _downsample(seq, x):
    """the (conv1x1, norm) pair of `_make_layer`; other module types run as they are"""
    if isinstance(seq, nn.Sequential) and len(seq) == 2 and isinstance(seq[1], nn.BatchNorm2d):
        return frozen_bn_act(seq[0](x), seq[1], relu=False)
    return seq(x)


class BasicBlock(nn.Module):
    expansion = 1

    def __init__(self, inplanes, planes, stride=1, downsample=None, groups=1, base_width=64, dilation=1,
                 norm_layer=None):
        super().__init__()
        norm_layer = norm_layer or nn.BatchNorm2d
        if groups != 1 or base_width != 64:
            raise ValueError("BasicBlock only supports groups=1 and base_width=64")
        if dilation > 1:
            raise NotImplementedError("Dilation > 1 not supported in BasicBlock")
        self.conv1 = conv3x3(inplanes, planes, stride)
        self.bn1 = norm_layer(planes)
        self.relu = nn.ReLU(inplace=True)
        self.conv2 = conv3x3(planes, planes)
        self.bn2 = norm_layer(planes)
        self.downsample = downsample
        self.stride = stride

    def forward(self, x):
        identity = x
        out = frozen_bn_act(self.conv1(x), self.bn1)
        if self.downsample is not None:
            identity = _downsample(self.downsample, x)
        return frozen_bn_act(self.conv2(out), self.bn2, residual=identity)


class Bottleneck(nn.Module):
    expansion = 4

    def __init__(self, inplanes, planes, stride=1, downsample=None, groups=1, base_width=64, dilation=1,
                 norm_layer=None):
        super().__init__()
        norm_layer = norm_layer or nn.BatchNorm2d
        width = int(planes * (base_width / 64.0)) * groups
        self.conv1 = conv1x1(inplanes, width)
        self.bn1 = norm_layer(width)
        self.conv2 = conv3x3(width, width, stride, groups, dilation)
        self.bn2 = norm_layer(width)
        self.conv3 = conv1x1(width, planes * self.expansion)
        self.bn3 = norm_layer(planes * self.expansion)
        self.relu = nn.ReLU(inplace=True)
        self.downsample = downsample
        self.stride = stride

    def forward(self, x):
        if conv_bn.fusable(self, x):
            # the block on this repo's kernels: each conv with its BatchNorm / identity / ReLU in the epilogue, forward
            # and backward (ops/conv_bn.py)
            return conv_bn.bottleneck(self, x)
        identity = x
        out = frozen_bn_act(self.conv1(x), self.bn1)
        out = frozen_bn_act(self.conv2(out), self.bn2)
        if self.downsample is not None:
            identity = _downsample(self.downsample, x)
        return frozen_bn_act(self.conv3(out), self.bn3, residual=identity)


@BACKBONES.register_module()
class ResNet(nn.Module):
    def __init__(self, block, layers, return_stages=["layer4"], frozen_stages=-1, norm_eval=True, num_classes=None,
                 groups=1, width_per_group=64, replace_stride_with_dilation=None, norm_layer=None):
        super().__init__()
        norm_layer = norm_layer or nn.BatchNorm2d
        self.frozen_stages = frozen_stages
        self.norm_eval = norm_eval
        self._norm_layer = norm_layer
        self.inplanes = 64
        self.dilation = 1
        if replace_stride_with_dilation is None:
            replace_stride_with_dilation = [False, False, False]
        if len(replace_stride_with_dilation) != 3:
            raise ValueError("replace_stride_with_dilation should be None or a 3-element tuple, got {}".format(
                replace_stride_with_dilation))
        self.groups = groups
        self.base_width = width_per_group
        self.conv1 = nn.Conv2d(3, self.inplanes, kernel_size=7, stride=2, padding=3, bias=False)
        nn.init.kaiming_normal_(self.conv1.weight, mode="fan_out", nonlinearity="relu")
        self.bn1 = norm_layer(self.inplanes)
        self.relu = nn.ReLU(inplace=True)
        self.maxpool = nn.MaxPool2d(kernel_size=3, stride=2, padding=1)
        self.layer1 = self._make_layer(block, 64, layers[0])
        self.layer2 = self._make_layer(block, 128, layers[1], stride=2, dilate=replace_stride_with_dilation[0])
        self.layer3 = self._make_layer(block, 256, layers[2], stride=2, dilate=replace_stride_with_dilation[1])
        self.layer4 = self._make_layer(block, 512, layers[3], stride=2, dilate=replace_stride_with_dilation[2])
        self.num_classes = num_classes
        self.return_stages = return_stages
        if num_classes is not None:
            self.avgpool = nn.AdaptiveAvgPool2d((1, 1))
            self.fc = nn.Linear(512 * block.expansion, num_classes)
        self._freeze_stages()

    def _make_layer(self, block, planes, blocks, stride=1, dilate=False):
        norm_layer = self._norm_layer
        downsample = None
        previous_dilation = self.dilation
        if dilate:
            self.dilation *= stride
            stride = 1
        if stride != 1 or self.inplanes != planes * block.expansion:
            downsample = nn.Sequential(conv1x1(self.inplanes, planes * block.expansion, stride),
                                       norm_layer(planes * block.expansion))
        layers = [block(self.inplanes, planes, stride, downsample, self.groups, self.base_width, previous_dilation,
                        norm_layer)]
        self.inplanes = planes * block.expansion
        for _ in range(1, blocks):
            layers.append(block(self.inplanes, planes, groups=self.groups, base_width=self.base_width,
                                dilation=self.dilation, norm_layer=norm_layer))
        return nn.Sequential(*layers)

    def _freeze_stages(self):
        if self.frozen_stages >= 0:
            self.bn1.eval()
            for m in [self.conv1, self.bn1]:
                for param in m.parameters():
                    param.requires_grad_(False)
        for i in range(1, self.frozen_stages + 1):
            m = getattr(self, "layer{}".format(i))
            m.eval()
            for param in m.parameters():
                param.requires_grad_(False)

    def forward(self, x):
        outputs = []
        x = self.conv1(x)
        x = frozen_bn_act(x, self.bn1) if isinstance(self.bn1, nn.BatchNorm2d) else self.relu(self.bn1(x))
        x = self.maxpool(x)
        if conv_bn.ENABLED and x.is_cuda and torch.is_grad_enabled():
            # the data-gradient weights of every trainable bottleneck, rewritten in one launch per step
            conv_bn.prepare([m for i in range(1, 5) for m in getattr(self, "layer%d" % i) if isinstance(m, Bottleneck)])
        for i in range(1, 5):
            name = f"layer{i}"
            x = getattr(self, name)(x)
            if name in self.return_stages:
                outputs.append(x)
        if self.num_classes is not None:
            x = self.fc(torch.flatten(self.avgpool(x), 1))
            if "fc" in self.return_stages:
                outputs.append(x)
        return tuple(outputs)

    execute = forward

    def train(self, mode=True):
        super().train(mode)
        if mode:
            self._freeze_stages()
            if self.norm_eval:
                for m in self.modules():
                    if isinstance(m, nn.modules.batchnorm._BatchNorm):
                        m.eval()
        return self


def _resnet(block, layers, pretrained=False, **kwargs):
    # `pretrained=True` means jittorhub://resnetNN.pkl in the reference: a network fetch that is not
    # available here; weights stay at their (same-distribution) random init.
    return ResNet(block, layers, **kwargs)


@BACKBONES.register_module()
def Resnet18(pretrained=False, **kwargs):
    return _resnet(BasicBlock, [2, 2, 2, 2], pretrained, **kwargs)


@BACKBONES.register_module()
def Resnet34(pretrained=False, **kwargs):
    return _resnet(BasicBlock, [3, 4, 6, 3], pretrained, **kwargs)


@BACKBONES.register_module()
def Resnet50(pretrained=False, **kwargs):
    return _resnet(Bottleneck, [3, 4, 6, 3], pretrained, **kwargs)


@BACKBONES.register_module()
def Resnet101(pretrained=False, **kwargs):
    return _resnet(Bottleneck, [3, 4, 23, 3], pretrained, **kwargs)


@BACKBONES.register_module()
def Resnet152(pretrained=False, **kwargs):
    return _resnet(Bottleneck, [3, 8, 36, 3], pretrained, **kwargs)
