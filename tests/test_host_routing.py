"""CPU: host-side routing of the fused conv / bias paths and the no-CPU-fallback rule of the new ops."""
import pytest
import torch

from jdet_amd import _lib as L


def test_conv_module_on_cpu_is_the_plain_framework_path():
    from jdet_amd.models.utils.modules import ConvModule
    torch.manual_seed(0)
    m = ConvModule(16, 32, 3, padding=1)
    x = torch.randn(1, 16, 8, 8, requires_grad=True)
    y = m(x)
    ref = torch.relu(torch.nn.functional.conv2d(x, m.conv.weight, m.conv.bias, padding=1))
    assert torch.equal(y, ref)
    y.sum().backward()
    assert x.grad is not None and m.conv.bias.grad is not None


def test_conv_module_helper_routes_by_device_and_shape():
    from jdet_amd.ops import conv_igemm as CI
    conv = torch.nn.Conv2d(16, 32, 3, padding=1)
    x = torch.randn(1, 16, 8, 8)
    assert not CI.preferred(x, conv.weight)                      # host tensor: never the kernel
    assert torch.equal(CI.conv_module(conv, x, relu=True), torch.relu(conv(x)))
    assert CI._bias_bwd_supported(256) and CI._bias_bwd_supported(2048) and not CI._bias_bwd_supported(15)
    assert not CI._bias_bwd_supported(60) and CI._bias_bwd_supported(64)
    g, y = torch.randn(2, 8, 3, 3), torch.relu(torch.randn(2, 8, 3, 3))
    gp, gb = CI.bias_act_backward(g, y, True)                    # host tensors: the framework expressions
    assert torch.equal(gp, g * (y > 0)) and torch.allclose(gb, gp.sum((0, 2, 3)))


@pytest.mark.parametrize("call", ["conv", "dcn_v2", "pool", "convex_iou", "min_area", "convex_sort"])
def test_new_ops_have_no_cpu_fallback(call):
    from jdet_amd.ops import conv_igemm, convex_sort, dcn_v2, reppoints_convex_iou, reppoints_min_area_bbox
    with pytest.raises(L.JDetHipError):
        if call == "conv":
            conv_igemm.conv3x3_nhwc(torch.zeros(1, 4, 4, 32), torch.zeros(16, 3, 3, 32))
        elif call == "dcn_v2":
            dcn_v2.dcn_v2_conv(torch.zeros(1, 4, 6, 6), torch.zeros(1, 18, 6, 6), torch.ones(1, 9, 6, 6),
                               torch.zeros(4, 4, 3, 3), torch.zeros(4), 1, 1, 1, 1)
        elif call == "pool":
            dcn_v2.dcn_v2_pooling(torch.zeros(1, 4, 6, 6), torch.zeros(1, 5), torch.zeros(1, 2, 3, 3), 1.0, 3, 4, False)
        elif call == "convex_iou":
            reppoints_convex_iou.reppoints_convex_iou(torch.zeros(2, 18), torch.zeros(3, 8))
        elif call == "min_area":
            reppoints_min_area_bbox.reppoints_min_area_bbox(torch.zeros(2, 18))
        else:
            convex_sort.convex_sort(torch.zeros(2, 5, 2), torch.ones(2, 5))


def test_bottleneck_on_cpu_takes_the_per_layer_path():
    """the fused bottleneck (ops/conv_bn.py) is a device path: on the host a Bottleneck is conv -> bn -> relu chains on
    framework ops (python/jdet/models/backbones/resnet.py:L61-93), gradients for every parameter"""
    import torch.nn.functional as F
    from jdet_amd.models.backbones.resnet import Bottleneck, conv1x1
    from jdet_amd.ops import conv_bn
    torch.manual_seed(0)
    ds = torch.nn.Sequential(conv1x1(32, 64, 2), torch.nn.BatchNorm2d(64))
    blk = Bottleneck(32, 16, 2, ds).eval()
    x = torch.randn(2, 32, 8, 8, requires_grad=True)
    assert not conv_bn.fusable(blk, x)
    y = blk(x)
    ref = F.relu(blk.bn1(F.conv2d(x, blk.conv1.weight)))
    ref = F.relu(blk.bn2(F.conv2d(ref, blk.conv2.weight, None, 2, 1)))
    ref = F.relu(blk.bn3(F.conv2d(ref, blk.conv3.weight)) + ds[1](F.conv2d(x, ds[0].weight, None, 2)))
    assert torch.allclose(y, ref, atol=1e-6)
    y.sum().backward()
    assert all(p.grad is not None for p in blk.parameters()) and x.grad is not None
    assert conv_bn._trainable(blk) is True
    blk.conv1.weight.requires_grad_(False)
    assert conv_bn._trainable(blk) is None          # mixed: the fused path steps aside
    for p in blk.parameters():
        p.requires_grad_(False)
    assert conv_bn._trainable(blk) is False
    assert conv_bn.out_size(15, 3, 2) == 8 and conv_bn.out_size(15, 1, 2) == 8 and conv_bn.out_size(16, 3, 1) == 16


def test_linear_with_own_bias_sum_is_f_linear_on_the_host():
    from jdet_amd.ops.linear import Linear, linear
    torch.manual_seed(1)
    m = Linear(12, 7)
    x = torch.randn(5, 12, requires_grad=True)
    y = m(x)
    assert torch.equal(y, torch.nn.functional.linear(x, m.weight, m.bias))
    y.sum().backward()
    assert torch.allclose(m.bias.grad, torch.full((7,), 5.0))
    assert torch.equal(linear(x, m.weight, None), torch.nn.functional.linear(x, m.weight))
    with pytest.raises(L.JDetHipError):
        from jdet_amd.ops import conv_bn
        conv_bn.conv_bn_nhwc(torch.zeros(1, 4, 4, 16), torch.zeros(16, 1, 1, 16))
