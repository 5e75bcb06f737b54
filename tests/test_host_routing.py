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
