"""GPU: the per-direction routing of the ResNet 1x1 convolutions (jdet_amd/models/backbones/resnet.py: Conv1x1 /
_Conv1x1Gemm -- forward / data gradient / weight gradient as GEMMs on the channels-last matrix view where that measured
faster) against the plain library convolution in float64."""
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("N,Ci,H,W,Co", [(2, 64, 40, 24, 256), (1, 512, 16, 16, 128), (2, 1024, 8, 8, 2048),
                                         (1, 256, 128, 160, 64)])
def test_conv1x1_routes_equal_the_convolution(dev, N, Ci, H, W, Co):
    from jdet_amd.models.backbones.resnet import conv1x1
    torch.manual_seed(Ci + Co)
    conv = conv1x1(Ci, Co).to(dev)
    x = torch.randn(N, Ci, H, W, device=dev).contiguous(memory_format=torch.channels_last).requires_grad_(True)
    g = torch.randn(N, Co, H, W, device=dev).contiguous(memory_format=torch.channels_last)
    y = conv(x)
    assert y.shape == (N, Co, H, W) and y.is_contiguous(memory_format=torch.channels_last)
    gx, gw = torch.autograd.grad(y, (x, conv.weight), g)
    xd, wd = x.detach().double().requires_grad_(True), conv.weight.detach().double().requires_grad_(True)
    yd = torch.nn.functional.conv2d(xd, wd)
    gxd, gwd = torch.autograd.grad(yd, (xd, wd), g.double())
    for a, b in ((y, yd), (gx, gxd), (gw, gwd)):
        assert float((a.double() - b).abs().max()) <= 2e-5 * max(1.0, float(b.abs().max()))
    # strided / no-grad calls take the plain convolution
    with torch.no_grad():
        assert torch.allclose(conv(x), y, atol=1e-4 * float(y.abs().max()))
