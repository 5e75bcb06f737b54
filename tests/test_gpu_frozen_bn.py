"""GPU: fused eval-mode BatchNorm (+residual) (+ReLU) against the framework ops it replaces -- values and all
gradients (x, residual, weight, bias), every flag combination, the channel counts of ResNet-50/101, and the
fall-through conditions."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


def _ref(x, bn, res, relu):
    out = F.batch_norm(x, bn.running_mean, bn.running_var, bn.weight, bn.bias, False, 0.0, bn.eps)
    if res is not None:
        out = out + res
    return F.relu(out) if relu else out


@pytest.mark.parametrize("C,hw", [(64, (24, 20)), (256, (9, 7)), (1024, (5, 6)), (2048, (3, 4)), (8, (11, 13))])
@pytest.mark.parametrize("relu", [True, False])
@pytest.mark.parametrize("with_res", [True, False])
@pytest.mark.parametrize("affine_grad", [True, False])
def test_frozen_bn_act_matches_framework_ops(dev, C, hw, relu, with_res, affine_grad):
    from jdet_amd.ops.frozen_bn import FrozenBNActFunction, frozen_bn_act
    torch.manual_seed(C + hw[0])
    bn = torch.nn.BatchNorm2d(C).to(dev).eval()
    with torch.no_grad():
        bn.weight.uniform_(0.5, 1.5)
        bn.bias.normal_()
        bn.running_mean.normal_()
        bn.running_var.uniform_(0.5, 2.0)
    bn.weight.requires_grad_(affine_grad)
    bn.bias.requires_grad_(affine_grad)

    def mk():
        x = torch.randn(3, C, *hw, device=dev).contiguous(memory_format=torch.channels_last).requires_grad_(True)
        r = torch.randn(3, C, *hw, device=dev).contiguous(memory_format=torch.channels_last).requires_grad_(True) \
            if with_res else None
        return x, r
    torch.manual_seed(1)
    x1, r1 = mk()
    torch.manual_seed(1)
    x2, r2 = mk()
    gy = torch.randn(3, C, *hw, device=dev).contiguous(memory_format=torch.channels_last)
    y1 = frozen_bn_act(x1, bn, residual=r1, relu=relu)
    assert isinstance(y1.grad_fn, FrozenBNActFunction._backward_cls)        # the fused path ran
    assert y1.is_contiguous(memory_format=torch.channels_last)
    y1.backward(gy)
    g1 = [x1.grad, r1.grad if with_res else None, bn.weight.grad, bn.bias.grad]
    bn.weight.grad = bn.bias.grad = None
    y2 = _ref(x2, bn, r2, relu)
    y2.backward(gy)
    g2 = [x2.grad, r2.grad if with_res else None, bn.weight.grad, bn.bias.grad]
    bn.weight.grad = bn.bias.grad = None
    torch.testing.assert_close(y1, y2, rtol=1e-5, atol=1e-5)
    for a, b in zip(g1, g2):
        assert (a is None) == (b is None)
        if a is not None:
            scale = max(1.0, float(b.abs().max()))
            torch.testing.assert_close(a, b, rtol=1e-4, atol=2e-5 * scale)


def test_frozen_bn_falls_through(dev):
    from jdet_amd.ops.frozen_bn import FrozenBNActFunction, frozen_bn_act
    bn = torch.nn.BatchNorm2d(12).to(dev)
    x = torch.randn(2, 12, 6, 6, device=dev).contiguous(memory_format=torch.channels_last).requires_grad_(True)
    bn.train()                       # batch statistics: framework op
    assert not isinstance(frozen_bn_act(x, bn).grad_fn, FrozenBNActFunction._backward_cls)
    bn.eval()                        # C = 12: C/4 = 3 does not divide 256: framework op
    assert not isinstance(frozen_bn_act(x, bn).grad_fn, FrozenBNActFunction._backward_cls)
    bn16 = torch.nn.BatchNorm2d(16).to(dev).eval()
    xn = torch.randn(2, 16, 6, 6, device=dev, requires_grad=True)   # NCHW: framework op
    assert not isinstance(frozen_bn_act(xn, bn16).grad_fn, FrozenBNActFunction._backward_cls)
    xc = xn.detach().contiguous(memory_format=torch.channels_last).requires_grad_(True)
    assert isinstance(frozen_bn_act(xc, bn16).grad_fn, FrozenBNActFunction._backward_cls)


def test_resnet50_train_mode_uses_fused_bn(dev):
    """backbone in train() mode (norm_eval): outputs / gradients equal the unfused module composition"""
    from jdet_amd.models.backbones.resnet import Resnet50
    import jdet_amd.ops.frozen_bn as FB
    torch.manual_seed(0)
    m = Resnet50(frozen_stages=1, return_stages=["layer1", "layer2", "layer3", "layer4"]).to(dev)
    for p in m.parameters():
        if p.dim() == 4:
            p.data = p.data.contiguous(memory_format=torch.channels_last)
    m.train()
    x = torch.randn(1, 3, 128, 128, device=dev).contiguous(memory_format=torch.channels_last)
    outs = m(x)
    sum(o.square().mean() for o in outs).backward()
    g_f = {n: p.grad.clone() for n, p in m.named_parameters() if p.grad is not None}
    m.zero_grad(set_to_none=True)
    orig = FB._fusable
    FB._fusable = lambda *a, **k: False
    try:
        outs_r = m(x)
        sum(o.square().mean() for o in outs_r).backward()
    finally:
        FB._fusable = orig
    for a, b in zip(outs, outs_r):
        torch.testing.assert_close(a, b, rtol=1e-3, atol=1e-3 * float(b.abs().max()))
    g_r = {n: p.grad for n, p in m.named_parameters() if p.grad is not None}
    assert set(g_f) == set(g_r) and "layer1.0.conv1.weight" not in g_f and "layer2.0.bn1.weight" in g_f
    for n in ("layer2.0.bn1.weight", "layer3.5.bn3.bias", "layer4.2.conv3.weight", "layer2.0.downsample.1.weight"):
        s = float(g_r[n].abs().max())
        torch.testing.assert_close(g_f[n], g_r[n], rtol=5e-3, atol=5e-3 * s)


@pytest.mark.parametrize("C", [5, 15, 7, 64, 255, 256])
@pytest.mark.parametrize("P", [1, 333, 2 * 128 * 128])
def test_channel_sum_any_channel_count(dev, C, P):
    """jdet_channel_sum: per-channel sum of channels-last rows for channel counts off the vector kernels' grid (the
    heads' 15- / 5-channel output convs) against the float64 sum; and through the conv module's backward: the bias
    gradient of such a conv equals autograd's."""
    from jdet_amd import _lib as L
    g = torch.Generator().manual_seed(C * 7 + P)
    x = torch.randn(P, C, generator=g).to(dev)
    nbytes = L.lib().jdet_channel_sum_workspace(P, C)
    ws = torch.empty((nbytes,), dtype=torch.uint8, device=dev)
    out = torch.empty(C, device=dev)
    L.check(L.lib().jdet_channel_sum(L.ptr(x), P, C, L.ptr(out), L.ptr(ws), nbytes, L.stream_ptr(x)), "channel_sum")
    ref = x.double().sum(0)
    assert float((out.double() - ref).abs().max()) <= 1e-5 * max(1.0, float(x.abs().sum(0).max())), C


def test_small_output_conv_bias_gradient_through_the_any_channel_sum(dev):
    from jdet_amd.ops import conv_igemm as CI
    torch.manual_seed(0)
    conv = torch.nn.Conv2d(64, 15, 3, padding=1).to(dev).to(memory_format=torch.channels_last)
    x = torch.randn(2, 64, 24, 20, device=dev).contiguous(memory_format=torch.channels_last).requires_grad_(True)
    y = CI.conv_module(conv, x)
    gy = torch.randn_like(y)
    y.backward(gy)
    got = conv.bias.grad.clone()
    assert torch.allclose(got, gy.double().sum((0, 2, 3)).float(), rtol=1e-5, atol=1e-4)


@pytest.mark.parametrize("rows,cin,cout", [(1024, 1024, 1024), (1024, 1024, 16), (300, 96, 80), (7, 64, 5)])
def test_linear_with_the_own_bias_sum_equals_autograd(dev, rows, cin, cout):
    """ops/linear.Linear: same output and gradients as nn.Linear; the bias gradient comes from the two-stage column
    sums (no framework reduce_kernel, whose multi-workgroup path clears its semaphores with a memset node)"""
    from jdet_amd.ops.linear import Linear
    torch.manual_seed(rows + cout)
    a = Linear(cin, cout).to(dev)
    b = torch.nn.Linear(cin, cout).to(dev)
    b.load_state_dict(a.state_dict())
    x = torch.randn(rows, cin, device=dev)
    xa, xb = x.clone().requires_grad_(True), x.clone().requires_grad_(True)
    ya, yb = a(xa), b(xb)
    assert torch.allclose(ya, yb, rtol=1e-5, atol=1e-5)
    g = torch.randn_like(ya)
    ya.backward(g)
    yb.backward(g)
    assert torch.allclose(xa.grad, xb.grad, rtol=1e-4, atol=1e-4)
    assert torch.allclose(a.weight.grad, b.weight.grad, rtol=1e-4, atol=1e-3)
    ref = g.double().sum(0)
    assert float((a.bias.grad.double() - ref).abs().max()) <= 1e-5 * float(g.abs().sum(0).max()) + 1e-5
