"""GPU: the HIP operators against closed forms that pass through neither the oracle restatement nor the
host-compiled reference text (tests/closed_form.py).  Every RoIAlign forward path is exercised."""
import numpy as np
import pytest
import torch

from tests import closed_form as CF

pytestmark = pytest.mark.gpu

V_ROT, V_ROT_V1, V_RI, V_HBB0, V_HBB1 = CF.V_ROT, CF.V_ROT_V1, CF.V_RI, CF.V_HBB0, CF.V_HBB1


def _layer(variant, hw, scale, s, nO=8):
    from jdet_amd.ops.riroi_align import RiRoIAlign
    from jdet_amd.ops.roi_align import ROIAlign
    from jdet_amd.ops.roi_align_rotated import ROIAlignRotated
    from jdet_amd.ops.roi_align_rotated_v1 import ROIAlignRotated_v1
    if variant == V_ROT:
        return ROIAlignRotated(hw, scale, s)
    if variant == V_ROT_V1:
        return ROIAlignRotated_v1(hw, scale, s)
    if variant == V_RI:
        return RiRoIAlign(hw, scale, s, nO)
    return ROIAlign(hw, scale, s, version=1 if variant == V_HBB1 else 0)


@pytest.fixture(params=["roi", "roi_cl"])
def path(request):
    from jdet_amd.ops import _roi_common as RC
    prev = RC.set_forward_path(request.param)
    yield request.param
    RC.set_forward_path(prev)


@pytest.mark.parametrize("variant", [V_ROT, V_ROT_V1, V_HBB0, V_HBB1, V_RI])
@pytest.mark.parametrize("hw,s", [((7, 7), 2), ((3, 5), 1), ((4, 4), 0)])
def test_roi_align_on_affine_map(dev, path, variant, hw, s):
    rng = np.random.default_rng(11 + variant)
    N, C, H, W, scale, nO = 2, 16, 48, 56, 0.25, 8
    feat, a, b, d = CF.affine_map(rng, N, C, H, W)
    rois = CF.interior_rois(rng, 40, N, H, W, scale, variant, max_wh=18.0)
    x = torch.from_numpy(feat).to(dev).contiguous(memory_format=torch.channels_last)
    y = _layer(variant, hw, scale, s, nO)(x, torch.from_numpy(rois).to(dev))
    ref = CF.roi_align_expected(variant, (a, b, d), rois, scale, hw[0], hw[1], nO)
    np.testing.assert_allclose(y.cpu().numpy(), ref, rtol=0, atol=2e-4)


def test_roi_align_affine_full_size(dev, path):
    """north-star shape (256x256x256 map, 2000 RoIs): interior RoIs pool the affine map to the closed form"""
    rng = np.random.default_rng(5)
    feat, a, b, d = CF.affine_map(rng, 1, 256, 256, 256)
    rois = CF.interior_rois(rng, 2000, 1, 256, 256, 0.25, V_ROT, max_wh=64.0)
    x = torch.from_numpy(feat).to(dev).contiguous(memory_format=torch.channels_last)
    y = _layer(V_ROT, (7, 7), 0.25, 2)(x, torch.from_numpy(rois).to(dev))
    ref = CF.roi_align_expected(V_ROT, (a, b, d), rois, 0.25, 7, 7)
    np.testing.assert_allclose(y.cpu().numpy(), ref, rtol=0, atol=1e-3)   # |f| up to ~600


@pytest.mark.parametrize("k,pad,stride,dil", [(3, 1, 1, 1), (3, 2, 2, 2), (1, 0, 1, 1)])
@pytest.mark.parametrize("cl", [False, True])
def test_deform_conv_integer_offsets(dev, k, pad, stride, dil, cl):
    """integer offsets = ordinary correlation on an integer-shifted image (numpy slicing); both layouts"""
    from jdet_amd.ops.dcn_v1 import DeformConv
    rng = np.random.default_rng(k + pad)
    B, C, Cout, H, W = 2, 8, 5, 13, 15
    x = rng.standard_normal((B, C, H, W)).astype(np.float32)
    dy = rng.integers(-3, 4, size=(k, k))
    dx = rng.integers(-3, 4, size=(k, k))
    Ho = (H + 2 * pad - (dil * (k - 1) + 1)) // stride + 1
    Wo = (W + 2 * pad - (dil * (k - 1) + 1)) // stride + 1
    off = CF.integer_offsets(dy, dx, B, Ho, Wo)
    conv = DeformConv(C, Cout, k, stride=stride, padding=pad, dilation=dil).to(dev)
    xt = torch.from_numpy(x).to(dev)
    if cl:
        xt = xt.contiguous(memory_format=torch.channels_last)
    y = conv(xt, torch.from_numpy(off).to(dev))
    ref = CF.deform_conv_integer_expected(x, conv.weight.detach().cpu().numpy(), dy, dx, pad, stride, dil)
    np.testing.assert_allclose(y.detach().cpu().numpy(), ref, rtol=1e-4, atol=1e-4)
