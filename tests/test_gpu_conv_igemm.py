"""fp32-MFMA implicit-GEMM 3x3 conv (csrc/conv_igemm.hip) against float64 convolutions on the host and, for the
deformable form, against the column-matrix path (jdet_deform_im2col_nhwc + GEMM) it replaces at inference.

Tolerance: fp32 products and fp32 accumulation over K = 9 * Cin terms in a different order than the reference GEMM:
|err| <= 2e-5 * sqrt(K) * max|x| * max|w| is generous (observed ~1e-6 relative to the output scale)."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


def _ref64(x, w, b, relu):
    y = F.conv2d(x.double().cpu(), w.double().cpu(), None if b is None else b.double().cpu(), padding=1)
    return torch.relu(y) if relu else y


CASES = [
    # N, H, W, Cin, Cout, bias, relu
    (2, 16, 16, 256, 256, True, True),
    (1, 7, 9, 32, 128, False, False),       # ragged M (63 positions: one partial tile)
    (2, 13, 5, 64, 15, True, False),        # Cout below one tile (the classification conv of a head)
    (1, 32, 32, 256, 5, True, False),       # the regression conv
    (3, 20, 12, 96, 200, True, True),       # Cout straddles two N tiles
    (1, 6, 6, 48, 40, True, True),          # Cin a multiple of 16 only
    (1, 1, 1, 32, 32, True, False),         # a single position: every tap but the centre is padding
    (1, 128, 128, 256, 256, False, True),   # S2ANet P3 tower conv at 1024^2
]


@pytest.mark.parametrize("N,H,W,Cin,Cout,bias,relu", CASES)
@pytest.mark.parametrize("tile", [0, 64, 65, 66, 128, 129, 130])     # +1: 16-deep K step, +2: no K split
def test_conv3x3_matches_float64(N, H, W, Cin, Cout, bias, relu, tile):
    from jdet_amd.ops import conv_igemm as CI
    if tile and N * H * W > 4096:
        pytest.skip("forced tile shapes are exercised on the small cases")
    g = torch.Generator().manual_seed(N * 1000 + H * 10 + Cout)
    x = torch.randn(N, Cin, H, W, generator=g)
    w = torch.randn(Cout, Cin, 3, 3, generator=g) * (2.0 / (9 * Cin)) ** 0.5
    b = torch.randn(Cout, generator=g) if bias else None
    y = CI.conv3x3(x.cuda(), w.cuda(), None if b is None else b.cuda(), relu, tile=tile)
    assert y.shape == (N, Cout, H, W)
    ref = _ref64(x, w, b, relu)
    err = (y.double().cpu() - ref).abs().max().item()
    assert err <= 2e-5 * ref.abs().max().item() + 1e-6, err


def test_rowmask_and_empty_batch():
    from jdet_amd.ops import conv_igemm as CI
    g = torch.Generator().manual_seed(3)
    x = torch.randn(2, 10, 6, 64, generator=g).cuda()
    w = torch.randn(128, 3, 3, 64, generator=g).cuda() * 0.05
    mask = (torch.rand(2 * 10 * 6, generator=g) > 0.3).float().cuda()
    y = CI.conv3x3_nhwc(x, w, None, True, mask)
    y0 = CI.conv3x3_nhwc(x, w, None, True)
    assert torch.equal(y, y0 * mask.view(2, 10, 6, 1))
    assert CI.conv3x3_nhwc(x[:0], w).shape == (0, 10, 6, 128)


def test_unsupported_shapes_raise():
    from jdet_amd import _lib as L
    from jdet_amd.ops import conv_igemm as CI
    assert not CI.supported(40, 256) and CI.supported(32, 15)
    x = torch.zeros(1, 4, 4, 40).cuda()
    w = torch.zeros(16, 3, 3, 40).cuda()
    with pytest.raises(L.JDetHipError):
        CI.conv3x3_nhwc(x, w)
    with pytest.raises(L.JDetHipError):
        CI.conv3x3_nhwc(torch.zeros(1, 4, 4, 32), torch.zeros(16, 3, 3, 32))     # host tensors: no CPU fallback


@pytest.mark.parametrize("tile", [64, 65, 66, 128, 129, 130])
@pytest.mark.parametrize("N,H,W,C,Cout,scale", [(2, 16, 16, 256, 256, 1.5), (1, 9, 11, 64, 96, 4.0),
                                                (1, 64, 64, 256, 256, 2.0)])
def test_deformable_matches_column_path(N, H, W, C, Cout, scale, tile):
    """same samples, same weights: the fused gather must agree with deformable im2col + GEMM, including samples that
    land outside the image (offsets up to `scale` pixels push border taps past (-1, H))"""
    from jdet_amd.ops import conv_igemm as CI
    from jdet_amd.ops import dcn_v1
    g = torch.Generator().manual_seed(C + H)
    x = torch.randn(N, H, W, C, generator=g).cuda()
    w = (torch.randn(Cout, 3, 3, C, generator=g) * (2.0 / (9 * C)) ** 0.5).cuda()
    off = (torch.randn(N, 18, H, W, generator=g) * scale).cuda()
    y = CI.conv3x3_nhwc(x, w, offset=off, tile=tile)
    cols = dcn_v1.deformable_im2col_nhwc(x, off, 3, 3, (1, 1), (1, 1), (1, 1))
    ref = (cols.double() @ w.reshape(Cout, -1).double().t()).view(N, H, W, Cout)
    err = (y.double() - ref).abs().max().item()
    assert err <= 2e-5 * ref.abs().max().item() + 1e-6, err


def test_zero_offset_is_the_plain_convolution():
    from jdet_amd.ops import conv_igemm as CI
    g = torch.Generator().manual_seed(11)
    x = torch.randn(1, 12, 12, 64, generator=g).cuda()
    w = (torch.randn(64, 3, 3, 64, generator=g) * 0.05).cuda()
    y0 = CI.conv3x3_nhwc(x, w, tile=66)           # same tile shape and K order on both sides
    y1 = CI.conv3x3_nhwc(x, w, offset=torch.zeros(1, 18, 12, 12).cuda(), tile=66)
    assert torch.equal(y0, y1)        # weights (1, 0, 0, 0): x * 1 + 0 + 0 + 0 is exact


@pytest.mark.parametrize("dgrad", [False, True])
@pytest.mark.parametrize("act", [True, False])
def test_conv_module_fused_path_matches_library_path(act, dgrad):
    """ConvModule routes 3x3 conv [+ ReLU] through the fused kernel above the measured break-even size: same output,
    same gradients (the backward is the library's convolution backward on the ReLU-masked gradient)"""
    from jdet_amd.models.utils.modules import ConvModule
    from jdet_amd.ops import conv_igemm as CI
    torch.manual_seed(5)
    m = ConvModule(64, 128, 3, padding=1, act_cfg=dict(type="ReLU") if act else None).cuda()
    torch.nn.init.normal_(m.conv.bias, std=0.1)
    x = torch.randn(2, 64, 72, 64, device="cuda").contiguous(memory_format=torch.channels_last)
    assert CI.preferred(x, m.conv.weight)
    g = torch.randn(2, 128, 72, 64, device="cuda")
    if act:      # outputs within rounding of the ReLU threshold may land on either side of it in the two summation
        CI.ENABLED = False          # orders: no upstream gradient there, so the mask decision cannot matter
        try:
            with torch.no_grad():
                g = g * (m(x).abs() > 1e-4)
        finally:
            CI.ENABLED = True
    outs = []
    for enabled in (True, False):
        CI.ENABLED, CI.TRAIN, CI.DGRAD = enabled, True, dgrad
        try:
            xi = x.clone().requires_grad_(True)
            m.zero_grad()
            y = m(xi)
            y.backward(g)
            outs.append((y.detach(), xi.grad, m.conv.weight.grad.clone(), m.conv.bias.grad.clone()))
        finally:
            CI.ENABLED, CI.TRAIN, CI.DGRAD = True, True, False
    for a, b in zip(*outs):
        assert (a - b).abs().max().item() <= 2e-4 * b.abs().max().item() + 1e-6
    with torch.no_grad():
        assert torch.equal(m(x), outs[0][0])          # the no-grad route is the same kernel
    assert not act or (outs[0][0] == 0).float().mean().item() > 0.2     # the ReLU is active


def test_dgrad_weights_of_the_towers_live_in_one_bank():
    """conv_igemm.dgrad_weight: channels-last weights are rewritten by ONE launch for all of them (conv_bn.DgradBank);
    a new version refreshes, a dead weight leaves the bank, other layouts take the per-weight form -- always
    weight.flip(2, 3).permute(1, 2, 3, 0)"""
    import gc
    from jdet_amd.ops import conv_igemm as CI
    torch.manual_seed(11)
    ws = [torch.nn.Parameter((torch.randn(co, ci, 3, 3, device="cuda") * 0.1).contiguous(memory_format=torch.channels_last))
          for co, ci in ((64, 32), (48, 64), (256, 256))]
    ref = lambda w: w.detach().flip(2, 3).permute(1, 2, 3, 0).contiguous()
    for w in ws:
        assert torch.equal(CI.dgrad_weight(w), ref(w))
    bank = CI._DGRAD_BANKS[ws[0].device]["bank"]
    assert all(any(v.weight is w for v in bank.convs) for w in ws)            # one bank holds the three
    with torch.no_grad():
        ws[1].mul_(3.0)
    for w in ws:
        assert torch.equal(CI.dgrad_weight(w), ref(w))                        # refreshed after the in-place update
    plain = torch.randn(16, 8, 3, 3, device="cuda")                           # NCHW-contiguous: per-weight form
    assert torch.equal(CI.dgrad_weight(plain), ref(plain))
    keep = ws[0]
    del ws, w
    gc.collect()
    assert torch.equal(CI.dgrad_weight(keep), ref(keep))                      # the dead weights are dropped on the way
    assert all(v.weight is not None for v in CI._DGRAD_BANKS[keep.device]["bank"].convs)


def test_deform_conv_inference_takes_the_fused_kernel():
    from jdet_amd.ops import conv_igemm as CI
    from jdet_amd.ops import dcn_v1
    torch.manual_seed(6)
    x = torch.randn(2, 32, 128, 128, device="cuda").contiguous(memory_format=torch.channels_last)
    w = torch.randn(48, 32, 3, 3, device="cuda") * 0.1
    off = torch.randn(2, 18, 128, 128, device="cuda") * 3
    with torch.no_grad():
        y = dcn_v1.deform_conv(x, off, w, 1, 1, 1)
        CI.ENABLED = False
        try:
            ref = dcn_v1.deform_conv(x, off, w, 1, 1, 1)
        finally:
            CI.ENABLED = True
    assert y.shape == ref.shape and (y - ref).abs().max().item() <= 2e-5 * ref.abs().max().item() + 1e-6
    # with gradients requested the column-matrix path (which has the backward) is taken
    xr = x.clone().requires_grad_(True)
    dcn_v1.deform_conv(xr, off, w, 1, 1, 1).sum().backward()
    assert xr.grad is not None


@pytest.mark.parametrize("relu", [True, False])
@pytest.mark.parametrize("N,C,H,W", [(2, 256, 33, 17), (1, 64, 8, 8), (3, 512, 5, 7), (1, 1024, 4, 4), (2, 60, 9, 9)])
def test_bias_act_backward_matches_framework(N, C, H, W, relu):
    """one-pass bias gradient + ReLU mask (jdet_bias_act_backward) vs threshold_backward + sum; C = 60: the framework
    fallback (channel count outside the kernel's shapes)"""
    from jdet_amd.ops import conv_igemm as CI
    g = torch.randn(N, C, H, W, device="cuda").contiguous(memory_format=torch.channels_last)
    y = torch.relu(torch.randn(N, C, H, W, device="cuda")).contiguous(memory_format=torch.channels_last)
    gp, gb = CI.bias_act_backward(g, y, relu)
    ref = g * (y > 0) if relu else g
    assert torch.equal(gp, ref)
    ref_b = ref.double().sum((0, 2, 3))
    assert (gb.double() - ref_b).abs().max().item() <= 1e-5 * ref.abs().sum((0, 2, 3)).max().item() + 1e-6


@pytest.mark.parametrize("k,pad,stride", [(1, 0, 1), (3, 1, 2), (3, 1, 1)])
def test_conv_module_library_forward_with_fused_bias_backward(k, pad, stride):
    """layers the implicit GEMM does not take (1x1, strided, small maps) still get the one-pass bias / ReLU backward"""
    from jdet_amd.models.utils.modules import ConvModule
    from jdet_amd.ops import conv_igemm as CI
    torch.manual_seed(9)
    m = ConvModule(64, 128, k, stride=stride, padding=pad).cuda()
    torch.nn.init.normal_(m.conv.bias, std=0.1)
    x = torch.randn(2, 64, 20, 24, device="cuda").contiguous(memory_format=torch.channels_last)
    outs = []
    for on in (True, False):
        CI.BIAS_ACT_BWD = on
        try:
            xi = x.clone().requires_grad_(True)
            m.zero_grad()
            y = m(xi)
            if on:
                g = torch.randn_like(y) * (y.detach().abs() > 1e-4)
            y.backward(g)
            outs.append((y.detach(), xi.grad, m.conv.weight.grad.clone(), m.conv.bias.grad.clone()))
        finally:
            CI.BIAS_ACT_BWD = True
    for a, b in zip(*outs):
        assert (a - b).abs().max().item() <= 2e-4 * b.abs().max().item() + 1e-6
