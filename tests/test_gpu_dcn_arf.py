"""GPU parity: deformable-conv sampling kernels, DeformConv autograd, ARF / ORConv2d."""
import numpy as np
import pytest
import torch

from oracle import oracle as O

pytestmark = pytest.mark.gpu


def _t(a, dev):
    return torch.from_numpy(np.ascontiguousarray(a)).to(dev)


def test_dcn_kernels_vs_golden(golden, dev):
    from jdet_amd.ops import dcn_v1 as D
    g = golden("deform_conv")
    for nm in "abc":
        k, pad, stride, dil, dg = [int(v) for v in g["cfg_" + nm]]
        a = (k, k, (pad, pad), (stride, stride), (dil, dil), dg)
        im, off, gcol = _t(g["im_" + nm], dev), _t(g["off_" + nm], dev), _t(g["gcol_" + nm], dev)
        np.testing.assert_array_equal(D.deformable_im2col(im, off, *a).cpu().numpy(), g["col_" + nm])
        np.testing.assert_allclose(D.deformable_col2im(gcol, off, im.shape, *a).cpu().numpy(), g["gim_" + nm], atol=2e-5)
        np.testing.assert_array_equal(D.deformable_col2im_coord(gcol, im, off, *a).cpu().numpy(), g["goff_" + nm])


def test_dcn_nhwc_kernels_vs_golden(golden, dev):
    """channels-last kernels against the same reference-generated vectors: im2col bit-exact (same
    arithmetic, other layout), col2im (sorted gather) to fp32 summation-order tolerance"""
    from jdet_amd.ops import dcn_v1 as D
    g = golden("deform_conv")
    for nm in "ac":     # deformable_groups == 1, C % 4 == 0
        k, pad, stride, dil, dg = [int(v) for v in g["cfg_" + nm]]
        a = (k, k, (pad, pad), (stride, stride), (dil, dil))
        im, off, gcol = g["im_" + nm], g["off_" + nm], g["gcol_" + nm]
        B, C, H, W = im.shape
        Ho, Wo = off.shape[2:]
        x = _t(im.transpose(0, 2, 3, 1), dev)
        cols = D.deformable_im2col_nhwc(x, _t(off, dev), *a).cpu().numpy()
        ref = g["col_" + nm].reshape(C, k * k, B, Ho, Wo).transpose(2, 3, 4, 1, 0).reshape(B * Ho * Wo, k * k * C)
        np.testing.assert_array_equal(cols, ref)
        gc = gcol.reshape(C, k * k, B, Ho, Wo).transpose(2, 3, 4, 1, 0).reshape(B * Ho * Wo, k * k * C)
        gx = D.deformable_col2im_nhwc(_t(gc, dev), _t(off, dev), (B, H, W, C), *a).cpu().numpy()
        np.testing.assert_allclose(gx.transpose(0, 3, 1, 2), g["gim_" + nm], atol=2e-5)


@pytest.mark.parametrize("cl", [False, True])
def test_deform_conv_nhwc_path_vs_oracle(dev, cl):
    """AlignConv usage: offsets detached -> the whole forward/backward runs on the channels-last path.
    Offsets large enough to leave the image; C = 260 exercises the second 256-channel chunk."""
    from jdet_amd.ops.dcn_v1 import DeformConv
    rng = np.random.default_rng(5)
    B, Cin, Cout, H, W = 2, 260, 12, 11, 14
    x = rng.standard_normal((B, Cin, H, W)).astype(np.float32)
    off = (rng.standard_normal((B, 18, H, W)) * 3).astype(np.float32)
    conv = DeformConv(Cin, Cout, 3, padding=1).to(dev)
    w = conv.weight.detach().cpu().numpy()
    xt = _t(x, dev)
    if cl:
        xt = xt.contiguous(memory_format=torch.channels_last)
    xt.requires_grad_(True)
    y = conv(xt, _t(off, dev))
    a = (3, 3, (1, 1), (1, 1), (1, 1), 1)
    col = O.deform_im2col(x, off, *a)
    ref = (w.reshape(Cout, -1).astype(np.float64) @ col.reshape(Cin * 9, -1).astype(np.float64)).reshape(Cout, B, H, W).transpose(1, 0, 2, 3)
    np.testing.assert_allclose(y.detach().cpu().numpy(), ref, rtol=1e-4, atol=2e-4)
    gy = rng.standard_normal(y.shape).astype(np.float32)
    y.backward(_t(gy, dev))
    gcol = (w.reshape(Cout, -1).T.astype(np.float64) @ gy.transpose(1, 0, 2, 3).reshape(Cout, -1).astype(np.float64)).astype(np.float32).reshape(col.shape)
    np.testing.assert_allclose(xt.grad.cpu().numpy(), O.deform_col2im(gcol, off, x.shape, *a), rtol=1e-4, atol=3e-4)
    gw = gy.transpose(1, 0, 2, 3).reshape(Cout, -1).astype(np.float64) @ col.reshape(Cin * 9, -1).astype(np.float64).T
    np.testing.assert_allclose(conv.weight.grad.cpu().numpy().reshape(Cout, -1), gw, rtol=1e-4, atol=3e-4)


def test_deform_conv_module_vs_oracle(dev):
    """S2ANet AlignConv shape family (3x3, pad 1, dg 1), small: forward = W . im2col, backward through
    the three kernels; GEMMs are rocBLAS fp32 so tolerance 1e-4 relative."""
    from jdet_amd.ops.dcn_v1 import DeformConv
    rng = np.random.default_rng(3)
    B, Cin, Cout, H, W = 2, 16, 12, 13, 17
    x = rng.standard_normal((B, Cin, H, W)).astype(np.float32)
    off = (rng.standard_normal((B, 18, H, W)) * 1.5).astype(np.float32)
    conv = DeformConv(Cin, Cout, 3, padding=1).to(dev)
    w = conv.weight.detach().cpu().numpy()
    xt, ot = _t(x, dev).requires_grad_(True), _t(off, dev).requires_grad_(True)
    y = conv(xt, ot)
    a = (3, 3, (1, 1), (1, 1), (1, 1), 1)
    col = O.deform_im2col(x, off, *a)
    ref = (w.reshape(Cout, -1).astype(np.float64) @ col.reshape(Cin * 9, -1).astype(np.float64)).reshape(Cout, B, H, W).transpose(1, 0, 2, 3)
    np.testing.assert_allclose(y.detach().cpu().numpy(), ref, rtol=1e-4, atol=1e-4)
    gy = rng.standard_normal(y.shape).astype(np.float32)
    y.backward(_t(gy, dev))
    gcol = (w.reshape(Cout, -1).T.astype(np.float64) @ gy.transpose(1, 0, 2, 3).reshape(Cout, -1).astype(np.float64)).astype(np.float32).reshape(col.shape)
    np.testing.assert_allclose(xt.grad.cpu().numpy(), O.deform_col2im(gcol, off, x.shape, *a), rtol=1e-4, atol=2e-4)
    np.testing.assert_allclose(ot.grad.cpu().numpy(), O.deform_col2im_coord(gcol, x, off, *a), rtol=1e-4, atol=2e-4)
    gw = gy.transpose(1, 0, 2, 3).reshape(Cout, -1).astype(np.float64) @ col.reshape(Cin * 9, -1).astype(np.float64).T
    np.testing.assert_allclose(conv.weight.grad.cpu().numpy().reshape(Cout, -1), gw, rtol=1e-4, atol=2e-4)


def test_deform_conv_zero_offset_equals_conv2d(dev):
    from jdet_amd.ops.dcn_v1 import DeformConv
    torch.manual_seed(0)
    conv = DeformConv(8, 6, 3, padding=1).to(dev)
    x = torch.randn(3, 8, 20, 24, device=dev)
    y = conv(x, torch.zeros(3, 18, 20, 24, device=dev))
    ref = torch.nn.functional.conv2d(x, conv.weight, padding=1)
    assert torch.allclose(y, ref, atol=1e-4, rtol=1e-4)
    with pytest.raises(ValueError):
        conv(torch.randn(8, 20, 24, device=dev), torch.zeros(3, 18, 20, 24, device=dev))


def test_arf_vs_golden_and_orconv(golden, dev):
    from jdet_amd.ops.orn import ORConv2d, RotationInvariantPooling, arf_backward, arf_forward, arf_indices
    g = golden("arf")
    np.testing.assert_array_equal(arf_indices(8, 8, (3, 3)).numpy(), g["idx"])
    np.testing.assert_array_equal(arf_indices(8, 8, (1, 1)).numpy(), g["idx1"])
    np.testing.assert_array_equal(arf_forward(_t(g["w"], dev), _t(g["idx"], dev)).cpu().numpy(), g["y"])
    np.testing.assert_array_equal(arf_backward(_t(g["idx"], dev), _t(g["g"], dev)).cpu().numpy(), g["gw"])
    np.testing.assert_array_equal(arf_forward(_t(g["w1"], dev), _t(g["idx1"], dev)).cpu().numpy(), g["y1"])
    # S2ANet's or_conv: ORConv2d(256, 32, 3, padding=1, arf_config=(1, 8)) -- scaled down
    conv = ORConv2d(16, 4, kernel_size=3, padding=1, arf_config=(1, 8)).to(dev)
    x = torch.randn(2, 16, 9, 9, device=dev, requires_grad=True)
    y = conv(x)
    assert y.shape == (2, 32, 9, 9)
    y.square().sum().backward()
    w = conv.weight.detach().cpu().numpy()
    wr = O.arf_forward(w, arf_indices(1, 8, (3, 3)).numpy())
    ref = torch.nn.functional.conv2d(x.detach(), _t(wr, dev), conv.bias, padding=1)
    assert torch.allclose(y, ref, atol=1e-5)
    assert conv.weight.grad is not None and conv.weight.grad.shape == conv.weight.shape
    rip = RotationInvariantPooling(32, 8).to(dev)
    p = rip(y)
    assert p.shape == (2, 4, 9, 9)
    assert torch.equal(p, y.view(2, 4, 8, 9, 9).max(2).values)
    assert any(k.startswith("conv.") for k in rip.state_dict())  # checkpoint-compatible unused submodule


def test_deform_conv_fused_training_path_vs_oracle(dev, monkeypatch):
    """Training without the column matrix (forward: gathered A operand of conv_igemm.hip; weight gradient: gathered B
    operand of conv_wgrad.hip): same oracle as the column path; the conv is used TWICE in one backward pass (a tower
    shared by two pyramid levels), so the second use adds into the gradient buffer of the first."""
    from jdet_amd.ops import dcn_v1
    from jdet_amd.ops.dcn_v1 import DeformConv
    monkeypatch.setattr(dcn_v1, "FUSED_TRAIN_MIN_POSITIONS", 1)
    rng = np.random.default_rng(11)
    Cin, Cout = 64, 12
    conv = DeformConv(Cin, Cout, 3, padding=1).to(dev)
    w = conv.weight.detach().cpu().numpy()
    a = (3, 3, (1, 1), (1, 1), (1, 1), 1)
    gw_ref = np.zeros((Cout, Cin * 9))
    xs, gxs = [], []
    total = 0
    for (B, H, W) in ((2, 11, 14), (1, 7, 20)):
        x = rng.standard_normal((B, Cin, H, W)).astype(np.float32)
        off = (rng.standard_normal((B, 18, H, W)) * 3).astype(np.float32)
        xt = _t(x, dev).contiguous(memory_format=torch.channels_last).requires_grad_(True)
        y = conv(xt, _t(off, dev))
        col = O.deform_im2col(x, off, *a)
        ref = (w.reshape(Cout, -1).astype(np.float64) @ col.reshape(Cin * 9, -1).astype(np.float64)).reshape(Cout, B, H, W).transpose(1, 0, 2, 3)
        np.testing.assert_allclose(y.detach().cpu().numpy(), ref, rtol=1e-4, atol=2e-4)
        gy = rng.standard_normal(y.shape).astype(np.float32)
        total = total + (y * _t(gy, dev)).sum()
        gcol = (w.reshape(Cout, -1).T.astype(np.float64) @ gy.transpose(1, 0, 2, 3).reshape(Cout, -1).astype(np.float64)).astype(np.float32).reshape(col.shape)
        xs.append(xt)
        gxs.append(O.deform_col2im(gcol, off, x.shape, *a))
        gw_ref += gy.transpose(1, 0, 2, 3).reshape(Cout, -1).astype(np.float64) @ col.reshape(Cin * 9, -1).astype(np.float64).T
    total.backward()
    for xt, gx in zip(xs, gxs):
        np.testing.assert_allclose(xt.grad.cpu().numpy(), gx, rtol=1e-4, atol=3e-4)
    np.testing.assert_allclose(conv.weight.grad.cpu().numpy().reshape(Cout, -1), gw_ref, rtol=1e-4, atol=3e-4)
    # a second backward pass starts a fresh buffer (nothing of the first pass's gradient leaks in)
    conv.weight.grad = None
    y = conv(xs[0].detach(), _t(np.zeros((2, 18, 11, 14), np.float32), dev))
    y.sum().backward()
    ref = torch.autograd.grad(torch.nn.functional.conv2d(xs[0].detach(), conv.weight, padding=1).sum(), conv.weight)[0]
    assert torch.allclose(conv.weight.grad, ref, rtol=1e-4, atol=1e-3)


@pytest.mark.parametrize("nO,C", [(8, 256), (4, 64)])
def test_rotation_invariant_pooling_kernels_match_amax(dev, nO, C):
    """RotationInvariantPooling on a channels-last map (csrc/arf.hip: jdet_rip_forward / _backward) against the
    framework's view + amax and its gradient, ties included (half the values are rounded so that maxima repeat)."""
    from jdet_amd.ops.orn import RotationInvariantPooling
    g = torch.Generator().manual_seed(nO)
    x = torch.randn(2, C, 9, 13, generator=g)
    x[:, :, ::2] = (x[:, :, ::2] * 2).round() / 2          # ties
    pool = RotationInvariantPooling(C, nO).to(dev)
    xa = x.to(dev).contiguous(memory_format=torch.channels_last).requires_grad_(True)
    xb = x.to(dev).requires_grad_(True)                    # NCHW-contiguous: the framework path
    ya, yb = pool(xa), pool(xb)
    assert torch.equal(ya, yb)
    gy = torch.randn(ya.shape, generator=g).to(dev)
    ya.backward(gy)
    yb.backward(gy)
    assert torch.allclose(xa.grad, xb.grad, rtol=0, atol=1e-7)


@pytest.mark.parametrize("arf_config,k", [((1, 8), 3), ((8, 8), 3), ((4, 4), 1)])
def test_orconv_channels_last_bank_equals_the_contiguous_one(dev, arf_config, k):
    """ORConv2d on a channels-last map expands its filters straight into channels-last memory (jdet_arf_forward_cl) and
    reads the bank's gradient from that layout (jdet_arf_backward_cl): same output and gradients as the contiguous bank
    (the values are copies, the weight gradient a sum of the same copies in the same order: bit-equal)."""
    from jdet_amd.ops.orn import ORConv2d, active_rotating_filter
    torch.manual_seed(3)
    nOri = arf_config[0]
    conv = ORConv2d(2 * nOri, 3, kernel_size=k, padding=k // 2, arf_config=arf_config).to(dev)
    bank = active_rotating_filter(conv.weight, conv.indices)
    bank_cl = conv.rotate_arf(channels_last=True)
    assert bank_cl.is_contiguous(memory_format=torch.channels_last) or k == 1
    assert torch.equal(bank, bank_cl)
    gb = torch.randn_like(bank)
    (g0,) = torch.autograd.grad(bank, conv.weight, gb)
    (g1,) = torch.autograd.grad(bank_cl, conv.weight, gb.contiguous(memory_format=torch.channels_last))
    assert torch.equal(g0, g1)
    x = torch.randn(2, conv.in_channels * nOri, 7, 9, device=dev)
    y_cl = conv(x.contiguous(memory_format=torch.channels_last))
    y = conv(x)
    assert torch.allclose(y_cl, y, atol=1e-5)
