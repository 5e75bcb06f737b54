"""GPU parity of the pair-merged RoIAlign forward (round 6: csrc/experimental/roi_align_pair.h -- the taps of two
neighbouring bins merged, two accumulators per row; libjdet_experimental.so, forward mode 5; a measured alternative, not a
product path): against the CPU oracle and the product kernel at the merged-tap tolerance (the bins' merged weights are the
product kernel's bit for bit; the order of a bin's fmaf chain differs)."""
import numpy as np
import pytest
import torch

from oracle import oracle as O
from tests import inputs as I

pytestmark = pytest.mark.gpu
ATOL = 2e-6


def _pair(variant, x, rois, hw, scale, n_orient=1):
    from jdet_amd import _experimental as X
    from jdet_amd import _lib as L
    N, C, H, W = x.shape
    R = rois.shape[0]
    out = torch.full((R, C) + tuple(hw), float("nan"), device=x.device).contiguous(memory_format=torch.channels_last)
    L.check(X.lib().jdet_roi_align_forward_cl_mode(5, variant, x.data_ptr(), N, C, H, W, rois.data_ptr(), R, hw[0], hw[1],
                                                   scale, 2, n_orient, None, out.data_ptr(), None, 0, L.stream_ptr(x)),
            "fwd_cl_mode 5")
    return out


def _product(variant, x, rois, hw, scale, n_orient=1):
    from jdet_amd import _lib as L
    N, C, H, W = x.shape
    R = rois.shape[0]
    out = torch.full((R, C) + tuple(hw), float("nan"), device=x.device).contiguous(memory_format=torch.channels_last)
    L.check(L.lib().jdet_roi_align_forward_cl_roi(variant, x.data_ptr(), N, C, H, W, rois.data_ptr(), R, hw[0], hw[1], scale,
                                                  2, n_orient, None, out.data_ptr(), L.stream_ptr(x)), "fwd_cl_roi")
    return out


@pytest.mark.parametrize("variant,C", [(O.V_ROT, 256), (O.V_ROT, 64), (O.V_ROT_V1, 128), (O.V_HBB0, 64), (O.V_HBB1, 192)])
@pytest.mark.parametrize("hw", [(7, 7), (4, 4), (5, 8), (8, 3), (1, 1), (2, 7)])
def test_pair_forward_matches_oracle_and_product(dev, variant, C, hw):
    rng = np.random.default_rng(500 + variant * 7 + C + hw[1])
    N, H, W, scale = 3, 40, 56, 0.25
    feat = rng.standard_normal((N, C, H, W)).astype(np.float32)
    R = 203
    rois = np.concatenate([I.rois_from_obbs(I.random_obbs(rng, R, extent=W / scale, wh=(2.0, 300.0)),
                                            rng.integers(0, N, R)), I.edge_rois(H, W, scale)], 0)
    rois[rng.random(rois.shape[0]) < 0.2, 0] = -1.0        # masked
    if variant in (O.V_HBB0, O.V_HBB1):
        rois = I.obb_to_hbb_rois(rois)
    x = torch.from_numpy(feat).to(dev).contiguous(memory_format=torch.channels_last)
    r = torch.from_numpy(rois).to(dev)
    a = _pair(variant, x, r, hw, scale).cpu().numpy()
    b = _product(variant, x, r, hw, scale).cpu().numpy()
    masked = rois[:, 0] < 0
    assert np.isnan(a[masked]).all()            # rows of masked RoIs untouched
    assert not np.isnan(a[~masked]).any()       # every other row written (bins without a valid sample: zeros)
    ref = O.roi_align_forward(variant, feat, rois[~masked], hw, scale, 2)
    np.testing.assert_allclose(a[~masked], ref, rtol=0, atol=ATOL)
    np.testing.assert_allclose(a[~masked], b[~masked], rtol=0, atol=ATOL)


@pytest.mark.parametrize("nO,C", [(8, 256), (4, 128)])
def test_pair_forward_riroi_matches_product(dev, nO, C):
    rng = np.random.default_rng(nO)
    feat = rng.standard_normal((2, C, 40, 56)).astype(np.float32)
    rois = I.rois_from_obbs(I.random_obbs(rng, 150, extent=224.0, wh=(4.0, 200.0)), rng.integers(0, 2, 150)).astype(np.float32)
    x = torch.from_numpy(feat).to(dev).contiguous(memory_format=torch.channels_last)
    r = torch.from_numpy(rois).to(dev)
    a = _pair(O.V_RI, x, r, (7, 7), 0.25, nO)
    b = _product(O.V_RI, x, r, (7, 7), 0.25, nO)
    torch.testing.assert_close(a, b, rtol=0, atol=ATOL)


def test_pair_forward_north_star_shape(dev):
    rng = np.random.default_rng(1000)
    feat = rng.standard_normal((1, 256, 256, 256)).astype(np.float32)
    rois = I.rois_from_obbs(I.random_obbs(rng, 2000), np.zeros(2000)).astype(np.float32)
    x = torch.from_numpy(feat).to(dev).contiguous(memory_format=torch.channels_last)
    r = torch.from_numpy(rois).to(dev)
    a = _pair(O.V_ROT, x, r, (7, 7), 0.25)
    b = _product(O.V_ROT, x, r, (7, 7), 0.25)
    assert not torch.isnan(a).any()
    torch.testing.assert_close(a, b, rtol=0, atol=ATOL)
    assert torch.equal(a, _pair(O.V_ROT, x, r, (7, 7), 0.25))      # no state between calls
