"""GPU parity of the footprint-staged RoIAlign forward (round 6: csrc/roi_align_stage.h -- distinct pixels of a line of
bins fetched once by LDS-DMA, taps served from LDS; libjdet_experimental.so, forward mode 4): against the CPU oracle and
the product's merged-tap kernel, merged-tap tolerance (2e-6 abs on N(0,1) maps: same weights, another summation order).
Covers the five dialects without orientation planes, several images, masked RoIs (rows untouched), RoIs over the border,
bin grids other than 7 x 7 (run-time geometry), RoIs too large for the line bitmaps (in-kernel direct path), both channel
pass widths, and the north-star shape."""
import os

import numpy as np
import pytest
import torch

from oracle import oracle as O
from tests import inputs as I

pytestmark = pytest.mark.gpu
ATOL = 2e-6


def _staged(variant, x, rois, hw, scale, order=None):
    from jdet_amd import _experimental as X
    from jdet_amd import _lib as L
    N, C, H, W = x.shape
    R = rois.shape[0]
    out = torch.full((R, C) + tuple(hw), float("nan"), device=x.device).contiguous(memory_format=torch.channels_last)
    L.check(X.lib().jdet_roi_align_forward_cl_mode(4, variant, x.data_ptr(), N, C, H, W, rois.data_ptr(), R, hw[0], hw[1],
                                                   scale, 2, 1, order.data_ptr() if order is not None else None,
                                                   out.data_ptr(), None, 0, L.stream_ptr(x)), "fwd_cl_mode 4")
    return out


def _product(variant, x, rois, hw, scale):
    from jdet_amd import _lib as L
    N, C, H, W = x.shape
    R = rois.shape[0]
    out = torch.full((R, C) + tuple(hw), float("nan"), device=x.device).contiguous(memory_format=torch.channels_last)
    L.check(L.lib().jdet_roi_align_forward_cl_roi(variant, x.data_ptr(), N, C, H, W, rois.data_ptr(), R, hw[0], hw[1],
                                                  scale, 2, 1, None, out.data_ptr(), L.stream_ptr(x)), "fwd_cl_roi")
    return out


@pytest.mark.parametrize("variant,C", [(O.V_ROT, 256), (O.V_ROT, 64), (O.V_ROT_V1, 128), (O.V_HBB0, 64), (O.V_HBB1, 192)])
@pytest.mark.parametrize("hw", [(7, 7), (4, 4), (5, 8), (8, 3)])
def test_staged_forward_matches_oracle_and_product(dev, variant, C, hw):
    rng = np.random.default_rng(300 + variant * 7 + C + hw[1])
    N, H, W, scale = 3, 40, 56, 0.25
    feat = rng.standard_normal((N, C, H, W)).astype(np.float32)
    R = 203
    rois = np.concatenate([I.rois_from_obbs(I.random_obbs(rng, R, extent=W / scale, wh=(4.0, 200.0)),
                                            rng.integers(0, N, R)), I.edge_rois(H, W, scale)], 0)
    rois[rng.random(rois.shape[0]) < 0.2, 0] = -1.0        # masked
    if variant in (O.V_HBB0, O.V_HBB1):
        rois = I.obb_to_hbb_rois(rois)
    x = torch.from_numpy(feat).to(dev).contiguous(memory_format=torch.channels_last)
    r = torch.from_numpy(rois).to(dev)
    a = _staged(variant, x, r, hw, scale).cpu().numpy()
    b = _product(variant, x, r, hw, scale).cpu().numpy()
    masked = rois[:, 0] < 0
    assert np.isnan(a[masked]).all()
    assert not np.isnan(a[~masked]).any()
    ref = O.roi_align_forward(variant, feat, rois[~masked], hw, scale, 2)
    np.testing.assert_allclose(a[~masked], ref, rtol=0, atol=ATOL)
    np.testing.assert_allclose(a[~masked], b[~masked], rtol=0, atol=ATOL)


def test_staged_forward_rois_beyond_the_line_bitmaps(dev):
    """sides of 75 ... 300 map pixels: a line's tap box exceeds 224 bitmap words -> the direct path inside the launch"""
    rng = np.random.default_rng(5)
    feat = rng.standard_normal((1, 64, 200, 300)).astype(np.float32)
    rois = I.rois_from_obbs(I.random_obbs(rng, 64, extent=1000.0, wh=(300.0, 1200.0)), np.zeros(64))
    x = torch.from_numpy(feat).to(dev).contiguous(memory_format=torch.channels_last)
    a = _staged(O.V_ROT, x, torch.from_numpy(rois).to(dev), (7, 7), 0.25).cpu().numpy()
    ref = O.roi_align_forward(O.V_ROT, feat, rois, (7, 7), 0.25, 2)
    np.testing.assert_allclose(a, ref, rtol=0, atol=ATOL)


def test_staged_forward_north_star_shape(dev):
    """1 x 256 x 256 x 256 map, 2000 RoIs, 7 x 7 under the XCD schedule: equal to the product kernel at the merged-tap
    tolerance, no row left unwritten, and a repeat gives the same bits (no state between calls)"""
    from jdet_amd.ops._roi_common import spatial_order
    rng = np.random.default_rng(0)
    R = 2000
    rois = torch.from_numpy(I.rois_from_obbs(I.random_obbs(rng, R), np.zeros(R))).to(dev)
    x = torch.randn(1, 256, 256, 256, device=dev).contiguous(memory_format=torch.channels_last)
    order = spatial_order(rois, 0.25, 1, 256, 256)
    a = _staged(O.V_ROT, x, rois, (7, 7), 0.25, order)
    b = _product(O.V_ROT, x, rois, (7, 7), 0.25)
    assert not torch.isnan(a).any()
    assert float((a - b).abs().max()) <= ATOL
    assert torch.equal(a, _staged(O.V_ROT, x, rois, (7, 7), 0.25, order))
