"""GPU parity of the EXPERIMENTAL register-cached RoIAlign forward (csrc/experimental/roi_align_pool.hip: plan kernel +
persistent pool kernel, libjdet_experimental.so -- not a product path) vs the CPU oracle.  Tolerance: 2e-6 abs on
N(0,1) maps (merged-tap weights, fma), the bound of the product's merged-tap kernel."""
import numpy as np
import pytest
import torch

from oracle import oracle as O
from tests import inputs as I

pytestmark = pytest.mark.gpu
ATOL = 2e-6


def _pool(variant, feat, rois, hw, scale, s, dev, fill=None):
    from jdet_amd import _experimental as X
    from jdet_amd import _lib as L
    lib = X.lib()
    N, C, H, W = feat.shape
    R = rois.shape[0]
    assert lib.jdet_roi_align_forward_pool_supported(variant, C, H, W, hw[0], hw[1], s) == 1
    x = torch.from_numpy(feat).to(dev).contiguous(memory_format=torch.channels_last)
    r = torch.from_numpy(rois).to(dev)
    out = torch.empty((R, C) + tuple(hw), device=dev).contiguous(memory_format=torch.channels_last)
    if fill is not None:
        out.fill_(fill)
    wsb = lib.jdet_roi_align_forward_pool_workspace(R)
    ws = torch.full((wsb,), 0xA5, dtype=torch.uint8, device=dev)     # no contract on the workspace contents
    L.check(lib.jdet_roi_align_forward_pool(variant, x.data_ptr(), N, C, H, W, r.data_ptr(), R, hw[0], hw[1], scale, s,
                                            out.data_ptr(), ws.data_ptr(), wsb, L.stream_ptr(x)), "pool")
    torch.cuda.synchronize()
    return out.cpu().numpy()


@pytest.mark.parametrize("variant,C,hw,s", [
    (O.V_ROT, 128, (7, 7), 2), (O.V_ROT, 256, (7, 7), 1), (O.V_ROT, 512, (8, 8), 2), (O.V_ROT_V1, 256, (7, 7), 2),
    (O.V_ROT_V1, 128, (3, 5), 2), (O.V_HBB0, 256, (7, 7), 2), (O.V_HBB0, 128, (1, 1), 2), (O.V_HBB1, 512, (7, 7), 2),
    (O.V_HBB1, 128, (8, 8), 1)])
def test_pool_vs_oracle(dev, variant, C, hw, s):
    """all four dialects, 1 / 2 / 4 channel slices of 128, odd bin grids up to the 64-bin limit,
    RoIs from sub-pixel to larger than the map, masked rows."""
    rng = np.random.default_rng(1000 * variant + C + hw[0])
    N, H, W, scale = 2, 40, 48, 0.25
    feat = rng.standard_normal((N, C, H, W)).astype(np.float32)
    obbs = I.random_obbs(rng, 90, extent=W / scale, wh=(2.0, 200.0))
    rois = np.concatenate([I.rois_from_obbs(obbs, rng.integers(0, N, 90)), I.edge_rois(H, W, scale)], 0)
    if variant in (O.V_HBB0, O.V_HBB1):
        rois = I.obb_to_hbb_rois(rois)
    rois[7, 0] = -1.0
    rois[40, 0] = -3.0
    y = _pool(variant, feat, rois, hw, scale, s, dev, fill=5.0)
    live = rois[:, 0] >= 0
    ref = O.roi_align_forward(variant, feat, rois[live], hw, scale, s)
    np.testing.assert_allclose(y[live], ref, rtol=0, atol=ATOL)
    assert (y[~live] == 5.0).all()


def test_pool_footprints_beyond_the_dedup_table(dev):
    """RoIs whose footprint exceeds the plan kernel's 92x92 position table take the no-sharing plan (every bin's
    taps get fresh slots): 300-pixel boxes on a 128x160 map, mixed with tiny ones (whole RoI = one group)."""
    rng = np.random.default_rng(11)
    N, C, H, W, scale = 1, 128, 128, 160, 1.0
    feat = rng.standard_normal((N, C, H, W)).astype(np.float32)
    big = I.random_obbs(rng, 40, extent=float(W), wh=(90.0, 300.0))
    tiny = I.random_obbs(rng, 40, extent=float(W), wh=(0.5, 6.0))
    rois = I.rois_from_obbs(np.concatenate([big, tiny], 0), np.zeros(80))
    y = _pool(O.V_ROT, feat, rois, (7, 7), scale, 2, dev)
    np.testing.assert_allclose(y, O.roi_align_forward(O.V_ROT, feat, rois, (7, 7), scale, 2), rtol=0, atol=ATOL)


def test_pool_cfg0_micro(dev):
    """BASELINE configs[0]: 1x256x256x256 map, 512 random OBBs, 7x7, sampling 2, scale 0.25 (R >= 64: XCD schedule on)"""
    rng = np.random.default_rng(0)
    feat = rng.standard_normal((1, 256, 256, 256)).astype(np.float32)
    rois = I.rois_from_obbs(I.random_obbs(rng, 512), np.zeros(512))
    y = _pool(O.V_ROT, feat, rois, (7, 7), 0.25, 2, dev)
    O.set_threads(8)
    np.testing.assert_allclose(y, O.roi_align_forward(O.V_ROT, feat, rois, (7, 7), 0.25, 2), rtol=0, atol=ATOL)
