"""CPU: the Oriented R-CNN box algebra and coders (pure torch programs restated from the reference) against
closed-form cases: wrap ranges, decode(encode(x)) = x, vertex conventions, anchor layout."""
import math

import numpy as np
import pytest
import torch

from jdet_amd.models.boxes.anchor_generator import AnchorGenerator
from jdet_amd.models.boxes.coder import MidpointOffsetCoder, OrientedDeltaXYWHTCoder
from jdet_amd.ops import bbox_transforms as T


def test_regular_theta_and_obb():
    th = torch.tensor([-10.0, -math.pi / 2, -0.1, 0.0, math.pi / 2 - 1e-4, math.pi / 2, 4.0])
    r = T.regular_theta(th)
    assert torch.all(r >= -math.pi / 2 - 1e-6) and torch.all(r < math.pi / 2 + 1e-6)
    assert torch.allclose(torch.sin(2 * (r - th)), torch.zeros_like(r), atol=1e-4)     # differs by k*pi
    assert abs(float(T.regular_theta(torch.tensor(math.pi / 2))) + math.pi / 2) < 1e-6  # upper end wraps
    r360 = T.regular_theta(torch.tensor([7.0]), mode="360", start=-math.pi)
    assert abs(float(r360) - (7.0 - 2 * math.pi)) < 1e-5
    o = T.regular_obb(torch.tensor([[0., 0., 2., 6., 0.3], [0., 0., 6., 2., 0.3]]))
    assert torch.allclose(o[0], torch.tensor([0., 0., 6., 2., 0.3 + math.pi / 2 - math.pi]), atol=1e-5)
    assert torch.allclose(o[1], torch.tensor([0., 0., 6., 2., 0.3]), atol=1e-6)


def test_obb_poly_hbb_conventions():
    obb = torch.tensor([[10., 20., 8., 4., 0.0], [10., 20., 8., 4., math.pi / 2]])
    p = T.obb2poly(obb)
    # theta=0: point1 = centre + (w/2,0) + (0,-h/2) = (14,18); then (14,22), (6,22), (6,18)
    assert torch.allclose(p[0], torch.tensor([14., 18., 14., 22., 6., 22., 6., 18.]), atol=1e-5)
    # theta=pi/2 with the (+w/2 cos, -w/2 sin) convention: vector1 = (0,-4), vector2 = (-2, 0)
    assert torch.allclose(p[1], torch.tensor([8., 16., 12., 16., 12., 24., 8., 24.]), atol=1e-4)
    h = T.obb2hbb(obb)
    assert torch.allclose(h[0], torch.tensor([6., 18., 14., 22.]), atol=1e-5)
    assert torch.allclose(h[1], torch.tensor([8., 16., 12., 24.]), atol=1e-4)
    assert torch.allclose(T.poly2hbb(p), h, atol=1e-4)
    back = T.rectpoly2obb(p)
    assert torch.allclose(back[:, :4], torch.tensor([[10., 20., 8., 4.], [10., 20., 8., 4.]]), atol=1e-4)
    assert torch.allclose(torch.sin(2 * (back[:, 4] - obb[:, 4])), torch.zeros(2), atol=1e-4)
    hb = torch.tensor([[0., 0., 4., 10.], [0., 0., 10., 4.]])
    o = T.hbb2obb(hb)
    assert torch.allclose(o[0], torch.tensor([2., 5., 10., 4., -math.pi / 2]), atol=1e-6)
    assert torch.allclose(o[1], torch.tensor([5., 2., 10., 4., 0.]), atol=1e-6)
    assert torch.allclose(T.hbb2poly(hb)[0], torch.tensor([0., 0., 4., 0., 4., 10., 0., 10.]))
    assert T.get_bbox_type(hb) == "hbb" and T.get_bbox_type(obb) == "obb" and T.get_bbox_type(p) == "poly"
    assert T.get_bbox_dim("obb", with_score=True) == 6
    assert T.bbox2type(obb, "hbb").shape == (2, 4) and T.bbox2type(obb, "obb") is obb
    assert torch.allclose(T.get_bbox_areas(p), torch.tensor([32., 32.]), atol=1e-3)
    with pytest.raises(ValueError):
        T.get_bbox_dim("nope")


def _rand_obbs(n, seed):
    rng = np.random.default_rng(seed)
    c = rng.uniform(100, 900, (n, 2))
    wh = np.exp(rng.uniform(np.log(16), np.log(200), (n, 2)))
    wh = np.stack([wh.max(1), wh.min(1)], 1)            # regular: w >= h
    th = rng.uniform(-math.pi / 2 + 0.05, math.pi / 2 - 0.05, (n, 1))
    return torch.from_numpy(np.concatenate([c, wh, th], 1).astype(np.float32))


def test_oriented_delta_coder_round_trip():
    coder = OrientedDeltaXYWHTCoder(target_means=[0.] * 5, target_stds=[0.1, 0.1, 0.2, 0.2, 0.1])
    p, g = _rand_obbs(400, 0), _rand_obbs(400, 1)
    d = coder.encode(p, g)
    assert torch.all(d[:, 4].abs() <= (math.pi / 4 + 1e-4) / 0.1)          # picks the smaller of dtheta, dtheta+pi/2
    back = coder.decode(p, d, wh_ratio_clip=1e-6)
    assert torch.allclose(back[:, :2], g[:, :2], atol=2e-2)
    assert torch.allclose(back[:, 2:4], g[:, 2:4], rtol=2e-4)
    assert torch.allclose(torch.sin(2 * (back[:, 4] - g[:, 4])), torch.zeros(400), atol=2e-4)


def test_midpoint_offset_coder_round_trip():
    coder = MidpointOffsetCoder(target_means=[0.] * 6, target_stds=[1., 1., 1., 1., 0.5, 0.5])
    g = _rand_obbs(300, 2)
    hb = T.obb2hbb(g)
    jitter = torch.from_numpy(np.random.default_rng(3).uniform(-8, 8, (300, 4)).astype(np.float32))
    anchors = hb + jitter
    d = coder.encode(anchors, g)
    assert d.shape == (300, 6) and torch.isfinite(d).all()
    assert torch.all(d[:, 4].abs() <= 1.0 + 1e-5) and torch.all(d[:, 5].abs() <= 1.0 + 1e-5)   # |da|,|db| <= 0.5 / 0.5
    back = coder.decode(anchors, d, wh_ratio_clip=1e-6)
    assert back.shape == (300, 5)
    # decode reproduces the gt as a *regular* obb (w >= h, theta in [-pi/2, pi/2)); compare via polygons' hbb + area
    assert torch.allclose(T.obb2hbb(back), hb, atol=0.5)
    assert torch.allclose(back[:, 2] * back[:, 3], g[:, 2] * g[:, 3], rtol=2e-2)
    assert torch.allclose(back[:, :2], g[:, :2], atol=0.3)


def test_horizontal_anchor_generator():
    g = AnchorGenerator(strides=[16], ratios=[1.], scales=[1.], base_sizes=[9])
    a = g.grid_anchors([(2, 2)])[0]                                  # docstring example of the reference
    assert torch.allclose(a, torch.tensor([[-4.5, -4.5, 4.5, 4.5], [11.5, -4.5, 20.5, 4.5], [-4.5, 11.5, 4.5, 20.5],
                                           [11.5, 11.5, 20.5, 20.5]]))
    g = AnchorGenerator(strides=[4, 8, 16, 32, 64], ratios=[0.5, 1.0, 2.0], scales=[8])
    assert g.num_levels == 5 and g.num_base_anchors == [3] * 5
    b = g.base_anchors[0]
    w, h = b[:, 2] - b[:, 0], b[:, 3] - b[:, 1]
    assert torch.allclose(w * h, torch.full((3,), 32.0 ** 2), rtol=1e-5)
    assert torch.allclose(h / w, torch.tensor([0.5, 1.0, 2.0]), rtol=1e-5)
    sizes = [(256 // (2 ** i), 256 // (2 ** i)) for i in range(5)]
    assert sum(x.shape[0] for x in g.grid_anchors(sizes)) == 261888   # 1024x1024 tile
    f = g.valid_flags([(4, 4)] + [(1, 1)] * 4, (12, 16))[0].view(4, 4, 3)
    assert f[:3, :, :].all() and not f[3].any()
