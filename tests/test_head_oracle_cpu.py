"""CPU: the pieces of oracle/head_oracle.py against independent statements (torch CPU ops, hand cases), so that the
head-level GPU parity tests compare against a checker whose losses / overlaps / selection rules are themselves held."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import head_oracle as HO


def test_sigmoid_focal_loss_equals_the_textbook_form():
    rng = np.random.default_rng(1)
    n, C = 200, 15
    pred = rng.normal(0, 2, (n, C)).astype(np.float32)
    target = rng.integers(0, C + 1, n)                       # 0 = background, c = class c (column c - 1)
    weight = rng.uniform(0, 1, n).astype(np.float32)
    t = torch.zeros(n, C, dtype=torch.float64)
    fg = torch.from_numpy(target) > 0
    t[fg, torch.from_numpy(target)[fg] - 1] = 1
    p = torch.from_numpy(pred).double()
    ce = F.binary_cross_entropy_with_logits(p, t, reduction="none")
    pt = torch.sigmoid(p) * t + (1 - torch.sigmoid(p)) * (1 - t)
    ref = ((0.25 * t + 0.75 * (1 - t)) * (1 - pt) ** 2.0 * ce * torch.from_numpy(weight).double()[:, None]).sum() / 37.0
    assert HO.sigmoid_focal_loss(pred, target, weight, 2.0, 0.25, 37.0) == pytest.approx(float(ref), rel=1e-9)


@pytest.mark.parametrize("beta", [1.0 / 9.0, 1.0])
def test_smooth_l1_and_l1_equal_torch(beta):
    rng = np.random.default_rng(2)
    a, b = rng.normal(0, 1, (50, 5)).astype(np.float32), rng.normal(0, 1, (50, 5)).astype(np.float32)
    w = rng.uniform(0, 1, (50, 5)).astype(np.float32)
    ref = (F.smooth_l1_loss(torch.from_numpy(a).double(), torch.from_numpy(b).double(), beta=beta, reduction="none")
           * torch.from_numpy(w).double()).sum() / 11.0
    assert float(HO.smooth_l1_loss(a, b, w, beta, 11.0)) == pytest.approx(float(ref), rel=1e-9)
    ref1 = ((torch.from_numpy(a).double() - torch.from_numpy(b).double()).abs() * torch.from_numpy(w).double()).sum() / 11.0
    assert HO.l1_loss(a, b, w, 11.0) == pytest.approx(float(ref1), rel=1e-9)


def test_softmax_cross_entropy_equals_torch():
    rng = np.random.default_rng(3)
    z = rng.normal(0, 3, (64, 16)).astype(np.float32)
    y = rng.integers(0, 16, 64)
    w = rng.uniform(0, 1, 64).astype(np.float32)
    ref = (F.cross_entropy(torch.from_numpy(z).double(), torch.from_numpy(y), reduction="none")
           * torch.from_numpy(w).double()).sum() / 64.0
    assert HO.softmax_cross_entropy(z, y, w, 64.0) == pytest.approx(float(ref), rel=1e-9)


def test_hbb_overlaps_literals_and_plus_one_convention():
    a = np.asarray([[0, 0, 10, 10], [10, 10, 20, 20], [32, 32, 38, 42]], np.float32)       # iou_calculator.py:L259-268
    b = np.asarray([[0, 0, 10, 20], [0, 10, 10, 19], [10, 10, 20, 20]], np.float32)
    v0 = HO.hbb_overlaps(a, b, version=0)
    assert v0[0, 0] == pytest.approx(0.5) and v0[1, 2] == pytest.approx(1.0) and v0[2, 0] == 0
    assert v0[0, 2] == 0                                      # touching corners: zero-area intersection
    v1 = HO.hbb_overlaps(a, b, version=1)
    assert v1[0, 2] == pytest.approx(1.0 / (121 + 121 - 1))   # +1 px: the shared corner pixel counts
    assert v1[0, 0] == pytest.approx(121.0 / 231.0)


def test_multiclass_nms_rotated_hand_case():
    boxes = np.asarray([[10, 10, 8, 4, 0.0], [10.5, 10, 8, 4, 0.0], [40, 40, 8, 4, 0.3], [10, 10, 8, 4, 0.0]], np.float32)
    scores = np.zeros((4, 3), np.float32)                      # column 0 = background
    scores[0, 1], scores[1, 1], scores[2, 1], scores[3, 2] = 0.9, 0.8, 0.7, 0.6
    scores[2, 2] = 0.04                                        # under the threshold
    b, s, l = HO.multiclass_nms_rotated(boxes, scores, 0.05, 0.1, 100)
    # class 0: box 1 overlaps box 0 and goes; box 2 stays.  class 1: box 3 has the class to itself (no cross-class NMS)
    assert list(s) == pytest.approx([0.9, 0.7, 0.6]) and list(l) == [0, 0, 1]
    assert np.array_equal(b[0], boxes[0]) and np.array_equal(b[2], boxes[3])
    assert HO.multiclass_nms_rotated(boxes, scores, 0.05, 0.1, 2)[1].shape == (2,)        # max_num cut, best first
    assert HO.multiclass_nms_rotated(boxes, scores * 0, 0.05, 0.1, 100)[0].shape == (0, 5)


def test_hbb_nms_is_greedy_in_score_order():
    boxes = np.asarray([[0, 0, 10, 10], [1, 1, 11, 11], [20, 20, 30, 30], [0, 0, 10, 10.5]], np.float32)
    scores = np.asarray([0.5, 0.9, 0.3, 0.8], np.float32)
    assert list(HO.hbb_nms(boxes, scores, 0.5)) == [1, 2]     # 3 and 0 overlap the winner above 0.5
    assert list(HO.hbb_nms(boxes, scores, 0.95)) == [1, 3, 2]          # IoU(3, 0) = 100 / 105 = 0.952
    assert list(HO.hbb_nms(boxes, scores, 0.96)) == [1, 3, 0, 2]


def test_retina_anchor_order_and_scales():
    (a,) = HO.retina_anchors([8], [(2, 3)])
    assert a.shape == (2 * 3 * 9, 5)
    # location-major, anchor-fastest; first location centred at (3.5, 3.5); ratio 1.0 first with its three octave scales
    np.testing.assert_allclose(a[0], [3.5, 3.5, 32, 32, 0], atol=1e-5)
    np.testing.assert_allclose(a[1, 2:4], [32 * 2 ** (1 / 3)] * 2, rtol=1e-6)
    np.testing.assert_allclose(a[3, 2:4], [32 / np.sqrt(0.5), 32 * np.sqrt(0.5)], rtol=1e-6)     # ratio 0.5: h = sqrt(r) * s
    np.testing.assert_allclose(a[9, :2], [11.5, 3.5], atol=1e-5)                                  # next location: x + stride
    np.testing.assert_allclose(a[27, :2], [3.5, 11.5], atol=1e-5)                                 # next row
