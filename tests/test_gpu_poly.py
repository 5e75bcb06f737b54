"""GPU: polygon IoU / polygon NMS kernels (csrc/poly_iou.hip) against the float64 restatement (oracle/poly_oracle.py;
parity unpinned by reference execution -- see its header) and closed forms."""
import math

import numpy as np
import pytest
import torch

from oracle import poly_oracle as PO

pytestmark = pytest.mark.gpu


def _quads(rng, n, extent, convex=True):
    out = []
    while len(out) < n:
        c = rng.uniform(0, extent, 2)
        ang = np.sort(rng.uniform(0, 2 * math.pi, 4))
        r = rng.uniform(4, 30, 4)
        q = (c + np.stack([r * np.cos(ang), r * np.sin(ang)], 1))
        if not convex:
            q[2] = c + 0.3 * (q[2] - c)          # pull one vertex towards the centre: a dart
        if PO.is_convex(q.reshape(8)) == convex:
            out.append(q.reshape(8) if rng.uniform() < 0.5 else q[::-1].reshape(8))   # both orientations
    return np.stack(out)


@pytest.mark.parametrize("convex", [True, False])
@pytest.mark.parametrize("offset", [0.0, 3000.0])
def test_poly_iou_matrix_vs_oracle(dev, convex, offset):
    """random quadrilaterals, both windings, packed so that most pairs overlap; far from the origin too (the fan is
    taken about the pair's centroid, so fp32 holds up at DOTA-size coordinates)"""
    from jdet_amd.ops.nms_poly import poly_iou_matrix
    rng = np.random.default_rng(11 + int(convex))
    a = _quads(rng, 40, 60.0, convex) + offset
    b = _quads(rng, 50, 60.0, True) + offset
    for mode in (0, 1):
        ref = PO.poly_iou_matrix(a, b, mode)
        got = poly_iou_matrix(torch.from_numpy(a.astype(np.float32)).to(dev),
                              torch.from_numpy(b.astype(np.float32)).to(dev), mode).cpu().numpy()
        assert (ref > 0.05).mean() > 0.05
        np.testing.assert_allclose(got, ref, rtol=0, atol=2e-4 if offset else 2e-5)


def test_poly_iou_closed_forms_and_degenerate(dev):
    from jdet_amd.ops.nms_poly import iou_poly, poly_iou_matrix
    sq = np.array([0, 0, 2, 0, 2, 2, 0, 2], np.float32)
    assert iou_poly(sq, sq) == 1.0
    assert abs(iou_poly(sq, sq + np.tile([1.0, 0.0], 4).astype(np.float32)) - 1.0 / 3.0) < 1e-6
    assert iou_poly(sq, sq + 5) == 0.0
    diamond = np.array([1, 0, 2, 1, 1, 2, 0, 1], np.float32)
    assert abs(iou_poly(sq, diamond) - 0.5) < 1e-6
    point = np.zeros(8, np.float32)
    m = poly_iou_matrix(torch.from_numpy(np.stack([point, sq])).to(dev), torch.from_numpy(point[None]).to(dev), 0)
    assert float(m[0, 0]) == 1.0 and float(m[1, 0]) == 0.0      # kernel rule: union == 0 -> 1
    m = poly_iou_matrix(torch.from_numpy(point[None]).to(dev), torch.from_numpy(point[None]).to(dev), 1)
    assert float(m[0, 0]) == 0.0                                 # iou_poly's rule: 0 / max(0, 0.01)
    assert poly_iou_matrix(torch.zeros((0, 8), device=dev), torch.from_numpy(sq[None]).to(dev)).shape == (0, 1)


def test_poly_nms_and_multiclass_vs_oracle(dev):
    from jdet_amd.ops.nms_poly import multiclass_poly_nms, poly_nms
    rng = np.random.default_rng(5)
    base = np.array([0, 0, 30, 0, 30, 12, 0, 12], np.float64).reshape(4, 2)
    polys = []
    for _ in range(300):
        ang = rng.uniform(-0.5, 0.5)
        c, s = math.cos(ang), math.sin(ang)
        p = base @ np.array([[c, s], [-s, c]]) + rng.uniform(0, 120, 2)
        p[rng.integers(0, 4)] += rng.uniform(-2, 2, 2)           # not a rectangle any more
        polys.append(p.reshape(8))
    polys = np.stack(polys)
    scores = rng.uniform(0, 1, 300)
    labels = rng.integers(0, 3, 300)
    thr = 0.25
    iou = PO.poly_iou_matrix(polys, polys, 0)
    assert np.abs(iou - thr).min() > 1e-4                         # no decision within fp32 reach of the threshold
    tp = torch.from_numpy(polys.astype(np.float32)).to(dev)
    ts = torch.from_numpy(scores.astype(np.float32)).to(dev)
    keep = poly_nms(torch.cat([tp, ts[:, None]], 1), thr).cpu().numpy().tolist()
    assert keep == PO.poly_nms(polys, scores.astype(np.float32), thr)
    dets, lab = multiclass_poly_nms(tp, ts, torch.from_numpy(labels).to(dev), thr)
    ref = PO.poly_nms(polys, scores.astype(np.float32), thr, labels=labels)
    assert dets.shape == (len(ref), 9) and lab.cpu().numpy().tolist() == labels[ref].tolist()
    np.testing.assert_array_equal(dets[:, :8].cpu().numpy(), polys[ref].astype(np.float32))
    assert poly_nms(torch.zeros((0, 9), device=dev), thr).numel() == 0


def test_evaluation_and_merging_take_general_quadrilaterals(dev):
    """device_iou_matrix / device_group_nms route non-rectangles to the polygon kernels (polygon front-end
    for foreign result files): equal to the restatement; rectangles keep the rotated-box route"""
    from jdet_amd.data.np_boxes import polys_are_rectangles
    from jdet_amd.data.result_merge import device_group_nms
    from jdet_amd.data.voc_eval import device_iou_matrix
    rng = np.random.default_rng(9)
    a, b = _quads(rng, 30, 80.0), _quads(rng, 20, 80.0)
    assert not polys_are_rectangles(a)
    np.testing.assert_allclose(device_iou_matrix(a, b, dev), PO.poly_iou_matrix(a, b, 1), rtol=0, atol=2e-5)
    scores = rng.uniform(0, 1, 30)
    groups = rng.integers(0, 2, 30)
    keep = device_group_nms(a, scores, groups, 0.2, device=dev)
    ref = sorted(PO.poly_nms(a, scores.astype(np.float32), 0.2, labels=groups))
    assert np.flatnonzero(keep).tolist() == ref
