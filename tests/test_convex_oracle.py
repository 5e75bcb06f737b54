"""CPU: pins of the RepPoints geometry restatement (oracle/jdet_oracle.cpp: jo_convex_iou, jo_min_area_bbox,
jo_convex_sort).  The reference text is CUDA only; the restatement is held to exact geometry: scipy's Qhull hulls,
analytic overlaps of rectangles, the pinned rotated-box IoU oracle, rectangles recovered from their own corners."""
import numpy as np
import pytest
from scipy.spatial import ConvexHull

from oracle import oracle as O


def _cyclic_equal(a, b, tol=1e-6):
    if len(a) != len(b):
        return False
    for s in range(len(b)):
        if np.abs(np.roll(b, s, axis=0) - a).max() <= tol:
            return True
    return False


def _rect_corners(cx, cy, w, h, t):
    c, s = np.cos(t), np.sin(t)
    d = np.asarray([[-w / 2, -h / 2], [w / 2, -h / 2], [w / 2, h / 2], [-w / 2, h / 2]])
    return d @ np.asarray([[c, s], [-s, c]]) + [cx, cy]


def _pointset_in_rect(rng, cx, cy, w, h, t):
    """the 4 corners + 5 interior points, shuffled: the hull is the rectangle"""
    corners = _rect_corners(cx, cy, w, h, t)
    u = rng.uniform(-0.4, 0.4, size=(5, 2)) * [w, h]
    c, s = np.cos(t), np.sin(t)
    inner = u @ np.asarray([[c, s], [-s, c]]) + [cx, cy]
    pts = np.concatenate([corners, inner], 0)
    return pts[rng.permutation(9)].reshape(18).astype(np.float32)


def test_hull_is_the_convex_hull_counter_clockwise_from_the_lowest_point():
    rng = np.random.default_rng(1)
    ps = rng.uniform(0, 100, size=(200, 18)).astype(np.float32)
    for p, h in zip(ps, O.convex_hull9(ps)):
        pts = p.reshape(9, 2).astype(np.float64)
        ref = pts[ConvexHull(pts).vertices]                  # counter-clockwise
        assert _cyclic_equal(h.astype(np.float64), ref), (h, ref)
        low = np.lexsort((pts[:, 0], pts[:, 1]))[0]
        assert np.array_equal(h[0].astype(np.float64), pts[low])


def test_hull_drops_collinear_and_duplicate_points():
    p = np.asarray([[0, 0, 4, 0, 2, 0, 4, 4, 0, 4, 2, 4, 2, 2, 0, 0, 4, 2]], np.float32)   # square + edge / inner points
    (h,) = O.convex_hull9(p)
    assert _cyclic_equal(h.astype(np.float64), np.asarray([[0, 0], [4, 0], [4, 4], [0, 4]], np.float64))


def test_convex_iou_axis_aligned_cases():
    rng = np.random.default_rng(2)
    ps = np.stack([_pointset_in_rect(rng, 10, 10, 8, 4, 0.0)])           # [6,14] x [8,12], area 32
    quads = np.asarray([
        [6, 8, 14, 8, 14, 12, 6, 12],          # identical              -> 1
        [6, 12, 14, 12, 14, 8, 6, 8],          # identical, clockwise   -> 1
        [10, 8, 18, 8, 18, 12, 10, 12],        # half overlap: 16 / 48
        [20, 20, 24, 20, 24, 24, 20, 24],      # disjoint               -> 0
        [8, 9, 12, 9, 12, 11, 8, 11],          # contained: 8 / 32
        [0, 0, 30, 0, 30, 30, 0, 30],          # containing: 32 / 900
    ], np.float32)
    np.testing.assert_allclose(O.convex_iou(ps, quads)[0], [1, 1, 16 / 48, 0, 0.25, 32 / 900], rtol=0, atol=1e-6)


def test_convex_iou_of_rotated_rectangles_equals_the_rotated_box_iou():
    rng = np.random.default_rng(3)
    n = 60
    b1 = np.stack([rng.uniform(30, 70, n), rng.uniform(30, 70, n), rng.uniform(10, 40, n), rng.uniform(5, 30, n),
                   rng.uniform(-np.pi / 2, np.pi / 2, n)], 1)
    b2 = b1 + np.stack([rng.normal(0, 6, n), rng.normal(0, 6, n), rng.normal(0, 3, n), rng.normal(0, 3, n),
                        rng.normal(0, 0.4, n)], 1)
    b2[:, 2:4] = np.abs(b2[:, 2:4]) + 2
    ps = np.stack([_pointset_in_rect(rng, *b) for b in b1])
    quads = np.stack([_rect_corners(*b).reshape(8) for b in b2]).astype(np.float32)
    got = O.convex_iou(ps, quads)
    ref = O.box_iou_rotated(b1.astype(np.float32), b2.astype(np.float32))
    np.testing.assert_allclose(got, ref, rtol=0, atol=2e-5)


def test_min_area_bbox_recovers_a_rectangle_from_its_corners():
    rng = np.random.default_rng(4)
    for _ in range(50):
        cx, cy, w, h, t = rng.uniform(20, 80), rng.uniform(20, 80), rng.uniform(8, 40), rng.uniform(4, 30), \
            rng.uniform(-np.pi / 2, np.pi / 2)
        box = O.min_area_bbox(_pointset_in_rect(rng, cx, cy, w, h, t)[None])[0].reshape(4, 2).astype(np.float64)
        ref = _rect_corners(cx, cy, w, h, t)
        # same corner set (order / starting corner are the rotated frame's), within float32 trigonometry
        d = np.abs(box[:, None, :] - ref[None, :, :]).sum(-1)
        assert d.min(1).max() < 2e-3 and d.min(0).max() < 2e-3, (box, ref)


def test_min_area_bbox_is_minimal_over_edge_aligned_rectangles_and_contains_the_points():
    rng = np.random.default_rng(5)
    ps = rng.uniform(0, 100, size=(100, 18)).astype(np.float32)
    boxes = O.min_area_bbox(ps).reshape(-1, 4, 2).astype(np.float64)
    for p, b in zip(ps, boxes):
        pts = p.reshape(9, 2).astype(np.float64)
        hull = pts[ConvexHull(pts).vertices]
        e0, e1 = b[0] - b[1], b[2] - b[1]                       # (xmax,ymin)-(xmin,ymin), (xmin,ymax)-(xmin,ymin)
        area = np.linalg.norm(e0) * np.linalg.norm(e1)
        assert abs(np.dot(e0, e1)) < 1e-2 * area                # a rectangle
        best = np.inf
        for i in range(len(hull)):
            d = hull[(i + 1) % len(hull)] - hull[i]
            d = d / np.linalg.norm(d)
            u, v = hull @ d, hull @ np.asarray([-d[1], d[0]])
            best = min(best, (u.max() - u.min()) * (v.max() - v.min()))
        assert area == pytest.approx(best, rel=2e-4)
        # every point inside (coordinates along the rectangle's axes)
        a0, a1 = e0 / np.linalg.norm(e0), e1 / np.linalg.norm(e1)
        u, v = (pts - b[1]) @ a0, (pts - b[1]) @ a1
        tol = 1e-3 * max(np.linalg.norm(e0), np.linalg.norm(e1))
        assert u.min() > -tol and u.max() < np.linalg.norm(e0) + tol and v.min() > -tol and v.max() < np.linalg.norm(e1) + tol


@pytest.mark.parametrize("circular", [True, False])
def test_convex_sort_is_the_masked_hull(circular):
    rng = np.random.default_rng(6)
    nbs, npts = 80, 12
    pts = rng.uniform(0, 50, size=(nbs, npts, 2)).astype(np.float32)
    masks = (rng.uniform(size=(nbs, npts)) > 0.3).astype(np.float32)
    masks[:, :4] = 1                                           # at least a few live points
    idx = O.convex_sort(pts, masks, circular)
    assert idx.shape == (nbs, npts + (1 if circular else 0))
    for b in range(nbs):
        live = np.nonzero(masks[b] > 0.5)[0]
        hull = live[ConvexHull(pts[b, live].astype(np.float64)).vertices]       # counter-clockwise original indices
        row = idx[b]
        used = row[:len(hull)]                     # the stack; slots past its end keep what earlier, deeper states
        if circular:                               # of the stack left there (the reference never clears them) or -1
            assert row[len(hull)] == row[0]
        assert set(used) == set(hull) and _cyclic_equal(pts[b, used].astype(np.float64), pts[b, hull].astype(np.float64))
        low = live[np.lexsort((np.arange(len(live)), pts[b, live, 1]))[0]]      # first lowest live point
        assert used[0] == low


def test_convex_sort_skips_duplicates_and_handles_tiny_sets():
    pts = np.asarray([[[0, 0], [2, 0], [2, 0], [1, 2], [0, 0]]], np.float32)
    idx = O.convex_sort(pts, np.ones((1, 5), np.float32), True)[0]
    # the duplicate of the stack top (index 2) is skipped; the duplicate of the START (index 4) is only compared with
    # the top of the stack and therefore pushed -- literally what the reference scan does
    assert list(idx) == [0, 1, 3, 4, 0, -1]
    one = O.convex_sort(pts, np.asarray([[0, 0, 0, 1, 0]], np.float32), True)[0]
    assert list(one) == [3, 3, -1, -1, -1, -1]
    assert O.convex_sort(np.zeros((2, 0, 2), np.float32), np.zeros((2, 0), np.float32), True).shape == (2, 1)


def test_convex_giou_definition_closed_forms():
    """oracle/convex_giou_oracle.py on squares: identical -> 1; half overlap -> I 2, U 6, C 6 -> 1/3; disjoint with a
    gap -> 0 - (10 - 8)/10; gradients by central differences vanish for interior points and push an overlapped hull
    corner the way that grows the intersection"""
    from oracle import convex_giou_oracle as G
    sq = np.array([[0, 0], [2, 0], [2, 2], [0, 2], [1, 1], [0.5, 0.5], [1.5, 0.5], [0.7, 1.2], [1.3, 1.6]], float).reshape(-1)
    assert abs(G.giou_value(sq, [0, 0, 2, 0, 2, 2, 0, 2]) - 1.0) < 1e-12
    assert abs(G.giou_value(sq, [1, 0, 3, 0, 3, 2, 1, 2]) - 1.0 / 3.0) < 1e-12
    assert abs(G.giou_value(sq, [3, 0, 5, 0, 5, 2, 3, 2]) + 0.2) < 1e-12
    assert abs(G.giou_value(sq, [1, 0, 1, 2, 3, 2, 3, 0]) - 1.0 / 3.0) < 1e-12        # clockwise quadrilateral
    v, g = G.convex_giou(sq[None], [[1, 0, 3, 0, 3, 2, 1, 2]])
    assert np.abs(g[0, 8:]).max() < 1e-9                      # the five interior points
    assert g[0, 2] > 0 and g[0, 4] > 0                        # corners (2,0), (2,2): moving right grows I
