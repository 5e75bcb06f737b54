"""CPU: the float64 polygon-IoU restatement (oracle/poly_oracle.py) against closed-form areas, its two independent
intersection routines against each other, and the host-side rectangle test that routes evaluation / merging."""
import math

import numpy as np

from oracle import poly_oracle as PO


def _rot(poly, ang, about=(0.0, 0.0)):
    p = np.asarray(poly, np.float64).reshape(4, 2) - about
    c, s = math.cos(ang), math.sin(ang)
    return (p @ np.array([[c, s], [-s, c]]) + about).reshape(8)


SQ = np.array([0, 0, 2, 0, 2, 2, 0, 2], np.float64)


def test_closed_form_areas():
    assert PO.poly_iou(SQ, SQ) == 1.0
    assert PO.poly_iou(SQ, SQ[::-1].reshape(4, 2)[:, ::-1].reshape(8)) == 1.0          # clockwise copy
    shifted = SQ + np.tile([1.0, 0.0], 4)
    assert abs(PO.poly_iou(SQ, shifted) - (2.0 / 6.0)) < 1e-12
    assert PO.poly_iou(SQ, SQ + 5) == 0.0
    inner = np.array([0.5, 0.5, 1.5, 0.5, 1.5, 1.5, 0.5, 1.5])
    assert abs(PO.poly_iou(SQ, inner) - 0.25) < 1e-12
    # unit-diagonal diamond inside the square: area 2 of 4
    diamond = np.array([1, 0, 2, 1, 1, 2, 0, 1], np.float64)
    assert abs(PO.poly_iou(SQ, diamond) - 0.5) < 1e-12
    # square rotated by 45 degrees about its centre: octagon of area 8 (sqrt 2 - 1) * ... = 4 * 2 (sqrt2 - 1)
    r45 = _rot(SQ, math.pi / 4, about=(1.0, 1.0))
    inter = 8.0 * (math.sqrt(2.0) - 1.0)
    assert abs(PO.poly_iou(SQ, r45) - inter / (8.0 - inter)) < 1e-12
    # non-convex chevron (arrow head) against a rectangle: only the fan routine handles it
    chevron = np.array([0, 0, 2, 1, 4, 0, 2, 3], np.float64)     # area 4
    assert not PO.is_convex(chevron) and abs(abs(PO._signed_area(chevron.reshape(4, 2))) - 4.0) < 1e-12
    strip = np.array([0, 0, 4, 0, 4, 1, 0, 1], np.float64)       # the band 0 <= y <= 1 cuts both prongs
    # expected area by quadrature: the chevron lies between y = x/2 | (4-x)/2 (below) and y = 1.5x | 1.5(4-x) (above)
    xs = np.linspace(0, 4, 400001)
    lower = np.where(xs <= 2, xs / 2, (4 - xs) / 2)
    upper = np.where(xs <= 2, 1.5 * xs, 1.5 * (4 - xs))
    seg = np.clip(np.minimum(upper, 1.0) - lower, 0, None)
    quad = float(np.sum((seg[1:] + seg[:-1]) / 2) * (xs[1] - xs[0]))
    got = PO.intersection_fan(chevron, strip)
    assert abs(got - quad) < 1e-6


def test_fan_equals_convex_clipping_on_random_convex_quads():
    rng = np.random.default_rng(0)
    for _ in range(300):
        def quad():
            c = rng.uniform(0, 50, 2)
            ang = np.sort(rng.uniform(0, 2 * math.pi, 4))
            r = rng.uniform(3, 15, 4)
            return (c + np.stack([r * np.cos(ang), r * np.sin(ang)], 1)).reshape(8)
        p, q = quad(), quad()
        if not (PO.is_convex(p) and PO.is_convex(q)):
            continue
        a, b = PO.intersection_convex(p, q), PO.intersection_fan(p, q)
        assert abs(a - b) <= 1e-9 * max(1.0, a)


def test_rectangle_detector_routes_correctly():
    from jdet_amd.data.np_boxes import polys_are_rectangles, rotated_box_to_poly_np
    rng = np.random.default_rng(1)
    rb = np.concatenate([rng.uniform(0, 1000, (50, 2)), rng.uniform(5, 200, (50, 2)), rng.uniform(-3, 3, (50, 1))], 1)
    polys = rotated_box_to_poly_np(rb.astype(np.float32))
    assert polys_are_rectangles(polys)
    skew = polys.copy()
    skew[7, 0] += 0.2 * rb[7, 2]
    assert not polys_are_rectangles(skew)
    assert polys_are_rectangles(np.zeros((0, 8)))


def test_poly_nms_restatement_properties():
    rng = np.random.default_rng(2)
    base = np.array([0, 0, 10, 0, 10, 6, 0, 6], np.float64)
    polys = np.stack([_rot(base, rng.uniform(-0.3, 0.3)) + np.tile(rng.uniform(0, 12, 2), 4) for _ in range(40)])
    scores = rng.uniform(0, 1, 40)
    keep = PO.poly_nms(polys, scores, 0.3)
    assert keep[0] == int(np.argmax(scores)) and len(set(keep)) == len(keep)
    iou = PO.poly_iou_matrix(polys[keep], polys[keep], 0)
    assert np.all(iou[np.triu_indices(len(keep), 1)] <= 0.3)
