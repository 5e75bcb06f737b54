"""float64 polygon IoU / polygon NMS.  TEST INFRASTRUCTURE ONLY (never imported by jdet_amd).

The reference has the polygon overlap twice: a CUDA kernel (python/jdet/ops/nms_poly.py:L79-133, fan-triangle
decomposition) and a CPU function `iou_poly` (L247-252) that calls shapely -- which is not installed here, and the
kernel has no CPU source, so neither can be executed: PARITY UNPINNED by reference execution.  This file restates the
DEFINITION (area of the intersection of two simple 4-point polygons over the area of their union) two independent
ways -- direct Sutherland-Hodgman clipping for convex polygons, and the signed fan-triangle sum for any simple
polygon -- checks them against each other and against closed-form areas (tests/test_poly_oracle.py), and is what the
HIP kernel is held to.
"""
import numpy as np


def _signed_area(p):
    x, y = p[:, 0], p[:, 1]
    return 0.5 * float(np.sum(x * np.roll(y, -1) - y * np.roll(x, -1)))


def _clip_left(poly, a, b):
    """part of the convex polygon `poly` (list of points) on the left of the directed line a -> b"""
    out = []
    n = len(poly)
    for i in range(n):
        cur, nxt = poly[i], poly[(i + 1) % n]
        dc = (b[0] - a[0]) * (cur[1] - a[1]) - (b[1] - a[1]) * (cur[0] - a[0])
        dn = (b[0] - a[0]) * (nxt[1] - a[1]) - (b[1] - a[1]) * (nxt[0] - a[0])
        if dc >= 0:
            out.append(cur)
        if (dc > 0 and dn < 0) or (dc < 0 and dn > 0):
            t = dc / (dc - dn)
            out.append((cur[0] + t * (nxt[0] - cur[0]), cur[1] + t * (nxt[1] - cur[1])))
    return out


def _area_of(points):
    if len(points) < 3:
        return 0.0
    return abs(_signed_area(np.asarray(points, np.float64)))


def is_convex(p):
    p = np.asarray(p, np.float64).reshape(4, 2)
    s = []
    for i in range(4):
        a, b, c = p[i], p[(i + 1) % 4], p[(i + 2) % 4]
        s.append((b[0] - a[0]) * (c[1] - b[1]) - (b[1] - a[1]) * (c[0] - b[0]))
    s = np.asarray(s)
    return bool(np.all(s >= 0) or np.all(s <= 0))


def intersection_convex(p, q):
    """both convex: clip p by every edge of q"""
    p = np.asarray(p, np.float64).reshape(4, 2)
    q = np.asarray(q, np.float64).reshape(4, 2)
    if _signed_area(p) < 0:
        p = p[::-1]
    if _signed_area(q) < 0:
        q = q[::-1]
    poly = [tuple(v) for v in p]
    for i in range(4):
        poly = _clip_left(poly, q[i], q[(i + 1) % 4])
        if len(poly) < 3:
            return 0.0
    return _area_of(poly)


def intersection_fan(p, q):
    """any simple polygons: signed sum over fan triangles about the common centroid"""
    p = np.asarray(p, np.float64).reshape(4, 2)
    q = np.asarray(q, np.float64).reshape(4, 2)
    c = (p.sum(0) + q.sum(0)) / 8.0
    p, q = p - c, q - c
    if _signed_area(p) < 0:
        p = p[::-1]
    if _signed_area(q) < 0:
        q = q[::-1]
    total = 0.0
    for i in range(4):
        a, b = p[i], p[(i + 1) % 4]
        wa = a[0] * b[1] - a[1] * b[0]
        if wa == 0:
            continue
        ta = [(0.0, 0.0), tuple(a), tuple(b)] if wa > 0 else [(0.0, 0.0), tuple(b), tuple(a)]
        for j in range(4):
            cc, d = q[j], q[(j + 1) % 4]
            wc = cc[0] * d[1] - cc[1] * d[0]
            if wc == 0:
                continue
            tb = [(0.0, 0.0), tuple(cc), tuple(d)] if wc > 0 else [(0.0, 0.0), tuple(d), tuple(cc)]
            poly = ta
            for k in range(3):
                poly = _clip_left(poly, tb[k], tb[(k + 1) % 3])
                if len(poly) < 3:
                    break
            area = _area_of(poly)
            total += area if (wa > 0) == (wc > 0) else -area
    return max(total, 0.0)


def poly_iou(p, q, mode=1):
    """mode 1: iou_poly's rule inter / max(union, 0.01) (nms_poly.py:L251); mode 0: the kernel's (L125-131)"""
    p = np.asarray(p, np.float64).reshape(4, 2)
    q = np.asarray(q, np.float64).reshape(4, 2)
    inter = intersection_convex(p, q) if (is_convex(p) and is_convex(q)) else intersection_fan(p, q)
    union = abs(_signed_area(p)) + abs(_signed_area(q)) - inter
    if mode == 1:
        return inter / max(union, 0.01)
    return (inter + 1.0) / (union + 1.0) if union == 0 else inter / union


def poly_iou_matrix(ps, qs, mode=1):
    ps, qs = np.asarray(ps, np.float64).reshape(-1, 8), np.asarray(qs, np.float64).reshape(-1, 8)
    return np.array([[poly_iou(p, q, mode) for q in qs] for p in ps]).reshape(len(ps), len(qs))


def poly_nms(polys, scores, thresh, labels=None):
    """greedy, descending score (stable), suppress at IoU > thresh (poly_nms_kernel L177); kept indices in score order"""
    polys = np.asarray(polys, np.float64).reshape(-1, 8)
    order = np.argsort(-np.asarray(scores, np.float64), kind="stable")
    removed = np.zeros(len(polys), bool)
    keep = []
    for a, i in enumerate(order):
        if removed[i]:
            continue
        keep.append(int(i))
        for j in order[a + 1:]:
            if removed[j] or (labels is not None and labels[i] != labels[j]):
                continue
            if poly_iou(polys[i], polys[j], 0) > thresh:
                removed[j] = True
    return keep
