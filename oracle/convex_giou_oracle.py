"""TEST INFRASTRUCTURE (never imported by jdet_amd): the definition behind `reppoints_convex_giou`
(/root/reference/python/jdet/ops/reppoints_convex_iou/convex_giou.py:L29-47; convex_giou_kernel.cu:L725-799 `devrIoU`):

    A = area(hull of the 9 points), B = area(quadrilateral), I = area(hull /\ quadrilateral), U = A + B - I,
    C = area(hull of the vertices of both), giou = I / U - (C - U) / C                       (kernel.cu:L764-779)
    point_grad = d giou / d (the 18 coordinates); zero for points that are not hull vertices (L782-797)

restated in float64 with independent tools -- Qhull (scipy.spatial.ConvexHull) for the hulls, a textbook convex clip
for the intersection -- and the gradient by central differences of that value.  Pinning: the reference's own kernel
text runs on the device (oracle/build_ref_hip.py -> refhip_convex_giou; tests/test_gpu_convex_ops.py compares the
product kernel with it AND with this file); on CPU this file is checked against closed forms (tests/test_convex_oracle.py)."""
import numpy as np
from scipy.spatial import ConvexHull, QhullError


def _hull(pts):
    pts = np.asarray(pts, np.float64)
    try:
        h = ConvexHull(pts)
    except QhullError:
        return pts[:0]
    return pts[h.vertices]                      # counter-clockwise in 2-D


def _area(p):
    if len(p) < 3:
        return 0.0
    x, y = p[:, 0], p[:, 1]
    return 0.5 * float(np.sum(x * np.roll(y, -1) - np.roll(x, -1) * y))


def _clip(p, q):
    """convex p clipped to the left of every edge of the counter-clockwise convex q"""
    p = [tuple(v) for v in p]
    for i in range(len(q)):
        a, b = q[i], q[(i + 1) % len(q)]
        out = []
        for k in range(len(p)):
            cur, nxt = p[k], p[(k + 1) % len(p)]
            sc = (b[0] - a[0]) * (cur[1] - a[1]) - (cur[0] - a[0]) * (b[1] - a[1])
            sn = (b[0] - a[0]) * (nxt[1] - a[1]) - (nxt[0] - a[0]) * (b[1] - a[1])
            if sc >= 0:
                out.append(cur)
            if (sc >= 0) != (sn >= 0):
                t = sc / (sc - sn)
                out.append((cur[0] + t * (nxt[0] - cur[0]), cur[1] + t * (nxt[1] - cur[1])))
        p = out
        if not p:
            break
    return np.asarray(p, np.float64).reshape(-1, 2)


def giou_value(points18, quad8):
    P = _hull(np.asarray(points18, np.float64).reshape(9, 2))
    Q = np.asarray(quad8, np.float64).reshape(4, 2)
    if _area(Q) < 0:
        Q = Q[::-1]
    A, B = _area(P), _area(Q)
    inter = abs(_area(_clip(P, Q))) if len(P) >= 3 else 0.0
    U = A + B - inter
    C = _area(_hull(np.concatenate([P, Q], 0)))
    return inter / U - (C - U) / C


def convex_giou(pointsets, polygons, h=1e-4):
    """(N, 18), (N, 8) -> giou (N,), point_grad (N, 18) by central differences with step h"""
    pointsets, polygons = np.asarray(pointsets, np.float64), np.asarray(polygons, np.float64)
    N = pointsets.shape[0]
    val, grad = np.zeros(N), np.zeros((N, 18))
    for n in range(N):
        val[n] = giou_value(pointsets[n], polygons[n])
        for j in range(18):
            e = np.zeros(18)
            e[j] = h
            grad[n, j] = (giou_value(pointsets[n] + e, polygons[n]) - giou_value(pointsets[n] - e, polygons[n])) / (2 * h)
    return val, grad
