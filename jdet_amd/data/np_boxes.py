"""numpy box algebra of the input pipeline.  Semantics of python/jdet/models/boxes/box_ops.py: `norm_angle` L176-178,
`poly_to_rotated_box_single/np` L436-482, `get_best_begin_point(_single)` L520-548, `rotated_box_to_poly_np` L568-590,
`rotated_box_to_bbox_np` L616-624 -- vectorised over boxes here (the reference loops in Python per box)."""
import math

import numpy as np


def norm_angle(angle, rng=(-math.pi / 4, math.pi)):
    return (angle - rng[0]) % rng[1] + rng[0]


def best_begin_point(polys):
    """(n,8) quads -> the cyclic rotation of each quad whose vertices are closest, in order, to the corners
    (xmin,ymin), (xmax,ymin), (xmax,ymax), (xmin,ymax) of its enclosing box; first minimum wins."""
    polys = np.asarray(polys, dtype=np.float64).reshape(-1, 4, 2)
    if polys.shape[0] == 0:
        return polys.reshape(0, 8)
    lo, hi = polys.min(1), polys.max(1)
    corners = np.stack([lo, np.stack([hi[:, 0], lo[:, 1]], 1), hi, np.stack([lo[:, 0], hi[:, 1]], 1)], 1)  # (n,4,2)
    cost = np.stack([np.sqrt(((np.roll(polys, -s, axis=1) - corners) ** 2).sum(-1)).sum(-1) for s in range(4)], 1)
    shift = np.argmin(cost, 1)                      # argmin takes the first minimum, like the `<` scan of the reference
    idx = (np.arange(4)[None, :] + shift[:, None]) % 4
    return np.take_along_axis(polys, idx[:, :, None], axis=1).reshape(-1, 8)


def rotated_box_to_poly_np(rrects):
    """(n,5) [xc,yc,w,h,theta] -> (n,8) float32, corners tl,tr,br,bl of the unrotated box rotated by theta, then
    `best_begin_point`"""
    rrects = np.asarray(rrects)
    if rrects.shape[0] == 0:
        return np.zeros((0, 8), dtype=np.float32)
    xc, yc, w, h, a = [rrects[:, i].astype(np.float32) for i in range(5)]     # fp32 throughout, as the reference
    xs = np.stack([-w / 2, w / 2, w / 2, -w / 2], 1)
    ys = np.stack([-h / 2, -h / 2, h / 2, h / 2], 1)
    c, s = np.cos(a)[:, None], np.sin(a)[:, None]
    px = c * xs - s * ys + xc[:, None]
    py = s * xs + c * ys + yc[:, None]
    return best_begin_point(np.stack([px, py], -1).reshape(-1, 8)).astype(np.float32)


def poly_to_rotated_box_np(polys):
    """(n,8) -> (n,5): long edge = w, angle of the long edge through norm_angle, centre = midpoint of p1 p3"""
    p = np.asarray(polys, dtype=np.float32)[:, :8].reshape(-1, 4, 2)
    if p.shape[0] == 0:
        return np.zeros((0, 5), dtype=np.float32)
    e1 = np.sqrt(((p[:, 0] - p[:, 1]) ** 2).sum(-1))
    e2 = np.sqrt(((p[:, 1] - p[:, 2]) ** 2).sum(-1))
    d12 = (p[:, 1] - p[:, 0]).astype(np.float64)
    d14 = (p[:, 3] - p[:, 0]).astype(np.float64)
    first = e1 > e2
    ang = np.where(first, np.arctan2(d12[:, 1], d12[:, 0]), np.arctan2(d14[:, 1], d14[:, 0]))
    ctr = (p[:, 0].astype(np.float64) + p[:, 2].astype(np.float64)) / 2
    return np.stack([ctr[:, 0], ctr[:, 1], np.maximum(e1, e2), np.minimum(e1, e2), norm_angle(ang)],
                    1).astype(np.float32)


def polys_are_rectangles(polys, rel_tol=1e-3):
    """(n,8): every polygon is a rectangle given corner by corner (diagonals bisect each other and have equal
    length) up to `rel_tol` of its size -- the test that routes evaluation / merging to the rotated-box kernels"""
    p = np.asarray(polys, dtype=np.float64).reshape(-1, 4, 2)
    if p.shape[0] == 0:
        return True
    mid = np.abs((p[:, 0] + p[:, 2]) - (p[:, 1] + p[:, 3])).max(-1) / 2
    d1 = np.sqrt(((p[:, 0] - p[:, 2]) ** 2).sum(-1))
    d2 = np.sqrt(((p[:, 1] - p[:, 3]) ** 2).sum(-1))
    size = np.maximum(np.maximum(d1, d2), 1e-9)
    return bool(np.all(mid <= rel_tol * size) and np.all(np.abs(d1 - d2) <= rel_tol * size))


def rotated_box_to_bbox_np(rboxes):
    """-> (enclosing (n,4) boxes, (n,8) polys)"""
    rboxes = np.asarray(rboxes)
    if rboxes.shape[0] == 0:
        return np.zeros((0, 4)), np.zeros((0, 8))
    polys = rotated_box_to_poly_np(rboxes)
    xs, ys = polys[:, 0::2], polys[:, 1::2]
    return np.stack([xs.min(1), ys.min(1), xs.max(1), ys.max(1)], 1), polys
