"""Seeded synthetic inputs shared by the parity tests, the golden generator and bench.py.

Distributions follow SURVEY.md §8(d): OBB centre U(0,extent)^2, w,h = exp(U(ln lo, ln hi)),
theta ~ U(-pi/2, pi/2); feature maps N(0,1).
"""
import math

import numpy as np


def random_obbs(rng, n, extent=1024.0, wh=(8.0, 256.0)):
    c = rng.uniform(0, extent, size=(n, 2))
    wh_ = np.exp(rng.uniform(math.log(wh[0]), math.log(wh[1]), size=(n, 2)))
    th = rng.uniform(-math.pi / 2, math.pi / 2, size=(n, 1))
    return np.concatenate([c, wh_, th], 1).astype(np.float32)


def clustered_obbs(rng, n, n_clusters=8, extent=256.0, jitter=6.0, wh=(16.0, 64.0)):
    """Heavily overlapping boxes (NMS / IoU stress): jittered copies of a few seeds."""
    seeds = random_obbs(rng, n_clusters, extent, wh)
    idx = rng.integers(0, n_clusters, size=n)
    b = seeds[idx].copy()
    b[:, :2] += rng.normal(0, jitter, size=(n, 2)).astype(np.float32)
    b[:, 2:4] *= np.exp(rng.normal(0, 0.15, size=(n, 2))).astype(np.float32)
    b[:, 4] += rng.normal(0, 0.2, size=n).astype(np.float32)
    return b.astype(np.float32)


def special_obbs():
    """Hand-built degenerate / structured cases for IoU (identical, shared edge, nested,
    touching corner, zero area, tiny, 45 deg, far apart)."""
    r = [
        [0, 0, 1, 1, 0], [0.5, 0.5, 1, 2, 0],           # reference literal box_iou_rotated.py:L513
        [10, 10, 4, 2, 0], [10, 10, 4, 2, 0],           # identical
        [14, 10, 4, 2, 0],                              # shares an edge with the previous
        [10, 10, 2, 1, 0.3],                            # nested, rotated
        [10, 10, 4, 2, math.pi / 4], [10, 10, 4, 2, -math.pi / 4],
        [10, 10, 4, 2, math.pi / 2], [12, 11, 4, 2, math.pi],
        [12, 12, 4, 4, 0.7853982],                      # diamond touching
        [50, 50, 0, 3, 0.1], [50, 50, 1e-8, 1e-8, 0.2],  # zero / sub-1e-14 area
        [50, 50, 5e-4, 7, 0.2],                         # "too small" for _v1 post-processing
        [1000, 1000, 30, 10, 1.0],                      # far away
        [10, 10.0001, 4, 2, 1e-4],                      # nearly identical
        [0, 0, 100, 100, 0.5], [3, -2, 7, 9, -1.2],     # big contains small
    ]
    return np.asarray(r, np.float32)


def rois_from_obbs(obbs, batch_idx):
    return np.concatenate([np.asarray(batch_idx, np.float32)[:, None], obbs], 1).astype(np.float32)


def edge_rois(H, W, scale):
    """RoIs (R,6) in image coordinates exercising the boundary rules of the RoIAlign dialects:
    fully outside, straddling each border, sub-pixel (forced to 1x1), larger than the map."""
    sx, sy = W / scale, H / scale
    r = [
        [0, -50, -50, 20, 20, 0.3],            # completely outside -> zeros
        [0, 0, 0, 30, 12, 0.0],                # centred on the corner
        [1, sx, sy, 40, 16, 1.0],              # bottom-right corner
        [0, sx / 2, -2, 64, 8, -0.4],          # top border
        [1, -1, sy / 2, 8, 64, 0.9],           # left border
        [0, sx / 2, sy / 2, 0.5, 0.25, 0.2],   # sub-pixel -> 1x1
        [1, sx / 2, sy / 2, 3 * sx, 3 * sy, 0.1],  # larger than map
        [0, sx / 3, sy / 3, 17.3, 5.1, -1.5707964],
        [1, sx / 4, sy / 1.5, 33.0, 9.0, 3.1],
        [0, 2.0, 2.0, 4.0, 4.0, 0.0],          # samples land exactly on <=0 / pixel centres
    ]
    return np.asarray(r, np.float32)


def obb_to_hbb_rois(rois6):
    """(R,6) -> (R,5) [b,x1,y1,x2,y2] enclosing axis-aligned boxes (for the hbb ROIAlign)."""
    b, xc, yc, w, h = rois6[:, 0], rois6[:, 1], rois6[:, 2], rois6[:, 3], rois6[:, 4]
    return np.stack([b, xc - w / 2, yc - h / 2, xc + w / 2, yc + h / 2], 1).astype(np.float32)


def arf_indices(n_orient=8, n_rot=8, k=3):
    """The ORConv2d index table (values as in reference ops/orn.py:L644-678; 1-based, uint8).
    Data, not code: which source tap feeds each rotated tap."""
    table = {
        1: {a: (1,) for a in range(0, 360, 45)},
        3: {
            0: (1, 2, 3, 4, 5, 6, 7, 8, 9),
            45: (2, 3, 6, 1, 5, 9, 4, 7, 8),
            90: (3, 6, 9, 2, 5, 8, 1, 4, 7),
            135: (6, 9, 8, 3, 5, 7, 2, 1, 4),
            180: (9, 8, 7, 6, 5, 4, 3, 2, 1),
            225: (8, 7, 4, 9, 5, 1, 6, 3, 2),
            270: (7, 4, 1, 8, 5, 2, 9, 6, 3),
            315: (4, 1, 2, 7, 5, 3, 8, 9, 6),
        },
    }
    d_or, d_rot = 360 / n_orient, 360 / n_rot
    idx = np.zeros((n_orient * k * k, n_rot), np.uint8)
    for i in range(n_orient):
        for j in range(k * k):
            for r in range(n_rot):
                angle = d_rot * r
                layer = (i + math.floor(angle / d_or)) % n_orient
                idx[i * k * k + j, r] = int(layer * k * k + table[k][int(angle)][j])
    return idx.reshape(n_orient, k, k, n_rot)
