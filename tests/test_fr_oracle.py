"""CPU: the feature-refinement restatement (oracle/fr_oracle.py) against a closed form, and its backward against
finite differences of its forward."""
import numpy as np

from oracle import fr_oracle as FO


def _case(rng, N=2, C=3, H=12, W=15, stride=8.0):
    ys, xs = np.mgrid[0:H, 0:W].astype(np.float32)
    boxes = np.stack([ys * stride + rng.normal(0, 12, (N, H, W)),      # column 0 plays the ROW coordinate (sic)
                      xs * stride + rng.normal(0, 12, (N, H, W)),
                      np.exp(rng.normal(3, 0.5, (N, H, W))), np.exp(rng.normal(3, 0.5, (N, H, W))),
                      rng.uniform(-1.5, 1.5, (N, H, W))], -1).astype(np.float32)
    return boxes


def test_affine_map_closed_form():
    """bilinear sampling reproduces an affine map exactly: out = f(h, w) + sum_i f(py_i, px_i) wherever every sample
    lies strictly inside the map (no clamping)"""
    rng = np.random.default_rng(0)
    N, C, H, W = 2, 3, 12, 15
    a = rng.integers(-4, 5, C) / 4.0
    b = rng.integers(-4, 5, C) / 4.0
    d = rng.integers(-8, 9, C) / 2.0
    ys, xs = np.mgrid[0:H, 0:W].astype(np.float64)
    f = (a[:, None, None] * xs + b[:, None, None] * ys + d[:, None, None])[None].repeat(N, 0).astype(np.float32)
    boxes = _case(rng, N, C, H, W)
    for points in (1, 5):
        py, px = FO.sample_points(boxes, 1 / 8.0, points)
        inside = np.all((py > 0) & (py < H - 1) & (px > 0) & (px < W - 1), axis=0)
        out = FO.feature_refine_forward(f, boxes, 1 / 8.0, points)
        expect = f.astype(np.float64).copy()
        for i in range(points):
            expect += (a[None, :, None, None] * px[i][:, None].astype(np.float64)
                       + b[None, :, None, None] * py[i][:, None].astype(np.float64) + d[None, :, None, None])
        assert inside.mean() > 0.2
        m = np.broadcast_to(inside[:, None], out.shape)
        np.testing.assert_allclose(out[m], expect[m], rtol=0, atol=2e-4)


def test_backward_is_the_transpose_of_forward():
    """the op is linear in the features: <forward(x), g> == <x, backward(g)>"""
    rng = np.random.default_rng(1)
    N, C, H, W = 2, 4, 9, 11
    boxes = _case(rng, N, C, H, W)
    boxes[0, 0, 0, :2] = -500.0                      # a sample far outside: contributes nothing
    x = rng.standard_normal((N, C, H, W)).astype(np.float32)
    g = rng.standard_normal((N, C, H, W)).astype(np.float32)
    for points in (1, 5):
        lhs = float((FO.feature_refine_forward(x, boxes, 1 / 8.0, points).astype(np.float64) * g).sum())
        rhs = float((x.astype(np.float64) * FO.feature_refine_backward(g, boxes, 1 / 8.0, points)).sum())
        assert abs(lhs - rhs) <= 1e-3 * max(1.0, abs(lhs))
