"""GPU: RepPoints geometry + convex_sort (csrc/convex_ops.hip) against the CPU restatement.  Index results are
bit-exact; the IoU is computed in double on both sides (1e-6); the rectangle goes through float cos / atan2 of two
different math libraries (1e-3 px on coordinates up to a few hundred)."""
import numpy as np
import pytest
import torch

from oracle import oracle as O

pytestmark = pytest.mark.gpu


def _pointsets(rng, n, extent=200.0, spread=40.0):
    c = rng.uniform(spread, extent - spread, size=(n, 1, 2))
    return (c + rng.normal(0, spread / 3, size=(n, 9, 2))).reshape(n, 18).astype(np.float32)


def _quads(rng, m, extent=200.0):
    c = rng.uniform(30, extent - 30, size=(m, 2))
    w, h, t = rng.uniform(10, 80, m), rng.uniform(6, 50, m), rng.uniform(-np.pi, np.pi, m)
    d = np.asarray([[-.5, -.5], [.5, -.5], [.5, .5], [-.5, .5]])
    out = []
    for i in range(m):
        r = np.asarray([[np.cos(t[i]), np.sin(t[i])], [-np.sin(t[i]), np.cos(t[i])]])
        out.append(((d * [w[i], h[i]]) @ r + c[i]).reshape(8))
    return np.asarray(out, np.float32)


@pytest.mark.parametrize("n,m", [(300, 17), (1, 1), (64, 64)])
def test_convex_iou(n, m):
    from jdet_amd.ops.reppoints_convex_iou import reppoints_convex_iou
    rng = np.random.default_rng(n + m)
    ps, q = _pointsets(rng, n), _quads(rng, m)
    q[::3] = q[::3].reshape(-1, 4, 2)[:, ::-1].reshape(-1, 8)          # some clockwise quadrilaterals
    got = reppoints_convex_iou(torch.from_numpy(ps).cuda(), torch.from_numpy(q).cuda()).cpu().numpy()
    ref = O.convex_iou(ps, q)
    assert got.shape == (n, m) and (n * m < 100 or ref.max() > 0.3)
    np.testing.assert_allclose(got, ref, rtol=0, atol=1e-6)


def test_convex_iou_degenerate_and_empty():
    from jdet_amd.ops.reppoints_convex_iou import reppoints_convex_iou
    rng = np.random.default_rng(9)
    ps = _pointsets(rng, 6)
    ps[0] = np.tile(ps[0, :2], 9)                       # nine identical points
    ps[1] = np.stack([np.linspace(10, 90, 9), np.linspace(20, 60, 9)], 1).reshape(18)     # collinear
    ps[2, 2:4] = ps[2, 0:2]                             # duplicates
    q = _quads(rng, 5)
    got = reppoints_convex_iou(torch.from_numpy(ps).cuda(), torch.from_numpy(q).cuda()).cpu().numpy()
    ref = O.convex_iou(ps, q)
    np.testing.assert_array_equal(np.isnan(got), np.isnan(ref))
    np.testing.assert_allclose(np.nan_to_num(got), np.nan_to_num(ref), rtol=0, atol=1e-6)
    e = reppoints_convex_iou(torch.zeros(0, 18).cuda(), torch.from_numpy(q).cuda())
    assert e.shape == (0, 5)
    assert reppoints_convex_iou(torch.from_numpy(ps).cuda(), torch.zeros(0, 8).cuda()).shape == (6, 0)


def test_min_area_bbox():
    from jdet_amd.ops.reppoints_min_area_bbox import reppoints_min_area_bbox
    rng = np.random.default_rng(10)
    ps = _pointsets(rng, 500)
    got = reppoints_min_area_bbox(torch.from_numpy(ps).cuda()).cpu().numpy()
    ref = O.min_area_bbox(ps)
    # two edge directions can give rectangles of (nearly) the same area: compare areas everywhere, corners where
    # the minimum is clear-cut
    area = lambda b: np.linalg.norm(b[:, 0:2] - b[:, 2:4], axis=1) * np.linalg.norm(b[:, 4:6] - b[:, 2:4], axis=1)
    np.testing.assert_allclose(area(got), area(ref), rtol=1e-4)
    same = np.abs(got - ref).max(1) < 2e-3
    assert same.mean() > 0.97
    assert reppoints_min_area_bbox(torch.zeros(0, 18).cuda()).shape == (0, 8)


@pytest.mark.parametrize("circular", [True, False])
@pytest.mark.parametrize("nbs,npts", [(200, 12), (33, 24), (5, 1), (3, 64)])
def test_convex_sort(nbs, npts, circular):
    from jdet_amd.ops.convex_sort import convex_sort
    rng = np.random.default_rng(nbs + npts)
    pts = rng.uniform(0, 50, size=(nbs, npts, 2)).astype(np.float32)
    pts[:, -1] = pts[:, 0]                                   # a duplicate point in every set
    masks = (rng.uniform(size=(nbs, npts)) > 0.3).astype(np.float32)
    masks[:, 0] = 1
    got = convex_sort(torch.from_numpy(pts).cuda(), torch.from_numpy(masks).cuda() > 0.5, circular).cpu().numpy()
    np.testing.assert_array_equal(got, O.convex_sort(pts, masks, circular))


def test_convex_sort_limits():
    from jdet_amd import _lib as L
    from jdet_amd.ops.convex_sort import convex_sort
    with pytest.raises(L.JDetHipError):
        convex_sort(torch.zeros(1, 65, 2).cuda(), torch.ones(1, 65).cuda())
    assert convex_sort(torch.zeros(2, 0, 2).cuda(), torch.ones(2, 0).cuda()).tolist() == [[-1], [-1]]


def test_convex_ops_vs_golden(golden):
    from jdet_amd.ops.convex_sort import convex_sort
    from jdet_amd.ops.reppoints_convex_iou import reppoints_convex_iou
    from jdet_amd.ops.reppoints_min_area_bbox import reppoints_min_area_bbox
    g = golden("convex_ops")
    ps, q = torch.from_numpy(g["pointsets"]).cuda(), torch.from_numpy(g["quads"]).cuda()
    got = reppoints_convex_iou(ps, q).cpu().numpy()
    np.testing.assert_array_equal(np.isnan(got), np.isnan(g["ious"]))
    np.testing.assert_allclose(np.nan_to_num(got), np.nan_to_num(g["ious"]), rtol=0, atol=1e-6)
    box, ref = reppoints_min_area_bbox(ps).cpu().numpy()[2:], g["boxes"][2:]          # rows 0, 1 are degenerate sets
    area = lambda b: np.linalg.norm(b[:, 0:2] - b[:, 2:4], axis=1) * np.linalg.norm(b[:, 4:6] - b[:, 2:4], axis=1)
    np.testing.assert_allclose(area(box), area(ref), rtol=1e-4)
    pts, masks = torch.from_numpy(g["pts"]).cuda(), torch.from_numpy(g["masks"]).cuda()
    np.testing.assert_array_equal(convex_sort(pts, masks, True).cpu().numpy(), g["sort_circular"])
    np.testing.assert_array_equal(convex_sort(pts, masks, False).cpu().numpy(), g["sort_open"])


def test_convex_giou_vs_definition():
    """reppoints_convex_giou (csrc/convex_giou.hip: forward-mode dual numbers, half a wave per pair) against the
    float64 restatement of the definition (oracle/convex_giou_oracle.py: Qhull hulls, textbook convex clip, central
    differences): value 1e-5, gradient 2e-3 relative to its scale (the finite-difference step against float inputs)"""
    from jdet_amd.ops.reppoints_convex_iou import reppoints_convex_giou
    from oracle import convex_giou_oracle as G
    rng = np.random.default_rng(5)
    n = 48
    ps = _pointsets(rng, n, spread=30.0)
    q = _quads(rng, n)
    q[: n // 2] += (ps[: n // 2].reshape(-1, 9, 2).mean(1) - q[: n // 2].reshape(-1, 4, 2).mean(1))[:, None, :].repeat(4, 1).reshape(-1, 8)
    dev = torch.device("cuda:0")
    giou, grad = reppoints_convex_giou(torch.from_numpy(ps).to(dev), torch.from_numpy(q).to(dev))
    assert giou.shape == (n,) and grad.shape == (n, 18)
    val, gref = G.convex_giou(ps.astype(np.float64), q.astype(np.float64), h=1e-3)
    np.testing.assert_allclose(giou.cpu().numpy(), val, rtol=0, atol=1e-5)
    assert (val[: n // 2] > 0).mean() > 0.8            # the centred half overlaps
    g = grad.cpu().numpy()
    np.testing.assert_allclose(g, gref, rtol=0, atol=2e-3 * max(1e-3, np.abs(gref).max()))
    hullsize = [len(G._hull(p.reshape(9, 2))) for p in ps.astype(np.float64)]
    assert all((np.abs(g[i].reshape(9, 2)).sum(1) > 0).sum() <= hullsize[i] for i in range(n))   # interior points: zero
    # empty call
    e, eg = reppoints_convex_giou(torch.zeros((0, 18), device=dev), torch.zeros((0, 8), device=dev))
    assert e.shape == (0,) and eg.shape == (0, 18)
