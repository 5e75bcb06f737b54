"""GPU parity: rotated IoU and rotated NMS vs golden fixtures and the CPU oracle.
Bar: bit-exact IoU values (same fp32 operation order as the reference CPU path, contraction off),
hence bit-exact NMS keep masks."""
import numpy as np
import pytest
import torch

from oracle import oracle as O
from tests import inputs as I

pytestmark = pytest.mark.gpu


def _iou(b1, b2, dev, version=0, sort_mode=0):
    from jdet_amd import _lib as L
    from jdet_amd.ops.box_iou_rotated import box_iou_rotated, box_iou_rotated_v1
    old = L.REFERENCE_SORT
    L.REFERENCE_SORT = sort_mode
    try:
        f = box_iou_rotated_v1 if version else box_iou_rotated
        return f(torch.from_numpy(b1).to(dev), torch.from_numpy(b2).to(dev)).cpu().numpy()
    finally:
        L.REFERENCE_SORT = old


def test_iou_vs_golden(golden, dev):
    g = golden("box_iou_rotated")
    np.testing.assert_array_equal(_iou(g["b1"], g["b2"], dev, 0, 0), g["iou"])
    np.testing.assert_array_equal(_iou(g["b1"], g["b2"], dev, 0, 1), g["iou_cudasort"])
    np.testing.assert_array_equal(_iou(g["lit"], g["lit"], dev), g["iou_lit"])
    big = _iou(g["big1"], g["big2"], dev)
    assert float(big.astype(np.float64).sum()) == float(g["iou_big_sum"])
    assert int((big > 0).sum()) == int(g["iou_big_nnz"])
    # _v1: golden holds the raw kernel output; the python-side "too small" zeroing is applied on top
    v1 = _iou(g["b1"], g["b2"], dev, 1, 0)
    exp = g["iou_v1"].copy()
    exp[g["b1"][:, 2:4].min(1) < 1e-3, :] = 0
    exp[:, g["b2"][:, 2:4].min(1) < 1e-3] = 0
    np.testing.assert_array_equal(v1, exp)


@pytest.mark.parametrize("seed,n1,n2", [(1, 512, 512), (2, 64, 21824), (3, 1, 1), (4, 130, 7)])
def test_iou_vs_oracle_random(dev, seed, n1, n2):
    """configs[0] (512x512 random OBBs) and the S2ANet assigner shape (64 gts x 21824 anchors)"""
    rng = np.random.default_rng(seed)
    b1 = I.random_obbs(rng, n1) if seed != 2 else I.random_obbs(rng, n1, wh=(16.0, 256.0))
    b2 = I.random_obbs(rng, n2)
    if seed == 2:  # anchors: a dense grid of squares, many exactly axis aligned / tied
        b2[:, 2:4] = b2[:, 2:3]
        b2[:, 4] = 0
        b2[:, :2] = np.round(b2[:, :2] / 8) * 8
    O.set_threads(8)
    np.testing.assert_array_equal(_iou(b1, b2, dev), O.box_iou_rotated(b1, b2))


def test_iou_clustered_heavy_overlap(dev):
    rng = np.random.default_rng(11)
    b1, b2 = I.clustered_obbs(rng, 700, 10, 300.0), I.clustered_obbs(rng, 650, 10, 300.0)
    got, ref = _iou(b1, b2, dev), O.box_iou_rotated(b1, b2)
    assert (ref > 0).mean() > 0.05
    np.testing.assert_array_equal(got, ref)


def test_iou_empty(dev):
    from jdet_amd.ops.box_iou_rotated import box_iou_rotated
    e = box_iou_rotated(torch.zeros((0, 5), device=dev), torch.zeros((3, 5), device=dev))
    assert e.shape == (0, 3)


def _nms(dets, scores, thr, dev, labels=None):
    from jdet_amd.ops.nms_rotated import ml_nms_rotated, nms_rotated
    d, s = torch.from_numpy(dets).to(dev), torch.from_numpy(scores).to(dev)
    if labels is None:
        return nms_rotated(d, s, thr).cpu().numpy()
    return ml_nms_rotated(d, s, torch.from_numpy(labels).to(dev), thr).cpu().numpy()


def test_nms_vs_golden(golden, dev):
    g = golden("nms_rotated")
    assert list(_nms(g["lit_dets"], g["lit_scores"], 0.3, dev)) == [2]
    assert list(_nms(g["lit_dets"], g["lit_scores"], 0.3, dev, np.ones(3, np.float32))) == [2]
    for nm in "abc":
        dets, scores, labels = g["dets_" + nm], g["scores_" + nm], g["labels_" + nm]
        for thr in (0.1, 0.5):
            np.testing.assert_array_equal(_nms(dets, scores, thr, dev), np.nonzero(g["keep5_%s_%g" % (nm, thr)])[0])
            np.testing.assert_array_equal(_nms(dets, scores, thr, dev, labels), np.nonzero(g["keep6_%s_%g" % (nm, thr)])[0])


@pytest.mark.parametrize("n,thr", [(512, 0.1), (512, 0.5), (2000, 0.1), (4097, 0.3)])
def test_nms_vs_oracle(dev, n, thr):
    """configs[0]: 512 random OBBs at thr 0.1 / 0.5; plus sizes crossing 64-box tile boundaries"""
    rng = np.random.default_rng(n)
    dets = np.concatenate([I.random_obbs(rng, n // 2), I.clustered_obbs(rng, n - n // 2, 24, 1024.0)], 0)
    scores = (rng.uniform(0, 1, n) + np.arange(n) * 1e-7).astype(np.float32)
    order = np.argsort(-scores, kind="stable").astype(np.int32)
    ref = np.nonzero(O.nms_rotated_keep(dets, order, thr, cmp_ge=1))[0]
    np.testing.assert_array_equal(_nms(dets, scores, thr, dev), ref)
    labels = rng.integers(0, 15, n).astype(np.float32)
    d6 = np.concatenate([dets, labels[:, None]], 1)
    ref6 = np.nonzero(O.nms_rotated_keep(d6, order, thr, cmp_ge=1))[0]
    np.testing.assert_array_equal(_nms(dets, scores, thr, dev, labels), ref6)


def test_nms_rules_and_edge_cases(dev):
    from jdet_amd.ops import nms_rotated as M
    dets = np.asarray([[0, 0, 2, 2, 0], [1, 0, 2, 2, 0]], np.float32)
    iou = float(O.box_iou_rotated(dets[:1], dets[1:])[0, 0])
    s = np.asarray([0.9, 0.8], np.float32)
    assert list(_nms(dets, s, iou, dev)) == [0]           # CPU rule >=
    M.REFERENCE_RULE = "cuda"
    try:
        assert list(_nms(dets, s, iou, dev)) == [0, 1]    # CUDA rule >
    finally:
        M.REFERENCE_RULE = "cpu"
    assert M.nms_rotated(torch.zeros((0, 5), device=dev), torch.zeros((0,), device=dev), 0.5).numel() == 0
    one = M.nms_rotated(torch.tensor([[5., 5, 2, 2, 0]], device=dev), torch.tensor([0.3], device=dev), 0.5)
    assert one.tolist() == [0]
    # all identical boxes -> only the best survives; result indices ascending (jt.where order)
    same = np.tile(np.asarray([[10, 10, 4, 2, 0.3]], np.float32), (130, 1))
    sc = np.linspace(0, 1, 130).astype(np.float32)
    assert list(_nms(same, sc, 0.5, dev)) == [129]


def test_nms_full_size_properties(dev):
    """n ~ 8.5k (RetinaNet-OBB pre-NMS size): idempotence and pairwise-IoU consistency of the kept set"""
    from jdet_amd.ops.box_iou_rotated import box_iou_rotated
    from jdet_amd.ops.nms_rotated import nms_rotated
    rng = np.random.default_rng(5)
    n = 8576
    dets = torch.from_numpy(np.concatenate([I.random_obbs(rng, n // 2), I.clustered_obbs(rng, n - n // 2, 64, 1024.0)], 0)).to(dev)
    scores = torch.from_numpy((rng.uniform(0, 1, n) + np.arange(n) * 1e-7).astype(np.float32)).to(dev)
    keep = nms_rotated(dets, scores, 0.1)
    assert torch.all(keep[1:] > keep[:-1])
    kd, ks = dets[keep], scores[keep]
    again = nms_rotated(kd, ks, 0.1)
    assert again.numel() == keep.numel()                                  # idempotent
    order = torch.argsort(ks, descending=True)
    iou = box_iou_rotated(kd[order], kd[order])
    assert torch.triu(iou, diagonal=1).max().item() < 0.1                 # no kept pair violates the rule
    # every suppressed box overlaps (>= thr) some kept box with a higher score
    sup = torch.ones(n, dtype=torch.bool, device=dev)
    sup[keep] = False
    si = torch.nonzero(sup)[:, 0][:512]
    io = box_iou_rotated(kd, dets[si])
    ok = ((io >= 0.1) & (ks[:, None] > scores[si][None, :])).any(0)
    assert bool(ok.all())


def test_multiclass_nms_rotated(dev):
    from jdet_amd.ops.nms_rotated import multiclass_nms_rotated
    rng = np.random.default_rng(9)
    n, ncls = 600, 15
    boxes = torch.from_numpy(I.clustered_obbs(rng, n, 12, 512.0)).to(dev)
    sc = torch.from_numpy(rng.uniform(0, 0.2, (n, ncls + 1)).astype(np.float32)).to(dev)
    det, lab = multiclass_nms_rotated(boxes, sc, 0.05, dict(type="nms_rotated", iou_thr=0.1), 2000)
    assert det.shape[1] == 6 and det.shape[0] == lab.shape[0] > 0
    assert torch.all(det[1:, 5] <= det[:-1, 5]) and lab.min() >= 0 and lab.max() < ncls
    # per-class check against the oracle: the kept IDENTITIES (box, score, class), not just how many
    scn, bn = sc.cpu().numpy()[:, 1:], boxes.cpu().numpy()
    expect = []
    for c in range(ncls):
        m = np.nonzero(scn[:, c] > 0.05)[0]
        if m.size == 0:
            continue
        order = np.argsort(-scn[m, c], kind="stable").astype(np.int32)
        keep = O.nms_rotated_keep(bn[m], order, 0.1)
        for i in m[keep]:
            expect.append((float(scn[i, c]), c, int(i)))
    assert len(expect) == det.shape[0]
    # the op returns the survivors of all classes sorted by score (no ties in this input)
    expect.sort(key=lambda t: -t[0])
    dn, ln = det.cpu().numpy(), lab.cpu().numpy()
    np.testing.assert_array_equal(ln, np.array([c for _, c, _ in expect]))
    np.testing.assert_array_equal(dn[:, 5], np.array([s for s, _, _ in expect], np.float32))
    np.testing.assert_array_equal(dn[:, :5], bn[[i for _, _, i in expect]])
    e_det, e_lab = multiclass_nms_rotated(boxes, sc * 0, 0.05, dict(iou_thr=0.1), 100)
    assert e_det.shape == (0, 6) and e_lab.shape == (0,)


def test_dota_evaluation_iou_runs_on_the_device(dev):
    """DOTADataset.evaluate's IoU matrices (polygons of rotated boxes -> box_iou_rotated on the device) equal the
    oracle's, and the AP of perfect detections is 1"""
    from jdet_amd.data.np_boxes import poly_to_rotated_box_np, rotated_box_to_poly_np
    from jdet_amd.data.voc_eval import device_iou_matrix, evaluate_dota
    rng = np.random.default_rng(21)
    a = rotated_box_to_poly_np(I.random_obbs(rng, 40))
    b = rotated_box_to_poly_np(I.clustered_obbs(rng, 30, 6, 1024.0))
    got = device_iou_matrix(a, b, dev)
    ref = O.box_iou_rotated(poly_to_rotated_box_np(a), poly_to_rotated_box_np(b))
    np.testing.assert_array_equal(got, ref)
    results = []
    for _ in range(3):
        g = I.random_obbs(rng, 6)
        gp = rotated_box_to_poly_np(g)
        labels = rng.integers(1, 4, 6)
        results.append(((gp.copy(), np.linspace(0.9, 0.5, 6), labels - 1),
                        dict(scale_factor=1.0, polys=gp, labels=labels, polys_ignore=np.zeros((0, 8)))))
    aps = evaluate_dota(results, ["a", "b", "c"], lambda x, y: device_iou_matrix(x, y, dev))
    assert abs(aps["eval/0_meanAP"] - 1.0) < 1e-12


def test_merge_nms_groups_on_the_device(dev):
    """result merging: all images of a class in ONE launch (image index as label, iou > thr suppresses) == greedy NMS
    image by image with the oracle"""
    from jdet_amd.data.np_boxes import poly_to_rotated_box_np, rotated_box_to_poly_np
    from jdet_amd.data.result_merge import device_group_nms
    rng = np.random.default_rng(31)
    n = 1500
    polys = rotated_box_to_poly_np(I.clustered_obbs(rng, n, 40, 2048.0))
    scores = (rng.uniform(0, 1, n) + np.arange(n) * 1e-7).astype(np.float32)
    groups = rng.integers(0, 7, n)
    for thr in (0.1, 0.3):
        got = device_group_nms(polys, scores, groups, thr, dev)
        ref = np.zeros(n, bool)
        for g in np.unique(groups):
            idx = np.nonzero(groups == g)[0]
            order = np.argsort(-scores[idx], kind="stable").astype(np.int32)
            ref[idx] = O.nms_rotated_keep(poly_to_rotated_box_np(polys[idx]), order, thr, cmp_ge=0).astype(bool)
        np.testing.assert_array_equal(got, ref)
        assert 0 < got.sum() < n


@pytest.mark.parametrize("horizontal", [0, 1])
def test_nms_one_scan_workgroup_per_label(dev, horizontal):
    """jdet_nms_labeled with n_labels > 1 (a scan workgroup per label) keeps exactly what the single scan keeps:
    label segments of sizes that are not multiples of 64 (row blocks shared by two labels), an absent label, a
    one-box label, heavy overlap inside the labels; for horizontal boxes additionally equal to the polygon-clipping
    kernel on the same boxes (rectangle formula vs clipping: same decisions on these boxes)."""
    from jdet_amd import _lib as L
    lib = L.lib()
    rng = np.random.default_rng(8)
    sizes = [700, 0, 1, 333, 64, 129]                # label 1 absent
    n, n_labels = sum(sizes), len(sizes)
    boxes = I.clustered_obbs(rng, n, n_clusters=20, extent=300.0, jitter=5.0, wh=(20.0, 70.0))
    if horizontal:
        boxes[:, 4] = 0.0
    labels = np.concatenate([np.full(s, i) for i, s in enumerate(sizes)]).astype(np.float32)
    perm = rng.permutation(n)
    labels = labels[perm]
    scores = rng.uniform(0, 1, n).astype(np.float32)
    dets = torch.from_numpy(np.concatenate([boxes, labels[:, None]], 1).astype(np.float32)).to(dev)
    order = np.argsort(-scores, kind="stable")
    order = order[np.argsort(labels[order], kind="stable")].astype(np.int32)
    o = torch.from_numpy(order).to(dev)
    wsb = lib.jdet_nms_rotated_workspace(n)

    def run(hz, nl):
        keep = torch.full((n,), 7, dtype=torch.uint8, device=dev)
        ws = torch.empty((wsb,), dtype=torch.uint8, device=dev)
        L.check(lib.jdet_nms_labeled(L.ptr(dets), n, 6, L.ptr(o), 0.3, 0, 0, hz, nl, L.ptr(keep), L.ptr(ws), wsb,
                                     L.stream_ptr(dets)), "nms_labeled")
        return keep.cpu().numpy()

    single = run(horizontal, 1)
    per_label = run(horizontal, n_labels)
    assert set(np.unique(single)) <= {0, 1} and np.array_equal(single, per_label)
    assert 0 < single.sum() < n
    if horizontal:
        assert np.array_equal(single, run(0, 1))
