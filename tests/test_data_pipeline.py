"""CPU: the input pipeline of the named configs -- numpy box algebra against per-box loops that
follow the reference line by line, transforms against closed forms, datasets / collate / loader on a synthetic
`dataset_dir` (images + labels.pkl) with the S2ANet config's transform list."""
import math
import os
import pickle

import numpy as np
import pytest
import torch
from PIL import Image


def _rboxes(rng, n, extent=200.0):
    return np.concatenate([rng.uniform(20, extent - 20, (n, 2)), rng.uniform(6, 60, (n, 2)),
                           rng.uniform(-math.pi / 4, 3 * math.pi / 4, (n, 1))], 1).astype(np.float32)


# ---- per-box restatements (python/jdet/models/boxes/box_ops.py:L436-470, L520-542, L568-590) ---------------------------
def _best_begin_single(c):
    x1, y1, x2, y2, x3, y3, x4, y4 = c
    xmin, ymin, xmax, ymax = min(x1, x2, x3, x4), min(y1, y2, y3, y4), max(x1, x2, x3, x4), max(y1, y2, y3, y4)
    comb = [[[x1, y1], [x2, y2], [x3, y3], [x4, y4]], [[x2, y2], [x3, y3], [x4, y4], [x1, y1]],
            [[x3, y3], [x4, y4], [x1, y1], [x2, y2]], [[x4, y4], [x1, y1], [x2, y2], [x3, y3]]]
    dst = [[xmin, ymin], [xmax, ymin], [xmax, ymax], [xmin, ymax]]
    force, flag = 100000000.0, 0
    for i in range(4):
        f = sum(math.sqrt((comb[i][k][0] - dst[k][0]) ** 2 + (comb[i][k][1] - dst[k][1]) ** 2) for k in range(4))
        if f < force:
            force, flag = f, i
    return np.array(comb[flag]).reshape(8)


def _r2p_single(r):
    x, y, w, h, a = r[:5]
    rect = np.array([[-w / 2, w / 2, w / 2, -w / 2], [-h / 2, -h / 2, h / 2, h / 2]])
    R = np.array([[np.cos(a), -np.sin(a)], [np.sin(a), np.cos(a)]])
    p = R.dot(rect)
    return np.array([p[0, 0] + x, p[1, 0] + y, p[0, 1] + x, p[1, 1] + y, p[0, 2] + x, p[1, 2] + y, p[0, 3] + x,
                     p[1, 3] + y], dtype=np.float32)


def _p2r_single(poly):
    from jdet_amd.data.np_boxes import norm_angle
    poly = np.array(poly[:8], dtype=np.float32)
    pt1, pt2, pt3, pt4 = (poly[0], poly[1]), (poly[2], poly[3]), (poly[4], poly[5]), (poly[6], poly[7])
    e1 = np.sqrt((pt1[0] - pt2[0]) ** 2 + (pt1[1] - pt2[1]) ** 2)
    e2 = np.sqrt((pt2[0] - pt3[0]) ** 2 + (pt2[1] - pt3[1]) ** 2)
    if e1 > e2:
        ang = np.arctan2(np.float64(pt2[1] - pt1[1]), np.float64(pt2[0] - pt1[0]))
    else:
        ang = np.arctan2(np.float64(pt4[1] - pt1[1]), np.float64(pt4[0] - pt1[0]))
    return np.array([np.float64(pt1[0] + pt3[0]) / 2, np.float64(pt1[1] + pt3[1]) / 2, max(e1, e2), min(e1, e2),
                     norm_angle(ang)], dtype=np.float32)


def test_np_box_algebra_matches_per_box_restatement():
    from jdet_amd.data import np_boxes as B
    rng = np.random.default_rng(0)
    r = _rboxes(rng, 300)
    r[:5, 4] = [0.0, math.pi / 2, -math.pi / 4, math.pi / 4, 3 * math.pi / 4 - 1e-4]     # axis-aligned / range ends
    r[5, 2:4] = 30.0                                                                       # square: tie of the edges
    polys = B.rotated_box_to_poly_np(r)
    ref = np.stack([_best_begin_single(_r2p_single(b).tolist()) for b in r]).astype(np.float32)
    np.testing.assert_allclose(polys, ref, rtol=0, atol=3e-5)    # same vertices, same begin point (1 ulp: dot vs a*b+c*d)
    back = B.poly_to_rotated_box_np(polys)
    np.testing.assert_allclose(back, np.stack([_p2r_single(p) for p in polys]), rtol=0, atol=1e-6)
    # same rectangle up to the (w,h,theta) <-> (h,w,theta +- pi/2) ambiguity: compare through the polygons' extents
    hb, _ = B.rotated_box_to_bbox_np(r)
    hb2, _ = B.rotated_box_to_bbox_np(back)
    np.testing.assert_allclose(hb, hb2, atol=2e-3)
    assert B.rotated_box_to_poly_np(np.zeros((0, 5), np.float32)).shape == (0, 8)
    assert B.rotated_box_to_bbox_np(np.zeros((0, 5)))[0].shape == (0, 4)
    a = np.array([-1.0, 0.0, 2.5, 4.0])
    np.testing.assert_allclose(B.norm_angle(a), (a + math.pi / 4) % math.pi - math.pi / 4)


def _target(r, w, h):
    from jdet_amd.data.np_boxes import rotated_box_to_bbox_np
    hb, polys = rotated_box_to_bbox_np(r)
    return dict(rboxes=r.copy(), hboxes=hb.astype(np.float32), polys=polys.copy(), labels=np.ones(len(r), np.int32),
                rboxes_ignore=np.zeros((0, 5), np.float32), hboxes_ignore=np.zeros((0, 4)),
                polys_ignore=np.zeros((0, 8)), img_size=(w, h), ori_img_size=(w, h), scale_factor=1.0)


def test_transforms_closed_forms():
    from jdet_amd.data import transforms as T
    rng = np.random.default_rng(1)
    w, h = 200, 160
    img = Image.fromarray(rng.integers(0, 255, (h, w, 3), dtype=np.uint8))
    r = _rboxes(rng, 20, 150.0)
    # flip twice = identity (angles through norm_angle); image pixels mirrored
    for direction in ("horizontal", "vertical"):
        f = T.RotatedRandomFlip(prob=1.0, direction=direction)
        im1, t1 = f(img, _target(r, w, h))
        assert t1["flip"] == direction
        a0, a1 = np.array(img), np.array(im1)
        np.testing.assert_array_equal(a1, a0[:, ::-1] if direction == "horizontal" else a0[::-1])
        im2, t2 = f(im1, t1)
        np.testing.assert_array_equal(np.array(im2), a0)
        np.testing.assert_allclose(t2["rboxes"][:, :4], r[:, :4], atol=1e-4)
        d = (t2["rboxes"][:, 4] - r[:, 4] + math.pi / 2) % math.pi - math.pi / 2
        np.testing.assert_allclose(d, 0, atol=1e-5)
        np.testing.assert_allclose(t2["polys"], _target(r, w, h)["polys"], atol=1e-4)
    t = T.RotatedRandomFlip(prob=1.0, direction="horizontal")(img, _target(r, w, h))[1]
    np.testing.assert_allclose(t["rboxes"][:, 0], w - r[:, 0] - 1, atol=1e-5)
    np.testing.assert_allclose(t["hboxes"][:, 0], w - _target(r, w, h)["hboxes"][:, 2], atol=1e-5)
    assert T.RotatedRandomFlip(prob=0.0)(img, _target(r, w, h))[1].get("flip") is None
    # resize: short side to 320 (x2), boxes scale with it
    rs = T.RotatedResize(min_size=320, max_size=4000)
    (oh, ow), sf = rs.get_size((w, h))
    assert (oh, ow) == (240, 300) and sf == 1.5           # clipped to 1.5 x the short side (160 -> 240)
    r = r.copy()
    r[:, :2] = rng.uniform(60, 100, (len(r), 2))          # fully inside the image: no vertex gets clipped
    im, t = rs(img, _target(r, w, h))
    assert im.size == (300, 240) and t["img_size"] == (300, 240) and t["pad_shape"] == (300, 240)
    assert t["scale_factor"] == 1.5 and t["keep_ratio"] is True
    np.testing.assert_allclose(t["rboxes"][:, :2], r[:, :2] * 1.5, rtol=2e-3, atol=0.2)
    # the refit from the polygon names the long edge w (poly_to_rotated_box): compare as (long, short)
    np.testing.assert_allclose(t["rboxes"][:, 2:4], np.sort(r[:, 2:4], 1)[:, ::-1] * 1.5, rtol=2e-3, atol=0.2)
    assert bool((t["rboxes"][:, 2] >= t["rboxes"][:, 3]).all())
    assert T.Resize(1024, 1024).get_size((1024, 1024)) == ((1024, 1024), 1.)
    assert T.Resize(800, 1333).get_size((2000, 1000))[0] == (666, 1332)     # long side capped by max_size (int(666 * 2))
    assert T.Resize(64, 128, keep_ratio=False).get_size((50, 40)) == ((64, 128), 64 / 40)
    # pad to a multiple of 32 with zeros, normalise with / without the BGR swap
    im, t = T.Pad(size_divisor=32)(img, dict())
    assert im.size == (224, 160) and t["pad_shape"] == (224, 160)
    assert np.array(im)[:, 200:].max() == 0
    mean, std = [123.675, 116.28, 103.53], [58.395, 57.12, 57.375]
    x, t = T.Normalize(mean, std, to_bgr=False)(img, dict())
    assert x.shape == (3, h, w) and x.dtype == np.float32 and t["to_bgr"] is False
    np.testing.assert_allclose(x[1], (np.array(img)[:, :, 1] - mean[1]) / std[1], rtol=1e-5, atol=1e-6)
    xb, _ = T.Normalize(mean, std, to_bgr=True)(img, dict())
    np.testing.assert_allclose(xb[0], (np.array(img)[:, :, 2] - mean[0]) / std[0], rtol=1e-5, atol=1e-6)
    with pytest.raises(AssertionError):
        T.Pad(size=(10, 10), size_divisor=32)
    with pytest.raises(TypeError):
        T.Compose([3])


S2ANET_TRAIN_TRANSFORMS = [   # configs/s2anet/s2anet_r50_fpn_1x_dota.py: dataset.train.transforms
    dict(type="RotatedResize", min_size=1024, max_size=1024),
    dict(type="RotatedRandomFlip", prob=0.5),
    dict(type="Pad", size_divisor=32),
    dict(type="Normalize", mean=[123.675, 116.28, 103.53], std=[58.395, 57.12, 57.375], to_bgr=False)]


def _make_dataset(root, sizes, rng, empty=()):
    os.makedirs(os.path.join(root, "images"))
    infos = []
    for i, (w, h) in enumerate(sizes):
        name = "P%04d.png" % i
        Image.fromarray(rng.integers(0, 255, (h, w, 3), dtype=np.uint8)).save(os.path.join(root, "images", name))
        n = 0 if i in empty else int(rng.integers(1, 6))
        infos.append(dict(filename=name, width=w, height=h, ann=dict(
            bboxes=_rboxes(rng, n, min(w, h)) if n else np.zeros((0, 5), np.float32),
            labels=rng.integers(1, 16, n).astype(np.int64), bboxes_ignore=np.zeros((0, 5), np.float32),
            labels_ignore=np.zeros((0,), np.int64))))
    with open(os.path.join(root, "labels.pkl"), "wb") as f:
        pickle.dump(infos, f)
    return infos


def test_dota_dataset_collate_loader_and_feeder(tmp_path):
    from jdet_amd.data import DeviceFeeder, DOTADataset, ImageDataset, collate_batch
    from jdet_amd.utils.registry import DATASETS, build_from_cfg
    rng = np.random.default_rng(5)
    root = str(tmp_path / "trainval")
    infos = _make_dataset(root, [(96, 64), (64, 64), (80, 120), (64, 96), (50, 40)], rng, empty=(1,))
    tfm = [dict(type="RotatedResize", min_size=64, max_size=128), dict(type="RotatedRandomFlip", prob=0.5),
           dict(type="Pad", size_divisor=32), S2ANET_TRAIN_TRANSFORMS[3]]
    ds = build_from_cfg(dict(type="DOTADataset", dataset_dir=root, transforms=tfm, batch_size=2, num_workers=2,
                             shuffle=False, filter_min_size=45), DATASETS)
    assert isinstance(ds, DOTADataset) and len(ds.CLASSES) == 15
    assert len(ds) == 3          # the empty record and the 50x40 image (min side < 45) are filtered out
    image, t = ds[0]
    assert image.dtype == np.float32 and image.shape[0] == 3 and image.shape[1] % 32 == 0 and image.shape[2] % 32 == 0
    assert set(t) >= {"rboxes", "hboxes", "polys", "labels", "rboxes_ignore", "hboxes_ignore", "polys_ignore",
                      "classes", "ori_img_size", "img_size", "scale_factor", "filename", "img_file", "pad_shape",
                      "mean", "std", "to_bgr", "keep_ratio"}
    assert t["rboxes"].dtype == np.float32 and t["labels"].dtype == np.int32 and t["ori_img_size"] == (96, 64)
    assert t["pad_shape"] == (image.shape[2], image.shape[1]) and t["rboxes"].shape[1] == 5
    imgs, ts = collate_batch([ds[0], ds[1]])
    assert imgs.shape[0] == 2 and imgs.shape[2] == max(ds[0][0].shape[1], ds[1][0].shape[1])
    seen = []
    feeder = DeviceFeeder(ds.loader(), "cpu")
    assert len(feeder) == 2
    for images, targets in feeder:
        assert torch.is_tensor(images) and images.dtype == torch.float32 and images.dim() == 4
        assert images.is_contiguous(memory_format=torch.channels_last)
        for tg in targets:
            assert torch.is_tensor(tg["rboxes"]) and tg["rboxes"].dtype == torch.float32
            assert tg["labels"].dtype == torch.int32 and tg["rboxes"].shape[0] == tg["labels"].shape[0] > 0
            seen.append(tg["filename"])
    assert sorted(seen) == ["P0000.png", "P0002.png", "P0003.png"]
    # category balancing repeats rare classes; DOTA submission files
    bal = DOTADataset(dataset_dir=root, transforms=tfm, balance_category=True)
    assert len(bal) >= len(ds)
    res = [((np.array([[30., 30., 20., 10., 0.3, 0.9], [50., 40., 8., 8., 0.0, 0.5]]), np.array([0, 14])),
            "P0000.png")]
    ds.parse_result(res, str(tmp_path / "sub"))
    rows = open(tmp_path / "sub" / "plane.txt").read().split()
    assert rows[0] == "P0000" and rows[1] == "0.9000" and len(rows) == 10
    assert os.path.exists(tmp_path / "sub" / "helicopter.txt")
    # test-time dataset: images only
    ids = ImageDataset(images_dir=os.path.join(root, "images"), transforms=[tfm[0], tfm[2], tfm[3]])
    assert len(ids) == 5
    im, tt = ids[2]
    assert set(tt) >= {"ori_img_size", "img_size", "scale_factor", "img_file", "pad_shape"} and "rboxes" not in tt
    assert tt["ori_img_size"] == (80, 120) and im.shape[0] == 3


def test_runner_fit_over_a_dataset(tmp_path):
    """Runner.fit: dataset -> loader -> DeviceFeeder -> train_step, with the production Runner and a tiny CPU model
    (the detectors need a HIP device)"""
    from tests.test_ddp_gloo import CFG, _register_tiny
    from jdet_amd.data import DOTADataset
    from jdet_amd.runner import Runner
    _register_tiny()
    rng = np.random.default_rng(9)
    root = str(tmp_path / "train")
    _make_dataset(root, [(64, 64)] * 6, rng)
    tfm = [dict(type="RotatedResize", min_size=64, max_size=64), dict(type="Pad", size_divisor=32),
           S2ANET_TRAIN_TRANSFORMS[3]]
    ds = DOTADataset(dataset_dir=root, transforms=tfm, batch_size=2, num_workers=0, shuffle=True)
    torch.manual_seed(0)
    r = Runner(CFG, device="cpu", channels_last=False, ddp=False, conv_autotune=False)
    loss, parts = r.fit(ds, max_epoch=2)
    assert r.iter == 6 and r.epoch == 2 and torch.isfinite(loss) and "loss_reg" in parts
    loss2, _ = r.fit(ds, max_epoch=5, max_iter=8)
    assert r.iter == 8 and torch.isfinite(loss2)


# ---- DOTA AP: batched-IoU implementation against the per-detection loop of devkits/voc_eval.py:L236-336 ---------------
def _oracle_iou_matrix(det_polys, gt_polys):
    from jdet_amd.data.np_boxes import poly_to_rotated_box_np
    from oracle import oracle as O
    return O.box_iou_rotated(poly_to_rotated_box_np(det_polys), poly_to_rotated_box_np(gt_polys))


def _voc_eval_loop(dets, gts, ovthresh=0.5):
    """restatement of the reference's loop (hbb prefilter with the +1 convention, pairwise IoU, take-once flags)"""
    from jdet_amd.data.voc_eval import voc_ap
    dets = np.array(dets.tolist())
    gts = {k: dict(box=v["box"].copy(), difficult=v["difficult"].copy(), det=[False] * len(v["box"]))
           for k, v in gts.items()}
    npos = sum([sum(~gts[k]["difficult"]) for k in gts])
    nd = len(dets)
    if nd == 0 or npos == 0:
        return 0., 0., 0.
    confidence, dets = dets[:, -1], dets[:, :-1]
    dets = dets[np.argsort(-confidence), :]
    tp, fp = np.zeros(nd), np.zeros(nd)
    for d, det in enumerate(dets):
        bb = det[1:].astype(float)
        ovmax, jmax = -np.inf, -1
        R = gts[int(det[0])]
        BBGT = R["box"].astype(float)
        if BBGT.size > 0:
            gx0, gy0 = BBGT[:, 0::2].min(1), BBGT[:, 1::2].min(1)
            gx1, gy1 = BBGT[:, 0::2].max(1), BBGT[:, 1::2].max(1)
            bx0, by0, bx1, by1 = bb[0::2].min(), bb[1::2].min(), bb[0::2].max(), bb[1::2].max()
            iw = np.maximum(np.minimum(gx1, bx1) - np.maximum(gx0, bx0) + 1., 0.)
            ih = np.maximum(np.minimum(gy1, by1) - np.maximum(gy0, by0) + 1., 0.)
            inters = iw * ih
            uni = (bx1 - bx0 + 1.) * (by1 - by0 + 1.) + (gx1 - gx0 + 1.) * (gy1 - gy0 + 1.) - inters
            keep = np.where(inters / uni > 0)[0]
            if len(keep) > 0:
                ov = [float(_oracle_iou_matrix(bb[None].astype(np.float32), BBGT[j][None].astype(np.float32))[0, 0])
                      for j in keep]
                ovmax, jmax = np.max(ov), keep[int(np.argmax(ov))]
        if ovmax > ovthresh:
            if not R["difficult"][jmax]:
                if not R["det"][jmax]:
                    tp[d] = 1.
                    R["det"][jmax] = 1
                else:
                    fp[d] = 1.
        else:
            fp[d] = 1.
    fp, tp = np.cumsum(fp), np.cumsum(tp)
    rec = tp / float(npos)
    prec = tp / np.maximum(tp + fp, np.finfo(np.float64).eps)
    return rec, prec, voc_ap(rec, prec)


def test_voc_eval_dota_matches_the_per_detection_loop():
    from jdet_amd.data.np_boxes import rotated_box_to_poly_np
    from jdet_amd.data.voc_eval import evaluate_dota, voc_ap, voc_eval_dota
    rng = np.random.default_rng(12)
    gts, dets = {}, []
    for img in range(6):
        k = int(rng.integers(0, 7))
        g = _rboxes(rng, k, 300.0)
        gts[img] = dict(box=rotated_box_to_poly_np(g).astype(np.float64).reshape(-1, 8),
                        difficult=rng.uniform(0, 1, k) < 0.25)
        # detections: jittered copies of gts (some twice -> duplicates), plus random false alarms
        cand = np.concatenate([g, g[: k // 2], _rboxes(rng, 4, 300.0)]) if k else _rboxes(rng, 3, 300.0)
        cand = cand + rng.normal(0, 1.0, cand.shape).astype(np.float32) * np.array([1, 1, 1, 1, 0.02], np.float32)
        p = rotated_box_to_poly_np(cand)
        sc = rng.uniform(0.05, 1, len(p)) + np.arange(len(p)) * 1e-6 + img * 1e-4     # no ties
        dets.append(np.concatenate([np.full((len(p), 1), img), p, sc[:, None]], 1))
    dets = np.concatenate(dets)
    rec, prec, ap = voc_eval_dota(dets, gts, _oracle_iou_matrix)
    rec2, prec2, ap2 = _voc_eval_loop(dets, gts)
    np.testing.assert_array_equal(rec, rec2)
    np.testing.assert_array_equal(prec, prec2)
    assert ap == ap2 and 0.2 < ap <= 1.0
    # closed forms: perfect detections -> AP 1; nothing -> 0; 11-point metric; PR envelope
    perfect = np.concatenate([np.concatenate([np.full((len(v["box"]), 1), k), v["box"],
                                              np.linspace(0.9, 0.5, len(v["box"]))[:, None]], 1)
                              for k, v in gts.items() if len(v["box"])])
    easy = {k: dict(box=v["box"], difficult=np.zeros(len(v["box"]), bool)) for k, v in gts.items()}
    assert abs(voc_eval_dota(perfect, easy, _oracle_iou_matrix)[2] - 1.0) < 1e-12
    assert voc_eval_dota(np.zeros((0, 10)), easy, _oracle_iou_matrix) == (0., 0., 0.)
    r, p = np.array([0.5, 0.5, 1.0]), np.array([1.0, 0.5, 2 / 3])
    assert abs(voc_ap(r, p) - (0.5 * 1.0 + 0.5 * 2 / 3)) < 1e-12
    assert abs(voc_ap(r, p, use_07_metric=True) - (6 * 1.0 + 5 * 2 / 3) / 11) < 1e-12
    # dataset-level wrapper: classes, 0-based detection labels, scale factor, difficult (ignore) polygons
    classes = ["a", "b", "c"]
    results = []
    for img in range(4):
        g = _rboxes(rng, 5, 300.0)
        labels = rng.integers(1, 3, 5)
        gp = rotated_box_to_poly_np(g)
        results.append(((gp.copy(), np.linspace(0.9, 0.6, 5), labels - 1),
                        dict(scale_factor=2.0, polys=gp * 2.0, labels=labels,
                             polys_ignore=rotated_box_to_poly_np(_rboxes(rng, 1, 300.0)) * 2.0)))
    aps = evaluate_dota(results, classes, _oracle_iou_matrix)
    assert set(aps) == {"eval/1_a_AP", "eval/2_b_AP", "eval/3_c_AP", "eval/0_meanAP"}
    assert abs(aps["eval/1_a_AP"] - 1.0) < 1e-12 and abs(aps["eval/2_b_AP"] - 1.0) < 1e-12 and aps["eval/3_c_AP"] == 0
    assert abs(aps["eval/0_meanAP"] - 2 / 3) < 1e-12
    assert evaluate_dota([], classes, _oracle_iou_matrix)["eval/0_meanAP"] == 0


# ---- merging tile detections (devkits/result_merge.py) ----------------------------------------------------------------
def _oracle_group_nms(polys, scores, groups, thresh):
    """per-group greedy NMS with the C++ oracle (cmp_ge=0: iou > thresh suppresses)"""
    from jdet_amd.data.np_boxes import poly_to_rotated_box_np
    from oracle import oracle as O
    keep = np.zeros(len(polys), bool)
    for g in np.unique(groups):
        idx = np.nonzero(groups == g)[0]
        boxes = poly_to_rotated_box_np(polys[idx])
        order = np.argsort(-np.asarray(scores)[idx], kind="stable").astype(np.int32)
        keep[idx] = O.nms_rotated_keep(boxes, order, thresh, cmp_ge=0).astype(bool)
    return keep


def _greedy_poly_nms(dets, thresh):
    """restatement of py_cpu_nms_poly_fast (L69-129) with the oracle as the polygon IoU"""
    x1, y1 = dets[:, 0:8:2].min(1), dets[:, 1:8:2].min(1)
    x2, y2 = dets[:, 0:8:2].max(1), dets[:, 1:8:2].max(1)
    areas = (x2 - x1 + 1) * (y2 - y1 + 1)
    order = dets[:, 8].argsort()[::-1]
    keep = []
    while order.size > 0:
        i = order[0]
        keep.append(i)
        w = np.maximum(0.0, np.minimum(x2[i], x2[order[1:]]) - np.maximum(x1[i], x1[order[1:]]))
        h = np.maximum(0.0, np.minimum(y2[i], y2[order[1:]]) - np.maximum(y1[i], y1[order[1:]]))
        ovr = w * h / (areas[i] + areas[order[1:]] - w * h)
        for j in np.where(ovr > 0)[0]:
            ovr[j] = _oracle_iou_matrix(dets[i:i + 1, :8].astype(np.float32),
                                        dets[order[j + 1]:order[j + 1] + 1, :8].astype(np.float32))[0, 0]
        order = order[np.where(ovr <= thresh)[0] + 1]
    return keep


def test_merge_tile_results(tmp_path):
    from jdet_amd.data.np_boxes import rotated_box_to_poly_np
    from jdet_amd.data.result_merge import mergebypoly, parse_tile_name
    assert parse_tile_name("P0001__1__824___0") == ("P0001", 824, 0, 1.0)
    assert parse_tile_name("P0706__0.5__1648___2472") == ("P0706", 1648, 2472, 0.5)
    rng = np.random.default_rng(3)
    src, dst = tmp_path / "raw", tmp_path / "merged"
    os.makedirs(src)
    expected = {}
    for cls in ("plane", "harbor"):
        rows, full = [], {}
        for img in ("P0001", "P0002"):
            objs = I_clustered(rng, 40)
            for (tx, ty) in ((0, 0), (824, 0), (0, 824)):
                for rate in (1.0, 0.5):
                    # every object is reported by every tile that "sees" it, with a little jitter
                    b = objs + rng.normal(0, 0.5, objs.shape).astype(np.float32) * np.array([1, 1, 1, 1, 0.01], np.float32)
                    polys = rotated_box_to_poly_np(b).astype(np.float64)
                    tile = polys * rate
                    tile[:, 0::2] -= tx
                    tile[:, 1::2] -= ty
                    sc = rng.uniform(0.1, 1.0, len(b))
                    name = "%s__%g__%d___%d" % (img, rate, tx, ty)
                    for p, s_ in zip(tile, sc):
                        rows.append(name + " " + "%.6f" % s_ + " " + " ".join("%.4f" % v for v in p))
                        back = p.copy()
                        back[0::2] = (np.array([float("%.4f" % v) for v in p[0::2]]) + tx) / rate
                        back[1::2] = (np.array([float("%.4f" % v) for v in p[1::2]]) + ty) / rate
                        full.setdefault(img, []).append(np.concatenate([back, [float("%.6f" % s_)]]))
        with open(src / (cls + ".txt"), "w") as f:
            f.write("\n".join(rows) + "\n")
        expected[cls] = full
    kept = mergebypoly(str(src), str(dst), threshold_type=1, group_nms=_oracle_group_nms)
    from jdet_amd.data.result_merge import NMS_THRESHOLD_BY_CLASS
    for cls, full in expected.items():
        out = [l.split() for l in open(dst / (cls + ".txt")).read().strip().split("\n")]
        assert len(out) == kept[cls] < sum(len(v) for v in full.values())
        for img, dets in full.items():
            dets = np.stack(dets)
            ref_keep = _greedy_poly_nms(dets, NMS_THRESHOLD_BY_CLASS[cls])
            got = np.array([[float(v) for v in r[1:]] for r in out if r[0] == img])
            np.testing.assert_allclose(got[:, 0], dets[ref_keep, 8], rtol=0, atol=1e-12)    # same boxes, same order
            np.testing.assert_allclose(got[:, 1:], dets[ref_keep, :8], rtol=0, atol=1e-9)


def I_clustered(rng, n):
    from tests import inputs as I
    return I.clustered_obbs(rng, n, 8, 1600.0)


def test_group_chunks_split_on_group_boundaries():
    """tile merging: the detections of a class are de-duplicated in runs of WHOLE images with a bounded number of boxes
    (the NMS workspace is quadratic); a run never cuts an image, every detection is in exactly one run"""
    from jdet_amd.data.result_merge import group_chunks
    rng = np.random.default_rng(0)
    groups = rng.integers(0, 40, 5000)
    runs = group_chunks(groups, max_boxes=600)
    allidx = np.concatenate(runs)
    assert sorted(allidx.tolist()) == list(range(5000))
    seen = set()
    for r in runs:
        gs = set(groups[r].tolist())
        assert not gs & seen            # an image lives in one run only
        seen |= gs
        sizes = [int((groups == g).sum()) for g in gs]
        assert len(r) <= 600 or len(gs) == 1, (len(r), sizes)
    assert len(runs) > 1
    one = group_chunks(np.zeros(1000, int), max_boxes=100)     # a single oversized image stays whole
    assert len(one) == 1 and len(one[0]) == 1000


def test_uint8_samples_for_device_side_normalisation(tmp_path):
    """Normalize(on_device=True) leaves the sample as the decoder's uint8 (h,w,3) array; collate_batch pads uint8 and
    records each image's extent -- the host half of the PCIe-saving path (the device half: tests/test_gpu_data.py)"""
    from jdet_amd.data import DOTADataset, collate_batch
    rng = np.random.default_rng(6)
    root = str(tmp_path / "trainval")
    _make_dataset(root, [(96, 64), (64, 64), (80, 120)], rng)
    norm = dict(S2ANET_TRAIN_TRANSFORMS[3])
    tfm = [dict(type="RotatedResize", min_size=64, max_size=128), dict(type="Pad", size_divisor=32)]
    host = DOTADataset(dataset_dir=root, transforms=tfm + [norm])
    dev = DOTADataset(dataset_dir=root, transforms=tfm + [dict(norm, on_device=True)])
    (xh, th), (xd, td) = host[0], dev[0]
    assert xd.dtype == np.uint8 and xd.shape == (xh.shape[1], xh.shape[2], 3) and td["normalize_on_device"] is True
    mean, std = np.float32(norm["mean"]).reshape(3, 1, 1), np.float32(norm["std"]).reshape(3, 1, 1)
    assert np.array_equal(xh, (xd.transpose(2, 0, 1) - mean) / std)          # same float32 arithmetic, same bits
    imgs, ts = collate_batch([dev[0], dev[2]])
    assert imgs.dtype == np.uint8 and imgs.shape[0] == 2 and imgs.shape[3] == 3
    assert ts[0]["canvas_hw"] == dev[0][0].shape[:2] and ts[1]["canvas_hw"] == dev[2][0].shape[:2]
    assert imgs[0, ts[0]["canvas_hw"][0]:].max(initial=0) == 0 and imgs[1, :, ts[1]["canvas_hw"][1]:].max(initial=0) == 0
