"""DOTA average precision.  Contract of python/jdet/data/devkits/voc_eval.py: `voc_ap` L39-71, `voc_eval_dota`
L236-336 (detections of one class over all images, sorted by confidence; a detection is a true positive if its best
IoU against the ground truths of its image exceeds `ovthresh` and that ground truth is neither difficult nor already
taken), and of `DOTADataset.evaluate` (data/dota.py:L89-146).

The reference walks the detections in Python and calls a polygon-IoU op once per (detection, candidate gt) pair.  A
detection's best ground truth and its IoU do not depend on the matching state, so here the IoU matrices of a class
are computed per image in one device launch each (detections and ground truths are rectangles: polygons of rotated
boxes -> `box_iou_rotated`), and only the take-once bookkeeping remains a host loop.
"""
import numpy as np

from .np_boxes import poly_to_rotated_box_np, polys_are_rectangles


def voc_ap(rec, prec, use_07_metric=False):
    if use_07_metric:                              # 11-point interpolation
        ap = 0.
        for t in np.arange(0., 1.1, 0.1):
            ap += (np.max(prec[rec >= t]) if np.sum(rec >= t) > 0 else 0.) / 11.
        return ap
    mrec = np.concatenate(([0.], rec, [1.]))
    mpre = np.concatenate(([0.], prec, [0.]))
    mpre = np.maximum.accumulate(mpre[::-1])[::-1]  # precision envelope
    i = np.where(mrec[1:] != mrec[:-1])[0]
    return float(np.sum((mrec[i + 1] - mrec[i]) * mpre[i + 1]))


def device_iou_matrix(det_polys, gt_polys, device=None):
    """(m,8) x (k,8) polygons given by their corners -> (m,k) IoU on the HIP device.  Rectangles (what the detectors
    emit and most DOTA labels are) go through the rotated-box kernel; anything else -- foreign submissions, skewed
    ground-truth quadrilaterals -- through the polygon kernel with `iou_poly`'s rule (nms_poly.py:L247-252)."""
    import torch
    dev = torch.device("cuda") if device is None else torch.device(device)
    if polys_are_rectangles(det_polys) and polys_are_rectangles(gt_polys):
        from jdet_amd.ops import box_iou_rotated
        a = torch.from_numpy(poly_to_rotated_box_np(det_polys)).to(dev)
        b = torch.from_numpy(poly_to_rotated_box_np(gt_polys)).to(dev)
        return box_iou_rotated(a, b).cpu().numpy()
    from jdet_amd.ops.nms_poly import poly_iou_matrix
    a = torch.from_numpy(np.asarray(det_polys, np.float32).reshape(-1, 8)).to(dev)
    b = torch.from_numpy(np.asarray(gt_polys, np.float32).reshape(-1, 8)).to(dev)
    return poly_iou_matrix(a, b, 1).cpu().numpy()


def voc_eval_dota(dets, gts, iou_matrix_fn=device_iou_matrix, ovthresh=0.5, use_07_metric=False):
    """dets (nd, 10) rows [image index, 8 polygon coordinates, confidence]; gts {image index: {"box" (k,8),
    "difficult" (k,) bool}} -> (recall curve, precision curve, ap); (0, 0, 0) without detections or positives."""
    dets = np.asarray(dets, dtype=np.float64).reshape(-1, 10)
    npos = int(sum(int(np.sum(~np.asarray(g["difficult"], bool))) for g in gts.values()))
    nd = dets.shape[0]
    if nd == 0 or npos == 0:
        return 0., 0., 0.
    order = np.argsort(-dets[:, -1])
    dets = dets[order]
    img_of = dets[:, 0].astype(np.int64)
    best_iou = np.full(nd, -np.inf)
    best_gt = np.zeros(nd, dtype=np.int64)
    for img in np.unique(img_of):
        g = gts.get(int(img))
        if g is None or np.asarray(g["box"]).size == 0:
            continue
        rows = np.nonzero(img_of == img)[0]
        iou = np.asarray(iou_matrix_fn(dets[rows, 1:9].astype(np.float32),
                                       np.asarray(g["box"], np.float32).reshape(-1, 8)), dtype=np.float64)
        best_gt[rows] = iou.argmax(1)
        best_iou[rows] = iou.max(1)
    taken = {k: np.zeros(len(np.asarray(g["difficult"])), bool) for k, g in gts.items()}
    tp, fp = np.zeros(nd), np.zeros(nd)
    for d in range(nd):
        if best_iou[d] > ovthresh:
            k, j = int(img_of[d]), int(best_gt[d])
            if not gts[k]["difficult"][j]:
                if not taken[k][j]:
                    tp[d] = 1.
                    taken[k][j] = True
                else:
                    fp[d] = 1.
        else:
            fp[d] = 1.
    fp, tp = np.cumsum(fp), np.cumsum(tp)
    rec = tp / float(npos)
    prec = tp / np.maximum(tp + fp, np.finfo(np.float64).eps)
    return rec, prec, voc_ap(rec, prec, use_07_metric)


def evaluate_dota(results, classes, iou_matrix_fn=device_iou_matrix):
    """results: [((det_polys (k,8), det_scores (k,), det_labels (k,) 0-based), target dict)] per image ->
    {"eval/<i>_<class>_AP": ap, ..., "eval/0_meanAP": mean} (data/dota.py:L89-146)"""
    dets, gts, hard = [], [], {}
    for img_idx, ((polys, scores, labels), target) in enumerate(results):
        polys, scores = np.asarray(polys, np.float64).reshape(-1, 8), np.asarray(scores, np.float64).reshape(-1)
        labels = np.asarray(labels).reshape(-1) + 1
        if polys.shape[0] > 0:
            dets.append(np.concatenate([np.full((len(labels), 1), img_idx), polys, scores[:, None],
                                        labels[:, None]], 1))
        sf = target["scale_factor"]
        gt_polys = np.asarray(target["polys"], np.float64).reshape(-1, 8) / sf
        if gt_polys.shape[0] > 0:
            gts.append(np.concatenate([np.full((len(gt_polys), 1), img_idx), gt_polys,
                                       np.asarray(target["labels"]).reshape(-1, 1)], 1))
        hard[img_idx] = np.asarray(target["polys_ignore"], np.float64).reshape(-1, 8) / sf
    aps = {}
    if not dets or not gts:
        for i, name in enumerate(classes):
            aps["eval/%d_%s_AP" % (i + 1, name)] = 0
    else:
        dets, gts = np.concatenate(dets), np.concatenate(gts)
        for i, name in enumerate(classes):
            c_dets = dets[dets[:, -1] == i + 1][:, :-1]
            c_gts = gts[gts[:, -1] == i + 1][:, :-1]
            per_img = {}
            for idx in np.unique(gts[:, 0]):
                g = c_gts[c_gts[:, 0] == idx][:, 1:]
                dg = hard[int(idx)]
                per_img[int(idx)] = {"box": np.concatenate([g, dg]),
                                     "difficult": np.concatenate([np.zeros(len(g), bool), np.ones(len(dg), bool)])}
            aps["eval/%d_%s_AP" % (i + 1, name)] = voc_eval_dota(c_dets, per_img, iou_matrix_fn)[2]
    aps["eval/0_meanAP"] = sum(aps.values()) / len(aps)
    return aps
