"""Merging per-tile detections into full-image DOTA submissions.

Contract of python/jdet/data/devkits/result_merge.py: one text file per class, rows
`<tile name> <score> x0 y0 ... x3 y3`; a tile name `P0001__1__824___0` carries the resize rate and the tile's
offset in the resized image (L221-232), so a polygon maps back as (v + offset) / rate (`poly2origpoly` L199-206);
detections of one original image then go through greedy NMS by descending score that drops a box when its IoU with
a kept one is > thresh (`py_cpu_nms_poly_fast` L69-129 keeps `iou <= thresh`), with one global threshold (0.1) or
the per-class table (L24-31); the survivors are written `<image> <score> <8 coordinates>` (L248-258).

The reference loops images in Python and calls a polygon-IoU op per candidate pair, 16 worker processes over the
class files.  Here all detections of a class are ONE rotated-NMS launch: the original-image index is the label
(boxes of different images never suppress each other), the polygons are rectangles (they were written from rotated
boxes) and go back to (xc, yc, w, h, theta) for the kernel.
"""
import os
import re

import numpy as np

from .np_boxes import poly_to_rotated_box_np, polys_are_rectangles

NMS_THRESHOLD = 0.1
NMS_THRESHOLD_BY_CLASS = {"roundabout": 0.1, "tennis-court": 0.3, "swimming-pool": 0.1, "storage-tank": 0.2,
                          "soccer-ball-field": 0.3, "small-vehicle": 0.2, "ship": 0.2, "plane": 0.3,
                          "large-vehicle": 0.1, "helicopter": 0.2, "harbor": 0.0001, "ground-track-field": 0.3,
                          "bridge": 0.0001, "basketball-court": 0.3, "baseball-diamond": 0.3,
                          "container-crane": 0.05, "airport": 0.1, "helipad": 0.1}
_OFFSET = re.compile(r"__(\d+)___(\d+)")
_RATE = re.compile(r"__([\d+\.]+)__\d+___")


def parse_tile_name(subname):
    """'P0001__0.5__824___1024' -> ('P0001', 824, 1024, 0.5)"""
    x, y = _OFFSET.findall(subname)[0]
    return subname.split("__")[0], int(x), int(y), float(_RATE.findall(subname)[0])


MAX_BOXES_PER_LAUNCH = 32768   # the suppression matrix is n x ceil(n / 64) words: 128 MiB at this size


def group_chunks(groups, max_boxes=MAX_BOXES_PER_LAUNCH):
    """Split the detections into runs of WHOLE groups with at most `max_boxes` boxes each (groups never interact, so
    every run is an independent NMS problem; the workspace of one launch is quadratic in its size).  A single group
    larger than `max_boxes` becomes a run of its own.  Returns index arrays."""
    groups = np.asarray(groups)
    order = np.argsort(groups, kind="stable")
    g_sorted = groups[order]
    starts = np.flatnonzero(np.r_[True, g_sorted[1:] != g_sorted[:-1]])
    ends = np.r_[starts[1:], len(groups)]
    runs, cur, cur_n = [], [], 0
    for a, b in zip(starts, ends):
        if cur and cur_n + (b - a) > max_boxes:
            runs.append(np.concatenate(cur))
            cur, cur_n = [], 0
        cur.append(order[a:b])
        cur_n += b - a
    if cur:
        runs.append(np.concatenate(cur))
    return runs


def device_group_nms(polys, scores, groups, thresh, device=None, max_boxes=MAX_BOXES_PER_LAUNCH):
    """greedy NMS inside every group (IoU > thresh suppresses) -> bool keep mask.  One launch per run of whole groups
    (all groups at once when they fit `max_boxes`)."""
    import torch
    from jdet_amd.ops.nms_rotated import nms_rotated_keep_mask
    dev = torch.device("cuda") if device is None else torch.device(device)
    rect = polys_are_rectangles(polys)     # general quadrilaterals (foreign result files): polygon kernel
    if rect:
        rb = np.concatenate([poly_to_rotated_box_np(polys), np.asarray(groups, np.float32)[:, None]], 1)
    else:
        from jdet_amd.ops.nms_poly import poly_nms_keep_mask
        rb = np.concatenate([np.asarray(polys, np.float32).reshape(-1, 8), np.asarray(groups, np.float32)[:, None]], 1)
    sc = np.asarray(scores, np.float32)
    keep = np.zeros(len(sc), bool)
    for idx in group_chunks(groups, max_boxes):
        if len(idx) > 524288:
            raise ValueError("%d detections of one original image in one class: beyond what one rotated-NMS launch "
                             "handles (524288); filter by score first" % len(idx))
        boxes = torch.from_numpy(rb[idx]).to(dev)
        s = torch.from_numpy(sc[idx]).to(dev)
        order = torch.argsort(s, descending=True, stable=True)
        order = order[torch.argsort(boxes[order, -1], stable=True)]
        mask = nms_rotated_keep_mask(boxes, order, thresh, rule="cuda") if rect else \
            poly_nms_keep_mask(boxes, order, thresh)
        keep[idx] = mask.cpu().numpy().astype(bool)
    return keep


def merge_class_file(src, dst, thresh, group_nms=device_group_nms):
    names, rows = [], []
    with open(src, "r") as f:
        for line in f:
            parts = line.strip().split(" ")
            if len(parts) < 10:
                continue
            ori, x, y, rate = parse_tile_name(parts[0])
            poly = np.array(list(map(float, parts[2:10])))
            poly[0::2] = (poly[0::2] + x) / rate
            poly[1::2] = (poly[1::2] + y) / rate
            names.append(ori)
            rows.append(np.concatenate([poly, [float(parts[1])]]))
    with open(dst, "w") as out:
        if not rows:
            return 0
        rows = np.stack(rows)
        first = {}
        for n in names:                             # image ids in order of first appearance, as the reference's dict
            first.setdefault(n, len(first))
        uniq = list(first)
        groups = np.array([first[n] for n in names])
        keep = np.asarray(group_nms(rows[:, :8].astype(np.float32), rows[:, 8], groups, thresh), bool)
        # per image, survivors in descending score (the order the reference's greedy loop emits them)
        for g in range(len(uniq)):
            idx = np.nonzero((groups == g) & keep)[0]
            for i in idx[np.argsort(-rows[idx, 8], kind="stable")]:
                out.write(uniq[g] + " " + str(rows[i, 8]) + " " + " ".join(map(str, rows[i, :8])) + "\n")
    return int(keep.sum())


def mergebypoly(srcpath, dstpath, threshold_type=0, group_nms=device_group_nms):
    """every `<class>.txt` under srcpath -> merged + NMS'd `<class>.txt` under dstpath; threshold_type 0: 0.1 for
    all classes, otherwise the per-class table (cfg.merge_nms_threshold_type in the reference)"""
    os.makedirs(dstpath, exist_ok=True)
    kept = {}
    for root, _, files in os.walk(srcpath):
        for fn in sorted(files):
            cls = os.path.splitext(fn)[0]
            thr = NMS_THRESHOLD if threshold_type == 0 else NMS_THRESHOLD_BY_CLASS[cls]
            kept[cls] = merge_class_file(os.path.join(root, fn), os.path.join(dstpath, cls + ".txt"), thr, group_nms)
    return kept
