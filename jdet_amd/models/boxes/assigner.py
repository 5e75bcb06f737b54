"""Max-IoU assignment.  Mirrors python/jdet/models/boxes/assigner.py: `AssignResult` L53-65,
`MaxIoUAssigner` L67-219, `MaxIoUAssignerRbbox` L222-274.

`assign_wrt_overlaps` is the reference's four steps (default -1; negatives; positives; low-quality
matches, later gts overwriting earlier ones) executed as two device launches without a host sync
(csrc/box_codec_assign.hip) instead of the reference's per-gt Python loop with `jt.sync_all()`.
"""
import math

import torch

from jdet_amd import _lib as L
from jdet_amd.utils.registry import BOXES, build_from_cfg


class AssignResult:
    def __init__(self, num_gts, gt_inds, max_overlaps, labels=None):
        self.num_gts = num_gts
        self.gt_inds = gt_inds
        self.max_overlaps = max_overlaps
        self.labels = labels

    def add_gt_(self, gt_labels):
        self_inds = torch.arange(1, len(gt_labels) + 1, dtype=self.gt_inds.dtype, device=self.gt_inds.device)
        self.gt_inds = torch.cat([self_inds, self.gt_inds])
        self.max_overlaps = torch.cat([self.max_overlaps.new_ones((self.num_gts,)), self.max_overlaps])
        if self.labels is not None:
            self.labels = torch.cat([gt_labels.to(self.labels.dtype), self.labels])


def assign_wrt_overlaps_device(overlaps, pos_iou_thr, neg_iou_thr, min_pos_iou, match_low_quality,
                               gt_max_assign_all, gt_labels, labels_filled):
    """overlaps (K,A) on a HIP device -> (gt_inds int32 (A), max_overlaps (A), labels int32 (A) | None)"""
    L.need_device(overlaps)
    ov = L.f32c(overlaps)
    K, A = ov.shape
    if isinstance(neg_iou_thr, float):
        lo, hi = 0.0, neg_iou_thr
    elif isinstance(neg_iou_thr, tuple):
        assert len(neg_iou_thr) == 2
        lo, hi = float(neg_iou_thr[0]), float(neg_iou_thr[1])
    else:  # neither branch of assigner.py:L187-193 fires: no negatives
        lo, hi = math.inf, -math.inf
    gt_inds = torch.empty((A,), dtype=torch.int32, device=ov.device)
    max_ov = torch.empty((A,), dtype=torch.float32, device=ov.device)
    labels = torch.empty((A,), dtype=torch.int32, device=ov.device) if gt_labels is not None else None
    gl = gt_labels.to(torch.int32).contiguous() if gt_labels is not None else None
    wsb = L.lib().jdet_assign_max_iou_workspace(K)
    ws = torch.empty((wsb,), dtype=torch.uint8, device=ov.device)
    L.check(L.lib().jdet_assign_max_iou(L.ptr(ov), K, A, float(pos_iou_thr), lo, hi, float(min_pos_iou),
                                        int(bool(match_low_quality)), int(bool(gt_max_assign_all)), L.ptr(gl),
                                        int(labels_filled), L.ptr(gt_inds), L.ptr(max_ov), L.ptr(labels),
                                        L.ptr(ws), wsb, L.stream_ptr(ov)), "jdet_assign_max_iou")
    return gt_inds, max_ov, labels


@BOXES.register_module()
class MaxIoUAssigner:
    """-1 don't care, 0 negative, k>0 positive matched to gt k-1."""

    def __init__(self, pos_iou_thr, neg_iou_thr, min_pos_iou=.0, gt_max_assign_all=True, ignore_iof_thr=-1,
                 ignore_wrt_candidates=True, match_low_quality=True, assigned_labels_filled=0,
                 iou_calculator=dict(type="BboxOverlaps2D")):
        self.pos_iou_thr = pos_iou_thr
        self.neg_iou_thr = neg_iou_thr
        self.min_pos_iou = min_pos_iou
        self.gt_max_assign_all = gt_max_assign_all
        self.ignore_iof_thr = ignore_iof_thr
        self.ignore_wrt_candidates = ignore_wrt_candidates
        self.match_low_quality = match_low_quality
        self.assigned_labels_filled = assigned_labels_filled
        self.iou_calculator = build_from_cfg(iou_calculator, BOXES)

    def assign(self, bboxes, gt_bboxes, gt_bboxes_ignore=None, gt_labels=None):
        if bboxes.shape[0] == 0 or gt_bboxes.shape[0] == 0:
            raise ValueError("No gt or bboxes")
        overlaps = self.iou_calculator(gt_bboxes, bboxes)
        if (self.ignore_iof_thr > 0) and (gt_bboxes_ignore is not None) and (gt_bboxes_ignore.numel() > 0):
            if self.ignore_wrt_candidates:
                ignore_overlaps = self.iou_calculator(bboxes, gt_bboxes_ignore, mode="iof")
                ignore_max_overlaps = ignore_overlaps.max(dim=1).values
            else:
                ignore_overlaps = self.iou_calculator(gt_bboxes_ignore, bboxes, mode="iof")
                ignore_max_overlaps = ignore_overlaps.max(dim=0).values
            overlaps[:, ignore_max_overlaps > self.ignore_iof_thr] = -1
        return self.assign_wrt_overlaps(overlaps, gt_labels)

    def assign_wrt_overlaps(self, overlaps, gt_labels=None):
        if overlaps.numel() == 0:
            raise ValueError("No gt or proposals")
        num_gts = overlaps.size(0)
        gt_inds, max_overlaps, labels = assign_wrt_overlaps_device(
            overlaps, self.pos_iou_thr, self.neg_iou_thr, self.min_pos_iou, self.match_low_quality,
            self.gt_max_assign_all, gt_labels, self.assigned_labels_filled)
        return AssignResult(num_gts, gt_inds, max_overlaps, labels=labels)


@BOXES.register_module()
class MaxIoUAssignerRbbox(MaxIoUAssigner):
    def __init__(self, pos_iou_thr, neg_iou_thr, min_pos_iou=.0, gt_max_assign_all=True, ignore_iof_thr=-1,
                 ignore_wrt_candidates=True, iou_calculator=dict(type="BboxOverlaps2D")):
        super().__init__(pos_iou_thr=pos_iou_thr, neg_iou_thr=neg_iou_thr, min_pos_iou=min_pos_iou,
                         gt_max_assign_all=gt_max_assign_all, ignore_iof_thr=ignore_iof_thr,
                         ignore_wrt_candidates=ignore_wrt_candidates, iou_calculator=iou_calculator)

    def assign(self, bboxes, gt_bboxes, gt_bboxes_ignore=None, gt_labels=None):
        if bboxes.shape[0] == 0 or gt_bboxes.shape[0] == 0:
            raise ValueError("No gt or bboxes")
        bboxes = bboxes[:, :5]
        overlaps = self.iou_calculator(gt_bboxes, bboxes)
        # the reference's ignore branch is `assert NotImplementedError` (a no-op), assigner.py:L267-273
        return self.assign_wrt_overlaps(overlaps, gt_labels)
