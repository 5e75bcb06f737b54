"""RoI-Transformer box heads (stage 1: horizontal RoI -> rotated RoI; stage 2: rotated RoI -> detection).
Contract of python/jdet/models/roi_heads/rbbox_head.py: target builders L9-146, `accuracy` L148-167,
`BBoxHeadRbbox` L169-448 (constructor arguments, parameter names, loss keys, decode functions).

Training runs on fixed-size row sets (models/boxes/fixed_shape.py: `StageRows`, always `sampler.num` rows per image
with valid / positive masks) instead of the reference's positives-first variable-length lists: targets, losses and
the stage-1 -> stage-2 refinement are masked dense arithmetic, no boolean indexing, no device -> host round trip.
"""
import torch
import torch.nn.functional as F
from torch import nn

from jdet_amd.models.boxes.box_ops import rotated_box_to_poly
from jdet_amd.ops.bbox_transforms import (best_match_dbbox2delta, choose_best_obb_batch, choose_best_Rroi_batch,
                                          dbbox2delta_v3, delta2dbbox_v2, delta2dbbox_v3, hbb2obb_v2)
from jdet_amd.ops.nms_rotated import multiclass_nms_rotated
from jdet_amd.utils.registry import HEADS, LOSSES, build_from_cfg

from jdet_amd.ops.linear import Linear

from .roi_feature_linear import RoIFeatureLinear


def _get(cfg, key):
    return cfg[key] if isinstance(cfg, dict) else getattr(cfg, key)


def _row_targets(rows, deltas, pos_weight):
    """the four target tensors of one image's StageRows (rbbox_head.py:L40-62 / L101-121 with masks instead of the
    positives-first slices): labels (num,) long, label_weights (num,), bbox_targets (num,5), bbox_weights (num,5)"""
    pw = 1.0 if pos_weight <= 0 else pos_weight
    valid, pos = rows.valid.float(), rows.is_pos.float()
    label_weights = valid * (pos * pw + (1.0 - pos))
    bbox_targets = torch.where(rows.is_pos[:, None], deltas, torch.zeros_like(deltas))
    return rows.labels, label_weights, bbox_targets, pos[:, None].expand(-1, 5)


def hbb_row_targets(rows, gt_obbs, cfg, target_means, target_stds, with_module=False):
    """stage 1 of RoI-Transformer: horizontal sampled boxes regress to the gt in its near-(-90 degree) form
    (`bbox_target_rbbox_single`, L9-63): dbbox2delta_v3(hbb2obb_v2(box), choose_best_obb_batch(gt))"""
    if with_module:
        # rbbox_head.py:L76 calls `dbbox2delta`, a name the reference never imports: unreachable there
        raise NameError("dbbox2delta (with_module=True) is not defined in the reference either")
    gts = choose_best_obb_batch(gt_obbs[rows.matched])
    boxes = hbb2obb_v2(rows.boxes) if rows.boxes.size(1) == 4 else rows.boxes
    return _row_targets(rows, dbbox2delta_v3(boxes, gts, target_means, target_stds), _get(cfg, "pos_weight"))


def obb_row_targets(rows, gt_rbboxes, cfg, target_means, target_stds):
    """stage 2: rotated sampled boxes regress to the best-matching of the gt's four equivalent forms
    (`rbbox_target_rbbox_single`, L88-121)"""
    assert rows.boxes.size(1) == 5
    deltas = best_match_dbbox2delta(rows.boxes, gt_rbboxes[rows.matched], target_means, target_stds)
    return _row_targets(rows, deltas, _get(cfg, "pos_weight"))


def _cat_images(parts):
    return tuple(torch.cat([p[k] for p in parts]) for k in range(4))


def accuracy(pred, target, topk=1, valid=None):
    """top-k accuracy in percent over the rows marked valid (all rows by default; L148-167)"""
    return_single = isinstance(topk, int)
    topk = (topk,) if return_single else topk
    _, pred_label = pred.topk(max(topk), 1, True, True)
    correct = pred_label.t() == target.view(1, -1).to(pred_label.dtype)
    if valid is None:
        n = float(pred.shape[0])
    else:
        correct = correct & valid.view(1, -1)
        n = torch.clamp(valid.sum().float(), min=1.0)
    res = [correct[:k].reshape(-1).float().sum(0, keepdim=True) * (100.0 / n) for k in topk]
    return res[0] if return_single else res


@HEADS.register_module()
class BBoxHeadRbbox(nn.Module):
    """two linear layers, classification and 5-parameter regression (L169-238)"""

    def __init__(self, with_avg_pool=False, with_cls=True, with_reg=True, roi_feat_size=7, in_channels=256,
                 num_classes=19, target_means=[0., 0., 0., 0., 0.], target_stds=[0.1, 0.1, 0.2, 0.2, 0.1],
                 reg_class_agnostic=False, with_module=True, hbb_trans="hbb2obb_v2",
                 loss_cls=dict(type="CrossEntropyLoss", use_sigmoid=False, loss_weight=1.0),
                 loss_bbox=dict(type="SmoothL1Loss", beta=1.0, loss_weight=1.0)):
        super().__init__()
        assert with_cls or with_reg
        self.with_avg_pool = with_avg_pool
        self.with_cls = with_cls
        self.with_reg = with_reg
        self.roi_feat_size = roi_feat_size
        self.in_channels = in_channels
        self.num_classes = num_classes
        self.target_means = target_means
        self.target_stds = target_stds
        self.reg_class_agnostic = reg_class_agnostic
        self.loss_cls = build_from_cfg(loss_cls, LOSSES)
        self.loss_bbox = build_from_cfg(loss_bbox, LOSSES)
        if self.with_avg_pool:
            self.avg_pool = nn.AvgPool2d(roi_feat_size)
        if self.with_cls:
            self.fc_cls = self._roi_linear(self.in_channels, num_classes)
        if self.with_reg:
            self.fc_reg = self._roi_linear(self.in_channels, 5 if reg_class_agnostic else 5 * num_classes)
        self.debug_imgs = None
        self.with_module = with_module
        self.hbb_trans = hbb_trans

    @property
    def roi_feat_area(self):
        size = self.roi_feat_size
        return size * size if isinstance(size, int) else size[0] * size[1]

    def _roi_linear(self, channels, out_features):
        """a layer on the pooled (R, C, PH, PW) features: after average pooling a plain Linear on C values, otherwise
        an FC over all C*PH*PW values that reads the channels-last memory in place (RoIFeatureLinear; state dicts
        keep the reference's (c, ph, pw) column order)"""
        if self.with_avg_pool:
            return Linear(channels, out_features)
        return RoIFeatureLinear(channels, self.roi_feat_area, out_features)

    def init_weights(self):
        if self.with_cls:
            nn.init.normal_(self.fc_cls.weight, 0, 0.01)
            nn.init.constant_(self.fc_cls.bias, 0)
        if self.with_reg:
            nn.init.normal_(self.fc_reg.weight, 0, 0.001)
            nn.init.constant_(self.fc_reg.bias, 0)

    def forward(self, x):
        if self.with_avg_pool:
            x = self.avg_pool(x).reshape(x.shape[0], -1)
        return (self.fc_cls(x) if self.with_cls else None), (self.fc_reg(x) if self.with_reg else None)

    execute = forward

    def get_target(self, stage_rows, gt_obbs, gt_labels, rcnn_train_cfg):
        """stage_rows: one StageRows per image (horizontal boxes) -> targets of all images, concatenated"""
        return _cat_images([hbb_row_targets(r, g, rcnn_train_cfg, self.target_means, self.target_stds,
                                            self.with_module) for r, g in zip(stage_rows, gt_obbs)])

    def get_target_rbbox(self, stage_rows, gt_rbboxes, gt_labels, rcnn_train_cfg):
        """stage_rows: one StageRows per image (rotated boxes); gt_rbboxes: what they were assigned against"""
        return _cat_images([obb_row_targets(r, g, rcnn_train_cfg, self.target_means, self.target_stds)
                            for r, g in zip(stage_rows, gt_rbboxes)])

    def _finish_dets(self, dbboxes, scores, scale_factor, rescale, cfg):
        if rescale:
            dbboxes = dbboxes.clone()
            for k in range(4):
                dbboxes[:, k::5] /= scale_factor
        if cfg is None:
            return dbboxes, scores
        det_bboxes, det_labels = multiclass_nms_rotated(dbboxes, scores, _get(cfg, "score_thr"), _get(cfg, "nms"),
                                                        _get(cfg, "max_per_img"))
        return torch.cat([rotated_box_to_poly(det_bboxes), det_bboxes[:, -1:]], -1), det_labels

    def get_det_bboxes(self, rois, cls_score, bbox_pred, img_shape, scale_factor, rescale=False, cfg=None):
        if isinstance(cls_score, list):
            cls_score = sum(cls_score) / float(len(cls_score))
        scores = F.softmax(cls_score, dim=1) if cls_score is not None else None
        assert rois.size(1) in (5, 6)
        obbs = hbb2obb_v2(rois[:, 1:]) if rois.size(1) == 5 else rois[:, 1:]
        dbboxes = obbs if bbox_pred is None else delta2dbbox_v3(obbs, bbox_pred, self.target_means, self.target_stds,
                                                                img_shape)
        assert cfg is not None
        return self._finish_dets(dbboxes, scores, scale_factor, rescale, cfg)

    def get_det_rbboxes(self, rrois, cls_score, rbbox_pred, img_shape, scale_factor, rescale=False, cfg=None,
                        alive=None):
        """`alive` (R,) bool: rows of a padded proposal table that are real; the others score 0 in every class"""
        if isinstance(cls_score, list):
            cls_score = sum(cls_score) / float(len(cls_score))
        scores = F.softmax(cls_score, dim=1) if cls_score is not None else None
        if alive is not None and scores is not None:
            scores = scores * alive[:, None].to(scores.dtype)
        dbboxes = rrois[:, 1:] if rbbox_pred is None else delta2dbbox_v2(rrois[:, 1:], rbbox_pred, self.target_means,
                                                                         self.target_stds, img_shape)
        return self._finish_dets(dbboxes, scores, scale_factor, rescale, cfg)

    def loss(self, cls_score, bbox_pred, labels, label_weights, bbox_targets, bbox_weights, reduce=True):
        """Rows with label_weight 0 are padding (fewer candidates than the sampler's `num`): they count nowhere.
        Classification = mean over the sampled rows; regression = positives only (bbox_weights), normalised by the
        number of sampled rows (L398-423, where every row is a sampled one and the positives are picked with a
        boolean index)."""
        losses = dict()
        sampled = label_weights > 0
        n_rows = torch.clamp(sampled.sum().float(), min=1.0)
        if cls_score is not None:
            losses["rbbox_loss_cls"] = self.loss_cls(cls_score, labels, label_weights, reduce=reduce)
            losses["rbbox_acc"] = accuracy(cls_score, labels, valid=sampled)
        if bbox_pred is not None:
            if self.reg_class_agnostic:
                pred = bbox_pred.view(bbox_pred.size(0), 5)
            else:
                pred = bbox_pred.view(bbox_pred.size(0), -1, 5)
                pred = pred.gather(1, labels.long()[:, None, None].expand(-1, 1, 5))[:, 0]
            losses["rbbox_loss_bbox"] = self.loss_bbox(pred, bbox_targets, bbox_weights, avg_factor=n_rows)
        return losses

    def refine_rbboxes(self, rois, labels, bbox_preds, stage_rows, img_metas):
        """every sampled row regressed by its label's deltas (L425-448).  rois (B*num, 6) [img, obb]; returns per
        image (boxes (num,5), alive (num,)): the gts that were added as proposals and the padding rows are dead
        (the reference drops the former with a boolean index)."""
        num = rois.shape[0] // len(img_metas)
        out = []
        for i, (rows, meta) in enumerate(zip(stage_rows, img_metas)):
            sl = slice(i * num, (i + 1) * num)
            boxes = self.regress_by_class_rbbox(rois[sl, 1:], labels[sl], bbox_preds[sl], meta)
            out.append((boxes, rows.valid & ~rows.is_gt))
        return out

    def regress_by_class_rbbox(self, rois, label, bbox_pred, img_meta):
        assert rois.size(1) == 5 or rois.size(1) == 6
        if not self.reg_class_agnostic:
            label = label.long() * 5
            inds = torch.stack((label, label + 1, label + 2, label + 3, label + 4), 1)
            bbox_pred = torch.gather(bbox_pred, 1, inds)
        assert bbox_pred.size(1) == 5
        if rois.size(1) == 5:
            new_rois = delta2dbbox_v3(rois, bbox_pred, self.target_means, self.target_stds, img_meta["img_shape"])
            return choose_best_Rroi_batch(new_rois)
        bboxes = delta2dbbox_v3(rois[:, 1:], bbox_pred, self.target_means, self.target_stds, img_meta["img_shape"])
        return torch.cat((rois[:, [0]], choose_best_Rroi_batch(bboxes)), dim=1)
