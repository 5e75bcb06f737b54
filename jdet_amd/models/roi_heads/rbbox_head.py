"""RoI-Transformer box heads (stage 1: horizontal RoI -> rotated RoI; stage 2: rotated RoI -> detection).
Mirrors python/jdet/models/roi_heads/rbbox_head.py: target builders L9-146, `accuracy` L148-167,
`BBoxHeadRbbox` L169-448."""
import torch
import torch.nn.functional as F
from torch import nn

from jdet_amd.models.boxes.box_ops import rotated_box_to_poly
from jdet_amd.ops.bbox_transforms import (best_match_dbbox2delta, choose_best_obb_batch, choose_best_Rroi_batch,
                                          dbbox2delta_v3, delta2dbbox_v2, delta2dbbox_v3, hbb2obb_v2)
from jdet_amd.ops.nms_rotated import multiclass_nms_rotated
from jdet_amd.utils.general import multi_apply
from jdet_amd.utils.registry import HEADS, LOSSES, build_from_cfg


def _get(cfg, key):
    return cfg[key] if isinstance(cfg, dict) else getattr(cfg, key)


def bbox_target_rbbox_single(pos_bboxes, neg_bboxes, pos_assigned_gt_inds, gt_obbs, pos_gt_labels, cfg, reg_classes=1,
                             target_means=(.0, .0, .0, .0, .0), target_stds=(1.0, 1.0, 1.0, 1.0, 1.0),
                             with_module=True, hbb_trans="hbb2obb_v2"):
    num_pos, num_neg = pos_bboxes.size(0), neg_bboxes.size(0)
    num_samples = num_pos + num_neg
    dev = pos_bboxes.device
    labels = torch.zeros(num_samples, dtype=torch.int32, device=dev)
    label_weights = torch.zeros(num_samples, device=dev)
    bbox_targets = torch.zeros((num_samples, 5), device=dev)
    bbox_weights = torch.zeros((num_samples, 5), device=dev)
    pos_gt_obbs = choose_best_obb_batch(gt_obbs[pos_assigned_gt_inds])
    pos_ext_bboxes = hbb2obb_v2(pos_bboxes) if pos_bboxes.size(1) == 4 else pos_bboxes
    if num_pos > 0:
        labels[:num_pos] = pos_gt_labels.to(labels.dtype)
        pos_weight = 1.0 if _get(cfg, "pos_weight") <= 0 else _get(cfg, "pos_weight")
        label_weights[:num_pos] = pos_weight
        if with_module:
            # rbbox_head.py:L76 calls `dbbox2delta`, a name the reference never imports: unreachable there
            raise NameError("dbbox2delta (with_module=True) is not defined in the reference either")
        bbox_targets[:num_pos, :] = dbbox2delta_v3(pos_ext_bboxes, pos_gt_obbs, target_means, target_stds)
        bbox_weights[:num_pos, :] = 1
    if num_neg > 0:
        label_weights[-num_neg:] = 1.0
    return labels, label_weights, bbox_targets, bbox_weights


def rbbox_target_rbbox_single(pos_rbboxes, neg_rbboxes, pos_gt_rbboxes, pos_gt_labels, cfg, reg_classes=1,
                              target_means=(.0, .0, .0, .0, .0), target_stds=(1.0, 1.0, 1.0, 1.0, 1.0)):
    assert pos_rbboxes.size(1) == 5
    num_pos, num_neg = pos_rbboxes.size(0), neg_rbboxes.size(0)
    num_samples = num_pos + num_neg
    dev = pos_rbboxes.device
    labels = torch.zeros(num_samples, dtype=torch.int32, device=dev)
    label_weights = torch.zeros(num_samples, device=dev)
    bbox_targets = torch.zeros((num_samples, 5), device=dev)
    bbox_weights = torch.zeros((num_samples, 5), device=dev)
    if num_pos > 0:
        labels[:num_pos] = pos_gt_labels.to(labels.dtype)
        pos_weight = 1.0 if _get(cfg, "pos_weight") <= 0 else _get(cfg, "pos_weight")
        label_weights[:num_pos] = pos_weight
        bbox_targets[:num_pos, :] = best_match_dbbox2delta(pos_rbboxes, pos_gt_rbboxes, target_means, target_stds)
        bbox_weights[:num_pos, :] = 1
    if num_neg > 0:
        label_weights[-num_neg:] = 1.0
    return labels, label_weights, bbox_targets, bbox_weights


def _concat_targets(parts, concat):
    return tuple(torch.cat(p, 0) for p in parts) if concat else parts


def bbox_target_rbbox(pos_bboxes_list, neg_bboxes_list, pos_assigned_gt_inds_list, gt_obbs_list, pos_gt_labels_list,
                      cfg, reg_classes=1, target_means=(.0, .0, .0, .0, .0), target_stds=(1.0, 1.0, 1.0, 1.0, 1.0),
                      concat=True, with_module=True, hbb_trans="hbb2obb_v2"):
    parts = multi_apply(bbox_target_rbbox_single, pos_bboxes_list, neg_bboxes_list, pos_assigned_gt_inds_list,
                        gt_obbs_list, pos_gt_labels_list, cfg=cfg, reg_classes=reg_classes,
                        target_means=target_means, target_stds=target_stds, with_module=with_module,
                        hbb_trans=hbb_trans)
    return _concat_targets(parts, concat)


def rbbox_target_rbbox(pos_rbboxes_list, neg_rbboxes_list, pos_gt_rbboxes_list, pos_gt_labels_list, cfg, reg_classes=1,
                       target_means=(.0, .0, .0, .0, 0), target_stds=(1.0, 1.0, 1.0, 1.0, 1.0), concat=True):
    parts = multi_apply(rbbox_target_rbbox_single, pos_rbboxes_list, neg_rbboxes_list, pos_gt_rbboxes_list,
                        pos_gt_labels_list, cfg=cfg, reg_classes=reg_classes, target_means=target_means,
                        target_stds=target_stds)
    return _concat_targets(parts, concat)


def accuracy(pred, target, topk=1):
    return_single = isinstance(topk, int)
    topk = (topk,) if return_single else topk
    _, pred_label = pred.topk(max(topk), 1, True, True)
    correct = pred_label.t() == target.view(1, -1).to(pred_label.dtype)
    res = [correct[:k].reshape(-1).float().sum(0, keepdim=True) * (100.0 / pred.shape[0]) for k in topk]
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
        in_channels = self.in_channels
        if self.with_avg_pool:
            self.avg_pool = nn.AvgPool2d(roi_feat_size)
        else:
            if isinstance(self.roi_feat_size, int):
                in_channels *= self.roi_feat_size * self.roi_feat_size
            else:
                assert len(self.roi_feat_size) == 2
                in_channels *= self.roi_feat_size[0] * self.roi_feat_size[1]
        if self.with_cls:
            self.fc_cls = nn.Linear(in_channels, num_classes)
        if self.with_reg:
            self.fc_reg = nn.Linear(in_channels, 5 if reg_class_agnostic else 5 * num_classes)
        self.debug_imgs = None
        self.with_module = with_module
        self.hbb_trans = hbb_trans

    def init_weights(self):
        if self.with_cls:
            nn.init.normal_(self.fc_cls.weight, 0, 0.01)
            nn.init.constant_(self.fc_cls.bias, 0)
        if self.with_reg:
            nn.init.normal_(self.fc_reg.weight, 0, 0.001)
            nn.init.constant_(self.fc_reg.bias, 0)

    def forward(self, x):
        if self.with_avg_pool:
            x = self.avg_pool(x)
        x = x.reshape(x.shape[0], -1)
        return (self.fc_cls(x) if self.with_cls else None), (self.fc_reg(x) if self.with_reg else None)

    execute = forward

    def get_target(self, sampling_results, gt_obbs, gt_labels, rcnn_train_cfg):
        reg_classes = 1 if self.reg_class_agnostic else self.num_classes
        return bbox_target_rbbox([r.pos_bboxes for r in sampling_results], [r.neg_bboxes for r in sampling_results],
                                 [r.pos_assigned_gt_inds for r in sampling_results], gt_obbs,
                                 [r.pos_gt_labels for r in sampling_results], rcnn_train_cfg, reg_classes,
                                 target_means=self.target_means, target_stds=self.target_stds,
                                 with_module=self.with_module, hbb_trans=self.hbb_trans)

    def get_target_rbbox(self, sampling_results, gt_bboxes, gt_labels, rcnn_train_cfg):
        reg_classes = 1 if self.reg_class_agnostic else self.num_classes
        return rbbox_target_rbbox([r.pos_bboxes for r in sampling_results], [r.neg_bboxes for r in sampling_results],
                                  [r.pos_gt_bboxes for r in sampling_results],
                                  [r.pos_gt_labels for r in sampling_results], rcnn_train_cfg, reg_classes,
                                  target_means=self.target_means, target_stds=self.target_stds)

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

    def get_det_rbboxes(self, rrois, cls_score, rbbox_pred, img_shape, scale_factor, rescale=False, cfg=None):
        if isinstance(cls_score, list):
            cls_score = sum(cls_score) / float(len(cls_score))
        scores = F.softmax(cls_score, dim=1) if cls_score is not None else None
        dbboxes = rrois[:, 1:] if rbbox_pred is None else delta2dbbox_v2(rrois[:, 1:], rbbox_pred, self.target_means,
                                                                         self.target_stds, img_shape)
        return self._finish_dets(dbboxes, scores, scale_factor, rescale, cfg)

    def loss(self, cls_score, bbox_pred, labels, label_weights, bbox_targets, bbox_weights, reduce=True):
        losses = dict()
        if cls_score is not None:
            losses["rbbox_loss_cls"] = self.loss_cls(cls_score, labels, label_weights, reduce=reduce)
            losses["rbbox_acc"] = accuracy(cls_score, labels)
        if bbox_pred is not None:
            pos_inds = labels > 0
            if self.reg_class_agnostic:
                pos_bbox_pred = bbox_pred.view(bbox_pred.size(0), 5)[pos_inds]
            else:
                pos_bbox_pred = bbox_pred.view(bbox_pred.size(0), -1, 5)[pos_inds, labels[pos_inds].long()]
            losses["rbbox_loss_bbox"] = self.loss_bbox(pos_bbox_pred, bbox_targets[pos_inds], bbox_weights[pos_inds],
                                                       avg_factor=bbox_targets.size(0))
        return losses

    def refine_rbboxes(self, rois, labels, bbox_preds, pos_is_gts, img_metas):
        """regress every sampled RoI by its label's deltas, drop the gts that were added as proposals"""
        img_ids = rois[:, 0].long().unique()
        assert img_ids.numel() == len(img_metas)
        bboxes_list = []
        for i in range(len(img_metas)):
            inds = torch.nonzero(rois[:, 0] == i)[:, 0]
            num_rois = inds.numel()
            bboxes = self.regress_by_class_rbbox(rois[inds, 1:], labels[inds], bbox_preds[inds], img_metas[i])
            keep = torch.ones(num_rois, dtype=torch.bool, device=rois.device)
            keep[:len(pos_is_gts[i])] = ~pos_is_gts[i].bool()
            bboxes_list.append(bboxes[keep])
        return bboxes_list

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
