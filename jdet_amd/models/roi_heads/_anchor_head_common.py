"""Pieces shared by the rotated single-stage anchor heads (S2ANetHead, RotatedRetinaHead): cached grid
anchors, valid flags from `pad_shape` (stored (w, h)), the per-level loss, the per-image
top-k -> decode -> multiclass rotated NMS post-processing and the target-dict parsing.  These are the
parts that are literally duplicated between s2anet_head.py and rotated_retina_head.py in the reference."""
import numpy as np
import torch

from jdet_amd.models.boxes.box_ops import delta2bbox_rotated, rotated_box_to_poly
from jdet_amd.ops.nms_rotated import multiclass_nms_rotated
from jdet_amd.utils.registry import BOXES, build_from_cfg


class RotatedAnchorHeadMixin:
    def _init_anchors(self, level, featmap_size, device):
        key = (level, tuple(featmap_size), str(device))
        if key not in self.base_anchors:
            self.base_anchors[key] = self.anchor_generators[level].grid_anchors(
                featmap_size, self.anchor_strides[level], device=device)
        return self.base_anchors[key]

    def _valid_flags(self, featmap_sizes, img_metas, device):
        valid_flag_list = []
        cache = self.__dict__.setdefault("_valid_flag_cache", {})
        for img_meta in img_metas:
            # the flags depend on (level sizes, pad_shape) only: DOTA tiles have one shape, so they are built once
            key = (tuple(featmap_sizes), tuple(img_meta["pad_shape"][:2]), str(device))
            if key not in cache:
                multi_level_flags = []
                all_valid = True
                for i in range(len(featmap_sizes)):
                    anchor_stride = self.anchor_strides[i]
                    feat_h, feat_w = featmap_sizes[i]
                    w, h = img_meta["pad_shape"][:2]
                    valid_feat_h = min(int(np.ceil(h / anchor_stride)), feat_h)
                    valid_feat_w = min(int(np.ceil(w / anchor_stride)), feat_w)
                    all_valid = all_valid and valid_feat_h == feat_h and valid_feat_w == feat_w
                    multi_level_flags.append(self.anchor_generators[i].valid_flags(
                        (feat_h, feat_w), (valid_feat_h, valid_feat_w), device=device))
                cache[key] = (multi_level_flags, all_valid)
            multi_level_flags, all_valid = cache[key]
            # host-side fact (no device sync): every anchor is valid -> anchor_target skips the mask gather
            img_meta["_all_valid"] = all_valid
            valid_flag_list.append(list(multi_level_flags))
        return valid_flag_list

    def get_init_anchors(self, featmap_sizes, img_metas, device):
        multi_level_anchors = [self._init_anchors(i, featmap_sizes[i], device) for i in range(len(featmap_sizes))]
        anchor_list = [list(multi_level_anchors) for _ in range(len(img_metas))]
        return anchor_list, self._valid_flags(featmap_sizes, img_metas, device)

    def _loss_single(self, loss_cls_fn, loss_bbox_fn, cls_score, bbox_pred, anchors, labels, label_weights,
                     bbox_targets, bbox_weights, num_total_samples, cfg):
        # The level's targets are column windows [:, s:e] of the per-image arrays.  FocalLoss / SmoothL1Loss / L1Loss read
        # such windows in place (one node per level and loss, models/losses/focal_loss.py: _FocalLevel); flattening them
        # here -- what the reference does, s2anet_head.py:L441-450 -- costs a copy per window (40 per S2ANet step).
        from jdet_amd.models.losses.focal_loss import FocalLoss
        from jdet_amd.models.losses.smooth_l1_loss import L1Loss, SmoothL1Loss
        if not isinstance(loss_cls_fn, FocalLoss):
            labels = labels.reshape(-1)
            label_weights = label_weights.reshape(-1)
        cls_score = cls_score.permute(0, 2, 3, 1).reshape(-1, self.cls_out_channels)
        loss_cls = loss_cls_fn(cls_score, labels, label_weights, avg_factor=num_total_samples)
        if not isinstance(loss_bbox_fn, (SmoothL1Loss, L1Loss)) or cfg.get("reg_decoded_bbox", False):
            bbox_targets = bbox_targets.reshape(-1, 5)
            bbox_weights = bbox_weights.reshape(-1, 5)
        bbox_pred = bbox_pred.permute(0, 2, 3, 1).reshape(-1, 5)
        if cfg.get("reg_decoded_bbox", False):
            bbox_coder_cfg = cfg.get("bbox_coder", "")
            if bbox_coder_cfg == "":
                bbox_coder_cfg = dict(type="DeltaXYWHBBoxCoder")
            bbox_coder = build_from_cfg(bbox_coder_cfg, BOXES)
            bbox_pred = bbox_coder.decode(anchors.reshape(-1, 5), bbox_pred)
        loss_bbox = loss_bbox_fn(bbox_pred, bbox_targets, bbox_weights, avg_factor=num_total_samples)
        return loss_cls, loss_bbox

    def get_bboxes_single(self, cls_score_list, bbox_pred_list, mlvl_anchors, img_shape, scale_factor, cfg,
                          rescale=False):
        assert len(cls_score_list) == len(bbox_pred_list) == len(mlvl_anchors)
        mlvl_bboxes, mlvl_scores = [], []
        for cls_score, bbox_pred, anchors in zip(cls_score_list, bbox_pred_list, mlvl_anchors):
            assert cls_score.shape[-2:] == bbox_pred.shape[-2:]
            cls_score = cls_score.permute(1, 2, 0).reshape(-1, self.cls_out_channels)
            scores = cls_score.sigmoid() if self.use_sigmoid_cls else cls_score.softmax(-1)
            bbox_pred = bbox_pred.permute(1, 2, 0).reshape(-1, 5)
            nms_pre = cfg.get("nms_pre", -1)
            if nms_pre > 0 and scores.shape[0] > nms_pre:
                max_scores = scores.max(dim=1).values if self.use_sigmoid_cls else scores[:, 1:].max(dim=1).values
                _, topk_inds = max_scores.topk(nms_pre)
                anchors = anchors[topk_inds, :]
                bbox_pred = bbox_pred[topk_inds, :]
                scores = scores[topk_inds, :]
            mlvl_bboxes.append(delta2bbox_rotated(anchors, bbox_pred, self.target_means, self.target_stds, img_shape))
            mlvl_scores.append(scores)
        mlvl_bboxes = torch.cat(mlvl_bboxes)
        if rescale:
            mlvl_bboxes[..., :4] /= scale_factor
        mlvl_scores = torch.cat(mlvl_scores)
        if self.use_sigmoid_cls:
            padding = mlvl_scores.new_zeros((mlvl_scores.shape[0], 1))
            mlvl_scores = torch.cat([padding, mlvl_scores], dim=1)
        det_bboxes, det_labels = multiclass_nms_rotated(mlvl_bboxes, mlvl_scores, cfg.score_thr, cfg.nms,
                                                        cfg.max_per_img)
        boxes, scores = det_bboxes[:, :5], det_bboxes[:, 5]
        return rotated_box_to_poly(boxes), scores, det_labels

    def parse_targets(self, targets, is_train=True):
        img_metas, gt_bboxes, gt_bboxes_ignore, gt_labels = [], [], [], []
        for target in targets:
            if is_train:
                gt_bboxes.append(target["rboxes"])
                gt_labels.append(target["labels"])
                gt_bboxes_ignore.append(target["rboxes_ignore"])
            img_metas.append(dict(img_shape=target["img_size"][::-1], scale_factor=target["scale_factor"],
                                  pad_shape=target["pad_shape"]))
        if not is_train:
            return img_metas
        return gt_bboxes, gt_labels, img_metas, gt_bboxes_ignore
