"""Anchor head and the Faster R-CNN RPN head used by RoI-Transformer.  Mirrors
python/jdet/models/roi_heads/fasterrcnn_head.py: `AnchorHead` L15-244, `FasterrcnnHead` L247-329.
Proposals: per level sigmoid scores -> top nms_pre -> delta2bbox -> horizontal NMS (`jt.nms`, parity
unpinned, see ops/nms.py) -> nms_post; levels concatenated -> top max_num by score."""
import numpy as np
import torch
import torch.nn.functional as F
from torch import nn

from jdet_amd.ops.bbox_transforms import delta2bbox
from jdet_amd.ops.nms import nms_dets
from jdet_amd.utils.general import multi_apply
from jdet_amd.utils.registry import HEADS, LOSSES, build_from_cfg

from .anchor_generator import AnchorGenerator
from .anchor_target import anchor_target


@HEADS.register_module()
class AnchorHead(nn.Module):
    def __init__(self, num_classes, in_channels, feat_channels=256, anchor_scales=[8, 16, 32],
                 anchor_ratios=[0.5, 1.0, 2.0], anchor_strides=[4, 8, 16, 32, 64], anchor_base_sizes=None,
                 target_means=(.0, .0, .0, .0), target_stds=(1.0, 1.0, 1.0, 1.0),
                 loss_cls=dict(type="CrossEntropyLoss", loss_weight=1.0, use_sigmoid=True),
                 loss_bbox=dict(type="SmoothL1Loss", beta=1.0 / 9.0, loss_weight=1.0)):
        super().__init__()
        self.in_channels = in_channels
        self.num_classes = num_classes
        self.feat_channels = feat_channels
        self.anchor_scales = anchor_scales
        self.anchor_ratios = anchor_ratios
        self.anchor_strides = anchor_strides
        self.anchor_base_sizes = list(anchor_strides) if anchor_base_sizes is None else anchor_base_sizes
        self.target_means = target_means
        self.target_stds = target_stds
        self.use_sigmoid_cls = loss_cls.get("use_sigmoid", False)
        self.sampling = loss_cls["type"] not in ["FocalLoss", "GHMC"]
        self.cls_out_channels = num_classes - 1 if self.use_sigmoid_cls else num_classes
        self.loss_cls = build_from_cfg(loss_cls, LOSSES)
        self.loss_bbox = build_from_cfg(loss_bbox, LOSSES)
        self.anchor_generators = [AnchorGenerator(b, anchor_scales, anchor_ratios) for b in self.anchor_base_sizes]
        self.num_anchors = len(self.anchor_ratios) * len(self.anchor_scales)
        self._init_layers()

    def _init_layers(self):
        self.conv_cls = nn.Conv2d(self.feat_channels, self.num_anchors * self.cls_out_channels, 1)
        self.conv_reg = nn.Conv2d(self.feat_channels, self.num_anchors * 4, 1)

    def _heads(self):
        return [self.conv_cls, self.conv_reg]

    def init_weights(self):
        for m in self._heads():
            nn.init.normal_(m.weight, 0, 0.01)
            nn.init.constant_(m.bias, 0.0)

    def forward_single(self, x):
        return self.conv_cls(x), self.conv_reg(x)

    def forward(self, feats):
        return multi_apply(self.forward_single, feats)

    execute = forward

    def get_anchors(self, featmap_sizes, img_metas, device=None):
        num_imgs, num_levels = len(img_metas), len(featmap_sizes)
        multi_level_anchors = [self.anchor_generators[i].grid_anchors(featmap_sizes[i], self.anchor_strides[i], device)
                               for i in range(num_levels)]
        anchor_list = [list(multi_level_anchors) for _ in range(num_imgs)]
        valid_flag_list = []
        for img_meta in img_metas:
            multi_level_flags = []
            for i in range(num_levels):
                stride = self.anchor_strides[i]
                feat_h, feat_w = featmap_sizes[i]
                h, w = img_meta["pad_shape"][0], img_meta["pad_shape"][1]
                valid_feat_h = min(int(np.ceil(h / stride)), feat_h)
                valid_feat_w = min(int(np.ceil(w / stride)), feat_w)
                multi_level_flags.append(self.anchor_generators[i].valid_flags((feat_h, feat_w),
                                                                               (valid_feat_h, valid_feat_w), device))
            valid_flag_list.append(multi_level_flags)
        return anchor_list, valid_flag_list

    def loss_single(self, cls_score, bbox_pred, labels, label_weights, bbox_targets, bbox_weights, num_total_samples,
                    cfg):
        labels = labels.reshape(-1)
        label_weights = label_weights.reshape(-1)
        cls_score = cls_score.permute(0, 2, 3, 1).reshape(-1, self.cls_out_channels)
        loss_cls = self.loss_cls(cls_score, labels, label_weights, avg_factor=num_total_samples)
        bbox_targets = bbox_targets.reshape(-1, 4)
        bbox_weights = bbox_weights.reshape(-1, 4)
        bbox_pred = bbox_pred.permute(0, 2, 3, 1).reshape(-1, 4)
        loss_bbox = self.loss_bbox(bbox_pred, bbox_targets, bbox_weights, avg_factor=num_total_samples)
        return loss_cls, loss_bbox

    def loss(self, cls_scores, bbox_preds, gt_bboxes, gt_labels, img_metas, cfg, gt_bboxes_ignore=None):
        featmap_sizes = [tuple(featmap.shape[-2:]) for featmap in cls_scores]
        assert len(featmap_sizes) == len(self.anchor_generators)
        anchor_list, valid_flag_list = self.get_anchors(featmap_sizes, img_metas, cls_scores[0].device)
        label_channels = self.cls_out_channels if self.use_sigmoid_cls else 1
        cls_reg_targets = anchor_target(anchor_list, valid_flag_list, gt_bboxes, img_metas, self.target_means,
                                        self.target_stds, cfg, gt_bboxes_ignore_list=gt_bboxes_ignore,
                                        gt_labels_list=gt_labels, label_channels=label_channels,
                                        sampling=self.sampling)
        if cls_reg_targets is None:
            return None
        (labels_list, label_weights_list, bbox_targets_list, bbox_weights_list, num_total_pos,
         num_total_neg) = cls_reg_targets
        num_total_samples = num_total_pos + num_total_neg if self.sampling else num_total_pos
        losses_cls, losses_bbox = multi_apply(self.loss_single, cls_scores, bbox_preds, labels_list,
                                              label_weights_list, bbox_targets_list, bbox_weights_list,
                                              num_total_samples=num_total_samples, cfg=cfg)
        return dict(loss_cls=losses_cls, loss_bbox=losses_bbox)

    def get_bboxes(self, cls_scores, bbox_preds, img_metas, cfg, rescale=False):
        assert len(cls_scores) == len(bbox_preds)
        num_levels = len(cls_scores)
        dev = cls_scores[0].device
        mlvl_anchors = [self.anchor_generators[i].grid_anchors(tuple(cls_scores[i].shape[-2:]),
                                                               self.anchor_strides[i], dev) for i in range(num_levels)]
        result_list = []
        for img_id in range(len(img_metas)):
            cls_score_list = [cls_scores[i][img_id].detach() for i in range(num_levels)]
            bbox_pred_list = [bbox_preds[i][img_id].detach() for i in range(num_levels)]
            result_list.append(self.get_bboxes_single(cls_score_list, bbox_pred_list, mlvl_anchors,
                                                      img_metas[img_id]["img_shape"],
                                                      img_metas[img_id]["scale_factor"], cfg, rescale))
        return result_list

    def get_bboxes_single(self, cls_scores, bbox_preds, mlvl_anchors, img_shape, scale_factor, cfg, rescale=False):
        raise NotImplementedError


@HEADS.register_module()
class FasterrcnnHead(AnchorHead):
    def __init__(self, in_channels, **kwargs):
        super().__init__(2, in_channels, **kwargs)

    def _init_layers(self):
        self.rpn_conv = nn.Conv2d(self.in_channels, self.feat_channels, 3, padding=1)
        self.rpn_cls = nn.Conv2d(self.feat_channels, self.num_anchors * self.cls_out_channels, 1)
        self.rpn_reg = nn.Conv2d(self.feat_channels, self.num_anchors * 4, 1)

    def _heads(self):
        return [self.rpn_conv, self.rpn_cls, self.rpn_reg]

    def forward_single(self, x):
        x = F.relu(self.rpn_conv(x))
        return self.rpn_cls(x), self.rpn_reg(x)

    def loss(self, cls_scores, bbox_preds, gt_bboxes, img_metas, cfg, gt_bboxes_ignore=None):
        losses = super().loss(cls_scores, bbox_preds, gt_bboxes, None, img_metas, cfg,
                              gt_bboxes_ignore=gt_bboxes_ignore)
        return dict(loss_rpn_cls=losses["loss_cls"], loss_rpn_bbox=losses["loss_bbox"])

    def get_bboxes_single(self, cls_scores, bbox_preds, mlvl_anchors, img_shape, scale_factor, cfg, rescale=False):
        mlvl_proposals = []
        for idx in range(len(cls_scores)):
            rpn_cls_score, rpn_bbox_pred = cls_scores[idx], bbox_preds[idx]
            assert rpn_cls_score.shape[-2:] == rpn_bbox_pred.shape[-2:]
            anchors = mlvl_anchors[idx]
            rpn_cls_score = rpn_cls_score.permute(1, 2, 0)
            if self.use_sigmoid_cls:
                scores = rpn_cls_score.reshape(-1).sigmoid()
            else:
                scores = rpn_cls_score.reshape(-1, 2).softmax(dim=1)[:, 1]
            rpn_bbox_pred = rpn_bbox_pred.permute(1, 2, 0).reshape(-1, 4)
            if cfg["nms_pre"] > 0 and scores.shape[0] > cfg["nms_pre"]:
                _, topk_inds = scores.topk(cfg["nms_pre"])
                rpn_bbox_pred = rpn_bbox_pred[topk_inds, :]
                anchors = anchors[topk_inds, :]
                scores = scores[topk_inds]
            proposals = delta2bbox(anchors, rpn_bbox_pred, self.target_means, self.target_stds, img_shape)
            if cfg["min_bbox_size"] > 0:
                w = proposals[:, 2] - proposals[:, 0] + 1
                h = proposals[:, 3] - proposals[:, 1] + 1
                valid = (w >= cfg["min_bbox_size"]) & (h >= cfg["min_bbox_size"])
                proposals, scores = proposals[valid, :], scores[valid]
            proposals = torch.cat([proposals, scores.unsqueeze(-1)], dim=-1)
            proposals = proposals[nms_dets(proposals, cfg["nms_thr"])]
            mlvl_proposals.append(proposals[:cfg["nms_post"], :])
        proposals = torch.cat(mlvl_proposals, 0)
        if cfg["nms_across_levels"]:
            proposals = proposals[nms_dets(proposals, cfg["nms_thr"])]
            proposals = proposals[:cfg["max_num"], :]
        else:
            num = min(cfg["max_num"], proposals.shape[0])
            _, topk_inds = proposals[:, 4].topk(num)
            proposals = proposals[topk_inds, :]
        return proposals
