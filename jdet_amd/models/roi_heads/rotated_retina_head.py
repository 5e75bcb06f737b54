"""RetinaNet-OBB head.  Mirrors python/jdet/models/roi_heads/rotated_retina_head.py:L13-398: 4+4 stacked 3x3
conv towers, 9 rotated anchors per location (3 ratios x 3 octave scales, angle 0), focal + L1/SmoothL1
loss on DeltaXYWHABBoxCoder targets, top-k -> decode -> multiclass rotated NMS at test time."""
import torch
from torch import nn

from jdet_amd.ops.conv_igemm import conv_module
from jdet_amd.models.boxes.anchor_generator import AnchorGeneratorRotatedRetinaNet
from jdet_amd.models.boxes.anchor_target import anchor_target, images_to_levels
from jdet_amd.models.utils.level_pack import run_levels
from jdet_amd.models.utils.modules import ConvModule
from jdet_amd.models.utils.weight_init import bias_init_with_prob, normal_init
from jdet_amd.utils.general import multi_apply
from jdet_amd.utils.registry import HEADS, LOSSES, build_from_cfg

from ._anchor_head_common import RotatedAnchorHeadMixin
from .s2anet_head import _DEFAULT_ASSIGN, _cfg


@HEADS.register_module()
class RotatedRetinaHead(RotatedAnchorHeadMixin, nn.Module):
    def __init__(self, num_classes, in_channels, feat_channels=256, stacked_convs=4, octave_base_scale=4,
                 scales_per_octave=3, anchor_ratios=[1.0, 0.5, 2.0], anchor_strides=[8, 16, 32, 64, 128],
                 anchor_base_sizes=None, target_means=(.0, .0, .0, .0, .0), target_stds=(1.0, 1.0, 1.0, 1.0, 1.0),
                 loss_cls=dict(type="FocalLoss", use_sigmoid=True, gamma=2.0, alpha=0.25, loss_weight=1.0),
                 loss_bbox=dict(type="SmoothL1Loss", beta=1.0 / 9.0, loss_weight=1.0),
                 test_cfg=dict(nms_pre=2000, min_bbox_size=0, score_thr=0.05, nms=dict(type="nms_rotated", iou_thr=0.1),
                               max_per_img=2000),
                 train_cfg=_DEFAULT_ASSIGN):
        super().__init__()
        self.num_classes = num_classes
        self.in_channels = in_channels
        self.feat_channels = feat_channels
        self.stacked_convs = stacked_convs
        self.anchor_ratios = anchor_ratios
        self.anchor_strides = list(anchor_strides)
        self.anchor_base_sizes = list(anchor_strides) if anchor_base_sizes is None else anchor_base_sizes
        self.target_means = target_means
        self.target_stds = target_stds
        self.use_sigmoid_cls = loss_cls.get("use_sigmoid", False)
        self.sampling = loss_cls["type"] not in ["FocalLoss", "GHMC"]
        self.cls_out_channels = num_classes - 1 if self.use_sigmoid_cls else num_classes
        if self.cls_out_channels <= 0:
            raise ValueError("num_classes={} is too small".format(num_classes))
        self.loss_cls = build_from_cfg(loss_cls, LOSSES)
        self.loss_bbox = build_from_cfg(loss_bbox, LOSSES)
        self.train_cfg = _cfg(train_cfg)
        self.test_cfg = _cfg(test_cfg)
        self.anchor_generators = [AnchorGeneratorRotatedRetinaNet(b, None, anchor_ratios,
                                                                  octave_base_scale=octave_base_scale,
                                                                  scales_per_octave=scales_per_octave)
                                  for b in self.anchor_base_sizes]
        self.num_anchors = self.anchor_generators[0].num_base_anchors
        self.base_anchors = dict()
        self._init_layers()

    def _init_layers(self):
        self.relu = nn.ReLU()
        self.reg_convs = nn.ModuleList()
        self.cls_convs = nn.ModuleList()
        for i in range(self.stacked_convs):
            chn = self.in_channels if i == 0 else self.feat_channels
            self.reg_convs.append(ConvModule(chn, self.feat_channels, 3, stride=1, padding=1))
            self.cls_convs.append(ConvModule(chn, self.feat_channels, 3, stride=1, padding=1))
        self.retina_reg = nn.Conv2d(self.feat_channels, self.num_anchors * 5, 1)
        self.retina_cls = nn.Conv2d(self.feat_channels, self.num_anchors * self.cls_out_channels, 1)
        self.init_weights()

    def init_weights(self):
        for m in self.reg_convs:
            normal_init(m.conv, std=0.01)
        for m in self.cls_convs:
            normal_init(m.conv, std=0.01)
        bias_cls = bias_init_with_prob(0.01)
        normal_init(self.retina_reg, std=0.01)
        normal_init(self.retina_cls, std=0.01, bias=bias_cls)

    def forward_single(self, x, stride=None, mask=None):
        """mask: set when x is a LevelPack of several small levels (models/utils/level_pack.py)"""
        reg_feat = cls_feat = x
        for conv in self.reg_convs:
            reg_feat = conv(reg_feat)
            if mask is not None:
                reg_feat = reg_feat * mask
        for conv in self.cls_convs:
            cls_feat = conv(cls_feat)
            if mask is not None:
                cls_feat = cls_feat * mask
        return conv_module(self.retina_cls, cls_feat), conv_module(self.retina_reg, reg_feat)

    def loss(self, cls_scores, bbox_preds, gt_bboxes, gt_labels, img_metas, gt_bboxes_ignore=None):
        cfg = self.train_cfg.copy()
        featmap_sizes = [tuple(featmap.shape[-2:]) for featmap in cls_scores]
        assert len(featmap_sizes) == len(self.anchor_generators)
        anchor_list, valid_flag_list = self.get_init_anchors(featmap_sizes, img_metas, cls_scores[0].device)
        num_level_anchors = [anchors.size(0) for anchors in anchor_list[0]]
        concat_anchor_list = [torch.cat(anchor_list[i]) for i in range(len(anchor_list))]
        all_anchor_list = images_to_levels(concat_anchor_list, num_level_anchors)
        label_channels = self.cls_out_channels if self.use_sigmoid_cls else 1
        cls_reg_targets = anchor_target(anchor_list, valid_flag_list, gt_bboxes, img_metas, self.target_means,
                                        self.target_stds, cfg, gt_bboxes_ignore_list=gt_bboxes_ignore,
                                        gt_labels_list=gt_labels, label_channels=label_channels, sampling=self.sampling)
        if cls_reg_targets is None:
            return None
        labels_list, label_weights_list, bbox_targets_list, bbox_weights_list, num_total_pos, num_total_neg = \
            cls_reg_targets
        num_total_samples = num_total_pos + num_total_neg if self.sampling else num_total_pos
        losses_cls, losses_bbox = multi_apply(self.loss_single, cls_scores, bbox_preds, all_anchor_list, labels_list,
                                              label_weights_list, bbox_targets_list, bbox_weights_list,
                                              num_total_samples=num_total_samples, cfg=cfg)
        return dict(loss_cls=losses_cls, loss_bbox=losses_bbox)

    def loss_single(self, cls_score, bbox_pred, anchors, labels, label_weights, bbox_targets, bbox_weights,
                    num_total_samples, cfg):
        return self._loss_single(self.loss_cls, self.loss_bbox, cls_score, bbox_pred, anchors, labels, label_weights,
                                 bbox_targets, bbox_weights, num_total_samples, cfg)

    def get_bboxes(self, cls_scores, bbox_preds, img_metas, rescale=True):
        assert len(cls_scores) == len(bbox_preds)
        cfg = self.test_cfg.copy()
        featmap_sizes = [tuple(featmap.shape[-2:]) for featmap in cls_scores]
        num_levels = len(cls_scores)
        anchor_list, _ = self.get_init_anchors(featmap_sizes, img_metas, cls_scores[0].device)
        result_list = []
        for img_id in range(len(img_metas)):
            cls_score_list = [cls_scores[i][img_id].detach() for i in range(num_levels)]
            bbox_pred_list = [bbox_preds[i][img_id].detach() for i in range(num_levels)]
            result_list.append(self.get_bboxes_single(cls_score_list, bbox_pred_list, anchor_list[img_id],
                                                      img_metas[img_id]["img_shape"],
                                                      img_metas[img_id]["scale_factor"], cfg, rescale))
        return result_list

    def forward(self, feats, targets):
        per_level = run_levels(list(feats), lambda x, mask: self.forward_single(x, mask=mask))
        outs = tuple(map(list, zip(*per_level)))
        if self.training:
            return self.loss(*outs, *self.parse_targets(targets))
        return self.get_bboxes(*outs, self.parse_targets(targets, is_train=False))

    execute = forward
