"""Oriented R-CNN RoI head.

Contract of python/jdet/models/roi_heads/oriented_head.py:L13-530 (constructor arguments and defaults, parameter
names `shared_fcs.N / cls_fcs / reg_fcs / fc_cls / fc_reg`, loss keys `loss_cls` / `orcnn_bbox_loss`, inference
output `(polys (k,8), scores (k,), labels (k,))` per image): assign proposals (+ the gts themselves) to gts by
rotated IoU (v1 convention), sample 512 RoIs per image (25 % positives), pool them with ROIAlignRotated_v1 on four
FPN levels, two shared FC layers, a (C+1)-way classifier (background = last class) and a class-agnostic 5-parameter
regressor on OrientedDeltaXYWHTCoder targets.

Execution (this file's own; the reference builds per-image SamplingResult index lists, L466-501):
  * the RPN hands over a proposal TABLE per image -- always `nms_post` rows [box5, score], padding rows with score
    < 0 -- and everything downstream keeps fixed shapes: padding rows get overlap -1 (ignored by the assigner), the
    sample is `num` rows per image drawn by random keys + top-k with a validity mask (models/boxes/fixed_shape.py),
    unused rows point at a small dummy box and carry weight 0.  No nonzero / boolean indexing / `.any()` in the
    train step, so no device -> host round trip and the step can be captured in a HIP graph.
  * RoI features arrive channels-last from the RoIAlign kernels; the first FC layer consumes them in that order
    (its weight columns are permuted once at load / save time, `RoIFeatureLinear`), so the 51 MB per step transposes
    in both directions of the reference layout disappear while checkpoints keep the reference's weight order.
"""
import torch
import torch.nn.functional as F
from torch import nn

from jdet_amd.models.boxes.fixed_shape import sample_rows
from jdet_amd.models.utils.modules import ConvModule
from jdet_amd.ops.bbox_transforms import get_bbox_dim, obb2poly
from jdet_amd.utils.general import const_like
from jdet_amd.utils.registry import BOXES, HEADS, LOSSES, ROI_EXTRACTORS, build_from_cfg

from jdet_amd.ops.linear import Linear

from .roi_feature_linear import RoIFeatureLinear


def _pair(x):
    return tuple(x) if isinstance(x, (tuple, list)) else (x, x)


@HEADS.register_module()
class OrientedHead(nn.Module):
    def __init__(self, num_classes=15, in_channels=256, num_shared_convs=0, num_shared_fcs=2, num_cls_convs=0,
                 num_cls_fcs=0, num_reg_convs=0, num_reg_fcs=0, fc_out_channels=1024, conv_out_channels=256,
                 score_thresh=0.05,
                 assigner=dict(type="MaxIoUAssigner", pos_iou_thr=0.5, neg_iou_thr=0.5, min_pos_iou=0.5,
                               ignore_iof_thr=-1, match_low_quality=False, assigned_labels_filled=-1,
                               iou_calculator=dict(type="BboxOverlaps2D_rotated_v1")),
                 sampler=dict(type="RandomSamplerRotated", num=512, pos_fraction=0.25, neg_pos_ub=-1,
                              add_gt_as_proposals=True),
                 bbox_coder=dict(type="OrientedDeltaXYWHTCoder", target_means=[0., 0., 0., 0., 0.],
                                 target_stds=[0.1, 0.1, 0.2, 0.2, 0.1]),
                 bbox_roi_extractor=dict(type="OrientedSingleRoIExtractor",
                                         roi_layer=dict(type="ROIAlignRotated_v1", output_size=7, sampling_ratio=2),
                                         out_channels=256, extend_factor=(1.4, 1.2), featmap_strides=[4, 8, 16, 32]),
                 loss_cls=dict(type="CrossEntropyLoss"), loss_bbox=dict(type="SmoothL1Loss", beta=1.0, loss_weight=1.0),
                 with_bbox=True, with_shared_head=False, with_avg_pool=False, with_cls=True, with_reg=True,
                 start_bbox_type="obb", end_bbox_type="obb", reg_dim=None, reg_class_agnostic=True,
                 reg_decoded_bbox=False, pos_weight=-1):
        super().__init__()
        assert with_bbox and with_cls and with_reg and not with_shared_head and not with_avg_pool
        assert start_bbox_type == "obb" and end_bbox_type == "obb" and not reg_decoded_bbox, \
            "the Oriented R-CNN configuration (obb proposals -> obb detections, encoded regression targets)"
        assert num_shared_convs + num_shared_fcs + num_cls_convs + num_cls_fcs + num_reg_convs + num_reg_fcs > 0
        self.num_classes, self.in_channels = num_classes, in_channels
        self.reg_class_agnostic, self.pos_weight, self.score_thresh = reg_class_agnostic, pos_weight, score_thresh
        self.start_bbox_type, self.end_bbox_type = start_bbox_type, end_bbox_type
        self.reg_dim = get_bbox_dim(end_bbox_type) if reg_dim is None else reg_dim
        self.roi_feat_size = _pair(7)
        self.roi_feat_area = self.roi_feat_size[0] * self.roi_feat_size[1]
        self.fc_out_channels, self.conv_out_channels = fc_out_channels, conv_out_channels
        self.bbox_coder = build_from_cfg(bbox_coder, BOXES)
        self.loss_cls = build_from_cfg(loss_cls, LOSSES)
        self.loss_bbox = build_from_cfg(loss_bbox, LOSSES)
        self.assigner = build_from_cfg(assigner, BOXES)
        self.sampler = build_from_cfg(sampler, BOXES)          # carries num / pos_fraction / neg_pos_ub / add_gt
        self.bbox_roi_extractor = build_from_cfg(bbox_roi_extractor, ROI_EXTRACTORS)
        # trunk: [shared convs] -> flatten -> [shared fcs]; then per branch [convs] -> [fcs] -> output layer
        self.shared_convs, self.shared_fcs, width, flat = self._branch(num_shared_convs, num_shared_fcs, in_channels,
                                                                       False)
        self.cls_convs, self.cls_fcs, cls_width, cls_flat = self._branch(num_cls_convs, num_cls_fcs, width, flat)
        self.reg_convs, self.reg_fcs, reg_width, reg_flat = self._branch(num_reg_convs, num_reg_fcs, width, flat)
        self.fc_cls = self._output_layer(cls_width, cls_flat, num_classes + 1)
        self.fc_reg = self._output_layer(reg_width, reg_flat,
                                         self.reg_dim if reg_class_agnostic else self.reg_dim * num_classes)
        self.init_weights()

    # ------------------------------------------------------------------ layers
    def _branch(self, n_convs, n_fcs, width, flat):
        """`flat`: the input is already a (R, width) matrix.  Returns (convs, fcs, output width, output is flat)."""
        convs, fcs = nn.ModuleList(), nn.ModuleList()
        for i in range(n_convs):
            assert not flat, "convolutions cannot follow a fully connected layer"
            convs.append(ConvModule(width, self.conv_out_channels, 3, padding=1, conv_cfg=None, norm_cfg=None))
            width = self.conv_out_channels
        for i in range(n_fcs):
            fcs.append(Linear(width, self.fc_out_channels) if flat else
                       RoIFeatureLinear(width, self.roi_feat_area, self.fc_out_channels))
            width, flat = self.fc_out_channels, True
        return convs, fcs, width, flat

    def _output_layer(self, width, flat, out):
        return Linear(width, out) if flat else RoIFeatureLinear(width, self.roi_feat_area, out)

    def init_weights(self):
        nn.init.normal_(self.fc_cls.weight, 0, 0.01)
        nn.init.constant_(self.fc_cls.bias, 0)
        nn.init.normal_(self.fc_reg.weight, 0, 0.001)
        nn.init.constant_(self.fc_reg.bias, 0)
        for fcs in (self.shared_fcs, self.cls_fcs, self.reg_fcs):
            for m in fcs:
                nn.init.xavier_uniform_(m.weight)
                nn.init.constant_(m.bias, 0)

    def _trunk(self, feats, rois):
        x = self.bbox_roi_extractor(feats[:self.bbox_roi_extractor.num_inputs], rois)
        for conv in self.shared_convs:
            x = conv(x)
        for fc in self.shared_fcs:
            x = F.relu(fc(x))
        outs = []
        for convs, fcs, last in ((self.cls_convs, self.cls_fcs, self.fc_cls), (self.reg_convs, self.reg_fcs, self.fc_reg)):
            y = x
            for conv in convs:
                y = conv(y)
            for fc in fcs:
                y = F.relu(fc(y))
            outs.append(last(y))
        return outs

    # ------------------------------------------------------------------ training
    @staticmethod
    def _dummy_box(like):
        return const_like([8.0, 8.0, 4.0, 4.0, 0.0], like)

    def _image_samples(self, table, target):
        """proposal table (P, 6) of one image -> the image's `num` sampled rows:
        boxes (num,5), labels (num,) long, label_weights (num,), bbox_targets (num,5), bbox_weights (num,5)"""
        gt = target["rboxes"].clone()
        gt[:, -1] *= -1                                   # Oriented R-CNN angle convention (L459-466)
        gt_labels = (target["labels"] - 1).long()         # 0-based, background = num_classes (L472)
        boxes, alive = table[:, :5], table[:, 5] >= 0
        overlaps = self.assigner.iou_calculator(gt, boxes)
        overlaps = torch.where(alive[None, :], overlaps, torch.full_like(overlaps, -1.0))   # padding rows: ignored
        assign = self.assigner.assign_wrt_overlaps(overlaps, gt_labels)
        gt_inds, labels = assign.gt_inds.long(), assign.labels.long()
        s = self.sampler
        if s.add_gt_as_proposals:                          # the gts join the candidates, matched to themselves (L92-99)
            k = gt.shape[0]
            boxes = torch.cat([gt, boxes])
            gt_inds = torch.cat([torch.arange(1, k + 1, device=gt.device), gt_inds])
            labels = torch.cat([gt_labels, labels])
        rows, valid, is_pos = sample_rows(gt_inds, s.num, s.pos_fraction, s.neg_pos_ub)
        sel = torch.where(valid[:, None], boxes[rows], self._dummy_box(boxes)[None, :])
        matched = gt[(gt_inds[rows] - 1).clamp(min=0)]
        bg = torch.full_like(rows, self.num_classes)
        out_labels = torch.where(is_pos, labels[rows], bg)
        pw = 1.0 if self.pos_weight <= 0 else self.pos_weight
        label_weights = valid.float() * torch.where(is_pos, torch.full_like(valid, pw, dtype=torch.float32),
                                                    torch.ones_like(valid, dtype=torch.float32))
        bbox_targets = self.bbox_coder.encode(sel, matched)
        bbox_weights = is_pos.float()[:, None].expand(-1, self.reg_dim)
        bbox_targets = torch.where(is_pos[:, None], bbox_targets, torch.zeros_like(bbox_targets))
        return sel, out_labels, label_weights, bbox_targets, bbox_weights, valid

    def forward_train(self, feats, proposal_tables, targets):
        per_image = [self._image_samples(t, tg) for t, tg in zip(proposal_tables, targets)]
        rois = torch.cat([torch.cat([b.new_full((b.shape[0], 1), float(i)), b], dim=1)
                          for i, (b, *_rest) in enumerate(per_image)])
        labels, label_w, box_t, box_w, valid = (torch.cat([p[k] for p in per_image]) for k in range(1, 6))
        cls_score, bbox_pred = self._trunk(feats, rois)
        n_rows = valid.sum().float()
        losses = dict()
        # classification: mean over the sampled rows (L318-325); regression: positives only, normalised by the
        # number of sampled rows (L326-343) -- both as weighted sums over the fixed-size row set
        losses["loss_cls"] = self.loss_cls(cls_score, labels, label_w,
                                           avg_factor=torch.clamp((label_w > 0).sum().float(), min=1.0))
        if self.reg_class_agnostic:
            pred = bbox_pred.view(bbox_pred.size(0), self.reg_dim)
        else:
            cls_of_row = labels.clamp(max=self.num_classes - 1)
            pred = bbox_pred.view(bbox_pred.size(0), -1, self.reg_dim)
            pred = pred.gather(1, cls_of_row[:, None, None].expand(-1, 1, self.reg_dim))[:, 0]
        losses["orcnn_bbox_loss"] = self.loss_bbox(pred, box_t, box_w, avg_factor=torch.clamp(n_rows, min=1.0))
        return losses

    # ------------------------------------------------------------------ inference
    def get_results(self, boxes, scores):
        """boxes (R, 5 | 5*C) decoded, scores (R, C+1) -> (dets (k,9) [poly8, score], labels (k,))"""
        num_classes = scores.size(1) - 1
        if boxes.shape[1] > 5:
            boxes = boxes.view(scores.size(0), -1, 5)
        else:
            boxes = boxes[:, None].expand(-1, num_classes, 5)
        fg = scores[:, :-1]
        hit = fg > self.score_thresh
        labels = hit.nonzero()[:, 1]                  # inference output has a data-dependent length
        if labels.numel() == 0:
            return boxes.new_zeros((0, 9)), boxes.new_zeros((0,), dtype=torch.long)
        return torch.cat([obb2poly(boxes[hit]), fg[hit].unsqueeze(1)], dim=1), labels

    def forward_test(self, feats, proposal_tables, targets):
        results = []
        for i, (table, target) in enumerate(zip(proposal_tables, targets)):
            alive = table[:, 5] >= 0
            boxes = torch.where(alive[:, None], table[:, :5], self._dummy_box(table)[None, :])
            rois = torch.cat([boxes.new_full((boxes.shape[0], 1), float(i)), boxes], dim=1)
            cls_score, bbox_pred = self._trunk(feats, rois)
            scores = F.softmax(cls_score, dim=1) * alive[:, None].float()       # padding rows score 0 everywhere
            decoded = self.bbox_coder.decode(boxes, bbox_pred, max_shape=target["img_size"])
            sf = target["scale_factor"]
            sf = const_like([sf] * 4 if isinstance(sf, (int, float)) else sf, decoded)
            decoded = decoded.view(decoded.size(0), -1, 5)
            decoded = torch.cat([decoded[..., :4] / sf, decoded[..., 4:]], dim=-1).view(decoded.size(0), -1)
            dets, labels = self.get_results(decoded, scores)
            results.append((dets[:, :8], dets[:, 8], labels))
        return results

    def forward(self, x, proposal_list, targets):
        if self.training:
            return self.forward_train(x, proposal_list, targets)
        return self.forward_test(x, proposal_list, targets)

    execute = forward
