"""Oriented R-CNN RPN head.

Contract of python/jdet/models/roi_heads/oriented_rpn_head.py:L9-492 (constructor arguments, parameter names
`rpn_conv / rpn_cls / rpn_reg`, loss keys, `forward(features, targets) -> (proposals per image, losses)`):
3x3 conv + 1x1 objectness (A per location) + 1x1 regression (A*6); targets = MaxIoUAssigner on the horizontal
anchors inside the image vs the gts' enclosing boxes, RandomSampler(256), MidpointOffsetCoder; proposals = per-level
top-k by score, midpoint-offset decode, per-level horizontal NMS on the enclosing boxes, best `nms_post` overall.

The execution is NOT the reference's (per-image index lists built with nonzero / boolean masks / randperm, each a
device -> host round trip, L281-366, L120-226).  Every tensor here has a shape fixed by the image size and the
number of gts:
  * targets are dense over ALL anchors of an image: anchors outside the image get overlap -1 (so the assigner
    ignores them and they can neither be sampled nor decide a low-quality match -- the same set the reference
    reaches by filtering first, L300-305); the sampled subset is drawn with random keys + top-k
    (models/boxes/fixed_shape.py) and written with a masked scatter; the sample counts that normalise the losses
    stay on the device.
  * proposals are always `nms_post` rows per image: [xc, yc, w, h, theta, score] sorted by score, rows beyond the
    survivors of NMS carry score -1 (consumers ignore them).  No boolean indexing, no `.all()` / `.any()`.
"""
import torch
import torch.nn.functional as F
from torch import nn

from jdet_amd.ops.conv_igemm import conv_module
from jdet_amd.models.boxes.anchor_target import anchor_inside_flags
from jdet_amd.models.boxes.fixed_shape import dense_anchor_targets, proposal_table
from jdet_amd.models.utils.level_pack import run_levels
from jdet_amd.ops import conv_igemm
from jdet_amd.ops.bbox_transforms import obb2hbb
from jdet_amd.utils.registry import BOXES, HEADS, LOSSES, build_from_cfg

INVALID_SCORE = -1.0   # score of a padding row in a proposal table


@HEADS.register_module()
class OrientedRPNHead(nn.Module):
    def __init__(self, in_channels, num_classes=1, min_bbox_size=0, nms_thresh=0.8, nms_pre=2000, nms_post=2000,
                 feat_channels=256, bbox_type="obb", reg_dim=6, background_label=0, reg_decoded_bbox=False,
                 pos_weight=-1,
                 anchor_generator=dict(type="AnchorGenerator", scales=[8], ratios=[0.5, 1.0, 2.0],
                                       strides=[4, 8, 16, 32, 64]),
                 bbox_coder=dict(type="MidpointOffsetCoder", target_means=[.0, .0, .0, .0, .0, .0],
                                 target_stds=[1.0, 1.0, 1.0, 1.0, 0.5, 0.5]),
                 loss_cls=dict(type="CrossEntropyLossForRcnn", use_sigmoid=True, loss_weight=1.0),
                 loss_bbox=dict(type="SmoothL1Loss", beta=1.0 / 9.0, loss_weight=1.0),
                 assigner=dict(type="MaxIoUAssigner", pos_iou_thr=0.7, neg_iou_thr=0.3, min_pos_iou=0.3,
                               ignore_iof_thr=-1, match_low_quality=True, assigned_labels_filled=-1),
                 sampler=dict(type="RandomSampler", num=256, pos_fraction=0.5, neg_pos_ub=-1,
                              add_gt_as_proposals=False)):
        super().__init__()
        assert bbox_type == "obb" and reg_dim == 6 and not reg_decoded_bbox, "the Oriented R-CNN configuration"
        assert loss_cls.get("use_sigmoid", False) and num_classes == 1, "objectness = one sigmoid per anchor"
        self.in_channels, self.feat_channels, self.num_classes = in_channels, feat_channels, num_classes
        self.min_bbox_size, self.nms_thresh, self.nms_pre, self.nms_post = min_bbox_size, nms_thresh, nms_pre, nms_post
        self.reg_dim, self.pos_weight = reg_dim, pos_weight
        self.background_label = num_classes if background_label is None else background_label
        self.cls_out_channels = num_classes
        self.bbox_coder = build_from_cfg(bbox_coder, BOXES)
        self.loss_cls = build_from_cfg(loss_cls, LOSSES)
        self.loss_bbox = build_from_cfg(loss_bbox, LOSSES)
        self.assigner = build_from_cfg(assigner, BOXES)
        self.sampler = build_from_cfg(sampler, BOXES)     # carries num / pos_fraction / neg_pos_ub
        self.anchor_generator = build_from_cfg(anchor_generator, BOXES)
        self.num_anchors = self.anchor_generator.num_base_anchors[0]
        self.rpn_conv = nn.Conv2d(in_channels, feat_channels, 3, padding=1)
        self.rpn_cls = nn.Conv2d(feat_channels, self.num_anchors * num_classes, 1)
        self.rpn_reg = nn.Conv2d(feat_channels, self.num_anchors * 6, 1)

    # ------------------------------------------------------------------ network
    def forward_single(self, x, mask=None):
        x = conv_igemm.conv3x3_module(self.rpn_conv, x, relu=True)      # the 1x1 layers below read no neighbours: a packed input needs no mask
        return conv_module(self.rpn_cls, x), conv_module(self.rpn_reg, x)

    @staticmethod
    def _per_anchor(t, width):
        """(N, A*width, H, W) -> (N, H*W*A, width): the anchor order of grid_anchors (location-major, A fastest)"""
        n = t.shape[0]
        return t.permute(0, 2, 3, 1).reshape(n, -1, width)

    # ------------------------------------------------------------------ targets (dense over all anchors)
    def _image_targets(self, anchors, inside, target):
        """anchors (A,4) of all levels, inside (A,) bool -> labels (A,) long, label_weights (A,), bbox_targets (A,6),
        bbox_weights (A,6), n_pos, n_neg (0-d device tensors)"""
        gt_obb = target["rboxes"].clone()
        gt_obb[:, -1] *= -1                      # Oriented R-CNN angle convention
        gt_hbb = obb2hbb(gt_obb)                 # the assigner sees the anchors' box type (L308-312)
        return dense_anchor_targets(anchors, inside, gt_hbb, gt_obb, self.assigner, self.sampler,
                                    self.bbox_coder.encode, self.reg_dim, self.background_label, self.pos_weight)

    def loss(self, cls_scores, bbox_preds, targets):
        sizes = [tuple(c.shape[-2:]) for c in cls_scores]
        dev = cls_scores[0].device
        level_anchors = self.anchor_generator.grid_anchors(sizes, device=dev)
        anchors = torch.cat(level_anchors)
        per_image = []
        for target in targets:
            valid = torch.cat(self.anchor_generator.valid_flags(sizes, target["pad_shape"], device=dev))
            inside = anchor_inside_flags(anchors, valid, target["img_size"][:2], allowed_border=0)
            per_image.append(self._image_targets(anchors, inside, target))
        labels, label_w, box_t, box_w = (torch.stack([p[k] for p in per_image]) for k in range(4))
        # sum over images of max(#pos, 1) + max(#neg, 1)  (L376-378), kept on the device
        n_samples = sum(torch.clamp(p[4], min=1) + torch.clamp(p[5], min=1) for p in per_image).float()
        losses_cls, losses_bbox = [], []
        start = 0
        for cls, reg, lvl in zip(cls_scores, bbox_preds, level_anchors):
            n = lvl.shape[0]
            sl = slice(start, start + n)
            start += n
            score = self._per_anchor(cls, self.cls_out_channels).reshape(-1, self.cls_out_channels)
            losses_cls.append(self.loss_cls(score, labels[:, sl].reshape(-1), label_w[:, sl].reshape(-1),
                                            avg_factor=n_samples))
            pred = self._per_anchor(reg, self.reg_dim).reshape(-1, self.reg_dim)
            losses_bbox.append(self.loss_bbox(pred, box_t[:, sl].reshape(-1, self.reg_dim),
                                              box_w[:, sl].reshape(-1, self.reg_dim), avg_factor=n_samples))
        return dict(loss_rpn_cls=losses_cls, loss_rpn_bbox=losses_bbox)

    # ------------------------------------------------------------------ proposals (always nms_post rows)
    def _image_proposals(self, level_scores, level_deltas, level_anchors, img_shape):
        scores, deltas, anchors, ids = [], [], [], []
        for lvl, (s, d, a) in enumerate(zip(level_scores, level_deltas, level_anchors)):
            s = s.sigmoid()
            k = s.shape[0] if self.nms_pre <= 0 else min(self.nms_pre, s.shape[0])
            s, top = torch.topk(s, k)         # always: proposal_table wants every level sorted by descending score
            d, a = d[top], a[top]
            scores.append(s)
            deltas.append(d)
            anchors.append(a)
            ids.append(torch.full((s.shape[0],), lvl, dtype=torch.long, device=s.device))
        scores, ids = torch.cat(scores), torch.cat(ids)
        boxes = self.bbox_coder.decode(torch.cat(anchors), torch.cat(deltas), max_shape=img_shape)
        alive = torch.ones_like(scores, dtype=torch.bool)
        if self.min_bbox_size >= 0:
            alive = (boxes[:, 2] > self.min_bbox_size) & (boxes[:, 3] > self.min_bbox_size)
        return proposal_table(obb2hbb(boxes), scores, ids, [int(x.shape[0]) for x in anchors], alive, self.nms_thresh,
                              None, self.nms_post, invalid_score=INVALID_SCORE, payload=boxes)

    def get_bboxes(self, cls_scores, bbox_preds, targets):
        sizes = [tuple(c.shape[-2:]) for c in cls_scores]
        level_anchors = self.anchor_generator.grid_anchors(sizes, device=cls_scores[0].device)
        scores = [self._per_anchor(c.detach(), 1)[..., 0] for c in cls_scores]          # (N, H*W*A) per level
        deltas = [self._per_anchor(r.detach(), self.reg_dim) for r in bbox_preds]
        return [self._image_proposals([s[i] for s in scores], [d[i] for d in deltas], level_anchors,
                                      target["img_size"]) for i, target in enumerate(targets)]

    def forward(self, features, targets):
        outs = run_levels(list(features), self.forward_single)
        cls_scores, bbox_preds = [o[0] for o in outs], [o[1] for o in outs]
        losses = self.loss(cls_scores, bbox_preds, targets) if self.training else dict()
        return self.get_bboxes(cls_scores, bbox_preds, targets), losses

    execute = forward
