"""RoI-Transformer detector.  Mirrors python/jdet/models/networks/roi_transformer.py:L9-203:
backbone -> FPN -> RPN (horizontal proposals) -> horizontal RoIAlign + head 0 (regresses a rotated RoI
per proposal) -> rotated RoIAlign on the enlarged RRoIs + head 1 (final rotated boxes)."""
import torch
from torch import nn

from jdet_amd.ops.bbox_transforms import bbox2roi, choose_best_Rroi_batch, dbbox2result, dbbox2roi, roi2droi
from jdet_amd.utils.registry import BACKBONES, BOXES, HEADS, MODELS, NECKS, ROI_EXTRACTORS, build_from_cfg


@MODELS.register_module()
class RoITransformer(nn.Module):
    def __init__(self, backbone, neck=None, rpn_head=None, bbox_roi_extractor=None, bbox_head=None,
                 rbbox_roi_extractor=None, rbbox_head=None, train_cfg=None, test_cfg=None, pretrained=None):
        super().__init__()
        self.backbone = build_from_cfg(backbone, BACKBONES)
        self.neck = build_from_cfg(neck, NECKS)
        self.rpn_head = build_from_cfg(rpn_head, HEADS)
        self.bbox_roi_extractor = build_from_cfg(bbox_roi_extractor, ROI_EXTRACTORS)
        self.bbox_head = build_from_cfg(bbox_head, HEADS)
        self.rbbox_roi_extractor = build_from_cfg(rbbox_roi_extractor, ROI_EXTRACTORS)
        self.rbbox_head = build_from_cfg(rbbox_head, HEADS)
        self.train_cfg = train_cfg
        self.test_cfg = test_cfg
        for m in (self.rpn_head, self.bbox_head, self.rbbox_head):
            if m is not None and hasattr(m, "init_weights"):
                m.init_weights()

    def _enlarge(self, rrois):
        out = rrois.clone()
        out[:, 3] = out[:, 3] * self.rbbox_roi_extractor.w_enlarge
        out[:, 4] = out[:, 4] * self.rbbox_roi_extractor.h_enlarge
        return out

    def execute_train(self, images, targets=None):
        image_meta, gt_labels, gt_bboxes, gt_bboxes_ignore, gt_obbs = [], [], [], [], []
        for target in targets:
            image_meta.append(dict(ori_shape=target["ori_img_size"], img_shape=target["img_size"],
                                   pad_shape=target["pad_shape"], img_file=target.get("img_file", ""),
                                   to_bgr=target.get("to_bgr", False), scale_factor=target["scale_factor"]))
            gt_bboxes.append(target["hboxes"])
            gt_labels.append(target["labels"])
            gt_bboxes_ignore.append(target.get("hboxes_ignore"))
            gt_obbs.append(target["rboxes"])
        losses = dict()
        features = self.backbone(images)
        if self.neck:
            features = self.neck(features)
        rpn_outs = self.rpn_head(features)
        losses.update(self.rpn_head.loss(*rpn_outs, gt_bboxes, image_meta, self.train_cfg["rpn"],
                                         gt_bboxes_ignore=gt_bboxes_ignore))
        proposal_cfg = self.train_cfg.get("rpn_proposal", self.test_cfg["rpn"])
        with torch.no_grad():
            proposal_list = self.rpn_head.get_bboxes(*rpn_outs, image_meta, proposal_cfg)

            bbox_assigner = build_from_cfg(self.train_cfg["rcnn"][0]["assigner"], BOXES)
            bbox_sampler = build_from_cfg(self.train_cfg["rcnn"][0]["sampler"], BOXES)
            sampling_results = []
            for proposal, gt_bbox, gt_bbox_ignore, gt_label in zip(proposal_list, gt_bboxes, gt_bboxes_ignore,
                                                                   gt_labels):
                assign_result = bbox_assigner.assign(proposal[:, :4], gt_bbox, gt_bbox_ignore, gt_label)
                sampling_results.append(bbox_sampler.sample(assign_result, proposal, gt_bbox, gt_label))
            rois = bbox2roi([res.bboxes for res in sampling_results])
        bbox_feats = self.bbox_roi_extractor(features[:self.bbox_roi_extractor.num_inputs], rois)
        cls_score, bbox_pred = self.bbox_head(bbox_feats)
        with torch.no_grad():
            rbbox_targets = self.bbox_head.get_target(sampling_results, gt_obbs, gt_labels,
                                                      self.train_cfg["rcnn"][0])
        for name, value in self.bbox_head.loss(cls_score, bbox_pred, *rbbox_targets).items():
            losses["s{}.{}".format(0, name)] = value

        with torch.no_grad():
            pos_is_gts = [res.pos_is_gt for res in sampling_results]
            roi_labels = rbbox_targets[0]
            rotated_proposal_list = self.bbox_head.refine_rbboxes(roi2droi(rois), roi_labels, bbox_pred.detach(),
                                                                  pos_is_gts, image_meta)
            bbox_assigner = build_from_cfg(self.train_cfg["rcnn"][1]["assigner"], BOXES)
            bbox_sampler = build_from_cfg(self.train_cfg["rcnn"][1]["sampler"], BOXES)
            sampling_results = []
            for rotated_proposal, gt_obb, gt_bbox_ignore, gt_label in zip(rotated_proposal_list, gt_obbs,
                                                                          gt_bboxes_ignore, gt_labels):
                gt_obbs_best_roi = choose_best_Rroi_batch(gt_obb)
                assign_result = bbox_assigner.assign(rotated_proposal, gt_obbs_best_roi, gt_bbox_ignore, gt_label)
                sampling_results.append(bbox_sampler.sample(assign_result, rotated_proposal, gt_obbs_best_roi,
                                                            gt_label))
            rrois = self._enlarge(dbbox2roi([res.bboxes for res in sampling_results]))
        rbbox_feats = self.rbbox_roi_extractor(features[:self.rbbox_roi_extractor.num_inputs], rrois)
        cls_score, rbbox_pred = self.rbbox_head(rbbox_feats)
        with torch.no_grad():
            rbbox_targets = self.rbbox_head.get_target_rbbox(sampling_results, gt_obbs, gt_labels,
                                                             self.train_cfg["rcnn"][1])
        for name, value in self.rbbox_head.loss(cls_score, rbbox_pred, *rbbox_targets).items():
            losses["s{}.{}".format(1, name)] = value
        return losses

    @torch.no_grad()
    def execute_test(self, images, targets=None, rescale=False):
        img_meta, img_shape, scale_factor = [], [], []
        for target in targets:
            ori = target["ori_img_size"]
            img_meta.append(dict(ori_shape=ori, img_shape=ori, pad_shape=ori, scale_factor=target["scale_factor"],
                                 img_file=target.get("img_file", "")))
            img_shape.append(target["img_size"])
            scale_factor.append(target["scale_factor"])
        x = self.backbone(images)
        if self.neck:
            x = self.neck(x)
        rpn_outs = self.rpn_head(x)
        proposal_list = self.rpn_head.get_bboxes(*rpn_outs, img_meta, self.test_cfg["rpn"])
        rois = bbox2roi(proposal_list)
        roi_feats = self.bbox_roi_extractor(x[:len(self.bbox_roi_extractor.featmap_strides)], rois)
        cls_score, bbox_pred = self.bbox_head(roi_feats)
        bbox_label = torch.argmax(cls_score, dim=1)
        rrois = self.bbox_head.regress_by_class_rbbox(roi2droi(rois), bbox_label, bbox_pred, img_meta[0])
        rbbox_feats = self.rbbox_roi_extractor(x[:len(self.rbbox_roi_extractor.featmap_strides)],
                                               self._enlarge(rrois))
        rcls_score, rbbox_pred = self.rbbox_head(rbbox_feats)
        sf = scale_factor[0] if len(scale_factor) == 1 else scale_factor
        det_rbboxes, det_labels = self.rbbox_head.get_det_rbboxes(rrois, rcls_score, rbbox_pred, img_shape, sf,
                                                                  rescale=rescale, cfg=self.test_cfg["rcnn"])
        return [dbbox2result(det_rbboxes, det_labels, self.rbbox_head.num_classes)]

    def forward(self, images, targets=None):
        return self.execute_train(images, targets) if self.training else self.execute_test(images, targets)

    execute = forward
