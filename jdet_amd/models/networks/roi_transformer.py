"""RoI-Transformer detector.  Mirrors python/jdet/models/networks/roi_transformer.py:L9-203:
backbone -> FPN -> RPN (horizontal proposals) -> horizontal RoIAlign + head 0 (regresses a rotated RoI
per proposal) -> rotated RoIAlign on the enlarged RRoIs + head 1 (final rotated boxes)."""
import torch
from torch import nn

from jdet_amd.models.boxes.fixed_shape import sample_stage_rows
from jdet_amd.ops.bbox_transforms import choose_best_Rroi_batch, dbbox2result, roi2droi
from jdet_amd.utils.general import const_like
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

    def _stage_rows(self, stage, cands, alive, gts, gt_labels, dummy):
        cfg = self.train_cfg["rcnn"][stage]
        key = "_stage%d" % stage
        if not hasattr(self, key):     # assigner / sampler objects are stateless: built once
            setattr(self, key, (build_from_cfg(cfg["assigner"], BOXES), build_from_cfg(cfg["sampler"], BOXES)))
        assigner, sampler = getattr(self, key)
        return sample_stage_rows(cands, alive, gts, gt_labels, assigner, sampler, const_like(dummy, cands))

    @staticmethod
    def _with_image_index(per_image_boxes):
        return torch.cat([torch.cat([b.new_full((b.shape[0], 1), float(i)), b], dim=1)
                          for i, b in enumerate(per_image_boxes)])

    def execute_train(self, images, targets=None):
        """Two sampled R-CNN stages on fixed-size row sets: every image contributes exactly `sampler.num` rows per
        stage (padding rows have weight 0 everywhere), so the step has static shapes and never waits for the device
        (the reference slices per-image index lists, roi_transformer.py:L60-134)."""
        image_meta = [dict(ori_shape=t["ori_img_size"], img_shape=t["img_size"], pad_shape=t["pad_shape"],
                           img_file=t.get("img_file", ""), to_bgr=t.get("to_bgr", False),
                           scale_factor=t["scale_factor"]) for t in targets]
        gt_bboxes = [t["hboxes"] for t in targets]
        gt_labels = [t["labels"] for t in targets]
        gt_obbs = [t["rboxes"] for t in targets]
        losses = dict()
        features = self.backbone(images)
        if self.neck:
            features = self.neck(features)
        rpn_outs = self.rpn_head(features)
        losses.update(self.rpn_head.loss(*rpn_outs, gt_bboxes, image_meta, self.train_cfg["rpn"],
                                         gt_bboxes_ignore=[t.get("hboxes_ignore") for t in targets]))
        proposal_cfg = self.train_cfg.get("rpn_proposal", self.test_cfg["rpn"])
        with torch.no_grad():
            tables = self.rpn_head.get_bboxes(*rpn_outs, image_meta, proposal_cfg)
            rows0 = [self._stage_rows(0, t[:, :4], t[:, 4] >= 0, g, l, (4.0, 4.0, 12.0, 12.0))
                     for t, g, l in zip(tables, gt_bboxes, gt_labels)]
            rois = self._with_image_index([r.boxes for r in rows0])
        bbox_feats = self.bbox_roi_extractor(features[:self.bbox_roi_extractor.num_inputs], rois)
        cls_score, bbox_pred = self.bbox_head(bbox_feats)
        with torch.no_grad():
            rbbox_targets = self.bbox_head.get_target(rows0, gt_obbs, gt_labels, self.train_cfg["rcnn"][0])
        for name, value in self.bbox_head.loss(cls_score, bbox_pred, *rbbox_targets).items():
            losses["s{}.{}".format(0, name)] = value

        with torch.no_grad():
            refined = self.bbox_head.refine_rbboxes(roi2droi(rois), rbbox_targets[0], bbox_pred.detach(), rows0,
                                                    image_meta)
            gt_best = [choose_best_Rroi_batch(g) for g in gt_obbs]
            rows1 = [self._stage_rows(1, boxes, alive, g, l, (8.0, 8.0, 4.0, 4.0, 0.0))
                     for (boxes, alive), g, l in zip(refined, gt_best, gt_labels)]
            rrois = self._enlarge(self._with_image_index([r.boxes for r in rows1]))
        rbbox_feats = self.rbbox_roi_extractor(features[:self.rbbox_roi_extractor.num_inputs], rrois)
        cls_score, rbbox_pred = self.rbbox_head(rbbox_feats)
        with torch.no_grad():
            rbbox_targets = self.rbbox_head.get_target_rbbox(rows1, gt_best, gt_labels, self.train_cfg["rcnn"][1])
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
        tables = self.rpn_head.get_bboxes(*rpn_outs, img_meta, self.test_cfg["rpn"])
        results = []
        for i, table in enumerate(tables):       # the reference handles one image per call (img_meta[0], L176-178)
            alive = table[:, 4] >= 0              # padding rows: pooled from a dummy box, scores zeroed below
            boxes = torch.where(alive[:, None], table[:, :4], const_like((4.0, 4.0, 12.0, 12.0), table)[None, :])
            rois = self._with_image_index([boxes])
            rois[:, 0] = float(i)
            roi_feats = self.bbox_roi_extractor(x[:len(self.bbox_roi_extractor.featmap_strides)], rois)
            cls_score, bbox_pred = self.bbox_head(roi_feats)
            bbox_label = torch.argmax(cls_score, dim=1)
            rrois = self.bbox_head.regress_by_class_rbbox(roi2droi(rois), bbox_label, bbox_pred, img_meta[i])
            rbbox_feats = self.rbbox_roi_extractor(x[:len(self.rbbox_roi_extractor.featmap_strides)],
                                                   self._enlarge(rrois))
            rcls_score, rbbox_pred = self.rbbox_head(rbbox_feats)
            det_rbboxes, det_labels = self.rbbox_head.get_det_rbboxes(
                rrois, rcls_score, rbbox_pred, img_shape[i], scale_factor[i], rescale=rescale,
                cfg=self.test_cfg["rcnn"], alive=alive)
            results.append(dbbox2result(det_rbboxes, det_labels, self.rbbox_head.num_classes))
        return results

    def forward(self, images, targets=None):
        return self.execute_train(images, targets) if self.training else self.execute_test(images, targets)

    execute = forward
