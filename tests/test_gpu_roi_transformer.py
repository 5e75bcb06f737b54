"""GPU: RoI-Transformer (configs/faster_rcnn_RoITrans_r50_fpn_1x_dota.py model section, built through the
registry) -- one train step with finite losses / gradients everywhere, inference output contract, and the
stage-1 -> stage-2 hand-off invariants."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def roitrans_cfg(backbone="Resnet50"):
    # configs/faster_rcnn_RoITrans_r50_fpn_1x_dota.py:L1-125
    return dict(
        type="RoITransformer",
        backbone=dict(type=backbone, frozen_stages=1, return_stages=["layer1", "layer2", "layer3", "layer4"],
                      pretrained=False),
        neck=dict(type="FPN", in_channels=[256, 512, 1024, 2048], out_channels=256, start_level=0,
                  add_extra_convs=False, num_outs=5),
        rpn_head=dict(type="FasterrcnnHead", in_channels=256, feat_channels=256, anchor_scales=[8],
                      anchor_ratios=[0.5, 1.0, 2.0], anchor_strides=[4, 8, 16, 32, 64],
                      target_means=[.0, .0, .0, .0], target_stds=[1.0, 1.0, 1.0, 1.0],
                      loss_cls=dict(type="CrossEntropyLossForRcnn", use_sigmoid=True, loss_weight=1.0),
                      loss_bbox=dict(type="SmoothL1Loss", beta=1.0 / 9.0, loss_weight=1.0)),
        bbox_roi_extractor=dict(type="SingleRoIExtractor",
                                roi_layer=dict(type="ROIAlign", output_size=7, sampling_ratio=2, version=1),
                                out_channels=256, featmap_strides=[4, 8, 16, 32]),
        bbox_head=dict(type="SharedFCBBoxHeadRbbox", num_fcs=2, in_channels=256, fc_out_channels=1024,
                       roi_feat_size=7, num_classes=16, target_means=[0., 0., 0., 0., 0.],
                       target_stds=[0.1, 0.1, 0.2, 0.2, 0.1], reg_class_agnostic=True, with_module=False,
                       loss_cls=dict(type="CrossEntropyLossForRcnn", use_sigmoid=False, loss_weight=1.0),
                       loss_bbox=dict(type="SmoothL1Loss", beta=1.0, loss_weight=1.0)),
        rbbox_roi_extractor=dict(type="RboxSingleRoIExtractor",
                                 roi_layer=dict(type="ROIAlignRotated", output_size=7, sampling_ratio=2),
                                 out_channels=256, featmap_strides=[4, 8, 16, 32]),
        rbbox_head=dict(type="SharedFCBBoxHeadRbbox", num_fcs=2, in_channels=256, fc_out_channels=1024,
                        roi_feat_size=7, num_classes=16, target_means=[0., 0., 0., 0., 0.],
                        target_stds=[0.05, 0.05, 0.1, 0.1, 0.05], reg_class_agnostic=False,
                        loss_cls=dict(type="CrossEntropyLossForRcnn", use_sigmoid=False, loss_weight=1.0),
                        loss_bbox=dict(type="SmoothL1Loss", beta=1.0, loss_weight=1.0)),
        train_cfg=dict(
            rpn=dict(assigner=dict(type="MaxIoUAssigner", pos_iou_thr=0.7, neg_iou_thr=0.3, min_pos_iou=0.3,
                                   ignore_iof_thr=-1, iou_calculator=dict(type="BboxOverlaps2D_v1")),
                     sampler=dict(type="RandomSampler", num=256, pos_fraction=0.5, neg_pos_ub=-1,
                                  add_gt_as_proposals=False),
                     allowed_border=0, pos_weight=-1, debug=False),
            rpn_proposal=dict(nms_across_levels=False, nms_pre=2000, nms_post=2000, max_num=2000, nms_thr=0.7,
                              min_bbox_size=0),
            rcnn=[dict(assigner=dict(type="MaxIoUAssigner", pos_iou_thr=0.5, neg_iou_thr=0.5, min_pos_iou=0.5,
                                     ignore_iof_thr=-1, iou_calculator=dict(type="BboxOverlaps2D_v1")),
                       sampler=dict(type="RandomSampler", num=512, pos_fraction=0.25, neg_pos_ub=-1,
                                    add_gt_as_proposals=True),
                       pos_weight=-1, debug=False),
                  dict(assigner=dict(type="MaxIoUAssignerRbbox", pos_iou_thr=0.5, neg_iou_thr=0.5, min_pos_iou=0.5,
                                     ignore_iof_thr=-1, iou_calculator=dict(type="BboxOverlaps2D_rotated")),
                       sampler=dict(type="RandomSamplerRotated", num=512, pos_fraction=0.25, neg_pos_ub=-1,
                                    add_gt_as_proposals=True),
                       pos_weight=-1, debug=False)]),
        test_cfg=dict(rpn=dict(nms_across_levels=False, nms_pre=2000, nms_post=2000, max_num=2000, nms_thr=0.7,
                               min_bbox_size=0),
                      rcnn=dict(score_thr=0.05, nms=dict(type="py_cpu_nms_poly_fast", iou_thr=0.1),
                                max_per_img=2000)))


def test_roi_transformer_train_step_and_inference(dev):
    import jdet_amd.models  # noqa: F401
    from jdet_amd.runner import synthetic_batch
    from jdet_amd.utils.general import parse_losses
    from jdet_amd.utils.registry import MODELS, build_from_cfg
    torch.manual_seed(0)
    m = build_from_cfg(roitrans_cfg(), MODELS).to(dev)
    m.train()
    images, targets = synthetic_batch(2, 256, dev, seed=7, num_gts=10)
    losses = m(images, targets)
    assert set(losses) == {"loss_rpn_cls", "loss_rpn_bbox", "s0.rbbox_loss_cls", "s0.rbbox_acc", "s0.rbbox_loss_bbox",
                           "s1.rbbox_loss_cls", "s1.rbbox_acc", "s1.rbbox_loss_bbox"}
    total, parsed = parse_losses(losses)
    assert torch.isfinite(total) and total.item() > 0
    # 16-way softmax at init: ~ log 16 up to the scale of the random backbone's features
    assert 1.0 < parsed["s0.rbbox_loss_cls"].item() < 12.0 and 1.0 < parsed["s1.rbbox_loss_cls"].item() < 12.0
    total.backward()
    g = {n: p.grad for n, p in m.named_parameters() if p.requires_grad}
    missing = [n for n, v in g.items() if v is None]
    assert not missing, missing
    assert all(torch.isfinite(v).all() for v in g.values())
    for n in ("bbox_head.shared_fcs.0.weight", "rbbox_head.shared_fcs.0.weight", "rpn_head.rpn_conv.weight",
              "neck.fpn_convs.0.conv.weight"):
        assert g[n].abs().sum() > 0, n
    m.eval()
    with torch.no_grad():
        res = m(images[:1], targets[:1])
    assert len(res) == 1
    polys, scores, labels = res[0]
    assert polys.shape[1] == 8 and polys.shape[0] == scores.shape[0] == labels.shape[0]
    assert polys.shape[0] <= 2000
    if scores.numel():
        assert float(scores.min()) > 0.05 and int(labels.min()) >= 0 and int(labels.max()) <= 14


def test_rpn_proposals_contract(dev):
    """FasterrcnnHead.get_bboxes: (<= max_num, 5) proposals inside the image, scores descending"""
    from jdet_amd.models.roi_heads import FasterrcnnHead
    torch.manual_seed(1)
    rpn = FasterrcnnHead(in_channels=16, feat_channels=16, anchor_scales=[8], anchor_ratios=[0.5, 1.0, 2.0],
                         anchor_strides=[4, 8, 16, 32, 64],
                         loss_cls=dict(type="CrossEntropyLossForRcnn", use_sigmoid=True)).to(dev)
    feats = [torch.randn(2, 16, 256 // s, 256 // s, device=dev) for s in (4, 8, 16, 32, 64)]
    metas = [dict(img_shape=(256, 256), pad_shape=(256, 256), scale_factor=1.0)] * 2
    cfg = dict(nms_across_levels=False, nms_pre=500, nms_post=300, max_num=400, nms_thr=0.7, min_bbox_size=0)
    with torch.no_grad():
        props = rpn.get_bboxes(*rpn(feats), metas, cfg)
    assert len(props) == 2
    for p in props:
        assert p.shape[1] == 5 and 0 < p.shape[0] <= 400
        assert float(p[:, :4].min()) >= 0 and float(p[:, :4].max()) <= 255
        s = p[:, 4].cpu().numpy()
        assert np.all(np.diff(s) <= 1e-7)
