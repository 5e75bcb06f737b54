"""Model / optimizer sections of the configs BASELINE.json names, as plain dicts (the reference's files are not
on the GPU box; `Config` loads them unchanged where they are present).  Sources:
configs/s2anet/s2anet_r50_fpn_1x_dota.py, configs/rotated_retinanet/rotated_retinanet_obb_r50_fpn_1x_dota.py,
configs/oriented_rcnn_r50_fpn_1x_dota_with_flip.py, configs/faster_rcnn_RoITrans_r50_fpn_1x_dota.py."""

_SGD_1X = dict(
    optimizer=dict(type="SGD", lr=0.01 / 4., momentum=0.9, weight_decay=0.0001, grad_clip=dict(max_norm=35, norm_type=2)),
    scheduler=dict(type="StepLR", warmup="linear", warmup_iters=500, warmup_ratio=1.0 / 3, milestones=[7, 10]))

S2ANET_CFG = dict(
    model=dict(
        type="S2ANet",
        backbone=dict(type="Resnet50", frozen_stages=1, return_stages=["layer1", "layer2", "layer3", "layer4"],
                      pretrained=True),
        neck=dict(type="FPN", in_channels=[256, 512, 1024, 2048], out_channels=256, start_level=1,
                  add_extra_convs="on_input", num_outs=5),
        bbox_head=dict(type="S2ANetHead", num_classes=16, in_channels=256, feat_channels=256, stacked_convs=2,
                       with_orconv=True, anchor_ratios=[1.0], anchor_strides=[8, 16, 32, 64, 128], anchor_scales=[4],
                       target_means=[.0, .0, .0, .0, .0], target_stds=[1.0, 1.0, 1.0, 1.0, 1.0])),
    # configs/s2anet/s2anet_r50_fpn_1x_dota.py:L151-166 (model section identical to L2-96 there)
    optimizer=dict(type="SGD", lr=0.01 / 4., momentum=0.9, weight_decay=0.0001, grad_clip=dict(max_norm=35, norm_type=2)),
    scheduler=dict(type="StepLR", warmup="linear", warmup_iters=500, warmup_ratio=1.0 / 3, milestones=[7, 10]))


RETINANET_CFG = dict(
    # configs/rotated_retinanet/rotated_retinanet_obb_r50_fpn_1x_dota.py:L2-57 (L1Loss only matters in training)
    model=dict(
        type="RotatedRetinaNet",
        backbone=dict(type="Resnet50", frozen_stages=1, return_stages=["layer1", "layer2", "layer3", "layer4"],
                      pretrained=True),
        neck=dict(type="FPN", in_channels=[256, 512, 1024, 2048], out_channels=256, start_level=1,
                  add_extra_convs="on_input", num_outs=5),
        bbox_head=dict(type="RotatedRetinaHead", num_classes=16, in_channels=256, feat_channels=256, stacked_convs=4,
                       octave_base_scale=4, scales_per_octave=3, anchor_ratios=[1.0, 0.5, 2.0],
                       anchor_strides=[8, 16, 32, 64, 128], loss_bbox=dict(type="L1Loss", loss_weight=1.0))))


ORCNN_CFG = dict(
    # configs/oriented_rcnn_r50_fpn_1x_dota_with_flip.py:L2-105 (head / rpn defaults are the config's values)
    model=dict(
        type="OrientedRCNN",
        backbone=dict(type="Resnet50", frozen_stages=1, return_stages=["layer1", "layer2", "layer3", "layer4"],
                      pretrained=True),
        neck=dict(type="FPN", in_channels=[256, 512, 1024, 2048], out_channels=256, num_outs=5),
        rpn=dict(type="OrientedRPNHead", in_channels=256, num_classes=1, nms_pre=2000, nms_post=2000),
        bbox_head=dict(type="OrientedHead", num_classes=15, in_channels=256, fc_out_channels=1024)),
    optimizer=dict(type="SGD", lr=0.005, momentum=0.9, weight_decay=0.0001, grad_clip=dict(max_norm=35, norm_type=2)),
    scheduler=dict(type="StepLR", warmup="linear", warmup_iters=500, warmup_ratio=1.0 / 3, milestones=[7, 10]))


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


def roitrans_train_cfg(backbone="Resnet50"):
    # optimizer / schedule: configs/faster_rcnn_RoITrans_r50_fpn_1x_dota.py:L196-213
    return dict(model=roitrans_cfg(backbone),
                optimizer=dict(type="SGD", lr=0.0025, momentum=0.9, weight_decay=0.0001,
                               grad_clip=dict(max_norm=35, norm_type=2)),
                scheduler=dict(type="StepLR", warmup="linear", warmup_iters=500, warmup_ratio=1.0 / 3,
                               milestones=[8, 11]))
