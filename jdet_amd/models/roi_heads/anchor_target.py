"""Horizontal-anchor target computation of the Faster R-CNN style RPN.  Mirrors
python/jdet/models/roi_heads/anchor_target.py:L23-208: same pipeline as models/boxes/anchor_target.py with
the encoder fixed to `bbox2delta(pos_bboxes, pos_gt_bboxes, target_means, target_stds)` (L157-159)."""
from functools import partial

from jdet_amd.models.boxes import anchor_target as _base
from jdet_amd.models.boxes.anchor_target import anchor_inside_flags, assign_and_sample, images_to_levels  # noqa: F401
from jdet_amd.ops.bbox_transforms import bbox2delta


def anchor_target(anchor_list, valid_flag_list, gt_bboxes_list, img_metas, target_means, target_stds, cfg,
                  gt_bboxes_ignore_list=None, gt_labels_list=None, label_channels=1, sampling=True,
                  unmap_outputs=True):
    enc = partial(bbox2delta, means=target_means, stds=target_stds)
    return _base.anchor_target(anchor_list, valid_flag_list, gt_bboxes_list, img_metas, target_means, target_stds,
                               cfg, gt_bboxes_ignore_list=gt_bboxes_ignore_list, gt_labels_list=gt_labels_list,
                               label_channels=label_channels, sampling=sampling, unmap_outputs=unmap_outputs,
                               encode_fn=enc)
