"""Anchor target computation.  Mirrors python/jdet/models/boxes/anchor_target.py: `anchor_target`
L18-87, `images_to_levels` L90-103, `anchor_target_single` L105-180, `anchor_inside_flags` L184-198.

Per image: inside flags -> assign -> (pseudo) sample -> encode positives -> scatter labels / weights /
targets by pos_inds / neg_inds -> unmap -> split per level.  Same structure as the reference; the
assigner and the encoder each are one fused launch here.
"""
import torch

from jdet_amd.utils.general import multi_apply, unmap
from jdet_amd.utils.registry import BOXES, build_from_cfg

from .sampler import PseudoSampler


def assign_and_sample(bboxes, gt_bboxes, gt_bboxes_ignore, gt_labels, cfg):
    bbox_assigner = build_from_cfg(cfg.get("assigner", ""), BOXES)
    bbox_sampler = build_from_cfg(cfg.get("sampler", ""), BOXES)
    assign_result = bbox_assigner.assign(bboxes, gt_bboxes, gt_bboxes_ignore, gt_labels)
    sampling_result = bbox_sampler.sample(assign_result, bboxes, gt_bboxes, gt_labels)
    return assign_result, sampling_result


def images_to_levels(target, num_level_anchors):
    """[target_img0, target_img1] -> [target_level0, target_level1, ...]"""
    target = torch.stack(target, 0)
    level_targets = []
    start = 0
    for n in num_level_anchors:
        end = start + n
        level_targets.append(target[:, start:end])
        start = end
    return level_targets


def anchor_inside_flags(flat_anchors, valid_flags, img_shape, allowed_border=0):
    img_h, img_w = img_shape[:2]
    if allowed_border >= 0:
        return valid_flags & (flat_anchors[:, 0] >= -allowed_border) & (flat_anchors[:, 1] >= -allowed_border) & \
            (flat_anchors[:, 2] < img_w + allowed_border) & (flat_anchors[:, 3] < img_h + allowed_border)
    return valid_flags


def anchor_target_single(flat_anchors, valid_flags, gt_bboxes, gt_bboxes_ignore, gt_labels, img_meta, target_means,
                         target_stds, cfg=None, label_channels=1, sampling=True, unmap_outputs=True, encode_fn=None):
    """encode_fn(pos_bboxes, pos_gt_bboxes): replaces the coder; the roi_heads/anchor_target.py twin of the
    reference hard-wires `bbox2delta(..., target_means, target_stds)` there (L157-159)."""
    if encode_fn is None:
        bbox_coder_cfg = cfg.get("bbox_coder", "")
        if bbox_coder_cfg == "":
            bbox_coder_cfg = dict(type="DeltaXYWHBBoxCoder")
        bbox_coder = build_from_cfg(bbox_coder_cfg, BOXES)
    reg_decoded_bbox = cfg.get("reg_decoded_bbox", False)
    allowed_border = cfg.get("allowed_border", -1)
    inside_flags = anchor_inside_flags(flat_anchors, valid_flags, img_meta["img_shape"][:2], allowed_border)
    all_inside = allowed_border < 0 and bool(img_meta.get("_all_valid", False))
    if not all_inside and not bool(inside_flags.any()):
        return (None,) * 6
    anchors = flat_anchors if all_inside else flat_anchors[inside_flags, :]

    if sampling:
        assign_result, sampling_result = assign_and_sample(anchors, gt_bboxes, gt_bboxes_ignore, None, cfg)
    else:
        bbox_assigner = build_from_cfg(cfg.get("assigner", ""), BOXES)
        assign_result = bbox_assigner.assign(anchors, gt_bboxes, gt_bboxes_ignore, gt_labels)
        sampling_result = PseudoSampler().sample(assign_result, anchors, gt_bboxes)

    num_valid_anchors = anchors.shape[0]
    bbox_targets = torch.zeros_like(anchors)
    bbox_weights = torch.zeros_like(anchors)
    labels = torch.zeros((num_valid_anchors,), dtype=torch.int32, device=anchors.device)
    label_weights = torch.zeros((num_valid_anchors,), dtype=torch.float32, device=anchors.device)
    pos_inds, neg_inds = sampling_result.pos_inds, sampling_result.neg_inds
    if len(pos_inds) > 0:
        if encode_fn is not None:
            pos_bbox_targets = encode_fn(sampling_result.pos_bboxes, sampling_result.pos_gt_bboxes)
        elif not reg_decoded_bbox:
            pos_bbox_targets = bbox_coder.encode(sampling_result.pos_bboxes, sampling_result.pos_gt_bboxes)
        else:
            pos_bbox_targets = sampling_result.pos_gt_bboxes
        bbox_targets[pos_inds, :] = pos_bbox_targets.to(bbox_targets.dtype)
        bbox_weights[pos_inds, :] = 1.0
        if gt_labels is None:
            labels[pos_inds] = 1
        else:
            labels[pos_inds] = gt_labels[sampling_result.pos_assigned_gt_inds].to(labels.dtype)
        pos_weight = cfg.get("pos_weight", -1)
        label_weights[pos_inds] = 1.0 if pos_weight <= 0 else pos_weight
    if len(neg_inds) > 0:
        label_weights[neg_inds] = 1.0

    if unmap_outputs and not all_inside:
        num_total_anchors = flat_anchors.size(0)
        labels = unmap(labels, num_total_anchors, inside_flags)
        label_weights = unmap(label_weights, num_total_anchors, inside_flags)
        bbox_targets = unmap(bbox_targets, num_total_anchors, inside_flags)
        bbox_weights = unmap(bbox_weights, num_total_anchors, inside_flags)
    return (labels, label_weights, bbox_targets, bbox_weights, pos_inds, neg_inds)


def _dense_ok(cfg, sampling, img_metas, gt_bboxes_ignore_list, anchors, encode_fn):
    """the case every single-stage rotated config hits: PseudoSampler, every anchor valid, no ignore
    regions, MaxIoUAssigner on 5-parameter boxes, DeltaXYWHABBoxCoder targets, device tensors"""
    if sampling or encode_fn is not None or cfg.get("reg_decoded_bbox", False) or cfg.get("allowed_border", -1) >= 0:
        return False
    if not all(bool(m.get("_all_valid", False)) for m in img_metas):
        return False
    if any(g is not None and g.numel() > 0 for g in gt_bboxes_ignore_list):
        return False
    coder, assigner = cfg.get("bbox_coder", ""), cfg.get("assigner", "")
    if coder == "" or assigner == "" or coder.get("type") != "DeltaXYWHABBoxCoder":
        return False
    if assigner.get("type") != "MaxIoUAssigner" or assigner.get("ignore_iof_thr", -1) > 0:
        return False
    return anchors.is_cuda and anchors.shape[-1] == 5 and anchors.dtype == torch.float32


def anchor_target_dense(anchor_list, gt_bboxes_list, gt_labels_list, cfg):
    """Fixed-shape twin of `anchor_target` (same values): per image one IoU launch, the fused assigner and
    ONE fused target launch (jdet_anchor_targets_rotated) instead of nonzero + 6 index scatters; the number of
    positives stays on the device.  anchor_list: per image a (A,5) tensor."""
    from jdet_amd import _lib as L
    assigner = build_from_cfg(cfg.get("assigner", ""), BOXES)
    coder = build_from_cfg(cfg.get("bbox_coder", ""), BOXES)
    pos_weight = cfg.get("pos_weight", -1)
    pos_weight = 1.0 if pos_weight <= 0 else float(pos_weight)
    num_imgs = len(anchor_list)
    dev = anchor_list[0].device
    A = anchor_list[0].shape[0]
    labels = torch.empty((num_imgs, A), dtype=torch.int32, device=dev)
    label_weights = torch.empty((num_imgs, A), dtype=torch.float32, device=dev)
    bbox_targets = torch.empty((num_imgs, A, 5), dtype=torch.float32, device=dev)
    bbox_weights = torch.empty((num_imgs, A, 5), dtype=torch.float32, device=dev)
    num_pos = torch.zeros((num_imgs,), dtype=torch.int32, device=dev)
    means, stds = L.vec5(coder.means), L.vec5(coder.stds)
    for i in range(num_imgs):
        anchors, gt = L.f32c(anchor_list[i]), L.f32c(gt_bboxes_list[i])
        res = assigner.assign(anchors, gt, None, None)
        gl = gt_labels_list[i].to(torch.int32).contiguous() if gt_labels_list[i] is not None else None
        L.check(L.lib().jdet_anchor_targets_rotated(
            L.ptr(anchors), L.ptr(gt), L.ptr(gl), L.ptr(res.gt_inds), A, gt.shape[0], means, stds, pos_weight,
            L.ptr(labels[i]), L.ptr(label_weights[i]), L.ptr(bbox_targets[i]), L.ptr(bbox_weights[i]),
            L.ptr(num_pos[i:i + 1]), L.stream_ptr(anchors)), "jdet_anchor_targets_rotated")
    # sum_img max(npos_img, 1) (anchor_target.py:L77) as a 0-dim device tensor: the loss normaliser never
    # forces a host sync
    return labels, label_weights, bbox_targets, bbox_weights, num_pos.clamp(min=1).sum().to(torch.float32)


def anchor_target(anchor_list, valid_flag_list, gt_bboxes_list, img_metas, target_means, target_stds, cfg,
                  gt_bboxes_ignore_list=None, gt_labels_list=None, label_channels=1, sampling=True,
                  unmap_outputs=True, encode_fn=None, dense=True):
    num_imgs = len(img_metas)
    assert len(anchor_list) == len(valid_flag_list) == num_imgs
    num_level_anchors = [anchors.size(0) for anchors in anchor_list[0]]
    for i in range(num_imgs):
        assert len(anchor_list[i]) == len(valid_flag_list[i])
        anchor_list[i] = torch.cat(anchor_list[i])
        valid_flag_list[i] = torch.cat(valid_flag_list[i])
    if gt_bboxes_ignore_list is None:
        gt_bboxes_ignore_list = [None for _ in range(num_imgs)]
    if gt_labels_list is None:
        gt_labels_list = [None for _ in range(num_imgs)]
    if dense and _dense_ok(cfg, sampling, img_metas, gt_bboxes_ignore_list, anchor_list[0], encode_fn):
        labels, label_weights, bbox_targets, bbox_weights, npos = anchor_target_dense(
            anchor_list, gt_bboxes_list, gt_labels_list, cfg)
        split = lambda t: list(torch.split(t, num_level_anchors, dim=1))  # noqa: E731
        return (split(labels), split(label_weights), split(bbox_targets), split(bbox_weights), npos, 0)
    (all_labels, all_label_weights, all_bbox_targets, all_bbox_weights, pos_inds_list, neg_inds_list) = multi_apply(
        anchor_target_single, anchor_list, valid_flag_list, gt_bboxes_list, gt_bboxes_ignore_list, gt_labels_list,
        img_metas, target_means=target_means, target_stds=target_stds, cfg=cfg, label_channels=label_channels,
        sampling=sampling, unmap_outputs=unmap_outputs, encode_fn=encode_fn)
    if any([labels is None for labels in all_labels]):
        return None
    num_total_pos = sum([max(inds.numel(), 1) for inds in pos_inds_list])
    num_total_neg = sum([max(inds.numel(), 1) for inds in neg_inds_list])
    labels_list = images_to_levels(all_labels, num_level_anchors)
    label_weights_list = images_to_levels(all_label_weights, num_level_anchors)
    bbox_targets_list = images_to_levels(all_bbox_targets, num_level_anchors)
    bbox_weights_list = images_to_levels(all_bbox_weights, num_level_anchors)
    return (labels_list, label_weights_list, bbox_targets_list, bbox_weights_list, num_total_pos, num_total_neg)
