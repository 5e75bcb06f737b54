"""S2ANet head (FAM + AlignConv + ODM).  Mirrors python/jdet/models/roi_heads/s2anet_head.py:
`S2ANetHead` L20-629, `bbox_decode` L631-654, `AlignConv` L657-723.

Layer names, channel plan (or_conv 256 -> 32x8, odm_cls tower fed by the 32-channel
orientation-pooled map), initialisation, target dict keys and the loss bookkeeping follow the
reference; the device work underneath is this repo's HIP path: DeformConv sampling, ARF gather,
rotated IoU, fused max-IoU assignment, fused delta codec, rotated NMS.
"""
import os

import torch
from torch import nn
try:    # what torch.func.functional_call is built on -- a private name whose signature has moved between releases:
    from torch.nn.utils.stateless import _reparametrize_module      # without it the packed branch stays on the main stream
except ImportError:      # pragma: no cover
    _reparametrize_module = None

from jdet_amd.models.boxes.anchor_generator import AnchorGeneratorRotatedS2ANet
from jdet_amd.models.boxes.anchor_target import anchor_target, images_to_levels
from jdet_amd.models.boxes.box_ops import delta2bbox_rotated
from jdet_amd.models.utils.level_pack import LevelPack

# Packed small levels on a side stream next to the big levels.  Default OFF since round 6: the parameter aliases the side
# branch needs cost 26 gradient-accumulation launches per step, and with the head's glue gone the overlap no longer pays
# for them (same-box A/B, 3 pairs: 27.26 ms off / 27.37 ms on; profiles/r06_glue.md).  JDET_HEAD_STREAMS=1 switches it on.
HEAD_STREAMS = os.environ.get("JDET_HEAD_STREAMS", "0") == "1"
# the pack's gap mask inside the tower convs' epilogue instead of a multiplication per layer and direction (A/B switch)
FUSED_PACK_MASK = os.environ.get("JDET_PACK_FUSED_MASK", "1") == "1"
_SIDE = {}


def _side_stream(device):
    s = _SIDE.get(device)
    if s is None:
        s = _SIDE[device] = torch.cuda.Stream(device)
    return s
from jdet_amd.models.utils.modules import ConvModule
from jdet_amd.ops.conv_igemm import conv_module
from jdet_amd.models.utils.weight_init import bias_init_with_prob, normal_init
from jdet_amd.ops.dcn_v1 import DeformConv
from jdet_amd.ops.orn import ORConv2d, RotationInvariantPooling
from jdet_amd.utils.general import multi_apply
from jdet_amd.utils.registry import HEADS, LOSSES, build_from_cfg

from ._anchor_head_common import RotatedAnchorHeadMixin


class _AttrDict(dict):
    """dict with Config-like attribute access (missing -> None), for the default train/test cfgs"""

    def __getattr__(self, name):
        return self.get(name)

    def copy(self):
        return _AttrDict(self)


def _cfg(d):
    if type(d) is dict:  # plain dicts (the signature defaults); Config objects already have attribute access
        return _AttrDict({k: _cfg(v) if isinstance(v, dict) else v for k, v in d.items()})
    return d


_DEFAULT_ASSIGN = dict(
    assigner=dict(type="MaxIoUAssigner", pos_iou_thr=0.5, neg_iou_thr=0.4, min_pos_iou=0, ignore_iof_thr=-1,
                  iou_calculator=dict(type="BboxOverlaps2D_rotated")),
    bbox_coder=dict(type="DeltaXYWHABBoxCoder", target_means=(0., 0., 0., 0., 0.), target_stds=(1., 1., 1., 1., 1.),
                    clip_border=True),
    allowed_border=-1, pos_weight=-1, debug=False)


@HEADS.register_module()
class S2ANetHead(RotatedAnchorHeadMixin, nn.Module):
    def __init__(self, num_classes, in_channels, feat_channels=256, stacked_convs=2, with_orconv=True,
                 anchor_scales=[4], anchor_ratios=[1.0], anchor_strides=[8, 16, 32, 64, 128], anchor_base_sizes=None,
                 target_means=(.0, .0, .0, .0, .0), target_stds=(1.0, 1.0, 1.0, 1.0, 1.0),
                 loss_fam_cls=dict(type="FocalLoss", use_sigmoid=True, gamma=2.0, alpha=0.25, loss_weight=1.0),
                 loss_fam_bbox=dict(type="SmoothL1Loss", beta=1.0 / 9.0, loss_weight=1.0),
                 loss_odm_cls=dict(type="FocalLoss", use_sigmoid=True, gamma=2.0, alpha=0.25, loss_weight=1.0),
                 loss_odm_bbox=dict(type="SmoothL1Loss", beta=1.0 / 9.0, loss_weight=1.0),
                 test_cfg=dict(nms_pre=2000, min_bbox_size=0, score_thr=0.05,
                               nms=dict(type="nms_rotated", iou_thr=0.1), max_per_img=2000),
                 train_cfg=dict(fam_cfg=_DEFAULT_ASSIGN, odm_cfg=_DEFAULT_ASSIGN)):
        super().__init__()
        self.num_classes = num_classes
        self.in_channels = in_channels
        self.feat_channels = feat_channels
        self.stacked_convs = stacked_convs
        self.with_orconv = with_orconv
        self.anchor_scales = anchor_scales
        self.anchor_ratios = anchor_ratios
        self.anchor_strides = list(anchor_strides)
        self.anchor_base_sizes = list(anchor_strides) if anchor_base_sizes is None else anchor_base_sizes
        self.target_means = target_means
        self.target_stds = target_stds
        self.use_sigmoid_cls = loss_odm_cls.get("use_sigmoid", False)
        self.sampling = loss_odm_cls["type"] not in ["FocalLoss", "GHMC"]
        self.cls_out_channels = num_classes - 1 if self.use_sigmoid_cls else num_classes
        if self.cls_out_channels <= 0:
            raise ValueError("num_classes={} is too small".format(num_classes))
        self.loss_fam_cls = build_from_cfg(loss_fam_cls, LOSSES)
        self.loss_fam_bbox = build_from_cfg(loss_fam_bbox, LOSSES)
        self.loss_odm_cls = build_from_cfg(loss_odm_cls, LOSSES)
        self.loss_odm_bbox = build_from_cfg(loss_odm_bbox, LOSSES)
        self.train_cfg = _cfg(train_cfg)
        self.test_cfg = _cfg(test_cfg)
        self.anchor_generators = [AnchorGeneratorRotatedS2ANet(b, anchor_scales, anchor_ratios)
                                  for b in self.anchor_base_sizes]
        self.base_anchors = dict()   # anchor cache, keyed by (level, featmap size, device)
        # levels of at most this many positions run their conv towers as ONE packed tensor (see LevelPack)
        # (round 4, 2-D placement: 64^2 + 32^2 + 16^2 + 8^2 in 64 x 97 -- with P4 in the pack the step is 0.18 ms shorter
        #  than with the three small levels alone, 29.16 vs 29.34 ms; all five levels in one tensor: 31.0 ms)
        self.pack_max_positions = int(os.environ.get("JDET_PACK_MAX_POS", "4096"))
        self._init_layers()

    def _init_layers(self):
        self.relu = nn.ReLU()
        self.fam_reg_convs = nn.ModuleList()
        self.fam_cls_convs = nn.ModuleList()
        for i in range(self.stacked_convs):
            chn = self.in_channels if i == 0 else self.feat_channels
            self.fam_reg_convs.append(ConvModule(chn, self.feat_channels, 3, stride=1, padding=1))
            self.fam_cls_convs.append(ConvModule(chn, self.feat_channels, 3, stride=1, padding=1))
        self.fam_reg = nn.Conv2d(self.feat_channels, 5, 1)
        self.fam_cls = nn.Conv2d(self.feat_channels, self.cls_out_channels, 1)
        self.align_conv = AlignConv(self.feat_channels, self.feat_channels, kernel_size=3)
        if self.with_orconv:
            self.or_conv = ORConv2d(self.feat_channels, int(self.feat_channels / 8), kernel_size=3, padding=1,
                                    arf_config=(1, 8))
        else:
            self.or_conv = nn.Conv2d(self.feat_channels, self.feat_channels, 3, padding=1)
        self.or_pool = RotationInvariantPooling(256, 8)
        self.odm_reg_convs = nn.ModuleList()
        self.odm_cls_convs = nn.ModuleList()
        for i in range(self.stacked_convs):
            chn = int(self.feat_channels / 8) if i == 0 and self.with_orconv else self.feat_channels
            self.odm_reg_convs.append(ConvModule(self.feat_channels, self.feat_channels, 3, stride=1, padding=1))
            self.odm_cls_convs.append(ConvModule(chn, self.feat_channels, 3, stride=1, padding=1))
        self.odm_cls = nn.Conv2d(self.feat_channels, self.cls_out_channels, 3, padding=1)
        self.odm_reg = nn.Conv2d(self.feat_channels, 5, 3, padding=1)
        self.init_weights()

    def init_weights(self):
        for m in self.fam_reg_convs:
            normal_init(m.conv, std=0.01)
        for m in self.fam_cls_convs:
            normal_init(m.conv, std=0.01)
        bias_cls = bias_init_with_prob(0.01)
        normal_init(self.fam_reg, std=0.01)
        normal_init(self.fam_cls, std=0.01, bias=bias_cls)
        self.align_conv.init_weights()
        normal_init(self.or_conv, std=0.01)
        for m in self.odm_reg_convs:
            normal_init(m.conv, std=0.01)
        for m in self.odm_cls_convs:
            normal_init(m.conv, std=0.01)
        normal_init(self.odm_cls, std=0.01, bias=bias_cls)
        normal_init(self.odm_reg, std=0.01)

    # ------------------------------------------------------------------ forward
    def _towers(self, x, convs, mask=None, rows=None):
        """rows: the pack's mask per position of the batch, flat -- the fused conv multiplies it into its finished rows
        (ConvModule.masked); layers it does not apply to are followed by the multiplication"""
        for conv in convs:
            y = conv.masked(x, rows) if rows is not None and hasattr(conv, "masked") else None
            if y is not None:
                x = y
                continue
            x = conv(x)
            if mask is not None:
                x = x * mask       # the gaps of a packed tensor stay zero: they are the zero padding of each level
        return x

    def _fam(self, x, mask=None, rows=None):
        # (prediction layers through conv_module: library forward, bias gradient by the own any-C column sum)
        fam_bbox_pred = conv_module(self.fam_reg, self._towers(x, self.fam_reg_convs, mask, rows))
        # the FAM classification tower only runs in training (L213-220)
        fam_cls_score = conv_module(self.fam_cls, self._towers(x, self.fam_cls_convs, mask, rows)) if self.training else None
        return fam_cls_score, fam_bbox_pred

    def _refine(self, x, fam_bbox_pred, stride):
        num_level = self.anchor_strides.index(stride)
        init_anchors = self._init_anchors(num_level, tuple(fam_bbox_pred.shape[-2:]), x.device)
        refine_anchor = bbox_decode(fam_bbox_pred.detach(), init_anchors, self.target_means, self.target_stds)
        return refine_anchor, self.align_conv(x, refine_anchor, stride)     # (read-only there: no copy needed)

    def _odm(self, align_feat, mask=None, rows=None):
        or_feat = self.or_conv(align_feat)
        if mask is not None:
            or_feat = or_feat * mask
        odm_cls_feat = self.or_pool(or_feat) if self.with_orconv else or_feat
        odm_cls_score = conv_module(self.odm_cls, self._towers(odm_cls_feat, self.odm_cls_convs, mask, rows))
        odm_bbox_pred = conv_module(self.odm_reg, self._towers(or_feat, self.odm_reg_convs, mask, rows))
        return odm_cls_score, odm_bbox_pred

    def forward_single(self, x, stride):
        fam_cls_score, fam_bbox_pred = self._fam(x)
        refine_anchor, align_feat = self._refine(x, fam_bbox_pred, stride)
        odm_cls_score, odm_bbox_pred = self._odm(align_feat)
        return fam_cls_score, fam_bbox_pred, refine_anchor, odm_cls_score, odm_bbox_pred

    def forward_packed(self, xs, strides):
        """The same five outputs per level for a group of SMALL levels (P5-P7 of a 1024 tile: 32x32, 16x16, 8x8),
        with every convolution of the FAM / ODM towers run once on a packed tensor instead of once per level.  The
        towers share their weights across levels, and on these maps a 3x3 conv over 256 channels is latency bound
        (a 2304-deep reduction for a handful of output tiles: ~48 us each whatever the size); one launch for the
        three levels costs about what one of them did, and the weight gradients need no per-level accumulation.
        AlignConv (per-level anchors / offsets) stays per level.  Equal to the per-level path up to the library's
        accumulation order (tests/test_gpu_s2anet.py)."""
        pack = LevelPack.cached([tuple(x.shape[-2:]) for x in xs], xs[0].device)
        mask = pack.mask.to(xs[0].dtype)
        rows = pack.row_mask(xs[0].shape[0]) if FUSED_PACK_MASK and xs[0].dtype == torch.float32 else None
        fam_cls_p, fam_box_p = self._fam(pack.pack(xs), mask, rows)
        fam_box = pack.unpack(fam_box_p)
        fam_cls = pack.unpack(fam_cls_p) if fam_cls_p is not None else [None] * len(xs)
        refined = [self._refine(x, b, stride) for x, b, stride in zip(xs, fam_box, strides)]
        odm_cls_p, odm_box_p = self._odm(pack.pack([r[1] for r in refined]), mask, rows)
        odm_cls, odm_box = pack.unpack(odm_cls_p), pack.unpack(odm_box_p)
        return [(fam_cls[i], fam_box[i], refined[i][0], odm_cls[i], odm_box[i]) for i in range(len(xs))]

    def get_refine_anchors(self, featmap_sizes, refine_anchors, img_metas, is_train=True, device=None):
        num_levels = len(featmap_sizes)
        refine_anchors_list = []
        for img_id in range(len(img_metas)):
            refine_anchors_list.append([refine_anchors[i][img_id].reshape(-1, 5) for i in range(num_levels)])
        valid_flag_list = self._valid_flags(featmap_sizes, img_metas, device) if is_train else []
        return refine_anchors_list, valid_flag_list

    # ------------------------------------------------------------------ loss
    def loss(self, fam_cls_scores, fam_bbox_preds, refine_anchors, odm_cls_scores, odm_bbox_preds, gt_bboxes,
             gt_labels, img_metas, gt_bboxes_ignore=None):
        cfg = self.train_cfg.copy()
        featmap_sizes = [tuple(featmap.shape[-2:]) for featmap in odm_cls_scores]
        assert len(featmap_sizes) == len(self.anchor_generators)
        device = odm_cls_scores[0].device
        anchor_list, valid_flag_list = self.get_init_anchors(featmap_sizes, img_metas, device)
        num_level_anchors = [anchors.size(0) for anchors in anchor_list[0]]
        # per-level anchors are only read by the loss when it regresses decoded boxes
        need_anchors = cfg.fam_cfg.get("reg_decoded_bbox", False) or cfg.odm_cfg.get("reg_decoded_bbox", False)
        all_anchor_list = [None] * len(num_level_anchors)
        if need_anchors:
            concat_anchor_list = [torch.cat(anchor_list[i]) for i in range(len(anchor_list))]
            all_anchor_list = images_to_levels(concat_anchor_list, num_level_anchors)

        label_channels = self.cls_out_channels if self.use_sigmoid_cls else 1
        cls_reg_targets = anchor_target(anchor_list, valid_flag_list, gt_bboxes, img_metas, self.target_means,
                                        self.target_stds, cfg.fam_cfg, gt_bboxes_ignore_list=gt_bboxes_ignore,
                                        gt_labels_list=gt_labels, label_channels=label_channels,
                                        sampling=self.sampling)
        if cls_reg_targets is None:
            return None
        labels_list, label_weights_list, bbox_targets_list, bbox_weights_list, num_total_pos, num_total_neg = \
            cls_reg_targets
        num_total_samples = num_total_pos + num_total_neg if self.sampling else num_total_pos
        losses_fam_cls, losses_fam_bbox = multi_apply(
            self.loss_fam_single, fam_cls_scores, fam_bbox_preds, all_anchor_list, labels_list, label_weights_list,
            bbox_targets_list, bbox_weights_list, num_total_samples=num_total_samples, cfg=cfg.fam_cfg)

        refine_anchors_list, valid_flag_list = self.get_refine_anchors(featmap_sizes, refine_anchors, img_metas,
                                                                       device=device)
        num_level_anchors = [anchors.size(0) for anchors in refine_anchors_list[0]]
        all_anchor_list = [None] * len(num_level_anchors)
        if need_anchors:
            concat_anchor_list = [torch.cat(refine_anchors_list[i]) for i in range(len(refine_anchors_list))]
            all_anchor_list = images_to_levels(concat_anchor_list, num_level_anchors)
        cls_reg_targets = anchor_target(refine_anchors_list, valid_flag_list, gt_bboxes, img_metas, self.target_means,
                                        self.target_stds, cfg.odm_cfg, gt_bboxes_ignore_list=gt_bboxes_ignore,
                                        gt_labels_list=gt_labels, label_channels=label_channels,
                                        sampling=self.sampling)
        if cls_reg_targets is None:
            return None
        (labels_list, label_weights_list, bbox_targets_list, bbox_weights_list, num_total_pos, num_total_neg) = \
            cls_reg_targets
        num_total_samples = num_total_pos + num_total_neg if self.sampling else num_total_pos
        losses_odm_cls, losses_odm_bbox = multi_apply(
            self.loss_odm_single, odm_cls_scores, odm_bbox_preds, all_anchor_list, labels_list, label_weights_list,
            bbox_targets_list, bbox_weights_list, num_total_samples=num_total_samples, cfg=cfg.odm_cfg)
        return dict(loss_fam_cls=losses_fam_cls, loss_fam_bbox=losses_fam_bbox, loss_odm_cls=losses_odm_cls,
                    loss_odm_bbox=losses_odm_bbox)

    def loss_fam_single(self, fam_cls_score, fam_bbox_pred, anchors, labels, label_weights, bbox_targets,
                        bbox_weights, num_total_samples, cfg):
        return self._loss_single(self.loss_fam_cls, self.loss_fam_bbox, fam_cls_score, fam_bbox_pred, anchors, labels,
                                 label_weights, bbox_targets, bbox_weights, num_total_samples, cfg)

    def loss_odm_single(self, odm_cls_score, odm_bbox_pred, anchors, labels, label_weights, bbox_targets,
                        bbox_weights, num_total_samples, cfg):
        return self._loss_single(self.loss_odm_cls, self.loss_odm_bbox, odm_cls_score, odm_bbox_pred, anchors, labels,
                                 label_weights, bbox_targets, bbox_weights, num_total_samples, cfg)

    # ------------------------------------------------------------------ inference
    def get_bboxes(self, fam_cls_scores, fam_bbox_preds, refine_anchors, odm_cls_scores, odm_bbox_preds, img_metas,
                   rescale=True):
        assert len(odm_cls_scores) == len(odm_bbox_preds)
        cfg = self.test_cfg.copy()
        featmap_sizes = [tuple(featmap.shape[-2:]) for featmap in odm_cls_scores]
        num_levels = len(odm_cls_scores)
        refine_anchors = self.get_refine_anchors(featmap_sizes, refine_anchors, img_metas, is_train=False)
        result_list = []
        for img_id in range(len(img_metas)):
            cls_score_list = [odm_cls_scores[i][img_id].detach() for i in range(num_levels)]
            bbox_pred_list = [odm_bbox_preds[i][img_id].detach() for i in range(num_levels)]
            img_shape = img_metas[img_id]["img_shape"]
            scale_factor = img_metas[img_id]["scale_factor"]
            result_list.append(self.get_bboxes_single(cls_score_list, bbox_pred_list, refine_anchors[0][img_id],
                                                      img_shape, scale_factor, cfg, rescale))
        return result_list

    def _level_outputs(self, feats):
        """per level (fam_cls_score, fam_bbox_pred, refine_anchor, odm_cls_score, odm_bbox_pred); the small levels
        go through forward_packed together"""
        small = [i for i, f in enumerate(feats)
                 if f.is_cuda and f.shape[-2] * f.shape[-1] <= self.pack_max_positions]
        outs = [None] * len(feats)
        side = None
        if len(small) >= 2:
            # The packed small levels and the big levels are independent until the losses, and the packed tower's
            # launches leave most of the chip idle (57 x 32 positions: 58 output tiles for 256 CUs): they run on a side
            # stream next to the big levels' (autograd replays each branch's backward on its forward stream, so the
            # backward overlaps the same way).  Not under graph capture (one capture stream), JDET_HEAD_STREAMS=0: off.
            if (HEAD_STREAMS and _reparametrize_module is not None and len(small) < len(feats)
                    and torch.is_grad_enabled() and not torch.cuda.is_current_stream_capturing()):
                side = _side_stream(feats[0].device)
                main = torch.cuda.current_stream(feats[0].device)
                side.wait_stream(main)
            if side is not None:
                # The towers' weights are shared by both branches.  The side branch reads them through ALIASES made
                # here, on the main stream: the gradients of the side branch then reach every parameter through a view
                # node that lives on the main stream, where its AccumulateGrad node (and DDP's bucket hook) lives too --
                # the autograd engine orders side -> main at that node like at any other cross-stream edge.  Reading
                # the parameters directly on the side stream made their AccumulateGrad nodes' stream disagree with one
                # of the two producers ("AccumulateGrad node's stream does not match ..." in the round-4 bench stderr).
                alias = {n: p.view_as(p) for n, p in self.named_parameters() if p.requires_grad}
                with torch.cuda.stream(side), _reparametrize_module(self, alias):
                    packed = self.forward_packed([feats[i] for i in small], [self.anchor_strides[i] for i in small])
            else:
                packed = self.forward_packed([feats[i] for i in small], [self.anchor_strides[i] for i in small])
            for i, o in zip(small, packed):
                outs[i] = o
        for i, f in enumerate(feats):
            if outs[i] is None:
                outs[i] = self.forward_single(f, self.anchor_strides[i])
        if side is not None:
            main.wait_stream(side)
            for i in small:                       # made on the side stream, consumed (and freed) on the main one
                feats[i].record_stream(side)
                for t in outs[i]:
                    if torch.is_tensor(t):
                        t.record_stream(main)
        return tuple(map(list, zip(*outs)))

    def forward(self, feats, targets):
        outs = self._level_outputs(feats)
        if self.training:
            return self.loss(*outs, *self.parse_targets(targets))
        return self.get_bboxes(*outs, self.parse_targets(targets, is_train=False))

    execute = forward


def bbox_decode(bbox_preds, anchors, means=[0, 0, 0, 0, 0], stds=[1, 1, 1, 1, 1]):
    """bbox_preds (N,5,H,W), anchors (H*W,5) -> (N,H,W,5) refined anchors (s2anet_head.py:L631-654;
    wh_ratio_clip = 1e-6).  All images in one fused decode launch."""
    num_imgs, _, H, W = bbox_preds.shape
    deltas = bbox_preds.permute(0, 2, 3, 1).reshape(num_imgs * H * W, 5)
    rois = anchors.unsqueeze(0).expand(num_imgs, H * W, 5).reshape(num_imgs * H * W, 5)
    bboxes = delta2bbox_rotated(rois, deltas, means, stds, wh_ratio_clip=1e-6)
    return bboxes.reshape(num_imgs, H, W, 5)


class AlignConv(nn.Module):
    def __init__(self, in_channels, out_channels, kernel_size=3, deformable_groups=1):
        super().__init__()
        self.kernel_size = kernel_size
        self.deform_conv = DeformConv(in_channels, out_channels, kernel_size=kernel_size,
                                      padding=(kernel_size - 1) // 2, deformable_groups=deformable_groups)
        self.relu = nn.ReLU()

    def init_weights(self):
        normal_init(self.deform_conv, std=0.01)

    @torch.no_grad()
    def get_offset(self, anchors, featmap_size, stride):
        """anchors (N, H*W, 5) -> offsets (N, 2*k*k, H, W), (dy,dx) per tap: the k x k sampling grid of the
        refined anchor minus the regular conv grid (s2anet_head.py:L676-713), batched over images."""
        dtype, device = anchors.dtype, anchors.device
        feat_h, feat_w = featmap_size
        if anchors.is_cuda and dtype == torch.float32:   # one fused launch (csrc/loss_offset.hip)
            from jdet_amd import _lib as L
            a = anchors.contiguous()
            n = a.shape[0]
            out = torch.empty((n, 2 * self.kernel_size ** 2, feat_h, feat_w), dtype=dtype, device=device)
            L.check(L.lib().jdet_align_conv_offset(L.ptr(a), n, feat_h, feat_w, float(stride), self.kernel_size,
                                                   L.ptr(out), L.stream_ptr(a)), "jdet_align_conv_offset")
            return out
        pad = (self.kernel_size - 1) // 2
        idx = torch.arange(-pad, pad + 1, dtype=dtype, device=device)
        yy, xx = torch.meshgrid(idx, idx, indexing="ij")
        xx, yy = xx.reshape(-1), yy.reshape(-1)
        xc = torch.arange(0, feat_w, dtype=dtype, device=device)
        yc = torch.arange(0, feat_h, dtype=dtype, device=device)
        yc, xc = torch.meshgrid(yc, xc, indexing="ij")
        xc, yc = xc.reshape(-1), yc.reshape(-1)
        x_conv = xc[:, None] + xx
        y_conv = yc[:, None] + yy
        x_ctr, y_ctr, w, h, a = torch.unbind(anchors, dim=-1)      # each (N, HW)
        x_ctr, y_ctr, w, h = x_ctr / stride, y_ctr / stride, w / stride, h / stride
        cos, sin = torch.cos(a), torch.sin(a)
        dw, dh = w / self.kernel_size, h / self.kernel_size
        x, y = dw[..., None] * xx, dh[..., None] * yy               # (N, HW, kk)
        xr = cos[..., None] * x - sin[..., None] * y
        yr = sin[..., None] * x + cos[..., None] * y
        x_anchor, y_anchor = xr + x_ctr[..., None], yr + y_ctr[..., None]
        offset_x = x_anchor - x_conv
        offset_y = y_anchor - y_conv
        offset = torch.stack([offset_y, offset_x], dim=-1)          # (N, HW, kk, 2)
        n = anchors.shape[0]
        return offset.reshape(n, feat_h * feat_w, -1).permute(0, 2, 1).reshape(n, -1, feat_h, feat_w)

    def forward(self, x, anchors, stride):
        num_imgs, H, W = anchors.shape[:3]
        offset_tensor = self.get_offset(anchors.reshape(num_imgs, H * W, 5), (H, W), stride)
        return self.relu(self.deform_conv(x, offset_tensor))

    execute = forward
