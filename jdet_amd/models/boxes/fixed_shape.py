"""Sampling with fixed tensor shapes and no host synchronisation.

The reference samplers (python/jdet/models/boxes/sampler.py:L133-233) pick positives / negatives with
`nonzero` + `randperm` + slicing: every step has data-dependent shapes, i.e. a device -> host round trip per image
and per stage, and nothing downstream can be captured in a HIP graph.  The same DISTRIBUTION -- a uniformly random
subset of at most `num * pos_fraction` positives and of `num - #sampled positives` negatives (capped by
`neg_pos_ub`) -- is drawn here with one random key per candidate and two `topk`s; the outputs have fixed lengths and
carry validity masks.  Which particular subset comes out is not pinned in the reference either (`jt.randperm`,
SURVEY 8c): tests check the counts, the class of every sampled index, the absence of duplicates and seeding.
"""
import torch


def _keys(n, device, generator):
    if generator is None:
        return torch.rand((n,), device=device)
    return torch.rand((n,), device=device, generator=generator)


def sample_fixed(gt_inds, num, pos_fraction, neg_pos_ub=-1, generator=None):
    """gt_inds (A,): > 0 positive, 0 negative, < 0 ignored (AssignResult.gt_inds).

    Returns (pos_idx (P,), pos_valid (P,) bool, neg_idx (Nn,), neg_valid (Nn,) bool) with P = min(int(num *
    pos_fraction), A), Nn = min(num, A).  Valid entries are distinct indices of the right class; their numbers are
    min(#pos, P) and min(#neg, num - n_pos [, int(neg_pos_ub * max(1, n_pos))]) -- sampler.py:L100-109."""
    A = gt_inds.numel()
    dev = gt_inds.device
    P = min(int(num * pos_fraction), A)
    Nn = min(int(num), A)
    keys = _keys(A, dev, generator)
    two = torch.full_like(keys, 2.0)
    pk, pos_idx = torch.topk(torch.where(gt_inds > 0, keys, two), P, largest=False)
    pos_valid = pk < 1.5
    n_pos = pos_valid.sum()
    n_neg = num - n_pos
    if neg_pos_ub >= 0:
        n_neg = torch.minimum(n_neg, (neg_pos_ub * torch.clamp(n_pos, min=1)).long())   # int(neg_pos_ub * max(1, n_pos))
    nk, neg_idx = torch.topk(torch.where(gt_inds == 0, keys, two), Nn, largest=False)
    neg_valid = (nk < 1.5) & (torch.arange(Nn, device=dev) < n_neg)
    return pos_idx, pos_valid, neg_idx, neg_valid


def sample_rows(gt_inds, num, pos_fraction, neg_pos_ub=-1, generator=None):
    """The sampled set as exactly `num` rows: (rows (num,) indices into the candidates, valid (num,), is_pos (num,)).
    Valid positives come first, then valid negatives (the order of SamplingResult.bboxes, sampler.py:L36-38), each in
    ASCENDING candidate index -- the reference passes both index lists through `.unique()` (sampler.py:L90, L104), which
    sorts them, so with `add_gt_as_proposals` the sampled gts lead the positives.  When the candidates run out the tail
    rows are invalid (valid False; their index is arbitrary but in range)."""
    pos_idx, pos_valid, neg_idx, neg_valid = sample_fixed(gt_inds, num, pos_fraction, neg_pos_ub, generator)
    A = gt_inds.numel()
    dev = gt_inds.device
    num = int(num)
    # selection masks over the candidates (slot A takes the invalid draws), then a prefix-sum partition: the k-th
    # selected candidate in index order goes to row k -- no device sort
    # (scatter_ with a scalar value: `mask[idx] = 1` copies the scalar from the host -- a synchronising operation)
    sel_p = torch.zeros((A + 1,), dtype=torch.int64, device=dev)
    sel_p.scatter_(0, torch.where(pos_valid, pos_idx, torch.full_like(pos_idx, A)), 1)
    sel_n = torch.zeros((A + 1,), dtype=torch.int64, device=dev)
    sel_n.scatter_(0, torch.where(neg_valid, neg_idx, torch.full_like(neg_idx, A)), 1)
    sel_p, sel_n = sel_p[:A], sel_n[:A]
    n_pos, n_neg = sel_p.sum(), sel_n.sum()
    dest = torch.where(sel_p > 0, torch.cumsum(sel_p, 0) - 1,
                       torch.where(sel_n > 0, n_pos + torch.cumsum(sel_n, 0) - 1, torch.full_like(sel_p, num)))
    rows = torch.zeros((num + 1,), dtype=torch.int64, device=dev)
    rows.scatter_(0, dest.clamp(max=num), torch.arange(A, device=dev))
    k = torch.arange(num, device=dev)
    return rows[:num], k < n_pos + n_neg, k < n_pos


def scatter_rows(dst, idx, valid, values):
    """dst[idx[k]] = values[k] for the valid k only, without boolean indexing: invalid rows are written to a dump
    slot appended to dst (and dropped).  dst (A, ...) is returned updated (a new tensor)."""
    A = dst.shape[0]
    tgt = torch.where(valid, idx, torch.full_like(idx, A))
    buf = torch.cat([dst, dst.new_zeros((1,) + tuple(dst.shape[1:]))], 0)
    if not torch.is_tensor(values):
        values = torch.full((idx.numel(),) + tuple(dst.shape[1:]), values, dtype=dst.dtype, device=dst.device)
    buf[tgt] = values.to(dst.dtype)
    return buf[:A]


def masked_overlaps(iou_calculator, gts, boxes, alive):
    """IoU matrix (K, A) with the columns of dead candidates (padding rows, anchors outside the image) at -1: the
    assigner ignores them and they can neither be sampled nor decide a low-quality match -- the same set the
    reference reaches by removing those candidates first."""
    from jdet_amd.models.boxes.iou_calculator import _HbbOverlaps
    if isinstance(iou_calculator, _HbbOverlaps):
        return iou_calculator(gts, boxes, alive=alive)        # mask fused into the overlap launch
    overlaps = iou_calculator(gts, boxes)
    return torch.where(alive[None, :], overlaps, torch.full_like(overlaps, -1.0))


def dense_anchor_targets(anchors, inside, gts_for_assign, gts_for_encode, assigner, sampler, encode, reg_dim,
                         background_label, pos_weight):
    """RPN targets of one image, dense over ALL anchors (anchor_target.py:L96-160 with the index lists replaced by
    masks): labels (A,) long [1 on sampled positives, `background_label` elsewhere], label_weights (A,), bbox_targets
    (A, reg_dim), bbox_weights (A, reg_dim), number of sampled positives / negatives (0-d device tensors).
    `encode(anchors (P, .), gts (P, .))` is the head's coder."""
    assign = assigner.assign_wrt_overlaps(masked_overlaps(assigner.iou_calculator, gts_for_assign, anchors, inside),
                                          None)
    pos_idx, pos_valid, neg_idx, neg_valid = sample_fixed(assign.gt_inds, sampler.num, sampler.pos_fraction,
                                                          sampler.neg_pos_ub)
    matched = (assign.gt_inds[pos_idx].long() - 1).clamp(min=0)
    pos_targets = encode(anchors[pos_idx], gts_for_encode[matched])
    A, dev = anchors.shape[0], anchors.device
    labels = scatter_rows(torch.full((A,), background_label, dtype=torch.long, device=dev), pos_idx, pos_valid, 1)
    pw = 1.0 if pos_weight <= 0 else pos_weight
    label_weights = scatter_rows(torch.zeros((A,), device=dev), pos_idx, pos_valid, pw)
    label_weights = scatter_rows(label_weights, neg_idx, neg_valid, 1.0)
    bbox_targets = scatter_rows(torch.zeros((A, reg_dim), device=dev), pos_idx, pos_valid, pos_targets)
    bbox_weights = scatter_rows(torch.zeros((A, reg_dim), device=dev), pos_idx, pos_valid, 1.0)
    return labels, label_weights, bbox_targets, bbox_weights, pos_valid.sum(), neg_valid.sum()


def proposal_table(boxes, scores, level_ids, level_sizes, alive, nms_thresh, nms_post_per_level, rows,
                   nms_across_levels=False, invalid_score=-1.0, payload=None):
    """Candidates of all levels (each level's slice sorted by descending score) -> a table of exactly `rows` rows
    [box..., score] sorted by score; rows beyond the survivors carry `invalid_score`.  Per-level NMS is ONE launch
    (the level id as label); `nms_post_per_level` caps the survivors of each level (None: no cap); with
    `nms_across_levels` a second, label-free pass runs over the survivors.  `boxes`: horizontal (x1,y1,x2,y2), what
    the NMS sees; `payload` (default: the boxes): what the table rows carry."""
    from jdet_amd.ops.nms import nms_keep_mask
    low = torch.full_like(scores, -2.0)
    # The candidates arrive level by level, descending score inside a level: that IS the visiting order (no device
    # sort).  Dropped boxes (too small) are shrunk to a point far outside the image: they overlap nothing, so they
    # suppress nothing that is kept (the reference removes them before the NMS), and are removed below.
    nowhere = boxes.new_full((4,), -1.0e4)
    keep, _ = nms_keep_mask(torch.where(alive[:, None], boxes, nowhere[None, :]), scores, nms_thresh,
                            labels=level_ids, n_labels=len(level_sizes),
                            visit_order=torch.arange(scores.shape[0], device=scores.device))
    ok = keep & alive
    if nms_post_per_level is not None:
        ranks, start = [], 0
        for n in level_sizes:
            ranks.append(torch.cumsum(ok[start:start + n].to(torch.int32), 0))
            start += n
        ok = ok & (torch.cat(ranks) <= nms_post_per_level)
    if nms_across_levels:
        keep2, _ = nms_keep_mask(boxes, torch.where(ok, scores, low), nms_thresh)
        ok = ok & keep2
    ranked = torch.where(ok, scores, torch.full_like(scores, invalid_score))
    k = min(rows, ranked.shape[0])
    top_scores, top = torch.topk(ranked, k)
    table = torch.cat([(boxes if payload is None else payload)[top], top_scores[:, None]], dim=1)
    if k < rows:
        pad = table.new_zeros((rows - k, table.shape[1]))
        pad[:, -1] = invalid_score
        table = torch.cat([table, pad])
    return table


class StageRows:
    """The sampled rows of one image for one R-CNN stage, always `num` of them (SamplingResult of the reference,
    sampler.py:L6-38, with masks instead of index lists): boxes (num, D) [invalid rows: a small dummy box],
    valid / is_pos / is_gt (num,) bool, labels (num,) long [background 0 off the positives], matched (num,) long
    [index of the assigned gt, 0 off the positives]."""

    def __init__(self, boxes, valid, is_pos, is_gt, labels, matched):
        self.boxes, self.valid, self.is_pos, self.is_gt, self.labels, self.matched = \
            boxes, valid, is_pos, is_gt, labels, matched


def sample_stage_rows(cands, alive, gts, gt_labels, assigner, sampler, dummy_box, background_label=0):
    """assign + sample one image's candidates (P, D) against its gts (K, D) with fixed shapes; `alive` (P,) marks
    the real candidates.  `sampler.add_gt_as_proposals`: the gts join the candidates, matched to themselves."""
    assign = assigner.assign_wrt_overlaps(masked_overlaps(assigner.iou_calculator, gts, cands, alive), gt_labels)
    gt_inds, labels = assign.gt_inds.long(), assign.labels.long()
    is_gt = torch.zeros_like(alive)
    boxes = cands
    if sampler.add_gt_as_proposals:
        k = gts.shape[0]
        boxes = torch.cat([gts.to(cands.dtype), cands])
        gt_inds = torch.cat([torch.arange(1, k + 1, device=gts.device), gt_inds])
        labels = torch.cat([gt_labels.long(), labels])
        is_gt = torch.cat([torch.ones((k,), dtype=torch.bool, device=gts.device), is_gt])
    rows, valid, is_pos = sample_rows(gt_inds, sampler.num, sampler.pos_fraction, sampler.neg_pos_ub)
    sel = torch.where(valid[:, None], boxes[rows], dummy_box[None, :])
    matched = torch.where(is_pos, gt_inds[rows] - 1, torch.zeros_like(rows))
    row_labels = torch.where(is_pos, labels[rows], torch.full_like(rows, background_label))
    return StageRows(sel, valid, is_pos, is_gt[rows] & valid, row_labels, matched)
