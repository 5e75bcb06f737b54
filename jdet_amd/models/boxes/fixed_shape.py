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
    Valid positives come first, then valid negatives (the order of SamplingResult.bboxes, sampler.py:L36-38); when the
    candidates run out the tail rows are invalid (valid False; their index is arbitrary but in range)."""
    pos_idx, pos_valid, neg_idx, neg_valid = sample_fixed(gt_inds, num, pos_fraction, neg_pos_ub, generator)
    idx = torch.cat([pos_idx, neg_idx])
    valid = torch.cat([pos_valid, neg_valid])
    is_pos = torch.cat([torch.ones_like(pos_valid), torch.zeros_like(neg_valid)])
    # stable partition: valid rows first, original (positives-then-negatives) order kept
    order = torch.argsort((~valid).to(torch.int8), stable=True)
    k = min(int(num), idx.numel())
    order = order[:k]
    rows, valid, is_pos = idx[order], valid[order], is_pos[order] & valid[order]
    if k < num:     # fewer candidates than rows asked for: pad (callers size their buffers by `num`)
        pad = num - k
        rows = torch.cat([rows, rows.new_zeros((pad,))])
        valid = torch.cat([valid, valid.new_zeros((pad,))])
        is_pos = torch.cat([is_pos, is_pos.new_zeros((pad,))])
    return rows, valid, is_pos


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
