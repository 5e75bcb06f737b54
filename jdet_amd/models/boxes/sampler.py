"""Samplers.  Mirrors python/jdet/models/boxes/sampler.py: `SamplingResult` L6-38, `BaseSampler`
L41-111, `PseudoSampler` L114-131, `RandomSampler` L133-177, `RandomSamplerRotated` L179-233."""
import torch

from jdet_amd.utils.registry import BOXES


class SamplingResult:
    def __init__(self, pos_inds, neg_inds, bboxes, gt_bboxes, assign_result, gt_flags):
        self.pos_inds = pos_inds
        self.neg_inds = neg_inds
        self.pos_bboxes = bboxes[pos_inds]
        self.neg_bboxes = bboxes[neg_inds]
        self.pos_is_gt = gt_flags[pos_inds]
        self.num_gts = gt_bboxes.shape[0]
        self.pos_assigned_gt_inds = assign_result.gt_inds[pos_inds].long() - 1
        box_dim = gt_bboxes.shape[-1] if gt_bboxes.dim() == 2 else 4
        if gt_bboxes.numel() == 0:
            assert self.pos_assigned_gt_inds.numel() == 0
            self.pos_gt_bboxes = gt_bboxes.new_empty((0, box_dim))
        else:
            if gt_bboxes.dim() < 2:
                gt_bboxes = gt_bboxes.view(-1, box_dim)
            self.pos_gt_bboxes = gt_bboxes[self.pos_assigned_gt_inds, :]
        self.pos_gt_labels = assign_result.labels[pos_inds] if assign_result.labels is not None else None

    @property
    def bboxes(self):
        return torch.cat([self.pos_bboxes, self.neg_bboxes])


class BaseSampler:
    """The list-shaped sampler interface of the reference (`sampler.sample(assign_result, bboxes, gt_bboxes,
    gt_labels)` -> SamplingResult with index lists; sampler.py:L41-111) for callers outside the fixed-shape train paths
    (`anchor_target(..., sampling=True)`).  The draw itself is `fixed_shape.sample_fixed` -- the one sampling
    implementation of this package: one random key per candidate, two top-k, the reference's counts -- compacted to
    ascending index lists here (what the reference's `.unique()` returns)."""
    box_dim = 4

    def __init__(self, num, pos_fraction, neg_pos_ub=-1, add_gt_as_proposals=True, **kwargs):
        self.num = num
        self.pos_fraction = pos_fraction
        self.neg_pos_ub = neg_pos_ub
        self.add_gt_as_proposals = add_gt_as_proposals

    def sample(self, assign_result, bboxes, gt_bboxes, gt_labels=None, generator=None, **kwargs):
        from .fixed_shape import sample_fixed
        gt_bboxes = gt_bboxes.to(bboxes.dtype)
        bboxes = bboxes.reshape(-1, bboxes.shape[-1])[:, :self.box_dim]
        is_gt = torch.zeros((bboxes.shape[0],), dtype=torch.bool, device=bboxes.device)
        if self.add_gt_as_proposals:            # the gts join the candidates, matched to themselves (L92-99)
            assign_result.add_gt_(gt_labels)
            is_gt = torch.cat([is_gt.new_ones((gt_bboxes.shape[0],)), is_gt])
            bboxes = torch.cat([gt_bboxes, bboxes], dim=0)
        pos, pos_ok, neg, neg_ok = sample_fixed(assign_result.gt_inds, self.num, self.pos_fraction, self.neg_pos_ub,
                                                generator)
        return SamplingResult(pos[pos_ok].sort().values, neg[neg_ok].sort().values, bboxes, gt_bboxes, assign_result,
                              is_gt)


@BOXES.register_module()
class PseudoSampler(BaseSampler):
    def __init__(self, **kwargs):
        pass

    def sample(self, assign_result, bboxes, gt_bboxes, **kwargs):
        # nonzero returns ascending unique indices already (= the reference's `.unique()` result)
        pos_inds = torch.nonzero(assign_result.gt_inds > 0)[:, 0]
        neg_inds = torch.nonzero(assign_result.gt_inds == 0)[:, 0]
        gt_flags = torch.zeros((bboxes.shape[0],), dtype=torch.bool, device=bboxes.device)
        return SamplingResult(pos_inds, neg_inds, bboxes, gt_bboxes, assign_result, gt_flags)


@BOXES.register_module()
class RandomSampler(BaseSampler):
    def __init__(self, num, pos_fraction, neg_pos_ub=-1, add_gt_as_proposals=True, **kwargs):
        super().__init__(num, pos_fraction, neg_pos_ub, add_gt_as_proposals)


@BOXES.register_module()
class RandomSamplerRotated(RandomSampler):
    """identical to RandomSampler but slices 5 box columns (sampler.py:L203)"""
    box_dim = 5
