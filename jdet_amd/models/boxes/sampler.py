"""Samplers.  Mirrors python/jdet/models/boxes/sampler.py: `SamplingResult` L6-38, `BaseSampler`
L41-111, `PseudoSampler` L114-131, `RandomSampler` L133-177, `RandomSamplerRotated` L179-233."""
from abc import ABCMeta, abstractmethod

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


class BaseSampler(metaclass=ABCMeta):
    box_dim = 4

    def __init__(self, num, pos_fraction, neg_pos_ub=-1, add_gt_as_proposals=True, **kwargs):
        self.num = num
        self.pos_fraction = pos_fraction
        self.neg_pos_ub = neg_pos_ub
        self.add_gt_as_proposals = add_gt_as_proposals

    @abstractmethod
    def _sample_pos(self, assign_result, num_expected, **kwargs):
        pass

    @abstractmethod
    def _sample_neg(self, assign_result, num_expected, **kwargs):
        pass

    def sample(self, assign_result, bboxes, gt_bboxes, gt_labels=None, **kwargs):
        gt_bboxes = gt_bboxes.to(bboxes.dtype)
        if bboxes.dim() < 2:
            bboxes = bboxes[None, :]
        bboxes = bboxes[:, :self.box_dim]
        gt_flags = torch.zeros((bboxes.shape[0],), dtype=torch.bool, device=bboxes.device)
        if self.add_gt_as_proposals:
            bboxes = torch.cat([gt_bboxes, bboxes], dim=0)
            assign_result.add_gt_(gt_labels)
            gt_flags = torch.cat([torch.ones((gt_bboxes.shape[0],), dtype=torch.bool, device=bboxes.device), gt_flags])
        num_expected_pos = int(self.num * self.pos_fraction)
        pos_inds = self._sample_pos(assign_result, num_expected_pos, bboxes=bboxes, **kwargs).unique()
        num_sampled_pos = pos_inds.numel()
        num_expected_neg = self.num - num_sampled_pos
        if self.neg_pos_ub >= 0:
            _pos = max(1, num_sampled_pos)
            neg_upper_bound = int(self.neg_pos_ub * _pos)
            if num_expected_neg > neg_upper_bound:
                num_expected_neg = neg_upper_bound
        neg_inds = self._sample_neg(assign_result, num_expected_neg, bboxes=bboxes, **kwargs).unique()
        return SamplingResult(pos_inds, neg_inds, bboxes, gt_bboxes, assign_result, gt_flags)


@BOXES.register_module()
class PseudoSampler(BaseSampler):
    def __init__(self, **kwargs):
        pass

    def _sample_pos(self, **kwargs):
        raise NotImplementedError

    def _sample_neg(self, **kwargs):
        raise NotImplementedError

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

    @staticmethod
    def random_choice(gallery, num):
        """jt.randperm in the reference (sampler.py:L153): nondeterministic by construction; seed torch
        for reproducible runs."""
        assert len(gallery) >= num
        perm = torch.randperm(gallery.numel(), device=gallery.device)[:num]
        return gallery[perm]

    def _sample_pos(self, assign_result, num_expected, **kwargs):
        pos_inds = torch.nonzero(assign_result.gt_inds > 0)[:, 0]
        if pos_inds.numel() <= num_expected:
            return pos_inds
        return self.random_choice(pos_inds, num_expected)

    def _sample_neg(self, assign_result, num_expected, **kwargs):
        neg_inds = torch.nonzero(assign_result.gt_inds == 0)[:, 0]
        if len(neg_inds) <= num_expected:
            return neg_inds
        return self.random_choice(neg_inds, num_expected)


@BOXES.register_module()
class RandomSamplerRotated(RandomSampler):
    """identical to RandomSampler but slices 5 box columns (sampler.py:L203)"""
    box_dim = 5
