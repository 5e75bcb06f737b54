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

    # The reference's extension points (sampler.py:L52-58): a subclass that overrides them (IoU-balanced / OHEM-style
    # samplers registered from a config) is sampled through them, in the reference's sequence; the stock samplers leave
    # them alone and draw through the fixed-shape sampler.
    def _sample_pos(self, assign_result, num_expected, **kwargs):
        raise NotImplementedError

    def _sample_neg(self, assign_result, num_expected, **kwargs):
        raise NotImplementedError

    def _hooks_overridden(self):
        return (type(self)._sample_pos is not BaseSampler._sample_pos
                or type(self)._sample_neg is not BaseSampler._sample_neg)

    def sample(self, assign_result, bboxes, gt_bboxes, gt_labels=None, generator=None, **kwargs):
        from .fixed_shape import sample_fixed
        gt_bboxes = gt_bboxes.to(bboxes.dtype)
        if bboxes.dim() < 2:                    # a single box (or none) handed over flat, as the reference accepts it
            bboxes = bboxes[None, :]
        bboxes = bboxes[:, :self.box_dim]
        is_gt = torch.zeros((bboxes.shape[0],), dtype=torch.bool, device=bboxes.device)
        if self.add_gt_as_proposals:            # the gts join the candidates, matched to themselves (L92-99)
            assign_result.add_gt_(gt_labels)
            is_gt = torch.cat([is_gt.new_ones((gt_bboxes.shape[0],)), is_gt])
            bboxes = torch.cat([gt_bboxes, bboxes], dim=0)
        if self._hooks_overridden():            # sampler.py:L86-110 through the subclass's hooks
            pos = torch.unique(self._sample_pos(assign_result, int(self.num * self.pos_fraction), bboxes=bboxes,
                                                **kwargs))
            n_neg = self.num - pos.numel()
            if self.neg_pos_ub >= 0:
                n_neg = min(n_neg, int(self.neg_pos_ub * max(1, pos.numel())))
            neg = torch.unique(self._sample_neg(assign_result, n_neg, bboxes=bboxes, **kwargs))
            return SamplingResult(pos, neg, bboxes, gt_bboxes, assign_result, is_gt)
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

    @staticmethod
    def random_choice(gallery, num):
        """`num` elements of `gallery` uniformly at random (public in the reference, sampler.py:L145-156; the stock
        samplers here draw through fixed_shape.sample_fixed and do not call it)"""
        assert len(gallery) >= num
        is_tensor = torch.is_tensor(gallery)
        g = gallery if is_tensor else torch.as_tensor(gallery, dtype=torch.int64)
        picked = g[torch.randperm(g.numel(), device=g.device)[:num]]
        return picked if is_tensor else picked.cpu().numpy()


@BOXES.register_module()
class RandomSamplerRotated(RandomSampler):
    """identical to RandomSampler but slices 5 box columns (sampler.py:L203)"""
    box_dim = 5
