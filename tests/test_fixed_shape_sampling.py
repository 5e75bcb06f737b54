"""CPU: the fixed-shape samplers (models/boxes/fixed_shape.py) draw what RandomSampler / RandomSamplerRotated draw
(python/jdet/models/boxes/sampler.py:L133-233) -- counts, classes, no duplicates, neg_pos_ub, add_gt_as_proposals
layout, seeding -- without data-dependent shapes."""
import numpy as np
import pytest
import torch

from jdet_amd.models.boxes.fixed_shape import sample_fixed, sample_rows, scatter_rows


def _gt_inds(n, n_pos, n_ign, seed=0):
    g = torch.Generator().manual_seed(seed)
    v = torch.zeros(n, dtype=torch.int32)
    perm = torch.randperm(n, generator=g)
    v[perm[:n_pos]] = torch.randint(1, 9, (n_pos,), generator=g, dtype=torch.int32)
    v[perm[n_pos:n_pos + n_ign]] = -1
    return v


@pytest.mark.parametrize("n,n_pos,n_ign,num,frac,ub", [
    (5000, 40, 100, 256, 0.5, -1),      # fewer positives than asked: negatives fill up to num
    (5000, 400, 100, 256, 0.5, -1),     # more positives than asked: exactly num * frac
    (5000, 0, 0, 256, 0.5, -1),         # no positive
    (300, 10, 280, 256, 0.5, -1),       # negatives run out
    (5000, 30, 0, 512, 0.25, 3),        # neg_pos_ub: at most 3 * n_pos negatives
    (5000, 0, 0, 512, 0.25, 3),         # ... with no positive: int(ub * max(1, 0)) = ub
    (100, 60, 0, 512, 0.25, -1),        # fewer candidates than num
])
def test_counts_classes_and_uniqueness(n, n_pos, n_ign, num, frac, ub):
    gi = _gt_inds(n, n_pos, n_ign)
    g = torch.Generator().manual_seed(5)
    pi, pv, ni, nv = sample_fixed(gi, num, frac, ub, generator=g)
    P = int(num * frac)
    exp_pos = min(n_pos, P)
    n_neg_avail = n - n_pos - n_ign
    exp_neg = num - exp_pos
    if ub >= 0:
        exp_neg = min(exp_neg, int(ub * max(1, exp_pos)))
    exp_neg = min(exp_neg, n_neg_avail)
    assert pi.numel() == min(P, n) and ni.numel() == min(num, n)
    assert int(pv.sum()) == exp_pos and int(nv.sum()) == exp_neg
    sp, sn = pi[pv], ni[nv]
    assert (gi[sp] > 0).all() and (gi[sn] == 0).all()
    assert sp.unique().numel() == sp.numel() and sn.unique().numel() == sn.numel()
    rows, valid, is_pos = sample_rows(gi, num, frac, ub, generator=torch.Generator().manual_seed(5))
    assert rows.numel() == valid.numel() == is_pos.numel() == num
    assert int(valid.sum()) == exp_pos + exp_neg and int(is_pos.sum()) == exp_pos
    assert is_pos[:exp_pos].all() and not is_pos[exp_pos:].any()          # positives first, then negatives
    assert valid[:exp_pos + exp_neg].all() and not valid[exp_pos + exp_neg:].any()
    assert set(rows[is_pos].tolist()) == set(sp.tolist())
    assert (rows >= 0).all() and (rows < n).all()


def test_seeded_and_uniform():
    gi = _gt_inds(2000, 500, 0, seed=3)
    a = sample_fixed(gi, 128, 0.5, generator=torch.Generator().manual_seed(11))
    b = sample_fixed(gi, 128, 0.5, generator=torch.Generator().manual_seed(11))
    assert all(torch.equal(x, y) for x, y in zip(a, b))
    c = sample_fixed(gi, 128, 0.5, generator=torch.Generator().manual_seed(12))
    assert not torch.equal(a[0], c[0])
    # every positive is picked with probability 64 / 500: the empirical frequencies are flat
    hits = torch.zeros(2000)
    g = torch.Generator().manual_seed(0)
    for _ in range(400):
        pi, pv, _, _ = sample_fixed(gi, 128, 0.5, generator=g)
        hits[pi[pv]] += 1
    f = hits[gi > 0] / 400
    assert abs(float(f.mean()) - 64 / 500) < 1e-6 and float(f.std()) < 0.03 and float(f.min()) > 0.05


def test_add_gt_as_proposals_layout():
    """gts are prepended with gt_inds 1..K (AssignResult.add_gt_, sampler.py:L92-99): all of them are positives and
    (being at most num * frac) all of them are sampled"""
    K = 12
    gi = torch.cat([torch.arange(1, K + 1, dtype=torch.int32), _gt_inds(2000, 30, 50)])
    rows, valid, is_pos = sample_rows(gi, 512, 0.25, generator=torch.Generator().manual_seed(1))
    assert set(range(K)) <= set(rows[is_pos].tolist())
    assert int(is_pos.sum()) == K + 30 and int(valid.sum()) == 512


def test_scatter_rows_ignores_invalid_rows():
    dst = torch.zeros(10, 3)
    idx = torch.tensor([2, 5, 5, 9])
    valid = torch.tensor([True, False, True, False])
    out = scatter_rows(dst, idx, valid, torch.arange(12.).view(4, 3))
    assert torch.equal(out[2], torch.tensor([0., 1., 2.])) and torch.equal(out[5], torch.tensor([6., 7., 8.]))
    assert float(out[9].abs().sum()) == 0 and float(out.sum()) == 3 + 21
    lab = scatter_rows(torch.zeros(10, dtype=torch.long), idx, valid, 1)
    assert lab.tolist() == [0, 0, 1, 0, 0, 1, 0, 0, 0, 0]


def test_roi_feature_linear_keeps_reference_checkpoint_order():
    """the first FC layer of the RoI heads stores its weight for channels-last features; state dicts carry the
    reference's (c, ph, pw) column order in both directions"""
    from jdet_amd.models.roi_heads.roi_feature_linear import RoIFeatureLinear
    torch.manual_seed(0)
    C, O, R = 6, 5, 3
    ref = torch.nn.Linear(C * 4, O)
    m = RoIFeatureLinear(C, 4, O)
    m.load_state_dict(ref.state_dict())
    x = torch.randn(R, C, 2, 2)
    y_ref = ref(x.flatten(1))
    assert torch.allclose(m(x), y_ref, atol=1e-6)
    assert torch.allclose(m(x.contiguous(memory_format=torch.channels_last)), y_ref, atol=1e-6)
    sd = m.state_dict()
    assert torch.equal(sd["weight"], ref.weight) and torch.equal(sd["bias"], ref.bias)
    assert not torch.equal(m.weight, ref.weight)       # ... while the parameter itself is stored permuted


def test_roi_transformer_head_keeps_the_reference_weight_order():
    """SharedFCBBoxHeadRbbox consumes (R, C, 7, 7) features through RoIFeatureLinear: a state dict in the reference's
    column order ((c, ph, pw), `x.reshape(R, -1)` of an NCHW tensor, convfc_rbbox_head.py:L136-144) gives the
    reference's outputs for contiguous and channels-last inputs alike, and comes back unchanged from state_dict()"""
    import torch.nn.functional as F
    import jdet_amd.models  # noqa: F401
    from jdet_amd.models.roi_heads import SharedFCBBoxHeadRbbox
    torch.manual_seed(0)
    C, R = 8, 6
    head = SharedFCBBoxHeadRbbox(num_fcs=2, in_channels=C, fc_out_channels=32, roi_feat_size=7, num_classes=5,
                                 reg_class_agnostic=True, with_module=False,
                                 loss_cls=dict(type="CrossEntropyLossForRcnn", use_sigmoid=False, loss_weight=1.0))
    ref = {k: torch.randn_like(v) for k, v in head.state_dict().items()}
    head.load_state_dict(ref)
    for k, v in head.state_dict().items():
        assert torch.equal(v, ref[k]), k
    x = torch.randn(R, C, 7, 7)
    h = F.relu(F.linear(x.reshape(R, -1), ref["shared_fcs.0.weight"], ref["shared_fcs.0.bias"]))
    h = F.relu(F.linear(h, ref["shared_fcs.1.weight"], ref["shared_fcs.1.bias"]))
    cls_ref = F.linear(h, ref["fc_cls.weight"], ref["fc_cls.bias"])
    reg_ref = F.linear(h, ref["fc_reg.weight"], ref["fc_reg.bias"])
    for inp in (x, x.contiguous(memory_format=torch.channels_last)):
        cls, reg = head(inp)
        assert torch.allclose(cls, cls_ref, atol=1e-5) and torch.allclose(reg, reg_ref, atol=1e-5)


@pytest.mark.parametrize("cls_name,dim", [("RandomSampler", 4), ("RandomSamplerRotated", 5)])
def test_list_shaped_sampler_interface_draws_through_the_fixed_shape_sampler(cls_name, dim):
    """`sampler.sample(assign_result, bboxes, gt_bboxes, gt_labels)` (python/jdet/models/boxes/sampler.py:L72-111) for
    callers outside the fixed-shape paths: SamplingResult with ascending index lists, the reference's counts, the gts
    prepended as positives matched to themselves, the box columns cut to box_dim."""
    from jdet_amd.models.boxes import sampler as S
    from jdet_amd.models.boxes.assigner import AssignResult
    n, k = 3000, 8          # (_gt_inds draws gt indices 1..8)
    gi = _gt_inds(n, 90, 200, seed=3).to(torch.int64)
    labels = torch.where(gi > 0, gi + 10, torch.zeros_like(gi))
    boxes = torch.rand(n, dim + 1)
    gts = torch.rand(k, dim)
    res = AssignResult(k, gi.clone(), torch.rand(n), labels.clone())
    smp = getattr(S, cls_name)(num=128, pos_fraction=0.25, neg_pos_ub=-1, add_gt_as_proposals=True)
    out = smp.sample(res, boxes, gts, torch.arange(k) + 100, generator=torch.Generator().manual_seed(1))
    assert out.pos_inds.numel() == 32 and out.neg_inds.numel() == 96
    assert torch.equal(out.pos_inds, out.pos_inds.unique()) and torch.equal(out.neg_inds, out.neg_inds.unique())
    full = torch.cat([torch.arange(1, k + 1), gi])
    assert (full[out.pos_inds] > 0).all() and (full[out.neg_inds] == 0).all()
    assert out.pos_bboxes.shape == (32, dim) and out.neg_bboxes.shape == (96, dim)
    assert torch.equal(out.pos_is_gt, out.pos_inds < k)
    assert torch.equal(out.pos_assigned_gt_inds, full[out.pos_inds] - 1)
    assert torch.equal(out.pos_gt_bboxes, gts[full[out.pos_inds] - 1])
    # a gt sampled as a positive carries its own label
    own = out.pos_inds < k
    assert torch.equal(out.pos_gt_labels[own], (out.pos_inds[own] + 100).to(out.pos_gt_labels.dtype))


def test_subclass_hooks_and_compatibility_surface():
    """advisor findings (round 5): a config-registered subclass that overrides `_sample_pos` / `_sample_neg` (the
    reference's extension points, sampler.py:L52-58) is sampled through them; `RandomSampler.random_choice` exists; a flat
    (1-D) box is accepted"""
    from types import SimpleNamespace

    from jdet_amd.models.boxes.sampler import RandomSampler

    class FirstK(RandomSampler):
        def _sample_pos(self, assign_result, num_expected, **kwargs):
            return torch.nonzero(assign_result.gt_inds > 0)[:, 0][:num_expected]

        def _sample_neg(self, assign_result, num_expected, **kwargs):
            return torch.nonzero(assign_result.gt_inds == 0)[:, 0][:num_expected]

    gi = torch.tensor([0, 1, 2, 0, 0, 1, -1, 0, 2, 0])
    res = SimpleNamespace(gt_inds=gi, labels=None, add_gt_=lambda labels: None)
    boxes = torch.arange(40, dtype=torch.float32).view(10, 4)
    s = FirstK(num=4, pos_fraction=0.5, add_gt_as_proposals=False).sample(res, boxes, boxes[:2])
    assert s.pos_inds.tolist() == [1, 2] and s.neg_inds.tolist() == [0, 3]
    picked = RandomSampler.random_choice(torch.arange(100), 10)
    assert picked.numel() == 10 and picked.unique().numel() == 10
    assert len(RandomSampler.random_choice(list(range(20)), 5)) == 5
    one = RandomSampler(num=4, pos_fraction=0.5, add_gt_as_proposals=False).sample(
        SimpleNamespace(gt_inds=torch.tensor([1]), labels=None, add_gt_=lambda labels: None), torch.tensor([1., 2., 3., 4.]),
        torch.tensor([[1., 2., 3., 4.]]))
    assert one.pos_inds.tolist() == [0]
