"""GPU: the fixed-shape samplers under REAL sub-sampling (fewer rows than candidates) against the numpy restatement of
the reference's RandomSampler / RandomSamplerRotated (oracle/sampler_oracle.py; sampler.py:L52-110, L114-233) fed the
same per-candidate keys: counts, classes, gt-first order and the rows themselves are bit-exact.  (The reference's own
randomness, `jt.randperm`, cannot be pinned; its use -- `gallery[perm[:num]]` then `.unique()` -- is what is restated.)"""
import numpy as np
import pytest
import torch

from jdet_amd.models.boxes import fixed_shape as FS
from oracle import sampler_oracle as SO
from tests.test_sampler_oracle import CASES, _gt_inds, check_rows_against_restatement

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("n,n_pos,n_ign,num,frac,ub", CASES)
def test_sample_rows_equal_the_restatement_on_the_device(dev, n, n_pos, n_ign, num, frac, ub):
    n_p, n_n = check_rows_against_restatement(_gt_inds(n, n_pos, n_ign, seed=11), num, frac, ub, dev)
    if n_pos > int(num * frac):
        assert n_p == int(num * frac) < n_pos                 # the regime the head-parity tests never reach


class _FixedAssigner:
    """assign_wrt_overlaps of MaxIoUAssigner (pos 0.5 / neg 0.5, no low-quality matches) on a given overlap matrix"""
    iou_calculator = None

    def __init__(self, overlaps):
        self.overlaps = overlaps

    def assign_wrt_overlaps(self, overlaps, gt_labels):
        from types import SimpleNamespace
        mx, arg = overlaps.max(0)
        gt_inds = torch.where(mx >= 0.5, arg + 1, torch.zeros_like(arg))
        gt_inds = torch.where(mx < 0, torch.full_like(arg, -1), gt_inds)
        labels = torch.where(gt_inds > 0, gt_labels[(gt_inds - 1).clamp(min=0)], torch.zeros_like(gt_inds))
        return SimpleNamespace(gt_inds=gt_inds, labels=labels)


@pytest.mark.parametrize("add_gt", [True, False])
def test_stage_rows_with_gts_as_proposals_under_subsampling(dev, add_gt, monkeypatch):
    """sample_stage_rows (the R-CNN stages): 2000 candidates, 24 gts, 512 rows at 0.25 -> 128 of ~600 positives and 384
    of ~1300 negatives are drawn; with add_gt_as_proposals the gts join as candidates 0..K-1 matched to themselves, and
    the SAMPLED gts lead the rows (ascending index)."""
    from types import SimpleNamespace
    rng = np.random.default_rng(3)
    P, K, D = 2000, 24, 5
    ov = rng.random((K, P)).astype(np.float32) * 0.45
    hit = rng.choice(P, 600, replace=False)
    ov[rng.integers(0, K, 600), hit] = 0.5 + 0.5 * rng.random(600).astype(np.float32)
    alive = np.ones(P, bool)
    alive[rng.choice(P, 100, replace=False)] = False
    cands = torch.from_numpy(rng.random((P, D)).astype(np.float32)).to(dev)
    gts = torch.from_numpy(rng.random((K, D)).astype(np.float32)).to(dev)
    gt_labels = torch.from_numpy(rng.integers(1, 16, K)).to(dev)
    overlaps = torch.from_numpy(ov).to(dev)
    monkeypatch.setattr(FS, "masked_overlaps", lambda calc, g, b, al: torch.where(al[None, :], overlaps,
                                                                                 torch.full_like(overlaps, -1.0)))
    sampler = SimpleNamespace(num=512, pos_fraction=0.25, neg_pos_ub=-1, add_gt_as_proposals=add_gt)
    seed = 99
    monkeypatch.setattr(FS, "_keys", lambda n, device, generator: torch.rand(
        (n,), device=device, generator=torch.Generator(device=device).manual_seed(seed)))
    A = P + (K if add_gt else 0)
    keys = torch.rand((A,), device=dev, generator=torch.Generator(device=dev).manual_seed(seed)).cpu().numpy()
    dummy = torch.zeros((D,), device=dev)
    sr = FS.sample_stage_rows(cands, torch.from_numpy(alive).to(dev), gts, gt_labels, _FixedAssigner(overlaps), sampler,
                              dummy)
    # restatement: the assigner's result, then the sampler
    ovm = np.where(alive[None, :], ov, -1.0)
    mx, arg = ovm.max(0), ovm.argmax(0)
    gi = np.where(mx >= 0.5, arg + 1, 0)
    gi = np.where(mx < 0, -1, gi)
    lab = np.where(gi > 0, gt_labels.cpu().numpy()[np.clip(gi - 1, 0, None)], 0)
    ref = SO.sample_with_gts(gi, lab, gt_labels.cpu().numpy(), 512, 0.25, -1, add_gt, keys)
    pos, neg = ref["pos_inds"], ref["neg_inds"]
    assert len(pos) == 128 and len(neg) == 384                      # both classes truly sub-sampled
    valid, is_pos = sr.valid.cpu().numpy(), sr.is_pos.cpu().numpy()
    assert valid.all() and is_pos[:128].all() and not is_pos[128:].any()
    boxes_all = np.concatenate([gts.cpu().numpy(), cands.cpu().numpy()]) if add_gt else cands.cpu().numpy()
    exp_rows = np.concatenate([pos, neg])
    assert np.array_equal(sr.boxes.cpu().numpy(), boxes_all[exp_rows])                     # the rows, in order
    assert np.array_equal(sr.is_gt.cpu().numpy(), ref["gt_flags"][exp_rows])
    assert np.array_equal(sr.matched.cpu().numpy()[:128], ref["gt_inds"][pos] - 1)
    assert np.array_equal(sr.labels.cpu().numpy()[:128], ref["labels"][pos])
    assert (sr.labels.cpu().numpy()[128:] == 0).all()
    if add_gt:
        n_gt = int(ref["gt_flags"][pos].sum())
        assert 0 < n_gt <= K and ref["gt_flags"][exp_rows][:n_gt].all()     # sampled gts first
