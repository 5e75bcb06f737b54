"""TEST INFRASTRUCTURE ONLY (never imported by jdet_amd): numpy restatement of the reference's RandomSampler /
RandomSamplerRotated (python/jdet/models/boxes/sampler.py:L52-110, L114-233) with the one unpinnable primitive made an
argument: `jt.randperm(n)` is replaced by `argsort(keys[gallery])`, keys = one number per candidate.  (A uniformly random
key per candidate and "the `num` smallest keys" IS a uniformly random subset of size `num`, which is what
`gallery[randperm(n)[:num]]` draws -- sampler.py:L145-156; with the keys fixed both sides draw the same subset.)
Parity unpinned by reference execution (Jittor is not importable here); checked against closed-form counts in
tests/test_sampler_oracle.py."""
import numpy as np


def random_choice(gallery, num, keys):
    """sampler.py:L145-156 with randperm := argsort of the candidates' keys"""
    assert len(gallery) >= num
    perm = np.argsort(keys[gallery], kind="stable")[:num]
    return gallery[perm]


def sample(gt_inds, num, pos_fraction, neg_pos_ub, keys):
    """BaseSampler.sample after the optional gt concatenation (sampler.py:L86-110): gt_inds (A,) > 0 positive, 0 negative,
    < 0 ignored.  -> (pos_inds, neg_inds), each ascending (`.unique()`, L90 / L104)."""
    gt_inds = np.asarray(gt_inds)
    num_expected_pos = int(num * pos_fraction)
    pos = np.nonzero(gt_inds > 0)[0]
    if len(pos) > num_expected_pos:                                  # _sample_pos, L158-166
        pos = random_choice(pos, num_expected_pos, keys)
    pos = np.unique(pos)
    num_expected_neg = num - len(pos)
    if neg_pos_ub >= 0:
        num_expected_neg = min(num_expected_neg, int(neg_pos_ub * max(1, len(pos))))
    neg = np.nonzero(gt_inds == 0)[0]
    if len(neg) > num_expected_neg:                                  # _sample_neg, L168-176
        neg = random_choice(neg, num_expected_neg, keys)
    neg = np.unique(neg)
    return pos, neg


def sample_with_gts(assigned_gt_inds, assigned_labels, gt_labels, num, pos_fraction, neg_pos_ub, add_gt_as_proposals,
                    keys):
    """RandomSampler(Rotated).sample incl. add_gt_as_proposals (L86-92, L203-211; AssignResult.add_gt_: the gts are
    PREPENDED, matched to themselves with their own labels).  keys: one per row of the concatenated candidate list.
    -> dict(pos_inds, neg_inds, gt_inds, labels, gt_flags) over the concatenated list."""
    gi, lab = np.asarray(assigned_gt_inds), np.asarray(assigned_labels)
    flags = np.zeros(len(gi), bool)
    if add_gt_as_proposals:
        k = len(gt_labels)
        gi = np.concatenate([np.arange(1, k + 1), gi])
        lab = np.concatenate([np.asarray(gt_labels), lab])
        flags = np.concatenate([np.ones(k, bool), flags])
    pos, neg = sample(gi, num, pos_fraction, neg_pos_ub, keys)
    return {"pos_inds": pos, "neg_inds": neg, "gt_inds": gi, "labels": lab, "gt_flags": flags}
