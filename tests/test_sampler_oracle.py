"""CPU: the sampler restatement (oracle/sampler_oracle.py) against closed-form counts, and the fixed-shape sampler
(models/boxes/fixed_shape.py) against it with the SAME keys: rows bit-exact (set, classes, order)."""
import numpy as np
import pytest
import torch

from jdet_amd.models.boxes import fixed_shape as FS
from oracle import sampler_oracle as SO


def _gt_inds(n, n_pos, n_ign, seed=0):
    rng = np.random.default_rng(seed)
    v = np.zeros(n, np.int64)
    perm = rng.permutation(n)
    v[perm[:n_pos]] = rng.integers(1, 9, n_pos)
    v[perm[n_pos:n_pos + n_ign]] = -1
    return v


CASES = [
    (5000, 40, 100, 256, 0.5, -1),      # fewer positives than asked: negatives fill up to num
    (5000, 400, 100, 256, 0.5, -1),     # REAL sub-sampling of both classes
    (2000, 700, 50, 512, 0.25, -1),     # the R-CNN stage shape: 128 of 700 positives, 384 of 1250 negatives
    (5000, 0, 0, 256, 0.5, -1),         # no positive
    (300, 10, 280, 256, 0.5, -1),       # negatives run out
    (5000, 30, 0, 512, 0.25, 3),        # neg_pos_ub
    (5000, 0, 0, 512, 0.25, 3),         # ... with no positive: int(ub * max(1, 0)) = ub
    (100, 60, 0, 512, 0.25, -1),        # fewer candidates than num
]


@pytest.mark.parametrize("n,n_pos,n_ign,num,frac,ub", CASES)
def test_restatement_counts(n, n_pos, n_ign, num, frac, ub):
    gi = _gt_inds(n, n_pos, n_ign)
    keys = np.random.default_rng(1).random(n)
    pos, neg = SO.sample(gi, num, frac, ub, keys)
    exp_pos = min(n_pos, int(num * frac))
    exp_neg = num - exp_pos
    if ub >= 0:
        exp_neg = min(exp_neg, int(ub * max(1, exp_pos)))
    exp_neg = min(exp_neg, n - n_pos - n_ign)
    assert len(pos) == exp_pos and len(neg) == exp_neg
    assert (gi[pos] > 0).all() and (gi[neg] == 0).all()
    assert (np.diff(pos) > 0).all() and (np.diff(neg) > 0).all()          # `.unique()`: ascending, no duplicates
    if n_pos > exp_pos:      # the kept positives are the exp_pos smallest keys among the positives
        allp = np.nonzero(gi > 0)[0]
        assert set(pos) == set(allp[np.argsort(keys[allp], kind="stable")[:exp_pos]])


def check_rows_against_restatement(gi_np, num, frac, ub, dev, seed=7):
    """one draw on `dev` with a seeded generator; the restatement gets the very keys the draw used"""
    gi = torch.from_numpy(gi_np).to(dev)
    g = torch.Generator(device=dev).manual_seed(seed)
    keys = torch.rand((gi.numel(),), device=dev, generator=g).cpu().numpy()
    rows, valid, is_pos = FS.sample_rows(gi, num, frac, ub, generator=torch.Generator(device=dev).manual_seed(seed))
    rows, valid, is_pos = rows.cpu().numpy(), valid.cpu().numpy(), is_pos.cpu().numpy()
    pos, neg = SO.sample(gi_np, num, frac, ub, keys)
    exp = np.concatenate([pos, neg])
    assert rows.shape == (num,) and valid.sum() == len(exp) and is_pos.sum() == len(pos)
    assert valid[:len(exp)].all() and not valid[len(exp):].any()
    assert is_pos[:len(pos)].all() and not is_pos[len(pos):].any()
    assert np.array_equal(rows[:len(exp)], exp)                             # set, classes AND order
    assert (rows >= 0).all() and (rows < max(1, gi_np.size)).all()
    return len(pos), len(neg)


@pytest.mark.parametrize("n,n_pos,n_ign,num,frac,ub", CASES)
def test_fixed_shape_rows_equal_the_restatement_cpu(n, n_pos, n_ign, num, frac, ub):
    check_rows_against_restatement(_gt_inds(n, n_pos, n_ign, seed=3), num, frac, ub, torch.device("cpu"))
