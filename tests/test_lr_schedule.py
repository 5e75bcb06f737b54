"""CPU: the learning-rate schedules against hand-evaluated values of the closed form in
python/jdet/optims/lr_scheduler.py (WarmUpLR L6-70, StepLR L196-235) and the checkpoint dictionary contract."""
import pytest

from jdet_amd.optims.lr_scheduler import StepLR, WarmUpLR


class _Opt:
    def __init__(self):
        self.lr = 0.01
        self.param_groups = [{"lr": 0.01}, {"lr": 0.02}, {}]


def test_steplr_linear_warmup_then_epoch_milestones():
    o = _Opt()
    s = StepLR(optimizer=o, milestones=[7, 10], warmup="linear", warmup_iters=500, warmup_ratio=1.0 / 3)
    assert o.lr == pytest.approx(0.01 / 3)                                  # it = 0
    s.step(250, 0)
    assert o.lr == pytest.approx(0.01 * (1 - 0.5 * (2.0 / 3)))             # 1 - (1 - it/W)(1 - ratio)
    assert [g["lr"] for g in o.param_groups] == pytest.approx([0.01 * 2 / 3, 0.02 * 2 / 3, 0.01 * 2 / 3])
    for ep, k in ((0, 1.0), (6, 1.0), (7, 0.1), (9, 0.1), (10, 0.01), (11, 0.01)):
        s.step(600, ep)
        assert o.lr == pytest.approx(0.01 * k) and o.param_groups[1]["lr"] == pytest.approx(0.02 * k)
    s.step(700, 0, by_epoch=False)                                          # by iteration: steps = it - warmup_iters
    assert o.lr == pytest.approx(0.01 * 0.01)


def test_other_modes_and_period_milestone():
    o = _Opt()
    s = StepLR(optimizer=o, milestones=3, gamma=0.5, min_lr=0.002, warmup="exp", warmup_iters=10, warmup_ratio=0.25)
    s.step(5, 0)
    assert o.lr == pytest.approx(0.01 * 0.25 ** 0.5)
    s.step(10, 4)
    assert o.lr == pytest.approx(0.005)
    s.step(10, 9)
    assert o.lr == pytest.approx(0.002)                                     # 0.01 / 8 floored at min_lr
    w = WarmUpLR(optimizer=_Opt(), warmup="constant", warmup_iters=4, warmup_ratio=0.1)
    assert w.optimizer.lr == pytest.approx(0.001)
    w.step(4, 0)
    assert w.optimizer.lr == pytest.approx(0.01)
    n = WarmUpLR(optimizer=_Opt())                                          # no warm-up mode: base rate throughout
    assert n.optimizer.lr == pytest.approx(0.01)
    with pytest.raises(TypeError):
        StepLR(optimizer=_Opt(), milestones="7")
    with pytest.raises(TypeError):
        StepLR(optimizer=_Opt(), milestones=[7, 0])


def test_checkpoint_dictionary_round_trip():
    s = StepLR(optimizer=_Opt(), milestones=[7, 10], warmup="linear", warmup_iters=500, warmup_ratio=1.0 / 3)
    p = s.parameters()
    assert set(p) == {"milestones", "gamma", "min_lr", "warmup_ratio", "warmup_iters", "warmup", "base_lr", "base_lr_pg"}
    t = StepLR(optimizer=_Opt(), milestones=[1], warmup=None)
    t.load_parameters(dict(p, unknown_key=1, optimizer="x"))
    assert t.parameters() == p and not isinstance(t.optimizer, str)
