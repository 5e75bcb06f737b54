"""Learning-rate schedules as closed-form functions of (iteration, epoch).

Contract mirrored from python/jdet/optims/lr_scheduler.py (registry names `WarmUpLR` L6-70 and `StepLR` L196-235,
constructor keywords, `step(iters, epochs, by_epoch)`, and the `parameters()` / `load_parameters()` dictionaries that
travel in checkpoints).  The schedule itself is stated once, as a function:

    lr(base, it, ep) = base * warm(it)                      while it < warmup_iters (and a warm-up mode is set)
                     = base * decay(ep)                     afterwards, counting by epoch
                     = base * decay(it - warmup_iters)      afterwards, counting by iteration

    warm(it): "constant" -> ratio;  "linear" -> 1 - (1 - it/W)(1 - ratio);  "exp" -> ratio ** (1 - it/W)
    decay(s): StepLR -> gamma ** (s // milestones)  or  gamma ** #{milestones m : not s < m, up to the first s < m}
"""
from jdet_amd.utils.registry import SCHEDULERS

_WARMUPS = {
    "constant": lambda ratio, frac: ratio,
    "linear": lambda ratio, frac: 1 - (1 - frac) * (1 - ratio),
    "exp": lambda ratio, frac: ratio ** (1 - frac),
}


def scheduled_lr(base, iters, epochs, by_epoch, warmup, warmup_iters, warmup_ratio, decay):
    """The whole schedule: `decay(steps)` is the post-warm-up factor of the concrete scheduler."""
    if warmup is not None and iters < warmup_iters:
        return _WARMUPS[warmup](warmup_ratio, iters / warmup_iters) * base
    if warmup is None:
        steps = epochs if by_epoch else iters
    else:
        steps = epochs if by_epoch else iters - warmup_iters
    return decay(base, steps)


class _Schedule(object):
    """Holder of the schedule's constants; every attribute except `optimizer` is checkpoint state."""

    def __init__(self, optimizer, warmup_ratio=1.0 / 3, warmup_iters=500, warmup=None):
        if warmup is not None and warmup not in _WARMUPS:
            raise ValueError("unknown warm-up mode %r" % (warmup,))
        self.optimizer = optimizer
        self.warmup_ratio, self.warmup_iters, self.warmup = warmup_ratio, warmup_iters, warmup
        self.base_lr = optimizer.lr
        self.base_lr_pg = [group.get("lr", optimizer.lr) for group in optimizer.param_groups]
        self.step(0, 0)

    def decay(self, base, steps):
        return base

    def lr_at(self, base, iters, epochs, by_epoch=True):
        return scheduled_lr(base, iters, epochs, by_epoch, self.warmup, self.warmup_iters, self.warmup_ratio, self.decay)

    def step(self, iters, epochs, by_epoch=True):
        self.optimizer.lr = self.lr_at(self.base_lr, iters, epochs, by_epoch)
        for group, base in zip(self.optimizer.param_groups, self.base_lr_pg):
            group["lr"] = self.lr_at(base, iters, epochs, by_epoch)

    def parameters(self):
        return {k: v for k, v in vars(self).items() if k != "optimizer"}

    def load_parameters(self, data):
        if isinstance(data, dict):
            vars(self).update({k: v for k, v in data.items() if k in vars(self) and k != "optimizer"})


@SCHEDULERS.register_module()
class WarmUpLR(_Schedule):
    """warm-up, then the base rate"""


@SCHEDULERS.register_module()
class StepLR(_Schedule):
    """warm-up, then base * gamma ** (number of milestones passed), floored at min_lr"""

    def __init__(self, milestones, gamma=0.1, min_lr=None, **kwargs):
        ok = (isinstance(milestones, int) and milestones > 0) or (
            isinstance(milestones, list) and all(m > 0 for m in milestones))
        if not ok:
            raise TypeError("milestones: a positive int (period) or a list of positive ints, got %r" % (milestones,))
        self.milestones, self.gamma, self.min_lr = milestones, gamma, min_lr
        super().__init__(**kwargs)

    def decay(self, base, steps):
        if isinstance(self.milestones, int):
            passed = steps // self.milestones
        else:
            passed = next((i for i, m in enumerate(self.milestones) if steps < m), len(self.milestones))
        lr = base * self.gamma ** passed
        return lr if self.min_lr is None else max(lr, self.min_lr)
