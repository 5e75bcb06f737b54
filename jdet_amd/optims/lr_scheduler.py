"""LR schedules.  Mirrors python/jdet/optims/lr_scheduler.py: `WarmUpLR` L6-70 (constant / linear / exp
warm-up over `warmup_iters`, then the schedule by epoch), `StepLR` L196-235."""
from jdet_amd.utils.registry import SCHEDULERS


@SCHEDULERS.register_module()
class WarmUpLR(object):
    def __init__(self, optimizer, warmup_ratio=1.0 / 3, warmup_iters=500, warmup=None):
        self.optimizer = optimizer
        self.warmup_ratio = warmup_ratio
        self.warmup_iters = warmup_iters
        self.warmup = warmup
        self.base_lr = optimizer.lr
        self.base_lr_pg = [pg.get("lr", optimizer.lr) for pg in optimizer.param_groups]
        self.step(0, 0)

    def get_warmup_lr(self, lr, cur_iters):
        if self.warmup == "constant":
            k = self.warmup_ratio
        elif self.warmup == "linear":
            k = 1 - (1 - cur_iters / self.warmup_iters) * (1 - self.warmup_ratio)
        elif self.warmup == "exp":
            k = self.warmup_ratio ** (1 - cur_iters / self.warmup_iters)
        return k * lr

    def get_lr(self, lr, steps):
        return lr

    def _update_lr(self, steps, get_lr_func):
        self.optimizer.lr = get_lr_func(self.base_lr, steps)
        for i, param_group in enumerate(self.optimizer.param_groups):
            param_group["lr"] = get_lr_func(self.base_lr_pg[i], steps)

    def step(self, iters, epochs, by_epoch=True):
        if self.warmup is not None:
            if iters >= self.warmup_iters:
                if by_epoch:
                    self._update_lr(epochs, self.get_lr)
                else:
                    self._update_lr(iters - self.warmup_iters, self.get_lr)
            else:
                self._update_lr(iters, self.get_warmup_lr)
        else:
            self._update_lr(epochs if by_epoch else iters, self.get_lr)

    def parameters(self):
        return {key: value for key, value in self.__dict__.items() if key != "optimizer"}

    def load_parameters(self, data):
        if isinstance(data, dict):
            for k, d in data.items():
                if k in self.__dict__:
                    self.__dict__[k] = d


@SCHEDULERS.register_module()
class StepLR(WarmUpLR):
    def __init__(self, milestones, gamma=0.1, min_lr=None, **kwargs):
        if isinstance(milestones, list):
            assert all([s > 0 for s in milestones])
        elif isinstance(milestones, int):
            assert milestones > 0
        else:
            raise TypeError('"step" must be a list or integer')
        self.milestones = milestones
        self.gamma = gamma
        self.min_lr = min_lr
        super().__init__(**kwargs)

    def get_lr(self, base_lr, steps):
        if isinstance(self.milestones, int):
            exp = steps // self.milestones
        else:
            exp = len(self.milestones)
            for i, s in enumerate(self.milestones):
                if steps < s:
                    exp = i
                    break
        lr = base_lr * (self.gamma ** exp)
        if self.min_lr is not None:
            lr = max(lr, self.min_lr)
        return lr
