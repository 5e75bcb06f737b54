"""SGD with gradient clipping.  Mirrors python/jdet/optims/optimizer.py:L8-36: `step(loss)` =
backward -> (data-parallel gradient all-reduce, inside Jittor there; DDP's bucketed RCCL all-reduce
overlapped with backward here) -> clip_grad_norm(**grad_clip) -> momentum SGD with weight decay."""
import torch

from jdet_amd.utils.registry import OPTIMS


@OPTIMS.register_module()
class SGD(torch.optim.SGD):
    def __init__(self, params, lr, momentum=0, weight_decay=0, dampening=0, nesterov=False, grad_clip=None):
        params = [p for p in params if p.requires_grad]
        super().__init__(params, lr=lr, momentum=momentum, weight_decay=weight_decay, dampening=dampening,
                         nesterov=nesterov, foreach=True)
        self.grad_clip = dict(grad_clip) if grad_clip is not None else None
        self.lr = lr

    def step(self, loss=None):
        """`optimizer.step(loss)` as in the reference runner (runner.py:L127)."""
        if loss is not None:
            self.zero_grad(set_to_none=True)
            loss.backward()
        if self.grad_clip is not None:
            params = [p for g in self.param_groups for p in g["params"] if p.grad is not None]
            torch.nn.utils.clip_grad_norm_(params, max_norm=self.grad_clip["max_norm"],
                                           norm_type=self.grad_clip.get("norm_type", 2), foreach=True)
        super().step()

    def cur_lr(self):
        return self.param_groups[0].get("lr", self.lr)

    def parameters(self):
        return self.state_dict()

    def load_parameters(self, data):
        if isinstance(data, dict) and "param_groups" in data:
            self.load_state_dict(data)
