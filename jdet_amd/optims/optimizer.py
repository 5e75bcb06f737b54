"""SGD with gradient clipping.  Mirrors python/jdet/optims/optimizer.py:L8-36: `step(loss)` =
backward -> (data-parallel gradient all-reduce, inside Jittor there; DDP's bucketed RCCL all-reduce
overlapped with backward here) -> clip_grad_norm(**grad_clip) -> momentum SGD with weight decay."""
import os

import torch

from jdet_amd.utils.registry import OPTIMS

# Device parameters: ONE multi-tensor kernel reads (p, g, momentum) and writes (p, momentum) -- the foreach form makes
# three passes (weight decay, momentum, update) -- and the clip coefficient rides along as the kernel's gradient scale
# instead of a separate pass over the gradients.  JDET_FUSED_SGD=0: the foreach form everywhere.
FUSED = os.environ.get("JDET_FUSED_SGD", "1") == "1"


@OPTIMS.register_module()
class SGD(torch.optim.SGD):
    def __init__(self, params, lr, momentum=0, weight_decay=0, dampening=0, nesterov=False, grad_clip=None):
        params = [p for p in params if p.requires_grad]
        self.fused_step = bool(FUSED and params and all(p.is_cuda and p.dtype == torch.float32 for p in params))
        kind = dict(fused=True) if self.fused_step else dict(foreach=True)
        super().__init__(params, lr=lr, momentum=momentum, weight_decay=weight_decay, dampening=dampening,
                         nesterov=nesterov, **kind)
        self.grad_clip = dict(grad_clip) if grad_clip is not None else None
        self.lr = lr

    def step(self, loss=None):
        """`optimizer.step(loss)` as in the reference runner (runner.py:L127)."""
        if loss is not None:
            self.zero_grad(set_to_none=True)
            loss.backward()
        if self.grad_clip is not None:
            params = [p for g in self.param_groups for p in g["params"] if p.grad is not None]
            norm_type = float(self.grad_clip.get("norm_type", 2))
            fused = self.fused_step and all(g.get("fused") for g in self.param_groups)   # (a loaded state may say otherwise)
            fused = fused and self._seed_late_momentum()
            if fused and params and norm_type == 2.0 and not any(p.grad.is_sparse for p in params):
                # clip_grad_norm_'s coefficient (same formula), applied inside the update kernel: it DIVIDES the
                # gradients by `grad_scale` (and writes them back), so the scale is the reciprocal
                norms = torch._foreach_norm([p.grad for p in params], 2.0)
                total = torch.linalg.vector_norm(torch.stack(norms), 2.0)
                coef = torch.clamp(self.grad_clip["max_norm"] / (total + 1e-6), max=1.0)
                self.grad_scale = torch.reciprocal(coef).to(torch.float32)
                try:
                    return super().step()
                finally:
                    del self.grad_scale
            torch.nn.utils.clip_grad_norm_(params, max_norm=self.grad_clip["max_norm"], norm_type=norm_type,
                                           foreach=True)
        elif self.fused_step:
            self._seed_late_momentum()
        super().step()

    def _seed_late_momentum(self):
        """torch's fused multi-tensor SGD takes "no momentum buffer yet" as a property of the whole group: a parameter
        whose FIRST gradient arrives after the others already hold buffers (a conditionally used branch, a partial
        state resume) makes it raise.  Such a parameter gets a zero buffer here -- `momentum * 0 + d` is the first-step
        rule `buf = d` when dampening is 0; with dampening the group drops to the foreach form, which seeds per tensor.
        Returns whether every group can still take the fused kernel."""
        ok = True
        for g in self.param_groups:
            if not g.get("fused") or g["momentum"] == 0:
                continue
            with_grad = [p for p in g["params"] if p.grad is not None]
            late = [p for p in with_grad if self.state[p].get("momentum_buffer") is None]
            if not late or len(late) == len(with_grad):
                continue
            if g["dampening"] == 0:
                for p in late:
                    self.state[p]["momentum_buffer"] = torch.zeros_like(p, memory_format=torch.preserve_format)
            else:
                g["fused"], g["foreach"] = False, True
                ok = False
        return ok

    def cur_lr(self):
        return self.param_groups[0].get("lr", self.lr)

    def parameters(self):
        return self.state_dict()

    def load_parameters(self, data):
        if isinstance(data, dict) and "param_groups" in data:
            self.load_state_dict(data)
