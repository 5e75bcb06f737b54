"""optims/optimizer.py on the host: `step(loss)` = backward -> clip_grad_norm -> momentum SGD with weight decay
(python/jdet/optims/optimizer.py:L8-36), checked against a hand-written update; host parameters take the foreach form
(the fused multi-tensor form is for device parameters and is exercised by the GPU suite's train-step tests)."""
import copy

import torch

from jdet_amd.optims.optimizer import SGD


def _manual(params, grads, bufs, lr, momentum, wd, max_norm):
    total = torch.sqrt(sum((g * g).sum() for g in grads))
    coef = torch.clamp(max_norm / (total + 1e-6), max=1.0)
    out = []
    for i, (p, g) in enumerate(zip(params, grads)):
        d = g * coef + wd * p
        bufs[i] = d.clone() if bufs[i] is None else momentum * bufs[i] + d
        out.append(p - lr * bufs[i])
    return out


def test_sgd_step_with_clip_matches_the_formula_over_three_steps():
    torch.manual_seed(0)
    ps = [torch.nn.Parameter(torch.randn(5, 3)), torch.nn.Parameter(torch.randn(7))]
    frozen = torch.nn.Parameter(torch.randn(2), requires_grad=False)
    opt = SGD(ps + [frozen], lr=0.05, momentum=0.9, weight_decay=1e-2, grad_clip=dict(max_norm=0.5, norm_type=2))
    assert not opt.fused_step                      # host parameters: foreach form
    assert all(p is not frozen for g in opt.param_groups for p in g["params"])
    ref = [p.detach().clone() for p in ps]
    bufs = [None, None]
    for step in range(3):
        x = torch.randn(5, 3)
        loss = ((ps[0] * x).sum() * 3 + (ps[1] ** 2).sum()) * (step + 1)
        rp = [r.clone().requires_grad_(True) for r in ref]
        rl = ((rp[0] * x).sum() * 3 + (rp[1] ** 2).sum()) * (step + 1)
        grads = torch.autograd.grad(rl, rp)
        ref = _manual(ref, grads, bufs, 0.05, 0.9, 1e-2, 0.5)
        opt.step(loss)
        for p, r in zip(ps, ref):
            assert torch.allclose(p.detach(), r, rtol=1e-6, atol=1e-7)
    assert abs(opt.cur_lr() - 0.05) < 1e-12


def test_state_round_trip_keeps_the_step_form():
    torch.manual_seed(1)
    p = torch.nn.Parameter(torch.randn(4))
    opt = SGD([p], lr=0.1, momentum=0.9, grad_clip=dict(max_norm=35, norm_type=2))
    opt.step((p ** 2).sum())
    state = copy.deepcopy(opt.parameters())      # (a checkpoint: the live state dict shares its tensors with `opt`)
    q = torch.nn.Parameter(p.detach().clone())
    opt2 = SGD([q], lr=0.1, momentum=0.9, grad_clip=dict(max_norm=35, norm_type=2))
    opt2.load_parameters(state)
    opt.step((p ** 2).sum())
    opt2.step((q ** 2).sum())
    assert torch.allclose(p.detach(), q.detach())
