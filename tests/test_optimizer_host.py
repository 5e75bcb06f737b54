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


def _force_fused(opt):
    """host parameters default to the foreach form; torch's fused kernel also runs on the host, which lets the
    fused + clip path (device default) be pinned without a GPU"""
    opt.fused_step = True
    for g in opt.param_groups:
        g["fused"], g["foreach"] = True, False
    return opt


def test_fused_clip_step_equals_foreach_with_a_late_first_gradient():
    """a parameter whose first gradient arrives at step 2 (torch's fused SGD raises on the mixed None / tensor momentum
    list): seeded with a zero buffer, and the whole fused + grad_scale update equals clip_grad_norm_ + foreach"""
    torch.manual_seed(2)
    init = [torch.randn(6, 4), torch.randn(5), torch.randn(3)]
    pa = [torch.nn.Parameter(t.clone()) for t in init]
    pb = [torch.nn.Parameter(t.clone()) for t in init]
    kw = dict(lr=0.05, momentum=0.9, weight_decay=1e-2, grad_clip=dict(max_norm=0.3, norm_type=2))
    fused, plain = _force_fused(SGD(pa, **kw)), SGD(pb, **kw)
    assert not plain.fused_step
    for step in range(4):
        x = torch.randn(6, 4)

        def loss(ps):
            out = (ps[0] * x).sum() * 2 + (ps[1] ** 2).sum()
            if step >= 1:                       # ps[2] takes part from the second step on
                out = out + (ps[2] ** 3).sum() * (step + 1)
            return out
        fused.step(loss(pa))
        plain.step(loss(pb))
        for a, b in zip(pa, pb):
            assert torch.allclose(a.detach(), b.detach(), rtol=1e-6, atol=1e-7), step
    assert all(g["fused"] for g in fused.param_groups)


def test_late_gradient_with_dampening_drops_to_foreach():
    torch.manual_seed(3)
    init = [torch.randn(4), torch.randn(3)]
    pa = [torch.nn.Parameter(t.clone()) for t in init]
    pb = [torch.nn.Parameter(t.clone()) for t in init]
    kw = dict(lr=0.1, momentum=0.8, dampening=0.5, grad_clip=dict(max_norm=10.0, norm_type=2))
    fused, plain = _force_fused(SGD(pa, **kw)), SGD(pb, **kw)
    for step in range(3):
        fused.step((pa[0] ** 2).sum() + ((pa[1] ** 2).sum() if step else 0))
        plain.step((pb[0] ** 2).sum() + ((pb[1] ** 2).sum() if step else 0))
        for a, b in zip(pa, pb):
            assert torch.allclose(a.detach(), b.detach(), rtol=1e-6, atol=1e-7), step
    assert not fused.param_groups[0]["fused"]
