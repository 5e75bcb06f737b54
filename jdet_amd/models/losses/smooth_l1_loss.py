"""SmoothL1 / L1.  Mirrors python/jdet/models/losses/smooth_l1_loss.py:L5-54 and l1_loss.py."""
import torch
from torch import nn

from jdet_amd.utils.registry import LOSSES


def smooth_l1_loss(pred, target, weight=None, beta=1.0, avg_factor=None, reduction="mean"):
    diff = torch.abs(pred - target)
    if beta != 0.0:
        flag = (diff < beta).to(diff.dtype)
        loss = flag * 0.5 * diff * diff / beta + (1 - flag) * (diff - 0.5 * beta)
    else:
        loss = diff
    if weight is not None:
        if weight.dim() == 1:
            weight = weight[:, None]
        loss = loss * weight
    if avg_factor is None:
        avg_factor = max(loss.shape[0], 1)
    if reduction == "mean":
        loss = loss.sum() / avg_factor
    elif reduction == "sum":
        loss = loss.sum()
    return loss


def l1_loss(pred, target, weight=None, avg_factor=None, reduction="mean"):
    loss = torch.abs(pred - target)
    if weight is not None:
        if weight.dim() == 1:
            weight = weight[:, None]
        loss = loss * weight
    if avg_factor is None:
        avg_factor = max(loss.shape[0], 1)
    if reduction == "mean":
        loss = loss.sum() / avg_factor
    elif reduction == "sum":
        loss = loss.sum()
    return loss


class _FusedSmoothL1(torch.autograd.Function):
    """sum of the weighted smooth-L1 (beta = 0: L1) + its gradient in one launch (csrc/loss_offset.hip)"""

    @staticmethod
    def forward(ctx, pred, target, weight, beta):
        from jdet_amd import _lib as L
        p, t = pred.contiguous(), target.to(torch.float32).contiguous()
        w = None
        if weight is not None:
            w = weight.to(torch.float32)
            if w.dim() == 1 and p.dim() == 2:
                w = w[:, None].expand_as(p)
            w = w.contiguous()
        out = torch.empty((), dtype=torch.float32, device=p.device)
        grad = torch.empty_like(p)
        wsb = L.lib().jdet_sigmoid_focal_loss_workspace()
        ws = torch.empty((wsb,), dtype=torch.uint8, device=p.device)
        L.check(L.lib().jdet_smooth_l1_loss(L.ptr(p), L.ptr(t), L.ptr(w), p.numel(), float(beta), out.data_ptr(),
                                            L.ptr(grad), L.ptr(ws), wsb, L.stream_ptr(p)), "jdet_smooth_l1_loss")
        ctx.save_for_backward(grad)
        return out

    @staticmethod
    def backward(ctx, grad_out):
        (grad,) = ctx.saved_tensors
        return grad * grad_out, None, None, None


class _SmoothL1Level(torch.autograd.Function):
    """(sum smooth-L1 / avg_factor) * loss_weight of one pyramid level as ONE node (see focal_loss._FocalLevel):
    pred (rows, E) contiguous, target / weight (rows, E) or (blocks, rows_per_block, E) windows read in place."""

    @staticmethod
    def forward(ctx, pred, target, weight, beta, avg_factor, loss_weight):
        from jdet_amd import _lib as L
        from .focal_loss import blocked_rows
        p = pred.contiguous()
        rows, E = p.shape
        tb = blocked_rows(target, E)
        wb = blocked_rows(weight, E) if weight is not None else (1, 0)
        out = torch.empty((), dtype=torch.float32, device=p.device)
        grad = torch.empty_like(p)
        wsb = L.lib().jdet_sigmoid_focal_loss_workspace()
        ws = torch.empty((wsb,), dtype=torch.uint8, device=p.device)
        L.check(L.lib().jdet_smooth_l1_loss_level(
            L.ptr(p), L.ptr(target), tb[0], tb[1], L.ptr(weight) if weight is not None else None, wb[0], wb[1], rows, E,
            float(beta), L.ptr(avg_factor), float(loss_weight), out.data_ptr(), L.ptr(grad), L.ptr(ws), wsb,
            L.stream_ptr(p)), "jdet_smooth_l1_loss_level")
        ctx.save_for_backward(grad, avg_factor)
        ctx.loss_weight = float(loss_weight)
        return out

    @staticmethod
    def backward(ctx, grad_out):
        from jdet_amd import _lib as L
        grad, avg = ctx.saved_tensors
        go = grad_out.to(torch.float32).contiguous()
        out = torch.empty_like(grad)
        L.check(L.lib().jdet_loss_grad_scale(L.ptr(grad), grad.numel(), L.ptr(go), L.ptr(avg), ctx.loss_weight,
                                             L.ptr(out), L.stream_ptr(grad)), "jdet_loss_grad_scale")
        return out, None, None, None, None, None


def _level_ok(pred, target, weight, avg_factor):
    from .focal_loss import LEVEL_NODES, blocked_rows, device_scalar
    if not (LEVEL_NODES and pred.is_cuda and pred.dtype == torch.float32 and pred.dim() == 2 and pred.numel() > 0 and
            not torch.is_autocast_enabled() and device_scalar(avg_factor) and not target.requires_grad):
        return False
    E = pred.shape[1]
    for t in (target, weight):
        if t is None:
            continue
        if t.dtype != torch.float32 or t.numel() != pred.numel() or blocked_rows(t, E) is None:
            return False
    return True


def _flat_rows(t, E):
    # (a level's window kept three-dimensional by the caller: the composed paths want rows)
    return t.reshape(-1, E) if t is not None and t.dim() > 2 else t


def _fusable(pred, target, weight):
    return (pred.is_cuda and pred.dtype == torch.float32 and pred.numel() > 0 and pred.shape == target.shape and
            not target.requires_grad and not torch.is_autocast_enabled() and
            (weight is None or weight.shape == pred.shape or (weight.dim() == 1 and pred.dim() == 2 and
                                                              weight.shape[0] == pred.shape[0])))


def _fused_loss(pred, target, weight, beta, reduction, avg_factor):
    total = _FusedSmoothL1.apply(pred, target, weight, beta)
    if reduction == "mean":
        total = total / (avg_factor if avg_factor is not None else max(pred.shape[0], 1))
    return total


@LOSSES.register_module()
class SmoothL1Loss(nn.Module):
    def __init__(self, beta=1.0, reduction="mean", loss_weight=1.0):
        super().__init__()
        self.beta = beta
        self.reduction = reduction
        self.loss_weight = loss_weight

    def forward(self, pred, target, weight=None, avg_factor=None, reduction_override=None):
        assert reduction_override in (None, "none", "mean", "sum")
        reduction = reduction_override if reduction_override else self.reduction
        if reduction == "mean" and _level_ok(pred, target, weight, avg_factor):
            return _SmoothL1Level.apply(pred, target, weight, self.beta, avg_factor, self.loss_weight)
        target, weight = _flat_rows(target, pred.shape[-1]), _flat_rows(weight, pred.shape[-1])
        if reduction in ("mean", "sum") and _fusable(pred, target, weight):
            return self.loss_weight * _fused_loss(pred, target, weight, self.beta, reduction, avg_factor)
        return self.loss_weight * smooth_l1_loss(pred, target, weight, beta=self.beta, reduction=reduction,
                                                 avg_factor=avg_factor)

    execute = forward


@LOSSES.register_module()
class L1Loss(nn.Module):
    def __init__(self, reduction="mean", loss_weight=1.0):
        super().__init__()
        self.reduction = reduction
        self.loss_weight = loss_weight

    def forward(self, pred, target, weight=None, avg_factor=None, reduction_override=None):
        assert reduction_override in (None, "none", "mean", "sum")
        reduction = reduction_override if reduction_override else self.reduction
        if reduction == "mean" and _level_ok(pred, target, weight, avg_factor):
            return _SmoothL1Level.apply(pred, target, weight, 0.0, avg_factor, self.loss_weight)
        target, weight = _flat_rows(target, pred.shape[-1]), _flat_rows(weight, pred.shape[-1])
        if reduction in ("mean", "sum") and _fusable(pred, target, weight):
            return self.loss_weight * _fused_loss(pred, target, weight, 0.0, reduction, avg_factor)
        return self.loss_weight * l1_loss(pred, target, weight, reduction=reduction, avg_factor=avg_factor)

    execute = forward
