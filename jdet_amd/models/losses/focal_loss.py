"""Sigmoid focal loss.  Mirrors python/jdet/models/losses/focal_loss.py:L5-96: BCE-with-logits in the
max_val-stable form with the log(max(.,1e-10)) floor, one-hot by `(class_index+1) == target`,
weight broadcast along classes, (1-p_t)^gamma, alpha weighting, sum / avg_factor."""
import os

import torch
from torch import nn

from jdet_amd.utils.registry import LOSSES

# one autograd node per (level, loss) with the targets read in place (A/B switch; profiles/r06_glue.md)
LEVEL_NODES = os.environ.get("JDET_LOSS_LEVEL_NODES", "1") == "1"


def binary_cross_entropy_with_logits(output, target, weight=None, pos_weight=None, reduction="none"):
    max_val = torch.clamp(-output, min=0)
    if pos_weight is not None:
        log_weight = (pos_weight - 1) * target + 1
        loss = (1 - target) * output + (
            log_weight * (torch.log(torch.clamp((-max_val).exp() + (-output - max_val).exp(), min=1e-10)) + max_val))
    else:
        loss = (1 - target) * output + max_val + torch.log(
            torch.clamp((-max_val).exp() + (-output - max_val).exp(), min=1e-10))
    if weight is not None:
        loss = loss * weight[:, None]
    if reduction == "mean":
        return loss.mean()
    elif reduction == "sum":
        return loss.sum()
    return loss


def sigmoid_focal_loss(inputs, targets, weight=None, alpha=-1, gamma=2, reduction="none", avg_factor=None):
    cls_index = torch.arange(1, inputs.shape[1] + 1, device=inputs.device, dtype=targets.dtype)
    targets = (cls_index[None, :] == targets[:, None]).to(inputs.dtype)
    p = inputs.sigmoid()
    ce_loss = binary_cross_entropy_with_logits(inputs, targets, weight, reduction="none")
    p_t = p * targets + (1 - p) * (1 - targets)
    loss = ce_loss * ((1 - p_t) ** gamma)
    if alpha >= 0:
        alpha_t = alpha * targets + (1 - alpha) * (1 - targets)
        loss = alpha_t * loss
    if reduction == "mean":
        if avg_factor is None:
            avg_factor = loss.numel()
        loss = loss.sum() / avg_factor
    elif reduction == "sum":
        loss = loss.sum()
    return loss


class _FusedSigmoidFocal(torch.autograd.Function):
    """sum of the focal loss over all (row, class) elements + its gradient in one launch
    (csrc/loss_offset.hip); the autograd backward only scales the stored gradient."""

    @staticmethod
    def forward(ctx, logits, labels, weight, alpha, gamma):
        from jdet_amd import _lib as L
        x = logits.contiguous()
        M, C = x.shape
        lab = labels.to(torch.int32).contiguous()
        w = weight.to(torch.float32).contiguous() if weight is not None else None
        out = torch.empty((), dtype=torch.float32, device=x.device)
        grad = torch.empty_like(x)
        wsb = L.lib().jdet_sigmoid_focal_loss_workspace()
        ws = torch.empty((wsb,), dtype=torch.uint8, device=x.device)
        L.check(L.lib().jdet_sigmoid_focal_loss(L.ptr(x), L.ptr(lab), L.ptr(w), M, C, float(alpha), float(gamma),
                                                out.data_ptr(), L.ptr(grad), L.ptr(ws), wsb, L.stream_ptr(x)),
                "jdet_sigmoid_focal_loss")
        ctx.save_for_backward(grad)
        return out

    @staticmethod
    def backward(ctx, grad_out):
        (grad,) = ctx.saved_tensors
        return grad * grad_out, None, None, None, None


def blocked_rows(t, inner=1):
    """(rows_per_block, block_stride) of `t` -- a (rows,) / (blocks, rows_per_block) array [with `inner` trailing contiguous
    values per row] whose rows of one block are contiguous -- or None.  A level's window [:, s:e] of the per-image target
    arrays is such an array: the level-loss kernels read it in place (include/jdet_hip.h: jdet_*_loss_level)."""
    shape, st = tuple(t.shape), t.stride()
    if inner > 1:
        if not shape or shape[-1] != inner or st[-1] != 1:
            return None
        shape, st = shape[:-1], st[:-1]
        if any(x % inner for x in st):
            return None
        st = tuple(x // inner for x in st)
    if len(shape) == 1:
        return (shape[0], shape[0]) if shape[0] <= 1 or st[0] == 1 else None
    if len(shape) == 2 and (shape[1] <= 1 or st[1] == 1) and st[0] >= 0:
        return (shape[1], st[0]) if shape[0] > 1 else (shape[1], shape[1])
    return None


def device_scalar(x):
    """avg_factor as the 0-dim fp32 device tensor the level-loss kernels divide by (a host number keeps the composed path:
    the framework divides by a host scalar through its reciprocal, another rounding)"""
    return (torch.is_tensor(x) and x.is_cuda and x.dtype == torch.float32 and x.numel() == 1 and not x.requires_grad)


class _FocalLevel(torch.autograd.Function):
    """(sum focal / avg_factor) * loss_weight of one pyramid level as ONE node: labels / weights read through their
    (block, row) windows, the scaling inside the finishing launch, one scaling launch in backward
    (csrc/loss_offset.hip: jdet_sigmoid_focal_loss_level, jdet_loss_grad_scale).  Bit-identical to
    `loss_weight * (_FusedSigmoidFocal(pred, labels.reshape(-1), weights.reshape(-1)) / avg_factor)`."""

    @staticmethod
    def forward(ctx, logits, labels, weight, alpha, gamma, avg_factor, loss_weight):
        from jdet_amd import _lib as L
        x = logits.contiguous()
        M, C = x.shape
        lb = blocked_rows(labels)
        wb = blocked_rows(weight) if weight is not None else (1, 0)
        out = torch.empty((), dtype=torch.float32, device=x.device)
        grad = torch.empty_like(x)
        wsb = L.lib().jdet_sigmoid_focal_loss_workspace()
        ws = torch.empty((wsb,), dtype=torch.uint8, device=x.device)
        L.check(L.lib().jdet_sigmoid_focal_loss_level(
            L.ptr(x), L.ptr(labels), lb[0], lb[1], L.ptr(weight) if weight is not None else None, wb[0], wb[1], M, C,
            float(alpha), float(gamma), L.ptr(avg_factor), float(loss_weight), out.data_ptr(), L.ptr(grad), L.ptr(ws),
            wsb, L.stream_ptr(x)), "jdet_sigmoid_focal_loss_level")
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
        return out, None, None, None, None, None, None


def _level_ok(pred, target, weight, avg_factor):
    return (pred.is_cuda and pred.dtype == torch.float32 and pred.dim() == 2 and pred.numel() > 0 and
            not torch.is_autocast_enabled() and device_scalar(avg_factor) and
            target.dtype == torch.int32 and target.numel() == pred.shape[0] and blocked_rows(target) is not None and
            (weight is None or (weight.dtype == torch.float32 and weight.numel() == pred.shape[0] and
                                blocked_rows(weight) is not None)))


def _fusable(pred, target, weight):
    return (pred.is_cuda and pred.dtype == torch.float32 and pred.dim() == 2 and pred.numel() > 0 and
            target.dim() == 1 and (weight is None or weight.dim() == 1) and not torch.is_autocast_enabled())


@LOSSES.register_module()
class FocalLoss(nn.Module):
    def __init__(self, use_sigmoid=True, gamma=2.0, alpha=0.25, reduction="mean", loss_weight=1.0):
        super().__init__()
        assert use_sigmoid is True, "Only sigmoid focal loss supported now."
        self.use_sigmoid = use_sigmoid
        self.gamma = gamma
        self.alpha = alpha
        self.reduction = reduction
        self.loss_weight = loss_weight

    def forward(self, pred, target, weight=None, avg_factor=None, reduction_override=None):
        assert reduction_override in (None, "none", "mean", "sum")
        reduction = reduction_override if reduction_override else self.reduction
        if reduction == "mean" and LEVEL_NODES and _level_ok(pred, target, weight, avg_factor):
            return _FocalLevel.apply(pred, target, weight, self.alpha, self.gamma, avg_factor, self.loss_weight)
        if target.dim() > 1:      # (a caller that kept its window un-flattened and did not get the level node)
            target = target.reshape(-1)
            weight = weight.reshape(-1) if weight is not None else None
        if reduction in ("mean", "sum") and _fusable(pred, target, weight):
            total = _FusedSigmoidFocal.apply(pred, target, weight, self.alpha, self.gamma)
            if reduction == "mean":
                total = total / (avg_factor if avg_factor is not None else pred.numel())
            return self.loss_weight * total
        return self.loss_weight * sigmoid_focal_loss(pred, target, weight, gamma=self.gamma, alpha=self.alpha,
                                                     reduction=reduction, avg_factor=avg_factor)

    execute = forward
