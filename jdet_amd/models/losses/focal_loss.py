"""Sigmoid focal loss.  Mirrors python/jdet/models/losses/focal_loss.py:L5-96: BCE-with-logits in the
max_val-stable form with the log(max(.,1e-10)) floor, one-hot by `(class_index+1) == target`,
weight broadcast along classes, (1-p_t)^gamma, alpha weighting, sum / avg_factor."""
import torch
from torch import nn

from jdet_amd.utils.registry import LOSSES


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
        if reduction in ("mean", "sum") and _fusable(pred, target, weight):
            total = _FusedSigmoidFocal.apply(pred, target, weight, self.alpha, self.gamma)
            if reduction == "mean":
                total = total / (avg_factor if avg_factor is not None else pred.numel())
            return self.loss_weight * total
        return self.loss_weight * sigmoid_focal_loss(pred, target, weight, gamma=self.gamma, alpha=self.alpha,
                                                     reduction=reduction, avg_factor=avg_factor)

    execute = forward
