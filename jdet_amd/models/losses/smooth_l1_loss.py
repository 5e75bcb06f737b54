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
        return self.loss_weight * l1_loss(pred, target, weight, reduction=reduction, avg_factor=avg_factor)

    execute = forward
