"""Inference-mode BatchNorm fused with ReLU / the residual add (csrc/frozen_bn.hip).

The reference keeps every backbone BatchNorm in eval mode while training (resnet.py:L177-185) with the affine
parameters trainable outside the frozen stages; a bottleneck is conv -> bn -> relu, conv -> bn -> relu,
conv -> bn -> (+identity) -> relu (resnet.py:L61-93).  `frozen_bn_act(x, bn, residual, relu)` computes one such
chain in a single channels-last pass forward and a single pass backward (per-channel sums fused).  It applies when
the BatchNorm is in eval mode and x is a channels-last fp32 HIP tensor with a supported channel count; in every other
case (BatchNorm in training mode, other layouts / dtypes) the framework ops run, exactly as written in the reference.
"""
import torch
import torch.nn.functional as F

from .. import _lib as L

__all__ = ["frozen_bn_act", "FrozenBNActFunction"]


def _supported_channels(C):
    q = C // 4
    return C % 4 == 0 and ((q <= 256 and 256 % q == 0) or (256 < q <= 1024))


def _fusable(x, bn, residual):
    if bn.training or not x.is_cuda or x.dtype != torch.float32 or x.dim() != 4:
        return False
    if bn.running_mean is None or not _supported_channels(x.shape[1]):
        return False
    if not x.is_contiguous(memory_format=torch.channels_last):
        return False
    if residual is not None and (residual.shape != x.shape or residual.dtype != torch.float32 or
                                 not residual.is_contiguous(memory_format=torch.channels_last)):
        return False
    return torch.is_autocast_enabled() is False


class FrozenBNActFunction(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, residual, weight, bias, mean, var, eps, relu):
        N, C, H, W = x.shape
        P = N * H * W
        y = torch.empty_like(x)    # preserves channels-last
        L.check(L.lib().jdet_frozen_bn_act_forward(L.ptr(x), L.ptr(residual), P, C, L.ptr(weight), L.ptr(bias),
                                                   L.ptr(mean), L.ptr(var), float(eps), int(relu), L.ptr(y),
                                                   L.stream_ptr(x)), "jdet_frozen_bn_act_forward")
        affine = weight is not None and (weight.requires_grad or (bias is not None and bias.requires_grad))
        ctx.cfg = (float(eps), bool(relu), residual is not None, affine)
        ctx.save_for_backward(x if affine else None, y if relu else None, weight, bias, mean, var)
        return y

    @staticmethod
    def backward(ctx, grad_y):
        x, y, weight, bias, mean, var = ctx.saved_tensors
        eps, relu, has_res, affine = ctx.cfg
        g = grad_y if grad_y.is_contiguous(memory_format=torch.channels_last) else \
            grad_y.contiguous(memory_format=torch.channels_last)
        N, C, H, W = g.shape
        P = N * H * W
        need_x, need_res = ctx.needs_input_grad[0], has_res and ctx.needs_input_grad[1]
        gx = torch.empty_like(g)
        gres = torch.empty_like(g) if need_res else None
        gw = torch.empty_like(weight) if affine else None
        gb = torch.empty_like(weight) if affine else None
        wsb = L.lib().jdet_frozen_bn_act_backward_workspace(P, C) if affine else 0
        ws = torch.empty((max(wsb, 4),), dtype=torch.uint8, device=g.device) if affine else None
        L.check(L.lib().jdet_frozen_bn_act_backward(L.ptr(g), L.ptr(y), L.ptr(x), P, C, L.ptr(weight), L.ptr(bias),
                                                    L.ptr(mean), L.ptr(var), eps, int(relu), L.ptr(gx), L.ptr(gres),
                                                    L.ptr(gw), L.ptr(gb), L.ptr(ws), wsb, L.stream_ptr(g)),
                "jdet_frozen_bn_act_backward")
        return (gx if need_x else None), gres, (gw if affine and ctx.needs_input_grad[2] else None), \
            (gb if affine and bias is not None and ctx.needs_input_grad[3] else None), None, None, None, None


def frozen_bn_act(x, bn, residual=None, relu=True):
    """act(bn(x) (+ residual)) for a torch.nn.BatchNorm2d `bn`"""
    if _fusable(x, bn, residual):
        return FrozenBNActFunction.apply(x, residual, bn.weight, bn.bias, bn.running_mean, bn.running_var, bn.eps,
                                         relu)
    out = bn(x)
    if residual is not None:
        out = out + residual
    return F.relu(out) if relu else out
