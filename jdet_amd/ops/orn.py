"""Oriented response network pieces used by S2ANet.  Mirrors python/jdet/ops/orn.py:
`active_rotating_filter` (ARF gather, L260-281 + Function), `ORConv2d` (L620-685),
`RotationInvariantPooling` (L595-617).  (The RIE encode kernels are out of scope: no named config
calls them)
"""
import math
import os

import torch
import torch.nn.functional as F
from torch import nn

from .. import _lib as L
from . import conv_igemm

__all__ = ["ORConv2d", "RotationInvariantPooling", "active_rotating_filter", "arf_forward", "arf_backward"]


def _pair(x):
    return tuple(x) if isinstance(x, (tuple, list)) else (x, x)


def arf_forward(input, indices):
    assert input.dim() == 5, "only supports a batch of ARFs."
    assert input.dtype == torch.float32 and indices.dtype == torch.uint8
    L.need_device(input, indices)
    w, idx = input.contiguous(), indices.contiguous()
    nOut, nIn, nOri, kH, kW = w.shape
    nRot = idx.shape[3]
    out = torch.empty((nOut * nRot, nIn * nOri, kH, kW), dtype=torch.float32, device=w.device)
    L.check(L.lib().jdet_arf_forward(L.ptr(w), L.ptr(idx), nOut, nIn, nOri, kH, kW, nRot, L.ptr(out),
                                     L.stream_ptr(w)), "jdet_arf_forward")
    return out


def arf_backward(indices, grad_output):
    assert indices.dim() == 4 and indices.dtype == torch.uint8 and grad_output.dtype == torch.float32
    L.need_device(indices, grad_output)
    idx, g = indices.contiguous(), grad_output.contiguous()
    nOri, kH, kW, nRot = idx.shape
    nOut = g.shape[0] // nRot
    nIn = g.shape[1] // nOri
    gw = torch.empty((nOut, nIn, nOri, kH, kW), dtype=torch.float32, device=g.device)
    L.check(L.lib().jdet_arf_backward(L.ptr(idx), L.ptr(g), nOut, nIn, nOri, kH, kW, nRot, L.ptr(gw),
                                      L.stream_ptr(g)), "jdet_arf_backward")
    return gw


class _ActiveRotatingFilter(torch.autograd.Function):
    @staticmethod
    def forward(ctx, input, indices):
        ctx.save_for_backward(indices)
        return arf_forward(input, indices)

    @staticmethod
    def backward(ctx, grad_output):
        (indices,) = ctx.saved_tensors
        return arf_backward(indices, grad_output), None


active_rotating_filter = _ActiveRotatingFilter.apply


class _ActiveRotatingFilterCL(torch.autograd.Function):
    """the expanded filter bank written straight in channels-last memory (and its gradient read from it): a
    channels-last convolution then takes the bank as it is -- the library converted a contiguous one inside the
    forward, the data gradient and the weight gradient of every call"""

    @staticmethod
    def forward(ctx, input, indices):
        L.need_device(input, indices)
        w, idx = input.contiguous(), indices.contiguous()
        nOut, nIn, nOri, kH, kW = w.shape
        nRot = idx.shape[3]
        out = torch.empty((nOut * nRot, kH, kW, nIn * nOri), dtype=torch.float32, device=w.device)
        L.check(L.lib().jdet_arf_forward_cl(L.ptr(w), L.ptr(idx), nOut, nIn, nOri, kH, kW, nRot, L.ptr(out),
                                            L.stream_ptr(w)), "jdet_arf_forward_cl")
        ctx.save_for_backward(idx)
        ctx.wshape = tuple(w.shape)
        return out.permute(0, 3, 1, 2)

    @staticmethod
    def backward(ctx, grad_output):
        (idx,) = ctx.saved_tensors
        nOut, nIn, nOri, kH, kW = ctx.wshape
        nRot = idx.shape[3]
        g = L.f32c(grad_output.permute(0, 2, 3, 1))       # free when the gradient is channels-last
        gw = torch.empty(ctx.wshape, dtype=torch.float32, device=g.device)
        L.check(L.lib().jdet_arf_backward_cl(L.ptr(idx), L.ptr(g), nOut, nIn, nOri, kH, kW, nRot, L.ptr(gw),
                                             L.stream_ptr(g)), "jdet_arf_backward_cl")
        return gw, None

# which source tap (1-based) feeds each tap of a 3x3 / 1x1 kernel rotated by k*45 degrees
_KERNEL_INDICES = {
    1: {a: (1,) for a in range(0, 360, 45)},
    3: {
        0: (1, 2, 3, 4, 5, 6, 7, 8, 9),
        45: (2, 3, 6, 1, 5, 9, 4, 7, 8),
        90: (3, 6, 9, 2, 5, 8, 1, 4, 7),
        135: (6, 9, 8, 3, 5, 7, 2, 1, 4),
        180: (9, 8, 7, 6, 5, 4, 3, 2, 1),
        225: (8, 7, 4, 9, 5, 1, 6, 3, 2),
        270: (7, 4, 1, 8, 5, 2, 9, 6, 3),
        315: (4, 1, 2, 7, 5, 3, 8, 9, 6),
    },
}


def arf_indices(n_orientation, n_rotation, kernel_size):
    """uint8 table (nOri, kH, kW, nRot), 1-based flat index into (nOri,kH,kW) (orn.py:L644-678)."""
    kH, kW = kernel_size
    d_or, d_rot = 360 / n_orientation, 360 / n_rotation
    idx = torch.zeros((n_orientation * kH * kW, n_rotation), dtype=torch.uint8)
    for i in range(n_orientation):
        for j in range(kH * kW):
            for k in range(n_rotation):
                angle = d_rot * k
                layer = (i + math.floor(angle / d_or)) % n_orientation
                idx[i * kH * kW + j, k] = int(layer * kH * kW + _KERNEL_INDICES[kW][int(angle)][j])
    return idx.view(n_orientation, kH, kW, n_rotation)


RIP_KERNEL = os.environ.get("JDET_RIP_KERNEL", "1") == "1"     # A/B switch
ARF_CL = os.environ.get("JDET_ARF_CL", "1") == "1"             # A/B switch


class _RipFunction(torch.autograd.Function):
    """max over the orientation channels of a channels-last map as one kernel per direction (the framework's `amax`
    backward is an equality mask, a count, a division and a product: four passes over the 8x larger tensor)"""

    @staticmethod
    def forward(ctx, x, nO):
        N, C, H, W = x.shape
        xn = x.permute(0, 2, 3, 1)                      # (N, H, W, C) contiguous view of a channels-last map
        y = torch.empty((N, H, W, C // nO), dtype=torch.float32, device=x.device)
        L.check(L.lib().jdet_rip_forward(L.ptr(xn), N * H * W, C, nO, L.ptr(y), L.stream_ptr(x)), "jdet_rip_forward")
        ctx.nO = nO
        ctx.save_for_backward(x, y)
        return y.permute(0, 3, 1, 2)

    @staticmethod
    def backward(ctx, g):
        x, y = ctx.saved_tensors
        N, C, H, W = x.shape
        gn = L.f32c(g.permute(0, 2, 3, 1))
        gx = torch.empty((N, H, W, C), dtype=torch.float32, device=x.device)
        L.check(L.lib().jdet_rip_backward(L.ptr(x.permute(0, 2, 3, 1)), L.ptr(y), L.ptr(gn), N * H * W, C, ctx.nO,
                                          L.ptr(gx), L.stream_ptr(x)), "jdet_rip_backward")
        return gx.permute(0, 3, 1, 2), None


class RotationInvariantPooling(nn.Module):
    def __init__(self, nInputPlane, nOrientation=8):
        super().__init__()
        self.nInputPlane = nInputPlane
        self.nOrientation = nOrientation
        # the reference keeps this never-applied submodule (orn.py:L602-605); its parameters are in
        # every S2ANet checkpoint, so it stays for state_dict compatibility
        hidden = int(nInputPlane / nOrientation)
        self.conv = nn.Sequential(nn.Conv2d(hidden, nInputPlane, 1, 1), nn.BatchNorm2d(nInputPlane))
        # never applied -> never receives a gradient; frozen so data-parallel wrappers do not wait for it
        for p in self.conv.parameters():
            p.requires_grad_(False)

    def forward(self, x):
        N, c, h, w = x.shape
        if (RIP_KERNEL and x.is_cuda and x.dtype == torch.float32 and self.nOrientation in (4, 8) and c % self.nOrientation == 0
                and x.is_contiguous(memory_format=torch.channels_last) and not torch.is_autocast_enabled()):
            return _RipFunction.apply(x, self.nOrientation)      # one pass forward, one backward (csrc/arf.hip)
        return x.view(N, -1, self.nOrientation, h, w).amax(dim=2)

    execute = forward


class ORConv2d(nn.Conv2d):
    def __init__(self, in_channels, out_channels, kernel_size=3, arf_config=None, stride=1, padding=0,
                 dilation=1, groups=1, bias=True):
        self.nOrientation, self.nRotation = _pair(arf_config)
        assert (math.log(self.nOrientation) + 1e-5) % math.log(2) < 1e-3, "invalid nOrientation {}".format(self.nOrientation)
        assert (math.log(self.nRotation) + 1e-5) % math.log(2) < 1e-3, "invalid nRotation {}".format(self.nRotation)
        super().__init__(in_channels, out_channels, kernel_size, stride, padding, dilation, groups, bias)
        self.register_buffer("indices", arf_indices(self.nOrientation, self.nRotation, self.kernel_size),
                             persistent=False)
        self.weight = nn.Parameter(torch.zeros(out_channels, in_channels, self.nOrientation, *self.kernel_size))
        if bias:
            self.bias = nn.Parameter(torch.zeros(out_channels * self.nRotation))
        self.reset_parameters()

    def reset_parameters(self):
        if not hasattr(self, "nOrientation") or self.weight.dim() != 5:
            return  # called by nn.Conv2d.__init__ before the ARF weight exists
        n = self.in_channels * self.nOrientation
        for k in self.kernel_size:
            n *= k
        nn.init.normal_(self.weight, 0, math.sqrt(2.0 / n))

    def rotate_arf(self, channels_last=False):
        if ARF_CL and channels_last and self.weight.is_cuda and self.weight.dtype == torch.float32:
            return _ActiveRotatingFilterCL.apply(self.weight, self.indices)
        return active_rotating_filter(self.weight, self.indices)

    def forward(self, input):
        cl = input.dim() == 4 and input.is_cuda and input.is_contiguous(memory_format=torch.channels_last)
        bank = self.rotate_arf(cl)
        if (cl and self.bias is not None and input.dtype == torch.float32 and torch.is_grad_enabled()
                and self.bias.requires_grad and not torch.is_autocast_enabled()
                and conv_igemm._bias_bwd_supported(bank.shape[0])):
            # the same library convolution, with the bias gradient from the own two-stage column sum: the library's
            # per-channel reduce of the (2, 256, 64, 97) packed gradient alone cost 198 us per step (scripts/copy_sources.py)
            return conv_igemm._ConvBiasAct.apply(input, bank, self.bias, False, self.stride, self.padding, self.dilation,
                                                 self.groups, False)
        return F.conv2d(input, bank, self.bias, self.stride, self.padding, self.dilation, self.groups)

    execute = forward
