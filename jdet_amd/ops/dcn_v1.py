"""Deformable convolution v1 (AlignConv's engine).  Mirrors python/jdet/ops/dcn_v1.py:
`DeformConvFunction` (L559-648), `deform_conv` (L650), `DeformConv` (L652-696).

y = W . im2col_deform(x, offset); offset (N, dg*2*kh*kw, Ho, Wo) ordered (dy,dx) per tap.  The
bilinear gather / scatter kernels are csrc/deform_arf.hip; the dense contraction is a library GEMM
(torch.bmm -> rocBLAS/hipBLASLt), as in the reference (`jt.matmul`, dcn_v1.py:L447,L490,L547).
"""
import math

import torch
from torch import nn

from .. import _lib as L

__all__ = ["DeformConv", "deform_conv", "DeformConvFunction"]


def _pair(x):
    return tuple(x) if isinstance(x, (tuple, list)) else (x, x)


def _out_hw(H, W, kh, kw, pad, stride, dil):
    Ho = (H + 2 * pad[0] - (dil[0] * (kh - 1) + 1)) // stride[0] + 1
    Wo = (W + 2 * pad[1] - (dil[1] * (kw - 1) + 1)) // stride[1] + 1
    return Ho, Wo


def _geom_args(B, C, H, W, kh, kw, pad, stride, dil, dg):
    return (B, C, H, W, kh, kw, pad[0], pad[1], stride[0], stride[1], dil[0], dil[1], dg)


def deformable_im2col(x, offset, kh, kw, pad, stride, dil, dg):
    """x (B,C,H,W), offset (B,dg*2*kh*kw,Ho,Wo) -> columns (C*kh*kw, B, Ho, Wo)"""
    B, C, H, W = x.shape
    Ho, Wo = _out_hw(H, W, kh, kw, pad, stride, dil)
    col = torch.empty((C * kh * kw, B, Ho, Wo), dtype=torch.float32, device=x.device)
    L.check(L.lib().jdet_deform_im2col(L.ptr(x), L.ptr(offset), *_geom_args(B, C, H, W, kh, kw, pad, stride, dil, dg),
                                       L.ptr(col), L.stream_ptr(x)), "jdet_deform_im2col")
    return col


def deformable_col2im(col, offset, im_shape, kh, kw, pad, stride, dil, dg):
    B, C, H, W = im_shape
    gim = torch.empty((B, C, H, W), dtype=torch.float32, device=col.device)
    L.check(L.lib().jdet_deform_col2im(L.ptr(col), L.ptr(offset), *_geom_args(B, C, H, W, kh, kw, pad, stride, dil, dg),
                                       L.ptr(gim), L.stream_ptr(col)), "jdet_deform_col2im")
    return gim


def deformable_col2im_coord(col, x, offset, kh, kw, pad, stride, dil, dg):
    B, C, H, W = x.shape
    goff = torch.empty_like(offset)
    L.check(L.lib().jdet_deform_col2im_coord(L.ptr(col), L.ptr(x), L.ptr(offset),
                                             *_geom_args(B, C, H, W, kh, kw, pad, stride, dil, dg),
                                             L.ptr(goff), L.stream_ptr(col)), "jdet_deform_col2im_coord")
    return goff


class DeformConvFunction(torch.autograd.Function):
    @staticmethod
    def forward(ctx, input, offset, weight, stride=1, padding=0, dilation=1, groups=1, deformable_groups=1,
                im2col_step=64):
        if input is not None and input.dim() != 4:
            raise ValueError("Expected 4D tensor as input, got {}D tensor instead.".format(input.dim()))
        L.need_device(input, offset, weight)
        stride, padding, dilation = _pair(stride), _pair(padding), _pair(dilation)
        x, off, w = L.f32c(input), L.f32c(offset), L.f32c(weight)
        B, Cin, H, W = x.shape
        Cout, Cin_g, kh, kw = w.shape
        Ho, Wo = _out_hw(H, W, kh, kw, padding, stride, dilation)
        if not (Ho > 0 and Wo > 0):
            raise ValueError("convolution input is too small (output would be {}x{}x{}x{})".format(B, Cout, Ho, Wo))
        assert off.shape[0] == B, "invalid batch size of offset"
        step = min(im2col_step, B)
        assert B % step == 0, "im2col step must divide batchsize"
        ctx.cfg = (stride, padding, dilation, groups, deformable_groups, step)
        ctx.save_for_backward(x, off, w)
        out = torch.empty((B, Cout, Ho, Wo), dtype=torch.float32, device=x.device)
        wg = w.view(groups, Cout // groups, Cin_g * kh * kw)
        for e in range(B // step):
            xs, os_ = x[e * step:(e + 1) * step], off[e * step:(e + 1) * step]
            col = deformable_im2col(xs, os_, kh, kw, padding, stride, dilation, deformable_groups)
            o = torch.bmm(wg, col.view(groups, Cin_g * kh * kw, step * Ho * Wo))
            out[e * step:(e + 1) * step] = o.view(Cout, step, Ho, Wo).permute(1, 0, 2, 3)
        return out

    @staticmethod
    def backward(ctx, grad_output):
        x, off, w = ctx.saved_tensors
        stride, padding, dilation, groups, dg, step = ctx.cfg
        B, Cin, H, W = x.shape
        Cout, Cin_g, kh, kw = w.shape
        Ho, Wo = _out_hw(H, W, kh, kw, padding, stride, dilation)
        go = L.f32c(grad_output)
        wg = w.view(groups, Cout // groups, Cin_g * kh * kw)
        need_x, need_off, need_w = ctx.needs_input_grad[0], ctx.needs_input_grad[1], ctx.needs_input_grad[2]
        grad_input = torch.empty_like(x) if need_x else None
        grad_offset = torch.empty_like(off) if need_off else None   # AlignConv offsets are detached: skipped
        grad_weight = torch.zeros_like(wg) if need_w else None
        for e in range(B // step):
            sl = slice(e * step, (e + 1) * step)
            xs, os_ = x[sl], off[sl]
            g = go[sl].permute(1, 0, 2, 3).reshape(groups, Cout // groups, step * Ho * Wo)
            if need_x or need_off:
                columns = torch.bmm(wg.transpose(1, 2), g).view(Cin * kh * kw, step, Ho, Wo)
                if need_off:
                    grad_offset[sl] = deformable_col2im_coord(columns, xs, os_, kh, kw, padding, stride, dilation, dg)
                if need_x:
                    grad_input[sl] = deformable_col2im(columns, os_, xs.shape, kh, kw, padding, stride, dilation, dg)
            if need_w:
                col = deformable_im2col(xs, os_, kh, kw, padding, stride, dilation, dg)
                grad_weight += torch.bmm(g, col.view(groups, Cin_g * kh * kw, step * Ho * Wo).transpose(1, 2))
        return (grad_input, grad_offset, grad_weight.view_as(w) if need_w else None, None, None, None, None, None,
                None)


def deform_conv(input, offset, weight, stride=1, padding=0, dilation=1, groups=1, deformable_groups=1,
                im2col_step=64):
    return DeformConvFunction.apply(input, offset, weight, stride, padding, dilation, groups,
                                    deformable_groups, im2col_step)


class DeformConv(nn.Module):
    def __init__(self, in_channels, out_channels, kernel_size, stride=1, padding=0, dilation=1, groups=1,
                 deformable_groups=1, bias=False):
        super().__init__()
        assert not bias
        assert in_channels % groups == 0, "in_channels {} cannot be divisible by groups {}".format(in_channels, groups)
        assert out_channels % groups == 0, "out_channels {} cannot be divisible by groups {}".format(out_channels, groups)
        self.in_channels = in_channels
        self.out_channels = out_channels
        self.kernel_size = _pair(kernel_size)
        self.stride = _pair(stride)
        self.padding = _pair(padding)
        self.dilation = _pair(dilation)
        self.groups = groups
        self.deformable_groups = deformable_groups
        self.weight = nn.Parameter(torch.zeros(out_channels, in_channels // self.groups, *self.kernel_size))
        self.reset_parameters()

    def reset_parameters(self):
        n = self.in_channels
        for k in self.kernel_size:
            n *= k
        stdv = 1.0 / math.sqrt(n)
        nn.init.uniform_(self.weight, -stdv, stdv)

    def forward(self, x, offset):
        return deform_conv(x, offset, self.weight, self.stride, self.padding, self.dilation, self.groups,
                           self.deformable_groups)

    execute = forward
