"""Deformable convolution v1 (AlignConv's engine).  Mirrors python/jdet/ops/dcn_v1.py:
`DeformConvFunction` (L559-648), `deform_conv` (L650), `DeformConv` (L652-696).

y = W . im2col_deform(x, offset); offset (N, dg*2*kh*kw, Ho, Wo) ordered (dy,dx) per tap.  The dense
contraction is a library GEMM (rocBLAS/hipBLASLt), as in the reference (`jt.matmul`, dcn_v1.py:L447,
L490,L547); the bilinear gather / scatter around it is hand-written:

* channels-last path (groups = 1, deformable_groups = 1, Cin % 4 == 0 -- every AlignConv): csrc/
  deform_nhwc.hip.  x NHWC, columns (B*Ho*Wo, kh*kw, Cin), so out_nhwc = cols @ Wt^T, grad_cols =
  grad_out_nhwc @ Wt and grad_Wt = grad_out_nhwc^T @ cols are plain row-major GEMMs with no layout
  copies, and the input gradient is a sorted gather (no fp atomics).
* general path (groups / deformable groups / offset gradient): csrc/deform_nchw.hip, the reference's
  NCHW column layout.
Both compute the same per-element arithmetic.

Training without the column matrix (optional: JDET_DCN_FUSED_TRAIN_MIN_POS; 3x3 / stride 1 / pad 1, offsets without
gradient, maps of at least that many positions): forward = csrc/conv_igemm.hip with a gathered A operand, weight gradient =
csrc/conv_wgrad.hip with a gathered B operand; only grad_input still forms grad_cols for the sorted gather.
"""
import math
import os

import torch
import torch.nn.functional as F
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


def deformable_im2col_nhwc(x_nhwc, offset, kh, kw, pad, stride, dil):
    """x_nhwc (B,H,W,C) contiguous, offset (B,2*kh*kw,Ho,Wo) -> columns (B*Ho*Wo, kh*kw*C)"""
    B, H, W, C = x_nhwc.shape
    Ho, Wo = _out_hw(H, W, kh, kw, pad, stride, dil)
    cols = torch.empty((B * Ho * Wo, kh * kw * C), dtype=torch.float32, device=x_nhwc.device)
    L.check(L.lib().jdet_deform_im2col_nhwc(L.ptr(x_nhwc), L.ptr(offset),
                                            *_geom_args(B, C, H, W, kh, kw, pad, stride, dil, 1)[:-1],
                                            L.ptr(cols), L.stream_ptr(x_nhwc)), "jdet_deform_im2col_nhwc")
    return cols


def deformable_col2im_nhwc(grad_cols, offset, nhwc_shape, kh, kw, pad, stride, dil):
    """grad_cols (B*Ho*Wo, kh*kw*C) -> grad_x (B,H,W,C)"""
    B, H, W, C = nhwc_shape
    geom = _geom_args(B, C, H, W, kh, kw, pad, stride, dil, 1)[:-1]
    nbytes = L.lib().jdet_deform_col2im_nhwc_workspace(*geom)
    ws = torch.empty((nbytes,), dtype=torch.uint8, device=grad_cols.device)
    gx = torch.empty((B, H, W, C), dtype=torch.float32, device=grad_cols.device)
    L.check(L.lib().jdet_deform_col2im_nhwc(L.ptr(grad_cols), L.ptr(offset), *geom, L.ptr(gx), L.ptr(ws), nbytes,
                                            L.stream_ptr(grad_cols)), "jdet_deform_col2im_nhwc")
    return gx


def _nhwc(t):
    """(B,C,H,W) logical -> (B,H,W,C) contiguous fp32; free when t is already channels-last"""
    return L.f32c(t.permute(0, 2, 3, 1))


# 288 GB of HBM: keep the forward's column matrix for grad_weight instead of re-sampling it (the
# reference recomputes, dcn_v1.py:L541-547).  S2ANet 1024^2 batch 2: ~0.8 GB over both AlignConvs.
SAVE_COLUMNS = True


def _weight_grad(g, cols, split_rows=16384, splits=4):
    """g (M, Cout)^T @ cols (M, K): 256 x 2304 outputs over a reduction of M = B*Ho*Wo rows.  At the finest FPN level
    (M = 32768) the library's single-pass kernel has 36 workgroups for 256 CUs (612 us, 40 % MFMA busy,
    profiles/r02_s2anet_mfma_utilisation.txt); as a 4-way split over M (batched GEMM + sum of the partials) it runs
    311 us.  Smaller M: the plain GEMM is faster (measured 97 vs 110 us at M = 8192)."""
    M = g.shape[0]
    if M >= split_rows and M % splits == 0:
        return torch.bmm(g.view(splits, M // splits, -1).transpose(1, 2), cols.view(splits, M // splits, -1)).sum(0)
    return torch.mm(g.t(), cols)


# Training without the column matrix (3x3 / stride 1 / pad 1, offsets without gradient): maps of at least this many
# positions take the gathered-operand kernels in the forward and the weight gradient.  0 = never (the default: the
# S2ANet step measured 29.39 ms with the finest level fused, 29.53 ms with the three finest, 29.28 ms with none --
# profiles/r04_conv_wgrad.md; what the fused path saves is memory, 0.8 GB of saved columns at 1024^2 batch 2).
FUSED_TRAIN_MIN_POSITIONS = int(os.environ.get("JDET_DCN_FUSED_TRAIN_MIN_POS", "0"))


def _fused_train_ok(x_nhwc, off, weight, kh, kw, stride, padding, dilation):
    from jdet_amd.ops import conv_igemm
    B, H, W, Cin = x_nhwc.shape
    return (FUSED_TRAIN_MIN_POSITIONS > 0 and B * H * W >= FUSED_TRAIN_MIN_POSITIONS
            and (kh, kw, tuple(stride), tuple(padding), tuple(dilation)) == (3, 3, (1, 1), (1, 1), (1, 1))
            and not off.requires_grad and tuple(off.shape) == (B, 18, H, W)
            and conv_igemm.supported(Cin, weight.shape[0]) and conv_igemm.wgrad_supported(Cin, weight.shape[0])
            and B * H * W * max(Cin, weight.shape[0]) < 2 ** 30 - 2 ** 20)


class DeformConvFunction(torch.autograd.Function):
    @staticmethod
    def _forward_nhwc(ctx, input, off, weight, stride, padding, dilation):
        B, Cin, H, W = input.shape
        Cout, _, kh, kw = weight.shape
        Ho, Wo = _out_hw(H, W, kh, kw, padding, stride, dilation)
        x = _nhwc(input)
        wt = L.f32c(weight.permute(0, 2, 3, 1)).view(Cout, kh * kw * Cin)   # K index = tap*Cin + c
        ctx.fused = _fused_train_ok(x, off, weight, kh, kw, stride, padding, dilation)
        if ctx.fused:
            # no column matrix in either direction: the forward gathers its A operand (csrc/conv_igemm.hip), the weight
            # gradient gathers its B operand (csrc/conv_wgrad.hip); only grad_input still goes through columns
            from jdet_amd.ops import conv_igemm
            ctx.save_for_backward(x, off, wt, None, weight)
            return conv_igemm.conv3x3_nhwc(x, wt.view(Cout, kh, kw, Cin), offset=off).permute(0, 3, 1, 2)
        cols = deformable_im2col_nhwc(x, off, kh, kw, padding, stride, dilation)
        out = F.linear(cols, wt)                                              # (B*Ho*Wo, Cout) == NHWC
        keep = SAVE_COLUMNS and weight.requires_grad
        ctx.save_for_backward(x, off, wt, cols if keep else None, None)
        return out.view(B, Ho, Wo, Cout).permute(0, 3, 1, 2)

    @staticmethod
    def _backward_nhwc(ctx, grad_output):
        x, off, wt, cols, weight = ctx.saved_tensors
        stride, padding, dilation = ctx.cfg[:3]
        Cout, Cin, kh, kw = ctx.wshape
        B, H, W, _ = x.shape
        need_x, need_off, need_w = ctx.needs_input_grad[:3]
        if need_off:   # coordinate gradient lives on the general path
            ctx.nhwc = False
            ctx.saved_nchw = (L.f32c(x.permute(0, 3, 1, 2)), off,
                              L.f32c(wt.view(Cout, kh, kw, Cin).permute(0, 3, 1, 2)))
            return DeformConvFunction.backward(ctx, grad_output)
        g = _nhwc(grad_output).view(-1, Cout)
        grad_input = grad_weight = None
        if need_x:
            gcols = torch.mm(g, wt)
            gx = deformable_col2im_nhwc(gcols, off, x.shape, kh, kw, padding, stride, dilation)
            grad_input = gx.permute(0, 3, 1, 2)
        if need_w and ctx.fused:
            from jdet_amd.ops import conv_igemm
            grad_weight = conv_igemm.shared_wgrad(weight, x, g.view(B, H, W, Cout), off)
        elif need_w:
            if cols is None:
                cols = deformable_im2col_nhwc(x, off, kh, kw, padding, stride, dilation)
            grad_weight = _weight_grad(g, cols).view(Cout, kh, kw, Cin).permute(0, 3, 1, 2)
        return grad_input, None, grad_weight, None, None, None, None, None, None

    @staticmethod
    def forward(ctx, input, offset, weight, stride=1, padding=0, dilation=1, groups=1, deformable_groups=1,
                im2col_step=64):
        if input is not None and input.dim() != 4:
            raise ValueError("Expected 4D tensor as input, got {}D tensor instead.".format(input.dim()))
        L.need_device(input, offset, weight)
        stride, padding, dilation = _pair(stride), _pair(padding), _pair(dilation)
        off = L.f32c(offset)
        B, Cin, H, W = input.shape
        Cout, Cin_g, kh, kw = weight.shape
        Ho, Wo = _out_hw(H, W, kh, kw, padding, stride, dilation)
        if not (Ho > 0 and Wo > 0):
            raise ValueError("convolution input is too small (output would be {}x{}x{}x{})".format(B, Cout, Ho, Wo))
        assert off.shape[0] == B, "invalid batch size of offset"
        step = min(im2col_step, B)
        assert B % step == 0, "im2col step must divide batchsize"
        ctx.cfg = (stride, padding, dilation, groups, deformable_groups, step)
        ctx.wshape = tuple(weight.shape)
        ctx.nhwc = groups == 1 and deformable_groups == 1 and Cin % 4 == 0 and Cin_g == Cin
        if ctx.nhwc:
            return DeformConvFunction._forward_nhwc(ctx, input, off, weight, stride, padding, dilation)
        x, w = L.f32c(input), L.f32c(weight)
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
        if ctx.nhwc:
            return DeformConvFunction._backward_nhwc(ctx, grad_output)
        x, off, w = getattr(ctx, "saved_nchw", None) or ctx.saved_tensors
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


def _fused_forward(input, offset, weight, stride, padding, dilation, groups, deformable_groups):
    """No gradient needed (inference): 3x3 / stride 1 / pad 1 deformable conv as ONE implicit GEMM whose A operand is
    gathered bilinearly (ops/conv_igemm.py) -- no column matrix.  Used where it measured faster than im2col + GEMM
    (398 vs 464 us on a 2 x 128^2 x 256 map); None = not applicable."""
    from jdet_amd.ops import conv_igemm
    if conv_igemm.needs_grad(input, weight, offset):
        return None
    if (_pair(stride), _pair(padding), _pair(dilation), groups, deformable_groups) != ((1, 1), (1, 1), (1, 1), 1, 1):
        return None
    if not conv_igemm.preferred(input, weight, conv_igemm.DEFORM_MIN_POSITIONS):
        return None
    if tuple(offset.shape) != (input.shape[0], 18) + tuple(input.shape[-2:]):
        return None
    return conv_igemm.conv3x3(input, weight, offset=L.f32c(offset))


def deform_conv(input, offset, weight, stride=1, padding=0, dilation=1, groups=1, deformable_groups=1,
                im2col_step=64):
    if input is not None and input.dim() == 4 and input.is_cuda:
        y = _fused_forward(input, offset, weight, stride, padding, dilation, groups, deformable_groups)
        if y is not None:
            return y
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
