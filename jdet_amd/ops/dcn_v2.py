"""DCN v2: modulated deformable convolution and deformable position-sensitive RoI pooling.

Drop-in for python/jdet/ops/dcn_v2.py: `dcn_v2_conv` (DCN_V2_CONV L786-806), `dcn_v2_pooling` (DCN_V2_POOLING
L1177-1212), modules `DeformConv` (L1214-1261), `DCNv2` (L1264-1299), `DCN` (L1302-1334, registered in HEADS),
`DCNv2Pooling` (L1337-1371), `DCNPooling` (L1374-1455): same constructor arguments, parameter names and call
signatures.  The sampling kernels are the gfx950 ones behind include/jdet_hip.h (csrc/deform_nchw.hip with a mask,
csrc/deform_psroi_pool.hip); the GEMMs around them are library GEMMs over the whole batch instead of the reference's
per-image cuBLAS loop.  No CPU fallback.
"""
import math

import torch
from torch import nn
from torch.nn.modules.utils import _pair

from jdet_amd import _lib as L
from jdet_amd.ops.dcn_v1 import _geom_args, _out_hw
from jdet_amd.utils.registry import HEADS

__all__ = ["DCN", "DCNv2", "DeformConv", "DCNv2Pooling", "DCNPooling", "dcn_v2_conv", "dcn_v2_pooling"]


def _im2col(x, off, mask, kh, kw, pad, stride, dil, dg):
    B, C, H, W = x.shape
    Ho, Wo = _out_hw(H, W, kh, kw, pad, stride, dil)
    col = torch.empty((C * kh * kw, B, Ho, Wo), dtype=torch.float32, device=x.device)
    L.check(L.lib().jdet_modulated_deform_im2col(L.ptr(x), L.ptr(off), L.ptr(mask),
                                                 *_geom_args(B, C, H, W, kh, kw, pad, stride, dil, dg),
                                                 L.ptr(col), L.stream_ptr(x)), "jdet_modulated_deform_im2col")
    return col


class DCN_V2_CONV(torch.autograd.Function):
    @staticmethod
    def forward(ctx, input, offset, mask, weight, bias, stride, padding, dilation, deformable_groups):
        L.need_device(input, offset, mask, weight, bias)
        stride, padding, dilation = _pair(stride), _pair(padding), _pair(dilation)
        x, off, m, w = L.f32c(input), L.f32c(offset), L.f32c(mask), L.f32c(weight)
        B, C, H, W = x.shape
        Cout, Cw, kh, kw = w.shape
        if Cw != C:
            raise ValueError("weight expects %d input channels, input has %d" % (Cw, C))
        Ho, Wo = _out_hw(H, W, kh, kw, padding, stride, dilation)
        kk = kh * kw
        if tuple(off.shape) != (B, 2 * deformable_groups * kk, Ho, Wo):
            raise ValueError("offset must be %r, got %r" % ((B, 2 * deformable_groups * kk, Ho, Wo), tuple(off.shape)))
        if tuple(m.shape) != (B, deformable_groups * kk, Ho, Wo):
            raise ValueError("mask must be %r, got %r" % ((B, deformable_groups * kk, Ho, Wo), tuple(m.shape)))
        col = _im2col(x, off, m, kh, kw, padding, stride, dilation, deformable_groups)
        out = torch.mm(w.view(Cout, C * kk), col.view(C * kk, B * Ho * Wo)).view(Cout, B, Ho, Wo).permute(1, 0, 2, 3)
        if bias is not None:
            out = out + L.f32c(bias).view(1, Cout, 1, 1)          # dcn_v2.py:L238-250 (ones x bias GEMM)
        ctx.save_for_backward(x, off, m, w)
        ctx.cfg = (stride, padding, dilation, deformable_groups, bias is not None)
        return out.contiguous()

    @staticmethod
    def backward(ctx, grad_output):
        x, off, m, w = ctx.saved_tensors
        stride, padding, dilation, dg, has_bias = ctx.cfg
        B, C, H, W = x.shape
        Cout, _, kh, kw = w.shape
        kk = kh * kw
        Ho, Wo = _out_hw(H, W, kh, kw, padding, stride, dilation)
        go = L.f32c(grad_output).permute(1, 0, 2, 3).reshape(Cout, B * Ho * Wo)
        lib, st = L.lib(), L.stream_ptr(x)
        geom = _geom_args(B, C, H, W, kh, kw, padding, stride, dilation, dg)
        columns = torch.mm(w.view(Cout, C * kk).t(), go).view(C * kk, B, Ho, Wo)        # L722-730
        grad_offset, grad_mask = torch.empty_like(off), torch.empty_like(m)
        L.check(lib.jdet_modulated_deform_col2im_coord(L.ptr(columns), L.ptr(x), L.ptr(off), L.ptr(m), *geom,
                                                       L.ptr(grad_offset), L.ptr(grad_mask), st),
                "jdet_modulated_deform_col2im_coord")
        grad_input = torch.empty_like(x)
        # L651-653: the reference hands modulated_deformable_col2im_gpu_kernel (pad_h, pad_h): the input gradient samples
        # at w_out * stride - pad_h + ... on the x axis too.  Reproduced for parity (identical to the true gradient for a
        # symmetric padding): the same positions through the true geometry = x offsets shifted by pad_w - pad_h.
        off_gi = off
        if padding[0] != padding[1]:
            off_gi = off.clone()
            off_gi[:, 1::2] += float(padding[1] - padding[0])
        L.check(lib.jdet_modulated_deform_col2im(L.ptr(columns), L.ptr(off_gi), L.ptr(m), *geom, L.ptr(grad_input), st),
                "jdet_modulated_deform_col2im")
        col = _im2col(x, off, m, kh, kw, padding, stride, dilation, dg)
        grad_weight = torch.mm(go, col.view(C * kk, B * Ho * Wo).t()).view_as(w)         # L746-762
        grad_bias = go.sum(1) if has_bias else None                                       # L768-776
        return grad_input, grad_offset, grad_mask, grad_weight, grad_bias, None, None, None, None


def dcn_v2_conv(input, offset, mask, weight, bias, stride, padding, dilation, deformable_groups):
    return DCN_V2_CONV.apply(input, offset, mask, weight, bias, stride, padding, dilation, deformable_groups)


class DCN_V2_POOLING(torch.autograd.Function):
    @staticmethod
    def forward(ctx, input, rois, offset, spatial_scale, pooled_size, output_dim, no_trans, group_size=1,
                part_size=None, sample_per_part=4, trans_std=.0):
        L.need_device(input, rois)
        part_size = pooled_size if part_size is None else part_size
        no_trans = int(bool(no_trans))
        from ._roi_common import to_nhwc
        x, r = to_nhwc(input), L.f32c(rois)          # channels-last memory: a bin's channels are one contiguous vector
        t = None if no_trans else L.f32c(offset)
        N, C, H, W = x.shape
        R = r.shape[0]
        if r.dim() != 2 or r.shape[1] != 5:
            raise ValueError("rois must be (R, 5) [batch, x1, y1, x2, y2], got %r" % (tuple(r.shape),))
        tch = 2 if no_trans else t.shape[1]
        if not no_trans and (t.dim() != 4 or t.shape[0] != R or tuple(t.shape[2:]) != (part_size, part_size)):
            raise ValueError("offset must be (R, 2*classes, part, part), got %r" % (tuple(t.shape),))
        out = torch.empty((R, output_dim, pooled_size, pooled_size), dtype=torch.float32, device=x.device,
                          memory_format=torch.channels_last)       # (R, P, P, output_dim) in memory
        cnt = torch.empty_like(out)
        ctx.args = (N, C, H, W, R, no_trans, float(spatial_scale), int(output_dim), int(group_size), int(pooled_size),
                    int(part_size), int(sample_per_part), float(trans_std), int(tch))
        L.check(L.lib().jdet_deform_psroi_pool_forward(L.ptr(x), L.ptr(r), L.ptr(t), *ctx.args, L.ptr(out), L.ptr(cnt),
                                                       L.stream_ptr(x)), "jdet_deform_psroi_pool_forward")
        ctx.save_for_backward(x, r, t, cnt)
        ctx.trans_shape = None if no_trans else tuple(t.shape)
        return out

    @staticmethod
    def backward(ctx, grad_output):
        x, r, t, cnt = ctx.saved_tensors
        go = grad_output.float().contiguous(memory_format=torch.channels_last)
        gi = torch.empty_like(x)                      # channels-last, as x
        gt = torch.empty(ctx.trans_shape, dtype=torch.float32, device=x.device) if ctx.trans_shape else None
        L.check(L.lib().jdet_deform_psroi_pool_backward(L.ptr(go), L.ptr(cnt), L.ptr(x), L.ptr(r), L.ptr(t), *ctx.args,
                                                        L.ptr(gi), L.ptr(gt), L.stream_ptr(x)),
                "jdet_deform_psroi_pool_backward")
        return gi, None, gt, None, None, None, None, None, None, None, None


def dcn_v2_pooling(input, rois, offset, spatial_scale, pooled_size, output_dim, no_trans, group_size=1,
                   part_size=None, sample_per_part=4, trans_std=.0):
    return DCN_V2_POOLING.apply(input, rois, offset, spatial_scale, pooled_size, output_dim, no_trans, group_size,
                                part_size, sample_per_part, trans_std)


def _uniform_fan_in(weight, in_channels, kernel_size):
    n = in_channels
    for k in kernel_size:
        n *= k
    stdv = 1. / math.sqrt(n)
    nn.init.uniform_(weight, -stdv, stdv)


class DeformConv(nn.Module):
    """dcn_v2.py:L1214-1261: the v1 interface (x, offset) on the v2 op with a mask of ones; `bias=False` keeps a
    constant zero bias (the reference stores a numpy array there: not a parameter)."""

    def __init__(self, in_channels, out_channels, kernel_size, stride=1, padding=0, dilation=1, deformable_groups=1,
                 bias=False):
        super().__init__()
        self.in_channels, self.out_channels = in_channels, out_channels
        self.kernel_size, self.stride = _pair(kernel_size), _pair(stride)
        self.padding, self.dilation = _pair(padding), _pair(dilation)
        self.deformable_groups = deformable_groups
        self.weight = nn.Parameter(torch.zeros(out_channels, in_channels, *self.kernel_size))
        if bias:
            self.bias = nn.Parameter(torch.zeros(out_channels))
        else:
            self.register_buffer("bias", torch.zeros(out_channels), persistent=False)
        self.reset_parameters()

    def reset_parameters(self):
        _uniform_fan_in(self.weight, self.in_channels, self.kernel_size)

    def forward(self, x, offset):
        assert x.size(2) > self.kernel_size[0] and x.size(3) > self.kernel_size[1]
        mask_shape = list(offset.size())
        mask_shape[1] //= 2
        mask = torch.ones(mask_shape, dtype=x.dtype, device=x.device)
        return dcn_v2_conv(x, offset, mask, self.weight, self.bias, self.stride, self.padding, self.dilation,
                           self.deformable_groups)

    execute = forward


class DCNv2(nn.Module):
    """dcn_v2.py:L1264-1299"""

    def __init__(self, in_channels, out_channels, kernel_size, stride, padding, dilation=1, deformable_groups=1):
        super().__init__()
        self.in_channels, self.out_channels = in_channels, out_channels
        self.kernel_size, self.stride = _pair(kernel_size), _pair(stride)
        self.padding, self.dilation = _pair(padding), _pair(dilation)
        self.deformable_groups = deformable_groups
        self.weight = nn.Parameter(torch.zeros(out_channels, in_channels, *self.kernel_size))
        self.bias = nn.Parameter(torch.zeros(out_channels))
        self.reset_parameters()

    def reset_parameters(self):
        nn.init.constant_(self.bias, 0.0)
        _uniform_fan_in(self.weight, self.in_channels, self.kernel_size)

    def forward(self, input, offset, mask):
        assert 2 * self.deformable_groups * self.kernel_size[0] * self.kernel_size[1] == offset.shape[1]
        assert self.deformable_groups * self.kernel_size[0] * self.kernel_size[1] == mask.shape[1]
        return dcn_v2_conv(input, offset, mask, self.weight, self.bias, self.stride, self.padding, self.dilation,
                           self.deformable_groups)

    execute = forward


@HEADS.register_module()
class DCN(DCNv2):
    """dcn_v2.py:L1302-1334: offsets and mask from a zero-initialised conv on the input (o1, o2, mask = chunks of 3)"""

    def __init__(self, in_channels, out_channels, kernel_size, stride, padding, dilation=1, deformable_groups=1):
        super().__init__(in_channels, out_channels, kernel_size, stride, padding, dilation, deformable_groups)
        channels_ = self.deformable_groups * 3 * self.kernel_size[0] * self.kernel_size[1]
        self.conv_offset_mask = nn.Conv2d(self.in_channels, channels_, kernel_size=self.kernel_size,
                                          stride=self.stride, padding=self.padding, bias=True)
        self.init_offset()

    def init_offset(self):
        nn.init.constant_(self.conv_offset_mask.weight, 0.0)
        nn.init.constant_(self.conv_offset_mask.bias, 0.0)

    def forward(self, input):
        out = self.conv_offset_mask(input)
        o1, o2, mask = torch.chunk(out, 3, dim=1)
        offset = torch.cat((o1, o2), dim=1)
        mask = torch.sigmoid(mask)
        return dcn_v2_conv(input, offset, mask, self.weight, self.bias, self.stride, self.padding, self.dilation,
                           self.deformable_groups)

    execute = forward


class DCNv2Pooling(nn.Module):
    """dcn_v2.py:L1337-1371"""

    def __init__(self, spatial_scale, pooled_size, output_dim, no_trans, group_size=1, part_size=None,
                 sample_per_part=4, trans_std=.0):
        super().__init__()
        self.spatial_scale, self.pooled_size, self.output_dim = spatial_scale, pooled_size, output_dim
        self.no_trans, self.group_size = no_trans, group_size
        self.part_size = pooled_size if part_size is None else part_size
        self.sample_per_part, self.trans_std = sample_per_part, trans_std

    def _pool(self, input, rois, offset, no_trans):
        return dcn_v2_pooling(input, rois, offset, self.spatial_scale, self.pooled_size, self.output_dim, no_trans,
                              self.group_size, self.part_size, self.sample_per_part, self.trans_std)

    def forward(self, input, rois, offset):
        assert input.shape[1] == self.output_dim
        if self.no_trans:
            offset = input.new_empty((0,) + tuple(input.shape[1:]))
        return self._pool(input, rois, offset, self.no_trans)

    execute = forward


class DCNPooling(DCNv2Pooling):
    """dcn_v2.py:L1374-1455: plain pooling -> three-layer MLP -> (offset, mask) -> deformable pooling x mask"""

    def __init__(self, spatial_scale, pooled_size, output_dim, no_trans, group_size=1, part_size=None,
                 sample_per_part=4, trans_std=.0, deform_fc_dim=1024):
        super().__init__(spatial_scale, pooled_size, output_dim, no_trans, group_size, part_size, sample_per_part,
                         trans_std)
        self.deform_fc_dim = deform_fc_dim
        if not no_trans:
            self.offset_mask_fc = nn.Sequential(
                nn.Linear(self.pooled_size * self.pooled_size * self.output_dim, self.deform_fc_dim), nn.ReLU(),
                nn.Linear(self.deform_fc_dim, self.deform_fc_dim), nn.ReLU(),
                nn.Linear(self.deform_fc_dim, self.pooled_size * self.pooled_size * 3))
            nn.init.constant_(self.offset_mask_fc[4].weight, 0.0)
            nn.init.constant_(self.offset_mask_fc[4].bias, 0.0)

    def forward(self, input, rois):
        offset = input.new_empty((0,) + tuple(input.shape[1:]))
        if self.no_trans:
            return self._pool(input, rois, offset, self.no_trans)
        n = rois.shape[0]
        roi = self._pool(input, rois, offset, True)
        offset_mask = self.offset_mask_fc(roi.reshape(n, -1)).view(n, 3, self.pooled_size, self.pooled_size)
        o1, o2, mask = torch.chunk(offset_mask, 3, dim=1)
        offset = torch.cat((o1, o2), dim=1)
        mask = torch.sigmoid(mask)
        return self._pool(input, rois, offset, self.no_trans) * mask

    execute = forward
