"""`(lateral + nearest_upsample(top)) / div` as one launch per direction (csrc/upsample_add.hip): the top-down step of
the FPN (python/jdet/models/necks/fpn.py:L160-171)."""
import torch

from .. import _lib as L


class _UpsampleAdd(torch.autograd.Function):
    @staticmethod
    def forward(ctx, lateral, top, div):
        N, C, H, W = lateral.shape
        Ht, Wt = top.shape[2], top.shape[3]
        out = torch.empty_like(lateral, memory_format=torch.channels_last)
        L.check(L.lib().jdet_upsample_add_nhwc_forward(L.ptr(lateral), L.ptr(top), N, C, H, W, Ht, Wt, float(div),
                                                       L.ptr(out), L.stream_ptr(lateral)), "jdet_upsample_add_nhwc_forward")
        ctx.shape = (N, C, H, W, Ht, Wt, float(div))
        return out

    @staticmethod
    def backward(ctx, g):
        N, C, H, W, Ht, Wt, div = ctx.shape
        g = g.contiguous(memory_format=torch.channels_last)
        g_lat = g_top = None
        if ctx.needs_input_grad[0]:
            g_lat = g if div == 1.0 else g / div
        if ctx.needs_input_grad[1]:
            g_top = torch.empty((N, C, Ht, Wt), dtype=g.dtype, device=g.device, memory_format=torch.channels_last)
            L.check(L.lib().jdet_upsample_add_nhwc_backward(L.ptr(g), N, C, H, W, Ht, Wt, div, L.ptr(g_top),
                                                            L.stream_ptr(g)), "jdet_upsample_add_nhwc_backward")
        return g_lat, g_top, None


def fusable(lateral, top):
    """channels-last fp32 device maps with C % 4 == 0 (what the conv stack produces)"""
    return (lateral.is_cuda and top.is_cuda and lateral.dtype == torch.float32 and top.dtype == torch.float32
            and lateral.dim() == 4 and lateral.shape[1] % 4 == 0 and lateral.shape[:2] == top.shape[:2]
            and lateral.is_contiguous(memory_format=torch.channels_last)
            and top.is_contiguous(memory_format=torch.channels_last))


def upsample_add(lateral, top, div=1.0):
    """(lateral + top resampled to lateral's size by the nearest rule) / div"""
    return _UpsampleAdd.apply(lateral, top, div)
