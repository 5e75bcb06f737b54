"""Fully connected layer whose bias gradient is this repo's deterministic per-channel sum.

Reference: the FC layers of the R-CNN heads (python/jdet/models/roi_heads/oriented_head.py:L173-189, L258-283 --
`nn.Linear` there).  The GEMMs stay hipBLASLt through torch.matmul; what changes is grad_bias = sum over the rows of the
incoming gradient: the framework's `reduce_kernel` takes its multi-workgroup path for a (1024, 1024) gradient -- a
scratch buffer plus semaphores cleared by a `hipMemsetAsync` per call.  Under HIP-graph replay that memset node is the
one thing that has failed on this stack before (csrc/common.h), and in round 5 the two FC bias gradients of Oriented
R-CNN came back as garbage (5e4 against 0.2) from EVERY replay once the backbone's allocation pattern changed
(scripts/graph_replay_diag.py); the two-stage sums of csrc/frozen_bn.hip (jdet_bias_act_backward / jdet_channel_sum)
launch plain kernels only, are deterministic, and cost 5 us instead of 30.
"""
import torch
import torch.nn.functional as F
from torch import nn

from jdet_amd import _lib as L


def _vec_supported(c):
    q = c // 4
    return c % 4 == 0 and ((q <= 256 and 256 % q == 0) or 256 < q <= 1024)


def rows_channel_sum(g):
    """g (P, C) contiguous fp32 on the device -> (C,) column sums (two-stage, fixed order)"""
    P, C = g.shape
    lib = L.lib()
    out = torch.empty((C,), dtype=torch.float32, device=g.device)
    if _vec_supported(C):
        nbytes = lib.jdet_frozen_bn_act_backward_workspace(P, C)
        ws = torch.empty((max(nbytes, 4),), dtype=torch.uint8, device=g.device)
        L.check(lib.jdet_bias_act_backward(L.ptr(g), None, P, C, 0, None, L.ptr(out), L.ptr(ws), ws.numel(),
                                           L.stream_ptr(g)), "jdet_bias_act_backward")
    else:
        nbytes = lib.jdet_channel_sum_workspace(P, C)
        ws = torch.empty((max(nbytes, 4),), dtype=torch.uint8, device=g.device)
        L.check(lib.jdet_channel_sum(L.ptr(g), P, C, L.ptr(out), L.ptr(ws), nbytes, L.stream_ptr(g)),
                "jdet_channel_sum")
    return out


class _LinearFunction(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, weight, bias):
        ctx.save_for_backward(x, weight)
        return torch.addmm(bias, x, weight.t())

    @staticmethod
    def backward(ctx, g):
        x, weight = ctx.saved_tensors
        g = g.contiguous()
        gx = g @ weight if ctx.needs_input_grad[0] else None
        gw = g.t() @ x if ctx.needs_input_grad[1] else None
        gb = rows_channel_sum(g) if ctx.needs_input_grad[2] else None
        return gx, gw, gb


def linear(x, weight, bias):
    """F.linear for a 2-D fp32 device input with a bias that needs a gradient; anything else is F.linear itself"""
    if (bias is not None and x.dim() == 2 and x.is_cuda and x.dtype == torch.float32 and weight.dtype == torch.float32
            and torch.is_grad_enabled() and bias.requires_grad and not torch.is_autocast_enabled()
            and 0 < bias.numel() <= 4096 and x.shape[0] > 0 and (_vec_supported(bias.numel()) or bias.numel() <= 256)):
        return _LinearFunction.apply(x, weight, bias)
    return F.linear(x, weight, bias)


class Linear(nn.Linear):
    """nn.Linear (same parameters / state-dict keys) through `linear`"""

    def forward(self, x):
        return linear(x, self.weight, self.bias)
