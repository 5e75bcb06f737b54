"""A tower of 3x3 conv + bias + ReLU layers (the FAM / ODM towers of S2ANet, python/jdet/models/roi_heads/
s2anet_head.py:L127-205; ConvModule, models/utils/modules.py:L91-175) as ONE autograd node.

Forward: csrc/conv_igemm.hip per layer (bias / ReLU / gap-row mask on the accumulators), as before.  Backward: the ReLU
mask + bias sum of a layer's output gradient used to be an elementwise pass per layer (`jdet_bias_act_backward`: read g
and y, write g'); inside a tower only the TOP layer still needs it -- the data gradient of layer j+1 runs on
csrc/conv_bn.hip (mode MASK on the flipped / transposed weights), whose epilogue applies [y_j > 0] to its accumulators
and leaves the column sums of the masked gradient = layer j's bias gradient.  Weight gradients: the library's, on the
same masked gradients.  JDET_TOWER_FUSED=0 switches back to one node per layer (A/B).
"""
import os
import weakref

import torch

from jdet_amd import _lib as L
from jdet_amd.ops import conv_bn, conv_igemm

ENABLED = os.environ.get("JDET_TOWER_FUSED", "1") == "1"
_BANKS = weakref.WeakKeyDictionary()       # first conv module of a tower group -> DgradBank


def _plain(conv):
    return (type(conv) is torch.nn.Conv2d and conv_igemm._is_igemm_conv(conv) and conv.bias is not None
            and conv.padding_mode == "zeros")


def applicable(modules, x):
    """modules: ConvModule list of a tower; x its (N, C, H, W) input"""
    if not (ENABLED and conv_bn.ENABLED and len(modules) >= 2 and x.is_cuda and x.dtype == torch.float32
            and torch.is_grad_enabled() and not torch.is_autocast_enabled() and conv_igemm.TRAIN
            and conv_igemm.BIAS_ACT_BWD):
        return False
    for m in modules:
        conv = getattr(m, "conv", None)
        if conv is None or getattr(m, "with_norm", True) or not getattr(m, "with_activation", False):
            return False
        if type(m.activate) is not torch.nn.ReLU or m.order.index("conv") > m.order.index("act") or not _plain(conv):
            return False
        if not (conv.weight.requires_grad and conv.bias.requires_grad and conv.weight.dtype == torch.float32):
            return False
        if conv.out_channels % 16 or not conv_igemm._bias_bwd_supported(conv.out_channels):
            return False
        if not conv.weight.permute(0, 2, 3, 1).is_contiguous():
            return False
    return conv_igemm.preferred(x, modules[0].conv.weight)


def prepare(conv_lists):
    """once per training step: the data-gradient weights of every tower conv of a head, one launch"""
    convs = [m.conv for ms in conv_lists for m in ms if hasattr(m, "conv")]
    if not convs:
        return
    bank = _BANKS.get(convs[0])
    if bank is None or [id(c) for c in bank.convs] != [id(c) for c in convs]:
        bank = conv_bn.DgradBank(convs)
        for c in convs:
            _BANKS[c] = bank
    bank.refresh()


def _bank(conv):
    bank = _BANKS.get(conv)
    if bank is None:
        bank = _BANKS[conv] = conv_bn.DgradBank([conv])
    return bank


def _column_sums(sums):
    """first halves of the partial rows (rows, 2, C) -> (C,)"""
    C = sums.shape[2]
    out = torch.empty((C,), dtype=torch.float32, device=sums.device)
    jobs = (L.BnSumsJob * 1)()
    jobs[0] = L.BnSumsJob(sums.data_ptr(), sums.shape[0], C, None, None, out.data_ptr())
    L.check(L.lib().jdet_bn_sums_finish(jobs, 1, L.stream_ptr(sums)), "jdet_bn_sums_finish")
    return out


class _TowerFunction(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, rowmask, convs, *wb):
        h = L.f32c(x.permute(0, 2, 3, 1))
        acts = []
        for k in range(len(convs)):
            h = conv_igemm.conv3x3_nhwc(h, conv_igemm.weight_krsc(wb[2 * k]), wb[2 * k + 1], True, rowmask)
            acts.append(h)
        ctx.convs = convs
        ctx.save_for_backward(x, *acts)
        return h.permute(0, 3, 1, 2)

    @staticmethod
    def backward(ctx, gout):
        convs = ctx.convs
        x, *acts = ctx.saved_tensors
        n = len(convs)
        grads = [None] * (2 * n)
        # top layer: the one elementwise pass (mask + bias sum)
        gp, gb = conv_igemm.bias_act_backward(gout, acts[-1].permute(0, 3, 1, 2), True)
        gp = L.f32c(gp.permute(0, 2, 3, 1))
        grads[2 * n - 1] = gb
        gx = None
        for j in range(n - 1, -1, -1):
            inp = acts[j - 1] if j > 0 else L.f32c(x.permute(0, 2, 3, 1))
            conv = convs[j]
            grads[2 * j] = torch.ops.aten.convolution_backward(gp.permute(0, 3, 1, 2), inp.permute(0, 3, 1, 2), conv.weight,
                                                               None, [1, 1], [1, 1], [1, 1], False, [0, 0], 1,
                                                               [False, True, False])[1]
            if j > 0:
                # data gradient of layer j with layer j-1's ReLU mask and bias sum in the epilogue
                gp, sums = conv_bn.conv_bn_nhwc(gp, _bank(conv).get(conv), 1, None, mode=L.EPI_MASK, act=acts[j - 1],
                                                want_sums=True)
                grads[2 * j - 1] = _column_sums(sums)
            elif ctx.needs_input_grad[0]:
                if conv.in_channels % 16 == 0 and conv.out_channels % 16 == 0:
                    gx = conv_bn.conv_bn_nhwc(gp, _bank(conv).get(conv), 1, None).permute(0, 3, 1, 2)
                else:
                    gx = torch.ops.aten.convolution_backward(gp.permute(0, 3, 1, 2), inp.permute(0, 3, 1, 2), conv.weight,
                                                             None, [1, 1], [1, 1], [1, 1], False, [0, 0], 1,
                                                             [True, False, False])[0]
        return (gx, None, None, *grads)


def tower(modules, x, rowmask=None):
    """relu(conv_k(... relu(conv_1(x)))) [* rowmask after every layer] for ConvModules that `applicable` accepted"""
    convs = [m.conv for m in modules]
    wb = [t for c in convs for t in (c.weight, c.bias)]
    return _TowerFunction.apply(x, rowmask, convs, *wb)
