"""The ResNet bottleneck on this repo's own kernels: every convolution with its BatchNorm / identity / ReLU in the
epilogue (csrc/conv_bn.hip), forward and backward.

Reference: Bottleneck.execute (python/jdet/models/backbones/resnet.py:L61-93) under `norm_eval` (L177-185: every
BatchNorm in eval mode, its weight / bias still trainable outside the frozen stages) and the gradients Jittor's autograd
derives for that chain.  One block =

    forward   y1 = relu(bn1(conv1(x)))            conv_bn  1x1
              y2 = relu(bn2(conv2(y1)))           conv_bn  3x3, stride
              id = x | bn_d(conv_d(x))            conv_bn  1x1, stride   (first block of a layer)
              y3 = relu(bn3(conv3(y2)) + id)      conv_bn  1x1, residual in the epilogue
    backward  g3' = g * [y3 > 0] * a3 (+ sums)                      one elementwise pass (jdet_bn_act_backward_from_output)
              g2' = dgrad3(g3') * [y2 > 0] * a2 (+ sums of bn2)     conv_bn, mode MASK
              g1' = dgrad2(g2') * [y1 > 0] * a1 (+ sums of bn1)     conv_bn, mode MASK  (stride 2: the library's data gradient)
              gx  = dgrad1(g1') + g * [y3 > 0]                      conv_bn, mode ADD   (downsample: + its data gradient)
              gW_k = wgrad(g_k', input_k)                           the library's kernels, or (JDET_BOTTLENECK_WGRAD=own)
                                                                    csrc/conv_wgrad.hip into one zero-filled buffer per block
              dgamma_k, dbeta_k                                     one finish launch for the block's BatchNorms

No conv output is ever stored (the normalised input of a BatchNorm is recovered from its activation: xhat = (y - beta) /
gamma wherever y > 0), no BatchNorm pass runs on its own, and the masked / scaled gradients are never formed by a
separate kernel except for the block's output.  A data gradient is the forward kernel on flipped / transposed weights:
`DgradBank` rewrites those for the whole backbone in ONE launch per step.

The fused path applies to channels-last fp32 device tensors with eval-mode BatchNorm2d layers whose parameters either all
train or are all frozen; everything else takes the per-layer path of models/backbones/resnet.py.
JDET_BOTTLENECK_FUSED=0 switches it off (A/B).
Determinism: with the own weight gradients (JDET_BOTTLENECK_WGRAD=own, the default) the K chunks of a layer meet in the
gradient buffer by float atomics, so the backbone's weight gradients differ in the last bits from run to run;
JDET_BOTTLENECK_WGRAD=lib takes the library's kernels instead (bitwise reproducible where the library is).
"""
import ctypes
import os
import struct
import weakref

import torch
from torch import nn

from jdet_amd import _lib as L

ENABLED = os.environ.get("JDET_BOTTLENECK_FUSED", "1") == "1"
# Weight gradients of the fused block: "own" (default) = the general (R, stride) kernel of this repo (csrc/conv_wgrad.hip,
# 64 x 64 tiles) accumulating into ONE zero-filled buffer per backbone and step -- 2-14 us per layer faster than the
# library's at the layer1-3 shapes of a 2 x 1024^2 step and no zero-fill launch per layer (profiles/r05_conv_bn.md);
# "lib" = the library's kernels on the same g' tensors (A/B).
OWN_WGRAD = os.environ.get("JDET_BOTTLENECK_WGRAD", "own") == "own"
# The block's weight gradients on a SIDE stream, concurrent with its data gradients (JDET_BOTTLENECK_WGRAD_STREAM=1): both
# kernel families run at ~60 % MFMA busy with two workgroups per CU on the deep layers, the weight gradient of layer k
# needs only g_k' and the saved input, and nothing reads it before the block's backward returns.  The side stream waits
# for an event behind the kernel that produced g_k'; the main stream waits for the side stream once, at the end of the
# block's backward (the gradients are handed to autograd -- and to DDP's bucket hooks -- only after that).
WGRAD_STREAM = os.environ.get("JDET_BOTTLENECK_WGRAD_STREAM", "0") == "1"
_SIDE = {}           # device index -> the side stream


def _side_stream(device):
    s = _SIDE.get(device.index)
    if s is None:
        s = _SIDE[device.index] = torch.cuda.Stream(device=device)
    return s
# stride-2 3x3 / 1x1 data gradients: the library's (a strided data gradient is a different kernel, not built here)
_PLAN = {}           # (N, H, W, Cin, Cout, R, stride) -> (workspace bytes, sums rows with it)


def out_size(n, R, stride):
    return (n + 2 * (R // 2) - R) // stride + 1


def _bn_params(bn):
    if bn is None:
        return L.BnParams(None, None, None, None, 0.0)
    return L.BnParams(L.ptr(bn.weight), L.ptr(bn.bias), L.ptr(bn.running_mean), L.ptr(bn.running_var), float(bn.eps))


def _plan(N, H, W, Cin, Cout, R, stride):
    key = (N, H, W, Cin, Cout, R, stride)
    hit = _PLAN.get(key)
    if hit is None:
        if len(_PLAN) >= 512:          # (multi-scale training cycles through a bounded set of shapes; keep the cache so)
            _PLAN.pop(next(iter(_PLAN)))
        lib = L.lib()
        ws = lib.jdet_conv_bn_workspace(N, H, W, Cin, Cout, R, stride)
        rows = lib.jdet_conv_bn_sums_rows(N, H, W, Cin, Cout, R, stride, 0, 1 if ws else 0)
        hit = _PLAN[key] = (ws, rows)
    return hit


def weight_krsc(weight):
    """(Cout, Cin, R, R) logical -> (Cout, R, R, Cin) contiguous view (channels-last weights: free)"""
    return L.f32c(weight.permute(0, 2, 3, 1))


def conv_bn_nhwc(x, w_krsc, stride=1, bn=None, residual=None, relu=False, mode=L.EPI_FORWARD, grad_out=None, act=None,
                 want_sums=False, tile=0):
    """One launch of the family.  x (N,H,W,Cin), w_krsc (Cout,R,R,Cin) contiguous fp32 device tensors.
    mode FORWARD: y = [relu]([bn](conv) [+ residual]);  ADD: y = conv + grad_out * [act > 0];
    MASK: g = conv * [act > 0], y = g * a(bn) -> returns (y, sums | None) with sums (rows, 2, Cout) partial column
    sums of g and g * (act - bn.bias) when want_sums.  Other modes return y."""
    L.need_device(x, w_krsc, residual, grad_out, act)
    N, H, W, Cin = x.shape
    Cout, R = w_krsc.shape[0], w_krsc.shape[1]
    if tuple(w_krsc.shape) != (Cout, R, R, Cin):
        raise ValueError("weight %r does not match input channels %d" % (tuple(w_krsc.shape), Cin))
    Ho, Wo = out_size(H, R, stride), out_size(W, R, stride)
    for t in (residual, grad_out, act):
        if t is not None and (tuple(t.shape) != (N, Ho, Wo, Cout) or not t.is_contiguous() or t.dtype != torch.float32):
            raise ValueError("epilogue tensors must be contiguous fp32 (N, Ho, Wo, Cout)")
    if not (x.is_contiguous() and w_krsc.is_contiguous() and x.dtype == torch.float32 and w_krsc.dtype == torch.float32):
        raise ValueError("x / w must be contiguous fp32")
    lib = L.lib()
    y = torch.empty((N, Ho, Wo, Cout), dtype=torch.float32, device=x.device)
    ws_bytes, rows = _plan(N, H, W, Cin, Cout, R, stride) if tile == 0 else (0, 0)
    ws = torch.empty((ws_bytes,), dtype=torch.uint8, device=x.device) if ws_bytes else None
    sums = None
    if mode == L.EPI_MASK and want_sums:
        if tile != 0:
            rows = lib.jdet_conv_bn_sums_rows(N, H, W, Cin, Cout, R, stride, tile, 0)
        sums = torch.empty((rows, 2, Cout), dtype=torch.float32, device=x.device)
    ep = L.ConvEpilogue(mode, 1 if (bn is not None and mode == L.EPI_FORWARD) else 0, 1 if relu else 0, _bn_params(bn),
                        L.ptr(residual), L.ptr(grad_out), L.ptr(act), L.ptr(sums))
    L.check(lib.jdet_conv_bn_forward(L.ptr(x), N, H, W, Cin, L.ptr(w_krsc), Cout, R, stride, ctypes.byref(ep), int(tile),
                                     L.ptr(y), L.ptr(ws), ws_bytes, L.stream_ptr(x)), "jdet_conv_bn_forward")
    return (y, sums) if mode == L.EPI_MASK else y


def conv_wgrad_nhwc(x, gy, R, stride, out, ksplit=0):
    """out (Cout, R, R, Cin) contiguous += weight gradient of conv(x; R, stride, pad R // 2) for the output gradient gy"""
    L.need_device(x, gy, out)
    N, H, W, Cin = x.shape
    Cout = gy.shape[3]
    if tuple(gy.shape) != (N, out_size(H, R, stride), out_size(W, R, stride), Cout):
        raise ValueError("gy %r does not match x %r (R %d, stride %d)" % (tuple(gy.shape), tuple(x.shape), R, stride))
    if tuple(out.shape) != (Cout, R, R, Cin) or not out.is_contiguous() or out.dtype != torch.float32:
        raise ValueError("out must be a contiguous fp32 (Cout, R, R, Cin) tensor")
    L.check(L.lib().jdet_conv_wgrad(L.ptr(L.f32c(x)), L.ptr(L.f32c(gy)), N, H, W, Cin, Cout, R, stride, L.ptr(out),
                                    int(ksplit), L.stream_ptr(x)), "jdet_conv_wgrad")
    return out


def bn_backward_from_output(g, y, bn, identity=None, own_output=None, want_sums=True):
    """g, y (and identity | own_output) contiguous (N,H,W,C): returns (grad_c = g * [y > 0] * a, sums | None); see
    jdet_bn_act_backward_from_output for what the second partial sum multiplies g with."""
    L.need_device(g, y, identity, own_output)
    C = y.shape[-1]
    P = y.numel() // C
    lib = L.lib()
    sums, nbytes = None, 0
    if want_sums:
        rows = lib.jdet_bn_act_backward_from_output_rows(P, C)
        if rows == 0:
            raise L.JDetHipError("bn_backward_from_output: unsupported channel count %d" % C)
        sums = torch.empty((rows, 2, C), dtype=torch.float32, device=g.device)
        nbytes = sums.numel() * 4
    gc = torch.empty_like(g)
    L.check(lib.jdet_bn_act_backward_from_output(L.ptr(g), L.ptr(y), L.ptr(identity), L.ptr(own_output), P, C,
                                                 L.ptr(bn.weight), L.ptr(bn.bias), L.ptr(bn.running_mean),
                                                 L.ptr(bn.running_var), float(bn.eps), L.ptr(gc), L.ptr(sums), nbytes,
                                                 L.stream_ptr(g)), "jdet_bn_act_backward_from_output")
    return gc, sums


def bn_sums_finish(items):
    """items: [(sums (rows, 2, C), BatchNorm2d)] (at most 4) -> [(grad_weight (C), grad_bias (C))] in ONE launch"""
    assert 1 <= len(items) <= 4
    dev = items[0][0].device
    total = sum(s.shape[2] for s, _ in items)
    flat = torch.empty((2 * total,), dtype=torch.float32, device=dev)
    jobs = (L.BnSumsJob * len(items))()
    out, off = [], 0
    for k, (s, bn) in enumerate(items):
        C = s.shape[2]
        gg, gb = flat[off:off + C], flat[off + C:off + 2 * C]
        off += 2 * C
        jobs[k] = L.BnSumsJob(s.data_ptr(), s.shape[0], C, L.ptr(bn.weight), gg.data_ptr(), gb.data_ptr())
        out.append((gg, gb))
    L.check(L.lib().jdet_bn_sums_finish(jobs, len(items), L.stream_ptr(items[0][0])), "jdet_bn_sums_finish")
    return out


class DgradBank:
    """The data-gradient weights (Cin, R, R, Cout)[ci][flipped tap][co] of a set of convolutions, rewritten from the live
    weights by ONE launch (`refresh`, once per training step from ResNet.forward; `get` refreshes again if a weight
    changed since).  The device job table holds raw pointers: it is rebuilt whenever a weight's storage moved."""

    def __init__(self, convs):
        self.convs = list(convs)
        self.index = {id(c): i for i, c in enumerate(self.convs)}
        self.buf = self.table = None
        self.views, self.ptrs, self.versions = [], [], []
        self.tiles = 0
        self.gw_flat, self.gw_claimed = None, set()

    def new_grad_buffer(self):
        """one zero fill for the weight gradients of every convolution of the bank (the kernels accumulate: the chunks
        of the position axis meet by atomics); handed out once per convolution by `claim_grad`"""
        w0 = self.convs[0].weight
        self.gw_flat = torch.zeros((sum(c.weight.numel() for c in self.convs),), dtype=torch.float32, device=w0.device)
        self.gw_claimed = set()

    def claim_grad(self, convs):
        """(Cout, R, R, Cin) views of the step's gradient buffer for these convolutions -- each view is given out ONCE
        per buffer (a second forward before the next `new_grad_buffer`, or a use outside a prepared backbone, gets its
        own zero-filled tensors: two autograd nodes must never accumulate into one buffer)"""
        if self.gw_flat is None or any(id(c) in self.gw_claimed for c in convs):
            flat = torch.zeros((sum(c.weight.numel() for c in convs),), dtype=torch.float32,
                               device=convs[0].weight.device)
            out, off = [], 0
            for c in convs:
                Co, Ci, R, _ = c.weight.shape
                out.append(flat[off:off + c.weight.numel()].view(Co, R, R, Ci))
                off += c.weight.numel()
            return out
        out = []
        for c in convs:
            i = self.index[id(c)]
            off = self._offs[i]
            Co, Ci, R, _ = c.weight.shape
            out.append(self.gw_flat[off:off + c.weight.numel()].view(Co, R, R, Ci))
            self.gw_claimed.add(id(c))
        return out

    def _build(self):
        dev = self.convs[0].weight.device
        sizes = [c.weight.numel() for c in self.convs]
        self.buf = torch.empty((sum(sizes),), dtype=torch.float32, device=dev)
        rec, off, tiles, self.views, self.ptrs = b"", 0, 0, [], []
        for c, n in zip(self.convs, sizes):
            w = c.weight
            Co, Ci, R, _ = w.shape
            if not w.permute(0, 2, 3, 1).is_contiguous() or w.dtype != torch.float32:
                raise L.JDetHipError("DgradBank needs fp32 weights in (Cout, R, R, Cin) memory order (channels-last)")
            v = self.buf[off:off + n].view(Ci, R, R, Co)
            rec += struct.pack("QQiiii", w.data_ptr(), v.data_ptr(), Co, Ci, R * R, tiles)
            tiles += ((Co + 31) // 32) * ((Ci + 31) // 32) * R * R
            off += n
            self.views.append(v)
            self.ptrs.append(w.data_ptr())
        self.tiles = tiles
        self.table = torch.frombuffer(bytearray(rec), dtype=torch.uint8).to(dev)
        self.versions = [None] * len(self.convs)
        self._offs, off = [], 0
        for n in sizes:
            self._offs.append(off)
            off += n

    def refresh(self):
        if self.buf is None or any(c.weight.data_ptr() != p for c, p in zip(self.convs, self.ptrs)):
            self._build()
        L.check(L.lib().jdet_conv_dgrad_weights(self.table.data_ptr(), len(self.convs), self.tiles,
                                                L.stream_ptr(self.buf)), "jdet_conv_dgrad_weights")
        self.versions = [c.weight._version for c in self.convs]

    def get(self, conv):
        i = self.index[id(conv)]
        if self.buf is None or self.versions[i] != conv.weight._version or conv.weight.data_ptr() != self.ptrs[i]:
            self.refresh()
        return self.views[i]


_BANKS = weakref.WeakKeyDictionary()      # Bottleneck module -> DgradBank (shared by the blocks of one ResNet)


def _block_convs(blk):
    convs = [blk.conv1, blk.conv2, blk.conv3]
    if blk.downsample is not None:
        convs.append(blk.downsample[0])
    return convs


# The fused backward never stores a conv output: x-hat is recovered as (y - beta) / gamma wherever y > 0, so the BatchNorm
# weight gradient carries an error of about ulp(y) / |gamma| per element -- harmless at the usual gammas, unbounded as a
# gamma approaches 0 (dead channels of a pretrained ResNet, gammas shrunk by weight decay), and the fused gradient clip
# would then scale EVERY gradient of the model by the poisoned norm (advisor finding, round 5).  Blocks holding a
# BatchNorm weight under GAMMA_MIN in magnitude take the per-layer path (which keeps the conv outputs).  The check is one
# device reduction + one host read for the whole backbone, made when the bank is built and every GAMMA_CHECK_EVERY
# training forwards after that (never inside a graph capture): gammas move by lr * grad per step, not by orders of
# magnitude.
GAMMA_MIN = 1e-3
GAMMA_CHECK_EVERY = 200
_SMALL_GAMMA = weakref.WeakSet()          # Bottleneck modules currently routed to the per-layer path
_gamma_clock = [0]


def _block_bns(blk):
    bns = [blk.bn1, blk.bn2, blk.bn3]
    if blk.downsample is not None:
        bns.append(blk.downsample[1])
    return bns


def check_gammas(blocks):
    """refresh _SMALL_GAMMA for these blocks (one host synchronisation)"""
    blocks = [b for b in blocks if all(isinstance(bn, nn.BatchNorm2d) and bn.affine for bn in _block_bns(b))]
    if not blocks:
        return
    mins = torch.stack([torch.stack([bn.weight.detach().abs().min() for bn in _block_bns(b)]).min() for b in blocks])
    for b, m in zip(blocks, mins.cpu().tolist()):
        if m < GAMMA_MIN:
            _SMALL_GAMMA.add(b)
        else:
            _SMALL_GAMMA.discard(b)


def prepare(blocks):
    """once per training forward of a backbone: one bank for all fusable trainable blocks, refreshed in one launch"""
    blocks = [b for b in blocks if _trainable(b) is True and all(
        c.weight.is_cuda and c.weight.dtype == torch.float32 and c.weight.permute(0, 2, 3, 1).is_contiguous()
        for c in _block_convs(b))]
    if not blocks:
        return
    if not torch.cuda.is_current_stream_capturing():
        fresh = _BANKS.get(blocks[0]) is None
        if fresh or _gamma_clock[0] % GAMMA_CHECK_EVERY == 0:
            check_gammas(blocks)
        _gamma_clock[0] += 1
    blocks = [b for b in blocks if b not in _SMALL_GAMMA]
    if not blocks:
        return
    bank = _BANKS.get(blocks[0])
    if bank is None or any(_BANKS.get(b) is not bank for b in blocks):
        bank = DgradBank([c for b in blocks for c in _block_convs(b)])
        for b in blocks:
            _BANKS[b] = bank
    bank.refresh()
    if OWN_WGRAD:
        bank.new_grad_buffer()


def _bank(blk):
    bank = _BANKS.get(blk)
    if bank is None:
        bank = _BANKS[blk] = DgradBank(_block_convs(blk))
    return bank


def _is_frozen_bn(bn):
    return isinstance(bn, nn.BatchNorm2d) and not bn.training and bn.running_mean is not None and bn.affine


def _params(blk):
    ps = [blk.conv1.weight, blk.bn1.weight, blk.bn1.bias, blk.conv2.weight, blk.bn2.weight, blk.bn2.bias,
          blk.conv3.weight, blk.bn3.weight, blk.bn3.bias]
    if blk.downsample is not None:
        ps += [blk.downsample[0].weight, blk.downsample[1].weight, blk.downsample[1].bias]
    return ps


def _trainable(blk):
    """True: every parameter trains; False: none does; None: mixed (the per-layer path handles it)"""
    flags = {p.requires_grad for p in _params(blk)}
    return flags.pop() if len(flags) == 1 else None


def fusable(blk, x):
    """can this Bottleneck run as fused launches on x?"""
    if not (ENABLED and x.is_cuda and x.dtype == torch.float32 and x.dim() == 4 and not torch.is_autocast_enabled()):
        return False
    if not x.is_contiguous(memory_format=torch.channels_last) or x.numel() == 0:
        return False
    ds = blk.downsample
    if ds is not None and not (isinstance(ds, nn.Sequential) and len(ds) == 2 and isinstance(ds[0], nn.Conv2d)
                               and ds[0].kernel_size == (1, 1) and ds[0].bias is None and ds[0].groups == 1
                               and ds[0].stride == blk.conv2.stride and ds[0].padding == (0, 0)):
        return False
    bns = [blk.bn1, blk.bn2, blk.bn3] + ([ds[1]] if ds is not None else [])
    if not all(_is_frozen_bn(b) for b in bns):
        return False
    c1, c2, c3 = blk.conv1, blk.conv2, blk.conv3
    if not (c2.groups == 1 and c2.dilation == (1, 1) and c2.padding == (1, 1) and c2.kernel_size == (3, 3)
            and c2.stride in ((1, 1), (2, 2)) and c1.stride == (1, 1) and c3.stride == (1, 1)
            and c1.kernel_size == (1, 1) and c3.kernel_size == (1, 1) and c1.padding == (0, 0) and c3.padding == (0, 0)
            and c1.bias is None and c2.bias is None and c3.bias is None):
        return False
    N, Cin, H, W = x.shape
    chans = (Cin, c1.out_channels, c3.out_channels)
    if any(c % 16 for c in chans) or N * H * W * max(chans) >= 2 ** 30:
        return False
    if not all(c.weight.dtype == torch.float32 and c.weight.permute(0, 2, 3, 1).is_contiguous()
               for c in _block_convs(blk)):
        return False
    if any(c.out_channels // 4 > 1024 or (c.out_channels // 4 <= 256 and 256 % (c.out_channels // 4))
           for c in _block_convs(blk)):       # the channel counts the elementwise BatchNorm backward takes
        return False
    grad = torch.is_grad_enabled() and (x.requires_grad or any(p.requires_grad for p in _params(blk)))
    if grad and _trainable(blk) is not True:
        return False          # gradients through a frozen or partly frozen block: the per-layer path
    if grad and blk in _SMALL_GAMMA:
        return False          # a BatchNorm weight too close to 0 for x-hat = (y - beta) / gamma (see GAMMA_MIN)
    return True


def _forward(blk, xn):
    st = blk.conv2.stride[0]
    y1 = conv_bn_nhwc(xn, weight_krsc(blk.conv1.weight), 1, blk.bn1, None, True)
    y2 = conv_bn_nhwc(y1, weight_krsc(blk.conv2.weight), st, blk.bn2, None, True)
    if blk.downsample is not None:
        idn = conv_bn_nhwc(xn, weight_krsc(blk.downsample[0].weight), st, blk.downsample[1], None, False)
    else:
        idn = xn
    y3 = conv_bn_nhwc(y2, weight_krsc(blk.conv3.weight), 1, blk.bn3, idn, True)
    return y1, y2, y3, idn


def _lib_dgrad(gy_nhwc, x_nhwc, weight, stride, padding):
    """the library's data gradient (stride-2 layers): NHWC views in, NHWC contiguous out"""
    gx = torch.ops.aten.convolution_backward(gy_nhwc.permute(0, 3, 1, 2), x_nhwc.permute(0, 3, 1, 2), weight, None,
                                             [stride, stride], [padding, padding], [1, 1], False, [0, 0], 1,
                                             [True, False, False])[0]
    return gx.permute(0, 2, 3, 1).contiguous()


class _BottleneckFunction(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, blk, *params):
        xn = x.permute(0, 2, 3, 1)
        y1, y2, y3, idn = _forward(blk, xn)
        ctx.blk = blk
        # the backward reads the weights / BatchNorm tensors LIVE from the module (nothing is copied): remember their
        # versions so that an in-place update between forward and backward raises as it would for a saved tensor
        ctx.versions = tuple(p._version for p in params)
        # the block's slices of the step's weight-gradient buffer, claimed at forward time: this node's backward is the
        # only writer of these views
        ctx.gws = _bank(blk).claim_grad(_block_convs(blk)) if OWN_WGRAD else None
        ctx.save_for_backward(x, y1, y2, y3, idn if blk.downsample is not None else None)
        return y3.permute(0, 3, 1, 2)

    @staticmethod
    def backward(ctx, gout):
        blk = ctx.blk
        x, y1, y2, y3, idn = ctx.saved_tensors
        if tuple(p._version for p in _params(blk)) != ctx.versions:
            raise RuntimeError("a parameter of this Bottleneck was modified in place between its forward and its backward "
                               "(optimizer step / load_state_dict / EMA swap): the fused backward reads the live weights")
        xn = x.permute(0, 2, 3, 1)
        g = gout.permute(0, 2, 3, 1)
        if not g.is_contiguous():
            g = g.contiguous()
        ds = blk.downsample
        st = blk.conv2.stride[0]
        bank = _bank(blk)
        need_gx = ctx.needs_input_grad[0]
        convs = _block_convs(blk)
        gws = [None] * len(convs)
        own = ctx.gws is not None
        if own:
            gws = list(ctx.gws)
            ctx.gws = None          # a second backward through this node (retain_graph) must not add onto the first's
        elif OWN_WGRAD:
            own = True
            gws = DgradBank(convs).claim_grad(convs)

        side = _side_stream(g.device) if (own and WGRAD_STREAM and g.is_cuda) else None
        main = torch.cuda.current_stream(g.device) if side is not None else None

        def wgrad(k, xin, gy, R, stride):
            c = convs[k]
            if own:
                if side is not None:
                    side.wait_stream(main)          # behind the kernel that wrote gy (and the step's zero fill)
                    with torch.cuda.stream(side):
                        conv_wgrad_nhwc(xin, gy, R, stride, gws[k])
                else:
                    conv_wgrad_nhwc(xin, gy, R, stride, gws[k])
                # the parameter's own strides: a 1x1 weight is plain (Cout, Cin, 1, 1) memory, a 3x3 one channels-last
                gws[k] = gws[k].view(c.weight.shape) if R == 1 else gws[k].permute(0, 3, 1, 2)
            else:
                gw = torch.ops.aten.convolution_backward(gy.permute(0, 3, 1, 2), xin.permute(0, 3, 1, 2), c.weight, None,
                                                         [stride, stride], [R // 2, R // 2], [1, 1], False, [0, 0], 1,
                                                         [False, True, False])[1]
                if R == 1 and gw.stride() != c.weight.stride():
                    # (Cout, Cin, 1, 1) is one memory order under two stride spellings; the parameter's own keeps the
                    # gradient layout contract (DDP's bucket views would otherwise copy and warn)
                    gw = gw.as_strided(c.weight.shape, c.weight.stride())
                gws[k] = gw
        # block output: y3 = relu(bn3(c3) + identity)
        g3p, s3 = bn_backward_from_output(g, y3, blk.bn3, identity=idn if ds is not None else xn)
        # (each weight gradient is launched BEFORE the data gradient that reads the same g_k': with WGRAD_STREAM the two run
        #  side by side; on one stream the order is immaterial)
        wgrad(2, y2, g3p, 1, 1)
        g2p, s2 = conv_bn_nhwc(g3p, bank.get(blk.conv3), 1, blk.bn2, mode=L.EPI_MASK, act=y2, want_sums=True)
        wgrad(1, y1, g2p, 3, st)
        if st == 1:
            g1p, s1 = conv_bn_nhwc(g2p, bank.get(blk.conv2), 1, blk.bn1, mode=L.EPI_MASK, act=y1, want_sums=True)
        else:
            g1p, s1 = bn_backward_from_output(_lib_dgrad(g2p, y1, blk.conv2.weight, st, 1), y1, blk.bn1)
        wgrad(0, xn, g1p, 1, 1)
        sums = [(s1, blk.bn1), (s2, blk.bn2), (s3, blk.bn3)]
        gx = None
        if ds is None:
            if need_gx:
                gx = conv_bn_nhwc(g1p, bank.get(blk.conv1), 1, None, mode=L.EPI_ADD, grad_out=g, act=y3)
        else:
            gdp, sd = bn_backward_from_output(g, y3, ds[1], own_output=idn)
            sums.append((sd, ds[1]))
            wgrad(3, xn, gdp, 1, st)
            if need_gx:
                if st == 1:
                    gxd = conv_bn_nhwc(gdp, bank.get(ds[0]), 1, None)
                else:
                    gxd = _lib_dgrad(gdp, xn, ds[0].weight, st, 0)
                gx = conv_bn_nhwc(g1p, bank.get(blk.conv1), 1, None, residual=gxd)
        bn_grads = bn_sums_finish(sums)
        if side is not None:
            # the temporaries the side kernels read (g_k', allocated on the main stream) go back to the main stream's pool
            # when this function returns: every later use of that memory is ordered behind this wait
            main.wait_stream(side)
        out = [gx.permute(0, 3, 1, 2) if gx is not None else None, None]
        for k in range(len(convs)):
            out += [gws[k], bn_grads[k][0], bn_grads[k][1]]
        return tuple(out)


def bottleneck(blk, x):
    """relu(bn3(conv3(relu(bn2(conv2(relu(bn1(conv1(x)))))))) + identity) for a Bottleneck `blk` that `fusable` accepted"""
    if torch.is_grad_enabled() and (x.requires_grad or any(p.requires_grad for p in _params(blk))):
        return _BottleneckFunction.apply(x, blk, *_params(blk))
    return _forward(blk, x.permute(0, 2, 3, 1))[2].permute(0, 3, 1, 2)
