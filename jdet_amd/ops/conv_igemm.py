"""3x3 convolution / deformable convolution forward as one fp32-MFMA implicit GEMM (csrc/conv_igemm.hip), and the
one-pass bias / ReLU backward of every conv + bias [+ ReLU] (csrc/frozen_bn.hip: jdet_bias_act_backward).

Replaces the `nn.Conv(3x3) [+ ReLU]` of ConvModule (python/jdet/models/utils/modules.py:L91-175) in the head towers,
the FPN and the RPNs, and -- where no gradient is needed -- the im2col + matmul pair of DeformConv.execute
(python/jdet/ops/dcn_v1.py:L412-454): no column matrix, bias / ReLU / gap-row mask applied to the accumulators.  The
kernel is a forward kernel: `_ConvBiasAct` gives it a backward through the library's data / weight gradient kernels
(the ReLU mask and the bias gradient come out of one pass over the incoming gradient); the training path of the
deformable conv keeps the column matrix, which its weight-gradient GEMM reads.

Measured on MI355X, 256 -> 256 channels, batch 2 (scripts/conv_igemm_timing.py, profiles/r03_conv_igemm.md):
128^2 map 296 us (130.6 TFLOP/s, MFMA busy 80.8 %) vs 394 us for library conv + bias + ReLU; 64^2 map 86-94 vs 126 us;
32^2 / 16^2 / 8^2 maps 37 / 18 / 16 us vs 55 / 29 / 26 us (K steps split over workgroups through a scratch buffer).
"""
import os
import weakref

import torch

from jdet_amd import _lib as L


def supported(cin, cout):
    return bool(L.lib().jdet_conv3x3_igemm_supported(int(cin), int(cout)))


def weight_krsc(weight):
    """(Cout, Cin, 3, 3) logical -> (Cout, 3, 3, Cin) contiguous fp32; free when the weight is channels_last"""
    return L.f32c(weight.permute(0, 2, 3, 1))


def conv3x3_nhwc(x_nhwc, w_krsc, bias=None, relu=False, rowmask=None, offset=None, tile=0):
    """x_nhwc (N,H,W,Cin) contiguous fp32, w_krsc (Cout,3,3,Cin), offset (N,18,H,W) or None -> (N,H,W,Cout)"""
    L.need_device(x_nhwc, w_krsc, bias, rowmask, offset)
    N, H, W, Cin = x_nhwc.shape
    Cout = w_krsc.shape[0]
    if w_krsc.shape[1:] != (3, 3, Cin):
        raise ValueError("weight %r does not match input channels %d" % (tuple(w_krsc.shape), Cin))
    if offset is not None and tuple(offset.shape) != (N, 18, H, W):
        raise ValueError("offset must be (N, 18, H, W), got %r" % (tuple(offset.shape),))
    x_nhwc, w_krsc = L.f32c(x_nhwc), L.f32c(w_krsc)
    y = torch.empty((N, H, W, Cout), dtype=torch.float32, device=x_nhwc.device)
    ws, nbytes = None, 0
    if tile == 0 and offset is None:      # small maps: K split over workgroups through a scratch buffer
        nbytes = L.lib().jdet_conv3x3_igemm_workspace(N, H, W, Cin, Cout)
        if nbytes:
            ws = torch.empty((nbytes,), dtype=torch.uint8, device=x_nhwc.device)
    L.check(L.lib().jdet_conv3x3_igemm_forward(
        L.ptr(x_nhwc), N, H, W, Cin, L.ptr(w_krsc), Cout,
        L.ptr(L.f32c(bias)) if bias is not None else None, int(bool(relu)),
        L.ptr(L.f32c(rowmask)) if rowmask is not None else None,
        L.ptr(L.f32c(offset)) if offset is not None else None,
        int(tile), L.ptr(y), L.ptr(ws), nbytes, L.stream_ptr(x_nhwc)), "jdet_conv3x3_igemm_forward")
    return y


def conv3x3(x, weight, bias=None, relu=False, offset=None, tile=0):
    """NCHW-logical convenience form: returns a channels_last (N, Cout, H, W) tensor"""
    y = conv3x3_nhwc(L.f32c(x.permute(0, 2, 3, 1)), weight_krsc(weight), bias, relu, None, offset, tile)
    return y.permute(0, 3, 1, 2)


# smallest map (positions = N*H*W) the fused path takes: it measured faster than library conv + bias + ReLU at every
# size once small maps split their K steps over workgroups; the deformable form only where it beats im2col + GEMM
MIN_POSITIONS = int(os.environ.get("JDET_CONV_MIN_POS", "1"))
DEFORM_MIN_POSITIONS = 32768
ENABLED = os.environ.get("JDET_CONV_IGEMM", "1") == "1"     # A/B switch for measurements
# Train step: the fused forward + the library's data / weight gradients (`_Conv3x3BiasAct`).  S2ANet step 30.05 ms with
# it, 30.37 ms without (two A/B pairs, profiles/r03_conv_igemm.md); JDET_CONV_IGEMM_TRAIN=0 switches it off.
TRAIN = os.environ.get("JDET_CONV_IGEMM_TRAIN", "1") == "1"

def needs_grad(*tensors):
    return torch.is_grad_enabled() and any(t is not None and t.requires_grad for t in tensors)


def preferred(x, weight, min_positions=None):
    """x (N, Cin, H, W) logical, weight (Cout, Cin, 3, 3): is the fused kernel the faster choice for this call?"""
    if not (ENABLED and x.is_cuda and x.dtype == torch.float32 and weight.dtype == torch.float32 and x.dim() == 4):
        return False
    N, Cin, H, W = x.shape
    if tuple(weight.shape[1:]) != (Cin, 3, 3) or not supported(Cin, weight.shape[0]):
        return False
    if N * H * W * max(Cin, weight.shape[0]) >= 2 ** 30:          # the kernel addresses with 32-bit byte offsets
        return False
    return N * H * W >= (MIN_POSITIONS if min_positions is None else min_positions)


def wgrad_supported(cin, cout):
    return bool(L.lib().jdet_conv3x3_wgrad_supported(int(cin), int(cout)))


def conv3x3_wgrad_nhwc(x_nhwc, gy_nhwc, offset=None, out=None, ksplit=0):
    """x (N,H,W,Cin), gy (N,H,W,Cout) contiguous fp32 [offset (N,18,H,W): the deformable form] -> the weight gradient
    (Cout,3,3,Cin).  `out`: a contiguous (Cout,3,3,Cin) buffer the gradient is ADDED to (csrc/conv_wgrad.hip), else a
    fresh zero-filled one."""
    L.need_device(x_nhwc, gy_nhwc, offset, out)
    N, H, W, Cin = x_nhwc.shape
    Cout = gy_nhwc.shape[3]
    if tuple(gy_nhwc.shape[:3]) != (N, H, W):
        raise ValueError("gy %r does not match x %r" % (tuple(gy_nhwc.shape), tuple(x_nhwc.shape)))
    if offset is not None and tuple(offset.shape) != (N, 18, H, W):
        raise ValueError("offset must be (N, 18, H, W), got %r" % (tuple(offset.shape),))
    x_nhwc, gy_nhwc = L.f32c(x_nhwc), L.f32c(gy_nhwc)
    if out is None:
        out = torch.zeros((Cout, 3, 3, Cin), dtype=torch.float32, device=x_nhwc.device)
    elif tuple(out.shape) != (Cout, 3, 3, Cin) or not out.is_contiguous() or out.dtype != torch.float32:
        raise ValueError("out must be a contiguous fp32 (Cout, 3, 3, Cin) tensor")
    L.check(_wgrad_call(x_nhwc, gy_nhwc, offset, N, H, W, Cin, Cout, L.ptr(out), ksplit), "jdet_conv3x3_wgrad")
    return out


def conv3x3_wgrad(x, gy, offset=None, ksplit=0):
    """NCHW-logical form: x (N,Cin,H,W), gy (N,Cout,H,W) -> (Cout,Cin,3,3) logical weight gradient (channels_last
    memory: a view of the kernel's (Cout,3,3,Cin) result)"""
    gw = conv3x3_wgrad_nhwc(L.f32c(x.permute(0, 2, 3, 1)), L.f32c(gy.permute(0, 2, 3, 1)), offset, None, ksplit)
    return gw.permute(0, 3, 1, 2)


# Weight gradient of the igemm layers by csrc/conv_wgrad.hip (default since round 6; JDET_CONV_WGRAD=0: the library's).  Several uses
# of ONE weight inside one backward pass (a tower shared by the pyramid levels) accumulate into the buffer the first use
# returned: the kernel adds in place, so the engine neither zero-fills per call nor sums the uses afterwards.
# Measured (profiles/r04_conv_wgrad.md): the kernel runs 91 % MFMA-busy in cycles and ties the library's in isolation
# (312 vs 314-326 us at 2 x 128^2 x 256), but inside the autotuned S2ANet step the library's pick is faster than its
# stand-alone time: 29.28 ms with this switch on against 29.02-29.06 ms off -- so it is off by default.  (Round 5: the same
# layers through the backbone's 64 x 64-tile entry point tie the library as well: 27.885 vs 27.904 ms, profiles/r05_conv_bn.md.)
# Round 6: ON by default.  With 128 x 64 tiles for the tower shape (310 us against 360 on 128 x 128 and the library's 326 incl.
# its zero fill), raw buffer atomics with SGPR row offsets in the epilogue and 32-bit prologue arithmetic the own kernel first
# tied the library in the step (26.48 vs 26.48-26.52 ms) and then beat it: 26.075 vs 26.173 ms (three same-box pairs, all three in
# its favour), 744 -> 727 launches (18 library zero fills and 8 accumulation adds fewer): profiles/r06_conv_prefetch.md.
# JDET_CONV_WGRAD=0: the library's.
WGRAD = os.environ.get("JDET_CONV_WGRAD", "1") == "1"
def _wgrad_call(x_nhwc, gy_nhwc, offset, N, H, W, Cin, Cout, out_ptr, ksplit):
    return L.lib().jdet_conv3x3_wgrad(L.ptr(x_nhwc), L.ptr(gy_nhwc), L.ptr(L.f32c(offset)) if offset is not None else None,
                                      N, H, W, Cin, Cout, out_ptr, int(ksplit), L.stream_ptr(x_nhwc))


_GW_ACC = {}           # weight.data_ptr() -> (backward pass id, device pointer of the (Cout,3,3,Cin) buffer, its shape)


def shared_wgrad(weight, x_nhwc, gy_nhwc, offset=None):
    """weight gradient of y = conv3x3(x, weight) [deformable with `offset`] for this use of `weight`, as autograd wants
    it: the first use in a backward pass returns a fresh (Cout, Cin, 3, 3) tensor (channels_last memory, so a
    channels_last parameter takes it without a copy); later uses add into that tensor's memory and return None.  Only
    the pointer is remembered -- a second reference would make AccumulateGrad clone the gradient instead of taking it."""
    tid = torch._C._current_graph_task_id()
    key = weight.data_ptr()          # (the saved tensor may come back in a new Python wrapper: the storage names the weight)
    Cout, Cin = weight.shape[0], weight.shape[1]
    hit = _GW_ACC.get(key)
    # sharing is per STREAM: a use replayed on another stream (the packed levels' side stream) must not add into a
    # buffer whose hand-over to AccumulateGrad the engine orders only against the first use's stream -- it gets its own
    # gradient tensor and autograd adds the two (advisor finding, round 4)
    stream = torch.cuda.current_stream(x_nhwc.device).cuda_stream if x_nhwc.is_cuda else 0
    if hit is not None and len(hit) > 3 and hit[3] != stream:
        hit = None
    if tid >= 0 and hit is not None and hit[0] == tid and hit[2] == (Cout, Cin, x_nhwc.device):
        N, H, W, _ = x_nhwc.shape
        x_nhwc, gy_nhwc = L.f32c(x_nhwc), L.f32c(gy_nhwc)
        L.check(_wgrad_call(x_nhwc, gy_nhwc, offset, N, H, W, Cin, Cout, hit[1], 0), "jdet_conv3x3_wgrad")
        return None
    buf = conv3x3_wgrad_nhwc(x_nhwc, gy_nhwc, offset)
    if tid >= 0 and (_GW_ACC.get(key) is None or _GW_ACC[key][0] != tid):
        _GW_ACC[key] = (tid, buf.data_ptr(), (Cout, Cin, x_nhwc.device), stream)
    return buf.permute(0, 3, 1, 2)


_FLIPPED = {}          # id(weight) -> (data_ptr, version, tensor): the data-gradient weights of the current step
# grad_x through this kernel (flipped weights): 296 vs 341 us against the library's data gradient in isolation
# (scripts/wgrad_bar.py), but no difference inside the autotuned train step (30.30 vs 30.30 ms): off by default.
DGRAD = os.environ.get("JDET_CONV_IGEMM_DGRAD", "0") == "1"


class _WeightRef:
    """what conv_bn.DgradBank reads of a "convolution": its weight (held weakly: the bank must not keep a model alive)"""

    def __init__(self, w):
        self.ref = weakref.ref(w)

    @property
    def weight(self):
        return self.ref()


_DGRAD_BANKS = {}      # device -> {"items": {id(weight): _WeightRef}, "bank": DgradBank}


def dgrad_weight(weight):
    """(Cout, Cin, 3, 3) -> (Cin, 3, 3, Cout) contiguous with the taps flipped: grad_x = conv3x3(grad_y, this).  The
    weights seen so far on a device live in ONE bank (conv_bn.DgradBank: one launch rewrites all of them when a version
    moved, i.e. once per training step -- the per-weight flip + copy pair was 2 launches x 19 weights per S2ANet step);
    weights that are not channels-last fp32 device tensors take the per-weight form."""
    st = None
    if weight.is_cuda and weight.dtype == torch.float32 and weight.permute(0, 2, 3, 1).is_contiguous() \
            and not torch.cuda.is_current_stream_capturing():
        from jdet_amd.ops.conv_bn import DgradBank
        st = _DGRAD_BANKS.setdefault(weight.device, {"items": {}, "bank": None})
        it = st["items"].get(id(weight))
        if it is None or it.weight is not weight or any(v.weight is None for v in st["items"].values()):
            live = {k: v for k, v in st["items"].items() if v.weight is not None}
            it = live[id(weight)] = _WeightRef(weight)
            st["items"] = live
            st["bank"] = DgradBank(list(live.values()))
        return st["bank"].get(it)
    if weight.is_cuda and torch.cuda.is_current_stream_capturing():
        st = _DGRAD_BANKS.get(weight.device)
        it = st["items"].get(id(weight)) if st else None
        if it is not None and it.weight is weight and st["bank"].buf is not None:
            return st["bank"].get(it)          # (built by the eager warm-up steps: a refresh is one kernel, capturable)
    key, stamp = id(weight), (weight.data_ptr(), weight._version)
    hit = _FLIPPED.get(key)
    if hit is None or hit[0] != stamp:
        hit = (stamp, weight.detach().flip(2, 3).permute(1, 2, 3, 0).contiguous())
        _FLIPPED[key] = hit
    return hit[1]


# bias [+ ReLU] backward as ONE pass (csrc/frozen_bn.hip: jdet_bias_act_backward) instead of the framework's
# threshold_backward + per-channel reduce pair; JDET_BIAS_ACT_BWD=0 switches back (A/B: profiles/r03_conv_igemm.md).
BIAS_ACT_BWD = os.environ.get("JDET_BIAS_ACT_BWD", "1") == "1"
_BWD_WS = {}


def _bias_bwd_supported(c):
    q = c // 4
    return c % 4 == 0 and ((q <= 256 and 256 % q == 0) or 256 < q <= 1024)


def bias_act_backward(g, y, relu):
    """g, y (N, C, H, W) logical -> (grad of the pre-activation (same logical shape, channels_last), grad_bias (C,)):
    grad_pre = g * [y > 0] when relu, else g itself; grad_bias = grad_pre.sum((0, 2, 3))."""
    N, C, H, W = g.shape
    if (BIAS_ACT_BWD and not relu and g.is_cuda and g.dtype == torch.float32 and not _bias_bwd_supported(C)
            and C <= 256 and N * H * W > 0):
        # channel counts off the vector kernels' grid (the heads' 15- / 5-channel output convs): the any-C column sum
        # (csrc/frozen_bn.hip: jdet_channel_sum) instead of the framework's per-channel reduce (25-60 us per call)
        gn = L.f32c(g.permute(0, 2, 3, 1))
        P = N * H * W
        nbytes = L.lib().jdet_channel_sum_workspace(P, C)
        ws = torch.empty((nbytes,), dtype=torch.uint8, device=g.device)
        gb = torch.empty((C,), dtype=torch.float32, device=g.device)
        L.check(L.lib().jdet_channel_sum(L.ptr(gn), P, C, L.ptr(gb), L.ptr(ws), nbytes, L.stream_ptr(gn)),
                "jdet_channel_sum")
        return g, gb
    if not (BIAS_ACT_BWD and g.is_cuda and g.dtype == torch.float32 and _bias_bwd_supported(C) and N * H * W > 0):
        gp = torch.ops.aten.threshold_backward(g, y, 0) if relu else g
        return gp, gp.sum((0, 2, 3))
    gn = L.f32c(g.permute(0, 2, 3, 1))
    P = N * H * W
    nbytes = L.lib().jdet_frozen_bn_act_backward_workspace(P, C)
    if torch.cuda.is_current_stream_capturing():
        # the address is baked into the graph: scratch from the capturing graph's own pool, never from (or into) the
        # cache -- an evicted cache entry would leave replays writing partial sums to freed memory
        ws = torch.empty((max(nbytes, 1),), dtype=torch.uint8, device=g.device)
    else:
        key = (g.device.index, torch.cuda.current_stream(g.device).cuda_stream)
        ws = _BWD_WS.get(key)
        if ws is None or ws.numel() < nbytes:
            while len(_BWD_WS) >= 8:       # streams come and go: keep the scratch cache bounded (oldest entry out)
                _BWD_WS.pop(next(iter(_BWD_WS)))
            ws = _BWD_WS[key] = torch.empty((max(nbytes, 1 << 20),), dtype=torch.uint8, device=g.device)
    gb = torch.empty((C,), dtype=torch.float32, device=g.device)
    gp = torch.empty_like(gn) if relu else gn
    yn = L.f32c(y.permute(0, 2, 3, 1)) if relu else None
    L.check(L.lib().jdet_bias_act_backward(L.ptr(gn), L.ptr(yn), P, C, int(bool(relu)), L.ptr(gp) if relu else None,
                                           L.ptr(gb), L.ptr(ws), ws.numel(), L.stream_ptr(gn)),
            "jdet_bias_act_backward")
    return gp.permute(0, 3, 1, 2), gb


CONV1X1_GEMM = os.environ.get("JDET_CONV1X1_GEMM", "1") == "1"


def _is_1x1(x, weight, stride, padding, groups):
    return (CONV1X1_GEMM and tuple(weight.shape[2:]) == (1, 1) and tuple(stride) == (1, 1) and tuple(padding) == (0, 0)
            and groups == 1 and x.is_cuda and x.dtype == torch.float32 and weight.shape[0] >= 64
            and x.is_contiguous(memory_format=torch.channels_last))


class _ConvBiasAct(torch.autograd.Function):
    """y = [relu](conv(x, w) + b).  Forward: the implicit-GEMM kernel (`igemm`: 3x3 / stride 1 / pad 1 only) or the
    library convolution; backward: bias gradient and ReLU mask in one pass (`bias_act_backward`), then the library's
    data / weight gradients (with DGRAD, for the igemm shapes: grad_x by the igemm kernel on the flipped weights)."""

    @staticmethod
    def forward(ctx, x, weight, bias, relu, stride, padding, dilation, groups, igemm, rowmask=None):
        # rowmask (igemm + ReLU only): the finished rows are multiplied by a 0 / 1 mask per position (the gap rows of a
        # LevelPack).  It needs no backward of its own: a masked row of y is 0, so the ReLU mask [y > 0] of
        # bias_act_backward already zeroes the gradient there.
        k1 = _is_1x1(x, weight, stride, padding, groups)
        if igemm:
            y = conv3x3_nhwc(L.f32c(x.permute(0, 2, 3, 1)), weight_krsc(weight), bias, relu,
                             rowmask).permute(0, 3, 1, 2)
        elif k1 and weight.shape[1] >= 256 and x.shape[0] * x.shape[2] * x.shape[3] <= 32768:
            # stride-1 1x1 convolution of a channels-last map = GEMM on its (positions, channels) matrix view; the
            # forward wins from 256 input channels up (scripts/conv1x1_probe.py, profiles/r04_conv1x1_probe.txt)
            N, Ci, H, W = x.shape
            Co = weight.shape[0]
            xm = x.permute(0, 2, 3, 1).reshape(-1, Ci)
            ym = torch.addmm(bias, xm, weight.view(Co, Ci).t()) if bias is not None else xm @ weight.view(Co, Ci).t()
            if relu:
                ym = torch.relu_(ym)
            y = ym.view(N, H, W, Co).permute(0, 3, 1, 2)
        else:
            y = torch.ops.aten.convolution(x, weight, bias, stride, padding, dilation, False, [0, 0], groups)
            if relu:
                y = torch.relu_(y)
        ctx.k1 = k1
        ctx.cfg = (bool(relu), list(stride), list(padding), list(dilation), groups, bool(igemm), bias is not None)
        ctx.save_for_backward(x, weight, y if relu else None)
        return y

    @staticmethod
    def backward(ctx, g):
        x, weight, y = ctx.saved_tensors
        relu, stride, padding, dilation, groups, igemm, has_bias = ctx.cfg
        gb = None
        if has_bias and ctx.needs_input_grad[2]:
            g, gb = bias_act_backward(g, y, relu)
        elif relu:
            g = torch.ops.aten.threshold_backward(g, y, 0)
        need = [ctx.needs_input_grad[0], ctx.needs_input_grad[1], False]
        gx = None
        if ctx.k1:
            # the data gradient of a 1x1 convolution as gy . W (18 % faster than the library's over the ResNet / FPN
            # shapes), the weight gradient as gy^T . x only on small maps (its reduction runs over the positions)
            N, Ci, H, W = x.shape
            Co = weight.shape[0]
            g = g.contiguous(memory_format=torch.channels_last)
            gm = g.permute(0, 2, 3, 1).reshape(-1, Co)
            gw1 = None
            if need[0]:
                gx = (gm @ weight.view(Co, Ci)).view(N, H, W, Ci).permute(0, 3, 1, 2)
                need[0] = False
            if need[1] and N * H * W <= 4096:
                gw1 = (gm.t() @ x.permute(0, 2, 3, 1).reshape(-1, Ci)).view(Co, Ci, 1, 1)
                need[1] = False
            lx, gw, _ = torch.ops.aten.convolution_backward(g, x, weight, None, stride, padding, dilation, False,
                                                            [0, 0], groups, need) if any(need) else (None, None, None)
            gw = gw1 if gw1 is not None else gw
            if gw is not None and gw.stride() != weight.stride() and gw.numel() == weight.numel():
                # (Cout, Cin, 1, 1) is one memory order under two stride spellings: the parameter's own keeps the
                # gradient layout contract (DDP's bucket views otherwise copy and warn)
                gw = gw.as_strided(weight.shape, weight.stride())
            return gx, gw, gb, None, None, None, None, None, None, None
        if need[0] and igemm and DGRAD and supported(weight.shape[0], weight.shape[1]):
            gx = conv3x3_nhwc(L.f32c(g.permute(0, 2, 3, 1)), dgrad_weight(weight)).permute(0, 3, 1, 2)
            need[0] = False
        own_gw = need[1] and igemm and WGRAD and wgrad_supported(weight.shape[1], weight.shape[0])
        if own_gw:
            gw = shared_wgrad(weight, x.permute(0, 2, 3, 1), g.permute(0, 2, 3, 1))
            need[1] = False
        lx, lw, _ = torch.ops.aten.convolution_backward(g, x, weight, None, stride, padding, dilation, False, [0, 0],
                                                        groups, need) if any(need) else (None, None, None)
        return (gx if gx is not None else lx), (gw if own_gw else lw), gb, None, None, None, None, None, None, None


_Conv3x3BiasAct = _ConvBiasAct      # (name used by the round-3 notes)


def conv3x3_bias_act(x, weight, bias=None, relu=False, rowmask=None):
    """(N, Cin, H, W) logical (channels_last memory is free) -> (N, Cout, H, W) channels_last; differentiable.
    rowmask: (N*H*W,) 0 / 1 floats multiplied into the finished rows (with relu=True and a bias when gradients flow:
    see _ConvBiasAct)"""
    if needs_grad(x, weight, bias):
        return _ConvBiasAct.apply(x, weight, bias, relu, (1, 1), (1, 1), (1, 1), 1, True, rowmask)
    if rowmask is not None:
        return conv3x3_nhwc(L.f32c(x.permute(0, 2, 3, 1)), weight_krsc(weight), bias, relu, rowmask).permute(0, 3, 1, 2)
    return conv3x3(x, weight, bias, relu)


def _is_igemm_conv(conv):
    return ((conv.kernel_size, conv.stride, conv.padding, conv.dilation, conv.groups)
            == ((3, 3), (1, 1), (1, 1), (1, 1), 1))


def masked_conv_module(conv, x, rowmask):
    """`relu(conv(x)) * mask` in ONE kernel for a plain 3x3 / stride 1 / pad 1 nn.Conv2d with a bias where the fused
    kernel applies (the towers on a LevelPack); None = not applicable here, the caller multiplies."""
    plain = (type(conv).__name__ == "Conv2d" and conv.padding_mode == "zeros" and not isinstance(conv.padding, str)
             and not torch.is_autocast_enabled())
    grad = needs_grad(x, conv.weight, conv.bias)
    if not (plain and _is_igemm_conv(conv) and preferred(x, conv.weight) and (TRAIN or not grad)):
        return None
    if grad and not (BIAS_ACT_BWD and conv.bias is not None and conv.bias.requires_grad
                     and _bias_bwd_supported(conv.out_channels)):
        return None          # (the ReLU mask of the one-pass bias backward is what zeroes the masked rows' gradient)
    return conv3x3_bias_act(x, conv.weight, conv.bias, True, rowmask)


def conv_module(conv, x, relu=False):
    """`[relu](conv(x))` for a plain nn.Conv2d (zeros padding).  3x3 / stride 1 / pad 1 layers take the fused kernel
    when `preferred()` says so; any layer WITH a bias that needs gradients goes through `_ConvBiasAct`, whose backward
    produces the bias gradient and the ReLU mask in one pass; everything else is the library call it always was."""
    plain = (type(conv).__name__ == "Conv2d" and conv.padding_mode == "zeros" and not isinstance(conv.padding, str)
             and not torch.is_autocast_enabled())
    grad = needs_grad(x, conv.weight, conv.bias)
    # (prediction layers of 5 / 15 / 18 channels would run a 64-wide output tile three quarters empty: the library's
    # narrow-tile kernels keep their forward; their bias gradient still comes from the own column sum below)
    if (plain and _is_igemm_conv(conv) and conv.out_channels >= 32 and preferred(x, conv.weight)
            and (TRAIN or not grad)):
        return conv3x3_bias_act(x, conv.weight, conv.bias, relu)
    if (plain and grad and BIAS_ACT_BWD and conv.bias is not None and x.is_cuda and x.dtype == torch.float32
            and (_bias_bwd_supported(conv.out_channels) or (not relu and conv.out_channels <= 256))):
        return _ConvBiasAct.apply(x, conv.weight, conv.bias, relu, conv.stride, conv.padding, conv.dilation,
                                  conv.groups, False)
    y = conv(x)
    return torch.relu(y) if relu else y


conv3x3_module = conv_module
