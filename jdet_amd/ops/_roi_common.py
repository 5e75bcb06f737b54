"""Shared autograd plumbing of the RoIAlign dialects (device work happens in csrc/roi_align.hip)."""
import torch

from .. import _lib as L

V_ROT, V_ROT_V1, V_RI, V_HBB0, V_HBB1 = 0, 1, 2, 3, 4


def _pair(x):
    return tuple(x) if isinstance(x, (tuple, list)) else (x, x)


def to_nhwc(x):
    """Return a (N,C,H,W)-shaped fp32 tensor whose memory is NHWC (torch.channels_last).

    The kernels read one bilinear tap as one contiguous C-vector, so NHWC is their native layout
    (DESIGN.md).  A channels_last input (what the conv stack produces) is used as is; an NCHW
    input is transposed once by jdet_nchw_to_nhwc.
    """
    if x.dtype != torch.float32:
        x = x.float()
    if x.is_contiguous(memory_format=torch.channels_last):
        return x
    x = x.contiguous()
    N, C, H, W = x.shape
    y = torch.empty((N, C, H, W), dtype=torch.float32, device=x.device,
                    memory_format=torch.channels_last)
    L.check(L.lib().jdet_nchw_to_nhwc(L.ptr(x), N, C, H, W, L.ptr(y), L.stream_ptr(x)), "jdet_nchw_to_nhwc")
    return y


# below this many RoIs the map traffic is too small for the XCD schedule to matter
SPATIAL_ORDER_MIN_ROIS = 64

# Forward path (process-wide; tests and bench switch it):
#   "roi_cl"     RoI-stationary kernels (csrc/roi_align.hip), channels-last result stored straight from registers
#                [default]
#   "roi"        the same kernels with the reference's (R,C,PH,PW)-contiguous result (transposed through LDS)
# The arithmetic of the kernels is chosen by the ENTRY POINT (`set_arithmetic`: "merged" = the product entry points,
# "reference" = jdet_roi_align_forward_reference / _cl_reference, the reference's operation order: the parity twin of
# the tests and of smoke()).  Shapes the channels-last store does not take (C % 4 != 0, RiRoIAlign with other than
# 4 / 8 orientations) fall back to "roi".
# Either way the result is the same logical (R, C, PH, PW) tensor; only its strides differ.
# (Two measured alternatives are not product paths: the tile-stationary kernels of round 2 -- removed, DESIGN.md 3.1 --
# and the register-cached plan + pool kernels, csrc/experimental/.)
_FORWARD_PATH = ["roi_cl"]


_ARITHMETIC = ["merged"]


def set_arithmetic(name):
    """ "merged" (default, the product kernels) | "reference" (the reference's operation order: bit-identical to the CPU
    oracle; parity tests / smoke); returns the previous choice.  The C library has no mode of its own."""
    assert name in ("merged", "reference")
    prev = _ARITHMETIC[0]
    _ARITHMETIC[0] = name
    return prev


def _fwd_entry():
    lib = L.lib()
    return lib.jdet_roi_align_forward_reference if _ARITHMETIC[0] == "reference" else lib.jdet_roi_align_forward


def set_forward_path(name):
    assert name in ("roi", "roi_cl")
    prev = _FORWARD_PATH[0]
    _FORWARD_PATH[0] = name
    return prev


def _roi_cl_ok(variant, C, H, W, n_orient=1):
    if variant == V_RI and n_orient not in (4, 8):
        return False
    return _FORWARD_PATH[0] == "roi_cl" and C % 4 == 0 and H * W * C * 4 < (1 << 31)


import collections

_BWD_WS = collections.OrderedDict()
_BWD_WS_MAX = 16   # (device, stream, map size) triples kept: an FPN has 4-5 maps per stream; captures add streams


def _kept_backward_workspace(dev, map_shape, nbytes):
    """Workspace of the channels-last backward, kept per (device, stream, map size): zero-filled once; the call hands
    its counters back zeroed (`workspace_clean` contract of jdet_roi_align_backward_cl), so no memset launch per
    step.  One buffer per map size: the zeroed region's length depends on it.
    Under HIP-graph capture nothing is kept: the buffer is allocated per call from the capturing graph's own pool (and
    passed as scratch of unknown content: the call zeroes its counters itself), so its address lives exactly as long
    as the graph that baked it in -- a cache entry created during one capture could otherwise be evicted (or re-used by
    the next capture on the same capture stream) while replays of the first graph still write to it.
    Returns (workspace, cache key | None = scratch)."""
    if torch.cuda.is_current_stream_capturing():
        return torch.empty((nbytes,), dtype=torch.uint8, device=dev), None
    key = (dev.index, torch.cuda.current_stream(dev).cuda_stream, tuple(map_shape))
    ws = _BWD_WS.get(key)
    if ws is None or ws.numel() < nbytes:
        ws = torch.zeros((nbytes,), dtype=torch.uint8, device=dev)
        _BWD_WS[key] = ws
        while len(_BWD_WS) > _BWD_WS_MAX:       # least recently used out: a dropped buffer is simply re-created zeroed
            _BWD_WS.popitem(last=False)
    _BWD_WS.move_to_end(key)
    return ws, key


# The plan of a backward (jdet_roi_align_backward_plan: the inversion of the scatter, a third of the backward's time)
# depends only on what the forward already knows.  In training it is built at the forward and the backward is the gather
# alone (jdet_roi_align_backward_cl_planned): 45.6 us instead of 66.5 us at the north-star point.
# JDET_ROI_BWD_PLAN=0: the self-contained backward; =side: the plan on a second stream beside the forward kernel (what
# bench.py's roi_align_rotated_pair does by hand) -- measured in the Oriented R-CNN step: same stream 27.57-27.82 ms = the
# self-contained backward's 27.54-27.57 ms, side stream 29.07-29.18 ms (four plan buffers per step handed between streams
# cost more than the 4 x 10 us they hide), so the forward's own stream is the default.
import os

_PLAN_ON = [os.environ.get("JDET_ROI_BWD_PLAN", "1") != "0"]
_PLAN_SIDE_STREAM = [os.environ.get("JDET_ROI_BWD_PLAN", "1") == "side"]
_PLAN_STREAMS = {}


def set_backward_plan(on):
    prev = _PLAN_ON[0]
    _PLAN_ON[0] = bool(on)
    return prev


class BackwardPlan:
    """device buffer + the event after which it is complete"""
    __slots__ = ("buf", "event", "key")

    def __init__(self, buf, event, key):
        self.buf, self.event, self.key = buf, event, key


def build_backward_plan(variant, rois_c, shape, PH, PW, scale, sample_num, n_orient=1):
    """-> BackwardPlan | None (shape served by the unplanned entries: RiRoIAlign, adaptive sampling, C % 4, R == 0)."""
    N, C, H, W = shape
    R = rois_c.shape[0]
    if not _PLAN_ON[0] or variant == V_RI or R == 0 or C % 4 != 0 or sample_num <= 0:
        return None
    lib = L.lib()
    nbytes = lib.jdet_roi_align_backward_plan_bytes(variant, R, N, H, W, PH, PW, sample_num)
    if nbytes == 0:
        return None
    dev = rois_c.device
    buf = torch.empty((nbytes,), dtype=torch.uint8, device=dev)
    cur = torch.cuda.current_stream(dev)
    if torch.cuda.is_current_stream_capturing() or not _PLAN_SIDE_STREAM[0]:
        side = cur          # inside a capture: one stream (the graph's own dependencies order plan and gather)
    else:
        side = _PLAN_STREAMS.get(dev.index)
        if side is None:
            side = _PLAN_STREAMS[dev.index] = torch.cuda.Stream(dev)
        side.wait_stream(cur)            # the RoIs (and the buffer's allocation) are ready
    with torch.cuda.stream(side):
        L.check(lib.jdet_roi_align_backward_plan(variant, L.ptr(rois_c), R, N, H, W, PH, PW, float(scale),
                                                 int(sample_num), L.ptr(buf), nbytes, side.cuda_stream),
                "jdet_roi_align_backward_plan")
        event = None
        if side is not cur:
            event = torch.cuda.Event()
            event.record(side)
            buf.record_stream(side)
            rois_c.record_stream(side)
    return BackwardPlan(buf, event, (variant, R, N, H, W, PH, PW, int(sample_num)))


def _backward_into(variant, g_out, rois_c, shape, PH, PW, scale, sample_num, n_orient, order, plan=None):
    """grad w.r.t. one feature map (NHWC memory).  A channels-last grad_out (what a channels-last forward result
    gets back from a layout-preserving consumer) feeds the sorted gather directly: no transpose pass; with a plan kept
    from the forward the call is the gather alone."""
    N, C, H, W = shape
    R = rois_c.shape[0]
    grad_in = torch.empty((N, C, H, W), dtype=torch.float32, device=g_out.device,
                          memory_format=torch.channels_last)
    if g_out.dtype != torch.float32:
        g_out = g_out.float()
    if (plan is not None and plan.key == (variant, R, N, H, W, PH, PW, int(sample_num)) and C % 4 == 0
            and g_out.is_contiguous(memory_format=torch.channels_last) and not g_out.is_contiguous()):
        if plan.event is not None:
            torch.cuda.current_stream(g_out.device).wait_event(plan.event)
        L.check(L.lib().jdet_roi_align_backward_cl_planned(variant, L.ptr(g_out), R, N, C, H, W, PH, PW,
                                                           int(sample_num), L.ptr(grad_in), L.ptr(plan.buf),
                                                           plan.buf.numel(), L.stream_ptr(g_out)),
                "jdet_roi_align_backward_cl_planned")
        return grad_in
    wsb = L.lib().jdet_roi_align_backward_workspace(variant, R, N, C, H, W, PH, PW, sample_num)
    if wsb and R and g_out.is_contiguous(memory_format=torch.channels_last) and not g_out.is_contiguous():
        ws, key = _kept_backward_workspace(g_out.device, (N, H, W), wsb)
        try:
            L.check(L.lib().jdet_roi_align_backward_cl(variant, L.ptr(g_out), L.ptr(rois_c), R, N, C, H, W, PH, PW,
                                                       scale, sample_num, int(n_orient), L.ptr(grad_in), L.ptr(ws),
                                                       ws.numel(), 1 if key is not None else 0, L.stream_ptr(g_out)),
                    "jdet_roi_align_backward_cl")
        except Exception:
            if key is not None:
                _BWD_WS.pop(key, None)      # state unknown after a failed call: start from a fresh zeroed buffer
            raise
        return grad_in
    ws = torch.empty((wsb,), dtype=torch.uint8, device=g_out.device) if wsb else None
    g = g_out.contiguous()
    L.check(L.lib().jdet_roi_align_backward(variant, L.ptr(g), L.ptr(rois_c), R, N, C, H, W, PH, PW,
                                            scale, sample_num, n_orient, L.ptr(order), L.ptr(grad_in),
                                            L.ptr(ws), wsb, L.stream_ptr(g)), "jdet_roi_align_backward")
    return grad_in


def spatial_order(rois_c, spatial_scale, N, H, W):
    """XCD-aware processing order (jdet_roi_spatial_order); a pure performance hint."""
    R, cols = rois_c.shape
    buf = torch.empty((2, R), dtype=torch.int32, device=rois_c.device)
    L.check(L.lib().jdet_roi_spatial_order(L.ptr(rois_c), R, cols, spatial_scale, N, H, W, buf[0].data_ptr(),
                                           buf[1].data_ptr(), L.stream_ptr(rois_c)), "jdet_roi_spatial_order")
    return buf[0]


def forward_cl(variant, feat, rois_c, out, PH, PW, spatial_scale, sample_num, n_orient):
    """Product forward into a channels-last `out` (jdet_roi_align_forward_cl): the RoI-stationary kernels under the
    XCD-aware order.  RoIs with a negative batch index are skipped (their rows of `out` stay as they are)."""
    N, C, H, W = feat.shape
    R = rois_c.shape[0]
    if R == 0:
        return
    wsb = L.lib().jdet_roi_align_forward_cl_workspace(R, PH, PW)
    ws = torch.empty((wsb,), dtype=torch.uint8, device=feat.device)
    fn = L.lib().jdet_roi_align_forward_cl_reference if _ARITHMETIC[0] == "reference" else L.lib().jdet_roi_align_forward_cl
    L.check(fn(variant, L.ptr(feat), N, C, H, W, L.ptr(rois_c), R, PH, PW, spatial_scale, sample_num, n_orient,
               L.ptr(out), L.ptr(ws), wsb, L.stream_ptr(feat)), "jdet_roi_align_forward_cl")


class RoIAlignFunction(torch.autograd.Function):
    """forward(input (N,C,H,W), rois (R,6|5)) -> (R,C,PH,PW); grad only w.r.t. input
    (reference: `return input_grad, None`, roi_align_rotated.py:L308)."""

    @staticmethod
    def forward(ctx, input, rois, variant, output_size, spatial_scale, sample_num, n_orient):
        L.need_device(input, rois)
        cols = 5 if variant in (V_HBB0, V_HBB1) else 6
        assert rois.dim() == 2 and rois.shape[1] == cols, "rois must be (R,%d)" % cols
        assert input.dim() == 4
        PH, PW = output_size
        feat = to_nhwc(input)
        rois_c = L.f32c(rois)
        N, C, H, W = feat.shape
        R = rois_c.shape[0]
        if _roi_cl_ok(variant, C, H, W, n_orient):
            out = torch.empty((R, C, PH, PW), dtype=torch.float32, device=feat.device,
                              memory_format=torch.channels_last)
            order = None
            forward_cl(variant, feat, rois_c, out, PH, PW, float(spatial_scale), int(sample_num), int(n_orient))
        else:
            out = torch.empty((R, C, PH, PW), dtype=torch.float32, device=feat.device)
            order = spatial_order(rois_c, float(spatial_scale), N, H, W) if R >= SPATIAL_ORDER_MIN_ROIS else None
            L.check(_fwd_entry()(variant, L.ptr(feat), N, C, H, W, L.ptr(rois_c), R, PH, PW, float(spatial_scale),
                                 int(sample_num), int(n_orient), L.ptr(order), L.ptr(out), L.stream_ptr(feat)),
                    "jdet_roi_align_forward")
        ctx.save_for_backward(rois_c, order)
        ctx.cfg = (variant, (N, C, H, W), PH, PW, float(spatial_scale), int(sample_num), int(n_orient))
        ctx.plan = None
        if ctx.needs_input_grad[0] and out.is_contiguous(memory_format=torch.channels_last) and not out.is_contiguous():
            ctx.plan = build_backward_plan(variant, rois_c, (N, C, H, W), PH, PW, float(spatial_scale), int(sample_num),
                                           int(n_orient))
        return out

    @staticmethod
    def backward(ctx, grad_output):
        rois_c, order = ctx.saved_tensors
        variant, shape, PH, PW, scale, sample_num, n_orient = ctx.cfg
        grad_in = _backward_into(variant, grad_output, rois_c, shape, PH, PW, scale, sample_num, n_orient, order,
                                 ctx.plan)
        ctx.plan = None
        return grad_in, None, None, None, None, None, None


class MultiLevelRoIAlignFunction(torch.autograd.Function):
    """FPN-routed RoIAlign: every RoI is pooled from the pyramid level `target_lvls[r]`.

    The reference (oriented_single_level.py:L91-114, rbox_single_level.py:L75-95, single_level.py:L70-85)
    loops over levels with a boolean mask, an `any_()` host sync, a gather, the kernel and a masked
    `+=`.  Here each level is one launch over ALL RoIs on the SAME output buffer, with off-level RoIs
    masked by a negative batch index (skipped inside the kernel): no sync, no gather / scatter-add.
    forward(rois, target_lvls, cfg, *feats) -> (R, C, PH, PW); gradients flow to every level map.
    """

    @staticmethod
    def forward(ctx, rois, target_lvls, cfg, *feats):
        variant, output_size, scales, sample_num, n_orient = cfg
        L.need_device(rois, *feats)
        PH, PW = output_size
        rois_c = L.f32c(rois)
        R = rois_c.shape[0]
        C = feats[0].shape[1]
        roi_cl = all(_roi_cl_ok(variant, C, f.shape[2], f.shape[3], n_orient) for f in feats)
        out = torch.empty((R, C, PH, PW), dtype=torch.float32, device=rois_c.device,
                          memory_format=torch.channels_last if roi_cl else torch.contiguous_format)
        lvl = target_lvls.to(rois_c.device)
        masked, shapes, plans = [], [], []
        for i, f in enumerate(feats):
            fm = to_nhwc(f)
            N, Ci, H, W = fm.shape
            assert Ci == C
            r_i = rois_c.clone()
            r_i[:, 0] = torch.where(lvl == i, rois_c[:, 0], torch.full_like(rois_c[:, 0], -1.0))
            if R and roi_cl:
                forward_cl(variant, fm, r_i, out, PH, PW, float(scales[i]), int(sample_num), int(n_orient))
            elif R:
                L.check(_fwd_entry()(variant, L.ptr(fm), N, C, H, W, L.ptr(r_i), R, PH, PW, float(scales[i]),
                                     int(sample_num), int(n_orient), None, L.ptr(out), L.stream_ptr(fm)),
                        "jdet_roi_align_forward")
            masked.append(r_i)
            shapes.append((N, C, H, W))
            plans.append(build_backward_plan(variant, r_i, (N, C, H, W), PH, PW, float(scales[i]), int(sample_num),
                                             int(n_orient))
                         if R and roi_cl and ctx.needs_input_grad[3 + i] else None)
        ctx.plans = plans
        ctx.save_for_backward(*masked)
        ctx.cfg = (variant, PH, PW, scales, int(sample_num), int(n_orient), shapes)
        return out

    @staticmethod
    def backward(ctx, grad_output):
        masked = ctx.saved_tensors
        variant, PH, PW, scales, sample_num, n_orient, shapes = ctx.cfg
        grads = []
        for i, (r_i, shape) in enumerate(zip(masked, shapes)):
            if not ctx.needs_input_grad[3 + i]:
                grads.append(None)
                continue
            grads.append(_backward_into(variant, grad_output, r_i, shape, PH, PW, float(scales[i]), sample_num,
                                        n_orient, None, ctx.plans[i]))
        ctx.plans = None
        return (None, None, None, *grads)


def multi_level_roi_align(variant, feats, rois, target_lvls, scales, output_size, sample_num, n_orient=1):
    return MultiLevelRoIAlignFunction.apply(rois, target_lvls, (variant, _pair(output_size), tuple(scales),
                                                               sample_num, n_orient), *feats)
