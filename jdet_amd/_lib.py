"""ctypes loader for libjdet_hip.so (the C ABI declared in include/jdet_hip.h).

The product path has NO fallback: if the HIP library is missing or a tensor is not on a HIP
device, the ops raise.  (The CPU restatement under oracle/ is test infrastructure and is never
imported from here.)  PyTorch is used only for device memory and the current stream.
"""
import ctypes
import os
import subprocess

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(_HERE, "csrc")
LIB_PATH = os.path.join(CSRC, "libjdet_hip.so")

_i, _f, _p, _sz, _l = ctypes.c_int, ctypes.c_float, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_long



class BnParams(ctypes.Structure):           # jdet_bn_params_t
    _fields_ = [("weight", _p), ("bias", _p), ("mean", _p), ("var", _p), ("eps", _f)]


class ConvEpilogue(ctypes.Structure):       # jdet_conv_epilogue_t
    _fields_ = [("mode", _i), ("affine", _i), ("relu", _i), ("bn", BnParams), ("residual", _p), ("grad_out", _p),
                ("act", _p), ("sums", _p)]


class BnSumsJob(ctypes.Structure):          # jdet_bn_sums_job_t
    _fields_ = [("partial", _p), ("rows", _l), ("C", _i), ("gamma", _p), ("grad_gamma", _p), ("grad_beta", _p)]


EPI_FORWARD, EPI_ADD, EPI_MASK = 0, 1, 2

# name -> (restype, argtypes); mirrors include/jdet_hip.h one to one (tests/test_abi.py checks it)
SIGNATURES = {
    "jdet_version": (_i, []),
    "jdet_nchw_to_nhwc": (_i, [_p, _i, _i, _i, _i, _p, _p]),
    "jdet_nhwc_to_nchw": (_i, [_p, _i, _i, _i, _i, _p, _p]),
    "jdet_roi_align_forward": (_i, [_i, _p, _i, _i, _i, _i, _p, _i, _i, _i, _f, _i, _i, _p, _p, _p]),
    "jdet_roi_align_forward_cl_roi": (_i, [_i, _p, _i, _i, _i, _i, _p, _i, _i, _i, _f, _i, _i, _p, _p, _p]),
    "jdet_roi_align_forward_cl_workspace": (_sz, [_i, _i, _i]),
    "jdet_roi_align_forward_cl": (_i, [_i, _p, _i, _i, _i, _i, _p, _i, _i, _i, _f, _i, _i, _p, _p, _sz, _p]),
    "jdet_roi_align_forward_reference": (_i, [_i, _p, _i, _i, _i, _i, _p, _i, _i, _i, _f, _i, _i, _p, _p, _p]),
    "jdet_roi_align_forward_cl_reference": (_i, [_i, _p, _i, _i, _i, _i, _p, _i, _i, _i, _f, _i, _i, _p, _p, _sz, _p]),
    "jdet_roi_align_backward_cl": (_i, [_i, _p, _p, _i, _i, _i, _i, _i, _i, _i, _f, _i, _i, _p, _p, _sz, _i, _p]),
    "jdet_roi_align_backward_clean_bytes": (_sz, [_i] * 9),
    "jdet_roi_align_backward_plan_bytes": (_sz, [_i] * 8),
    "jdet_roi_align_backward_plan": (_i, [_i, _p, _i, _i, _i, _i, _i, _i, _f, _i, _p, _sz, _p]),
    "jdet_roi_align_backward_cl_planned": (_i, [_i, _p, _i, _i, _i, _i, _i, _i, _i, _i, _p, _p, _sz, _p]),
    "jdet_roi_align_backward_workspace": (_sz, [_i] * 9),
    "jdet_roi_align_backward": (_i, [_i, _p, _p, _i, _i, _i, _i, _i, _i, _i, _f, _i, _i, _p, _p, _p, _sz, _p]),
    "jdet_roi_spatial_order": (_i, [_p, _i, _i, _f, _i, _i, _i, _p, _p, _p]),
    "jdet_box_iou_rotated": (_i, [_p, _i, _p, _i, _i, _i, _i, _p, _p]),
    "jdet_nms_rotated_workspace": (_sz, [_i]),
    "jdet_nms_rotated": (_i, [_p, _i, _i, _p, _f, _i, _i, _p, _p, _sz, _p]),
    "jdet_upsample_add_nhwc_forward": (_i, [_p, _p, _i, _i, _i, _i, _i, _i, _f, _p, _p]),
    "jdet_upsample_add_nhwc_backward": (_i, [_p, _i, _i, _i, _i, _i, _i, _f, _p, _p]),
    "jdet_zero_fill": (_i, [_p, _sz, _p]),
    "jdet_graph_replace_memset_nodes": (_i, [_p, ctypes.POINTER(ctypes.c_int)]),
    "jdet_sum_squares_workspace": (_sz, []),
    "jdet_sum_squares": (_i, [_p, _sz, _i, _p, _p, _sz, _p]),
    "jdet_normalize_u8_nhwc": (_i, [_p, _p, _i, _i, _i, _p, _p, _i, _p, _p]),
    "jdet_feature_refine_forward": (_i, [_p, _p, _i, _i, _i, _i, _f, _i, _p, _p]),
    "jdet_feature_refine_backward_workspace": (_sz, [_i] * 5),
    "jdet_feature_refine_backward": (_i, [_p, _p, _i, _i, _i, _i, _f, _i, _p, _p, _sz, _p]),
    "jdet_poly_iou": (_i, [_p, _i, _i, _p, _i, _i, _i, _p, _p]),
    "jdet_nms_poly": (_i, [_p, _i, _i, _p, _f, _i, _p, _p, _sz, _p]),
    "jdet_bbox_overlaps_hbb": (_i, [_p, _i, _p, _i, _i, _i, _i, _f, _p, _p, _p]),
    "jdet_nms_labeled": (_i, [_p, _i, _i, _p, _f, _i, _i, _i, _i, _p, _p, _sz, _p]),
    "jdet_deform_im2col": (_i, [_p, _p] + [_i] * 13 + [_p, _p]),
    "jdet_deform_col2im": (_i, [_p, _p] + [_i] * 13 + [_p, _p]),
    "jdet_deform_col2im_coord": (_i, [_p, _p, _p] + [_i] * 13 + [_p, _p]),
    "jdet_modulated_deform_im2col": (_i, [_p, _p, _p] + [_i] * 13 + [_p, _p]),
    "jdet_modulated_deform_col2im": (_i, [_p, _p, _p] + [_i] * 13 + [_p, _p]),
    "jdet_modulated_deform_col2im_coord": (_i, [_p, _p, _p, _p] + [_i] * 13 + [_p, _p, _p]),
    "jdet_deform_psroi_pool_forward": (_i, [_p, _p, _p] + [_i] * 6 + [_f] + [_i] * 5 + [_f, _i, _p, _p, _p]),
    "jdet_deform_psroi_pool_backward": (_i, [_p, _p, _p, _p, _p] + [_i] * 6 + [_f] + [_i] * 5 + [_f, _i, _p, _p, _p]),
    "jdet_deform_im2col_nhwc": (_i, [_p, _p] + [_i] * 12 + [_p, _p]),
    "jdet_deform_col2im_nhwc_workspace": (_sz, [_i] * 12),
    "jdet_deform_col2im_nhwc": (_i, [_p, _p] + [_i] * 12 + [_p, _p, _sz, _p]),
    "jdet_convex_iou": (_i, [_p, _i, _p, _i, _p, _p]),
    "jdet_convex_giou": (_i, [_p, _p, _i, _p, _p]),
    "jdet_min_area_bbox": (_i, [_p, _i, _p, _p]),
    "jdet_convex_sort": (_i, [_p, _p, _i, _i, _i, _p, _p]),
    "jdet_conv3x3_igemm_supported": (_i, [_i, _i]),
    "jdet_conv3x3_igemm_forward": (_i, [_p, _i, _i, _i, _i, _p, _i, _p, _i, _p, _p, _i, _p, _p, _sz, _p]),
    "jdet_conv3x3_igemm_workspace": (_sz, [_i] * 5),
    "jdet_conv3x3_wgrad_supported": (_i, [_i, _i]),
    "jdet_conv3x3_wgrad": (_i, [_p, _p, _p, _i, _i, _i, _i, _i, _p, _i, _p]),
    "jdet_conv_bn_supported": (_i, [_i] * 4),
    "jdet_conv_bn_workspace": (_sz, [_i] * 7),
    "jdet_conv_bn_sums_rows": (_sz, [_i] * 9),
    "jdet_conv_bn_forward": (_i, [_p, _i, _i, _i, _i, _p, _i, _i, _i, ctypes.POINTER(ConvEpilogue), _i, _p, _p, _sz, _p]),
    "jdet_conv_dgrad_weights": (_i, [_p, _i, _i, _p]),
    "jdet_conv_wgrad": (_i, [_p, _p, _i, _i, _i, _i, _i, _i, _i, _p, _i, _p]),
    "jdet_bn_act_backward_from_output_rows": (_sz, [_l, _i]),
    "jdet_bn_act_backward_from_output": (_i, [_p, _p, _p, _p, _l, _i, _p, _p, _p, _p, _f, _p, _p, _sz, _p]),
    "jdet_bn_sums_finish": (_i, [ctypes.POINTER(BnSumsJob), _i, _p]),
    "jdet_sigmoid_focal_loss_workspace": (_sz, []),
    "jdet_sigmoid_focal_loss": (_i, [_p, _p, _p, _l, _i, _f, _f, _p, _p, _p, _sz, _p]),
    "jdet_smooth_l1_loss": (_i, [_p, _p, _p, _l, _f, _p, _p, _p, _sz, _p]),
    "jdet_sigmoid_focal_loss_level": (_i, [_p, _p, _l, _l, _p, _l, _l, _l, _i, _f, _f, _p, _f, _p, _p, _p, _sz, _p]),
    "jdet_smooth_l1_loss_level": (_i, [_p, _p, _l, _l, _p, _l, _l, _l, _i, _f, _p, _f, _p, _p, _p, _sz, _p]),
    "jdet_loss_grad_scale": (_i, [_p, _l, _p, _p, _f, _p, _p]),
    "jdet_level_pack_nhwc": (_i, [_p, _p, _p, _i, _i, _i, _i, _i, _p, _p]),
    "jdet_align_conv_offset": (_i, [_p, _i, _i, _i, _f, _i, _p, _p]),
    "jdet_frozen_bn_act_forward": (_i, [_p, _p, _l, _i, _p, _p, _p, _p, _f, _i, _p, _p]),
    "jdet_frozen_bn_act_backward_workspace": (_sz, [_l, _i]),
    "jdet_frozen_bn_act_backward": (_i, [_p, _p, _p, _l, _i, _p, _p, _p, _p, _f, _i, _p, _p, _p, _p, _p, _sz, _p]),
    "jdet_bias_act_backward": (_i, [_p, _p, _l, _i, _i, _p, _p, _p, _sz, _p]),
    "jdet_channel_sum_workspace": (_sz, [_l, _i]),
    "jdet_channel_sum": (_i, [_p, _l, _i, _p, _p, _sz, _p]),
    "jdet_arf_forward": (_i, [_p, _p] + [_i] * 6 + [_p, _p]),
    "jdet_arf_backward": (_i, [_p, _p] + [_i] * 6 + [_p, _p]),
    "jdet_arf_forward_cl": (_i, [_p, _p] + [_i] * 6 + [_p, _p]),
    "jdet_arf_backward_cl": (_i, [_p, _p] + [_i] * 6 + [_p, _p]),
    "jdet_rip_forward": (_i, [_p, _l, _i, _i, _p, _p]),
    "jdet_rip_backward": (_i, [_p, _p, _p, _l, _i, _i, _p, _p]),
    "jdet_delta2bbox_rotated": (_i, [_p, _p, _i, _i, _p, _p, _f, _p, _p]),
    "jdet_bbox2delta_rotated": (_i, [_p, _p, _i, _p, _p, _p, _p]),
    "jdet_midpoint_offset_decode": (_i, [_p, _p, _l, _p, _p, _f, _p, _p]),
    "jdet_midpoint_offset_encode": (_i, [_p, _p, _l, _p, _p, _p, _p]),
    "jdet_oriented_delta_decode": (_i, [_p, _p, _l, _i, _p, _p, _f, _p, _p]),
    "jdet_oriented_delta_encode": (_i, [_p, _p, _l, _p, _p, _p, _p]),
    "jdet_anchor_targets_rotated": (_i, [_p, _p, _p, _p, _i, _i, _p, _p, _f, _p, _p, _p, _p, _p, _p]),
    "jdet_assign_max_iou_workspace": (_sz, [_i]),
    "jdet_assign_max_iou": (_i, [_p, _i, _i, _f, _f, _f, _f, _i, _i, _p, _i, _p, _p, _p, _p, _sz, _p]),
}

_lib = None

# Hull-point ordering inside the rotated IoU: 0 = the reference's CPU path (std::sort,
# box_iou_rotated.py:L316-325), 1 = its CUDA exchange sort (L338-351).  They differ only on
# degenerate hulls; 0 is "the Jittor CPU reference" BASELINE.json asks parity against.
REFERENCE_SORT = 0


def build(force=False, verbose=False):
    """Compile every .hip under csrc/ for gfx950 and link libjdet_hip.so in-tree."""
    cmd = ["make", "-C", CSRC, "-j8"] + (["-B"] if force else [])
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError("building libjdet_hip.so failed:\n" + r.stdout[-4000:] + r.stderr[-4000:])
    if verbose:
        print(r.stdout[-2000:])
    return LIB_PATH


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                "jdet_amd: %s not found.  Build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                "or `make -C jdet_amd/csrc` (needs hipcc).  There is no CPU fallback." % LIB_PATH)
        # torch is imported above, so its libamdhip64 (SONAME libamdhip64.so.7) is already in the
        # process and our DT_NEEDED entry resolves to that same runtime: one HIP context, shared
        # streams and allocations.
        l = ctypes.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(l, name)  # AttributeError here == ABI drift: fail loudly
            fn.restype = res
            fn.argtypes = args
        _lib = l
    return _lib


class JDetHipError(RuntimeError):
    pass


_ERR = {-1: "bad argument", -2: "unsupported shape", -3: "workspace too small"}


def check(status, what):
    if status != 0:
        msg = _ERR.get(status, "hipError_t %d" % status)
        raise JDetHipError("%s failed: %s" % (what, msg))


def stream_ptr(t=None):
    """launch stream of the op whose tensor is `t`.  The C entry points launch on the CURRENT HIP device, so the
    tensor's device is made current here (every launch site fetches its stream through this function)."""
    if t is not None and t.is_cuda and t.device.index != torch.cuda.current_device():
        torch.cuda.set_device(t.device)
    return torch.cuda.current_stream(t.device if t is not None else None).cuda_stream


def need_device(*tensors):
    for t in tensors:
        if t is None:
            continue
        if not t.is_cuda:
            raise JDetHipError(
                "jdet_amd ops run only on a HIP device (got a %s tensor); there is no CPU fallback "
                "in the product path" % t.device)


def f32c(t):
    """contiguous fp32 view/copy (the reference asserts dtypes instead; fp32 is its only dtype)"""
    if t.dtype != torch.float32:
        t = t.float()
    return t.contiguous()


def vec5(v):
    """host float[5] argument (codec means / stds)"""
    return (ctypes.c_float * 5)(*[float(x) for x in v])


def vecn(v, n):
    """host float[n] argument"""
    v = [float(x) for x in v]
    assert len(v) == n
    return (ctypes.c_float * n)(*v)


def ptr(t):
    return t.data_ptr() if t is not None and t.numel() > 0 else None


def zero_(t):
    """t.zero_() as a plain kernel (jdet_zero_fill): the framework's form records a memset node for large tensors, and
    memset nodes do not reliably re-execute when a captured step is replayed (csrc/graph_safe.hip).  t: a dense fp32 /
    int32 / ... device tensor whose byte size is a multiple of 4."""
    need_device(t)
    nbytes = t.numel() * t.element_size()
    if nbytes % 4 or not (t.is_contiguous() or t.is_contiguous(memory_format=torch.channels_last)):
        return t.zero_()
    check(lib().jdet_zero_fill(ptr(t), nbytes, stream_ptr(t)), "jdet_zero_fill")
    return t


def norm2(flat, ws=None):
    """2-norm of a dense fp32 device tensor as a 0-dim tensor (jdet_sum_squares: two plain kernels, fixed summation order)"""
    need_device(flat)
    n = flat.numel()
    wsb = lib().jdet_sum_squares_workspace()
    if ws is None:
        ws = torch.empty((wsb,), dtype=torch.uint8, device=flat.device)
    out = torch.empty((1,), dtype=torch.float32, device=flat.device)
    check(lib().jdet_sum_squares(ptr(flat), n, 1, ptr(out), ptr(ws), ws.numel(), stream_ptr(flat)), "jdet_sum_squares")
    return out[0]


def new_graph():
    """a torch CUDAGraph whose hipGraph_t stays editable until its first replay (keep_graph)"""
    return torch.cuda.CUDAGraph(keep_graph=True)


def harden_graph(g):
    """run right after capture, before the first replay: memset nodes -> fill-kernel nodes
    (jdet_graph_replace_memset_nodes); returns how many were replaced"""
    n = ctypes.c_int(0)
    check(lib().jdet_graph_replace_memset_nodes(ctypes.c_void_p(g.raw_cuda_graph()), ctypes.byref(n)),
          "jdet_graph_replace_memset_nodes")
    return n.value
