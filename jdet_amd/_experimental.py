"""ctypes loader for libjdet_experimental.so (include/jdet_experimental.h): measured alternatives and calibration
probes.  NOT the product path -- nothing in jdet_amd.ops / jdet_amd.models imports this module; bench.py
(JDET_ROI_FWD_PATH=pool), scripts/ and tests/test_gpu_experimental_pool.py do."""
import ctypes
import os

from . import _lib as L

LIB_PATH = os.path.join(L.CSRC, "experimental", "libjdet_experimental.so")
_i, _f, _p, _sz = ctypes.c_int, ctypes.c_float, ctypes.c_void_p, ctypes.c_size_t

SIGNATURES = {
    "jdet_roi_align_forward_pool_supported": (_i, [_i] * 7),
    "jdet_roi_align_forward_pool_workspace": (_sz, [_i]),
    "jdet_roi_align_forward_pool": (_i, [_i, _p, _i, _i, _i, _i, _p, _i, _i, _i, _f, _i, _p, _p, _sz, _p]),
    "jdet_roi_align_forward_cl_mode_workspace": (_sz, [_i] * 4),
    "jdet_roi_align_forward_cl_mode": (_i, [_i, _i, _p, _i, _i, _i, _i, _p, _i, _i, _i, _f, _i, _i, _p, _p, _p, _sz, _p]),
    "jdet_debug_gather_probe": (_i, [_p, ctypes.c_long, _i, _i, _i, _i, _i, _p, _p]),
    "jdet_debug_gather_width_probe": (_i, [_p, ctypes.c_long, _i, _i, _i, _i, _p, _p]),
    "jdet_debug_gather_accumulate_probe": (_i, [_p, ctypes.c_long, _i, _i, _i, _i, _p, _p]),
    "jdet_debug_dma_probe": (_i, [_p, ctypes.c_long, _i, _i, _i, _i, _i, _i, _i, _p, _p]),
    "jdet_debug_mfma_probe": (_i, [_i, _p, _i, _i, _p, _p, _p]),
}

_lib = None


def lib():
    global _lib
    if _lib is None:
        L.lib()   # the product library first: the experimental one links against it
        if not os.path.exists(LIB_PATH):
            raise RuntimeError("%s not found: `make -C jdet_amd/csrc`" % LIB_PATH)
        l = ctypes.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(l, name)
            fn.restype = res
            fn.argtypes = args
        _lib = l
    return _lib
