"""CPU: libjdet_hip.so builds, loads and exports every symbol include/jdet_hip.h declares, with the
argument counts the ctypes binding uses.  No compute calls (no GPU here)."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    src = open(os.path.join(ROOT, "include", "jdet_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    out = {}
    for m in re.finditer(r"\b(?:int|size_t)\s+(jdet_\w+)\s*\(([^)]*)\)\s*;", src):
        args = [a for a in m.group(2).split(",") if a.strip() and a.strip() != "void"]
        out[m.group(1)] = len(args)
    return out


@pytest.fixture(scope="module")
def built_lib():
    from jdet_amd import _lib
    import shutil
    if shutil.which("hipcc") or os.path.exists("/opt/rocm/bin/hipcc"):
        _lib.build()
    if not os.path.exists(_lib.LIB_PATH):
        pytest.skip("libjdet_hip.so not built and hipcc absent")
    return _lib


def test_header_declares_expected_surface():
    d = _declared()
    for name in ("jdet_roi_align_forward", "jdet_roi_align_backward", "jdet_box_iou_rotated", "jdet_nms_rotated",
                 "jdet_nms_rotated_workspace", "jdet_deform_im2col", "jdet_deform_col2im",
                 "jdet_deform_col2im_coord", "jdet_arf_forward", "jdet_arf_backward", "jdet_nchw_to_nhwc",
                 "jdet_nhwc_to_nchw", "jdet_version"):
        assert name in d, name


def test_every_declared_symbol_is_exported(built_lib):
    raw = ctypes.CDLL(built_lib.LIB_PATH)
    for name in _declared():
        assert hasattr(raw, name), "missing export " + name


def test_ctypes_signatures_match_header(built_lib):
    d = _declared()
    assert set(d) == set(built_lib.SIGNATURES), set(d) ^ set(built_lib.SIGNATURES)
    for name, nargs in d.items():
        assert len(built_lib.SIGNATURES[name][1]) == nargs, name
    lib = built_lib.lib()
    assert lib.jdet_version() >= 1
    assert lib.jdet_nms_rotated_workspace(0) == 0
    assert lib.jdet_nms_rotated_workspace(65) >= 65 * 2 * 8 + 2 * 4


def test_experimental_library_is_separate(built_lib):
    """include/jdet_experimental.h: every declared symbol is exported by libjdet_experimental.so, none of them by the
    product library, and the product python never loads it"""
    from jdet_amd import _experimental as X
    src = re.sub(r"/\*.*?\*/", "", open(os.path.join(ROOT, "include", "jdet_experimental.h")).read(), flags=re.S)
    names = re.findall(r"\b(?:int|size_t)\s+(jdet_\w+)\s*\(", src)
    assert set(names) == set(X.SIGNATURES)
    if not os.path.exists(X.LIB_PATH):
        pytest.skip("libjdet_experimental.so not built")
    raw, prod = ctypes.CDLL(X.LIB_PATH), ctypes.CDLL(built_lib.LIB_PATH)
    for n in names:
        assert hasattr(raw, n) and not hasattr(prod, n), n
    for dirpath, _, files in os.walk(os.path.join(ROOT, "jdet_amd")):
        for f in files:
            if f.endswith(".py") and f != "_experimental.py":
                assert "_experimental" not in open(os.path.join(dirpath, f)).read(), f


def test_product_path_has_no_cpu_fallback():
    import torch
    from jdet_amd._lib import JDetHipError
    from jdet_amd.ops import box_iou_rotated
    from jdet_amd.ops.roi_align_rotated import ROIAlignRotated
    with pytest.raises(JDetHipError):
        box_iou_rotated(torch.zeros(2, 5), torch.zeros(3, 5))
    with pytest.raises(JDetHipError):
        ROIAlignRotated(7, 0.25, 2)(torch.zeros(1, 4, 8, 8), torch.zeros(1, 6))


def test_product_does_not_import_oracle():
    """jdet_amd must never reach into oracle/ (the oracle is the checker, not a fallback)."""
    import subprocess
    import sys
    code = ("import sys; import jdet_amd, jdet_amd.ops; import jdet_amd.ops.nms_rotated, jdet_amd.ops.dcn_v1, "
            "jdet_amd.ops.orn; bad=[m for m in sys.modules if m=='oracle' or m.startswith('oracle.')]; "
            "sys.exit(1 if bad else 0)")
    assert subprocess.run([sys.executable, "-c", code], cwd=ROOT).returncode == 0
    for dirpath, _, files in os.walk(os.path.join(ROOT, "jdet_amd")):
        for f in files:
            if f.endswith((".py", ".hip", ".h")):
                txt = open(os.path.join(dirpath, f)).read()
                assert "import oracle" not in txt and "from oracle" not in txt and "libjdet_oracle" not in txt, f


def test_bench_contract_pieces_importable_without_gpu():
    """bench.py imports on a CPU-only host; the objects it adds to the JSON line carry the contract's keys; the
    workload table names the four BASELINE configs"""
    import importlib
    bench = importlib.import_module("bench")
    assert {"s2anet_train", "orcnn_train", "roitrans_train", "roitrans_r50_train"} <= set(bench.TRAIN_WORKLOADS)
    r = bench.roofline_obj("roi_align_rotated", 167508864, 0.064, "k")
    assert {"bound", "achieved", "peak", "unit", "frac", "traffic"} <= set(r)
    assert r["bound"] == "hbm" and r["unit"] == "GB/s" and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-12
    assert abs(r["achieved"] - 167508864 / 1e9 / 0.064e-3) < 1e-6
    assert r["traffic"] is None or r["traffic"] > 100e6          # PMC bytes per launch, when profiles/ has them
    assert r["traffic"] is None or "not live" in r["traffic_source"]     # the line says where the constant comes from
    assert "peak_measured" not in r                              # (measured on a device only: bench passes `dev`)
    assert callable(bench.measured_copy_peak) and callable(bench.secondary_lines)
    for name in ("S2ANET_CFG", "RETINANET_CFG", "ORCNN_CFG"):
        cfg = getattr(bench, name)
        assert cfg["model"]["type"] and cfg["model"]["backbone"]["type"].startswith("Resnet")
    assert bench.roitrans_train_cfg("Resnet101")["model"]["backbone"]["type"] == "Resnet101"


def test_bench_gpus_flag_launches_that_many_ranks():
    """`python bench.py --gpus 2` started by hand (no WORLD_SIZE) re-launches itself as 2 ranks of one node and the
    JSON line reports the size of the initialised group (the flag used to be parsed and ignored).
    Driven on CPU ranks through the plumbing-only `selftest_cpu` workload."""
    import json
    import subprocess
    import sys
    import importlib
    bench = importlib.import_module("bench")
    cmd = bench.launch_command(["--gpus", "2", "--workload", "s2anet_train"], 2, port=12345)
    assert cmd[1:3] == ["-m", "torch.distributed.run"] and "--nproc-per-node" in cmd and cmd[cmd.index("--nproc-per-node") + 1] == "2"
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1" and cmd[-4:] == ["--gpus", "2", "--workload", "s2anet_train"]
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--workload", "selftest_cpu",
                        "--steps", "5", "--warmup", "1"], capture_output=True, text=True, env=env, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.strip().splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout
    line = json.loads(lines[0])
    assert line["n_gpus"] == 2 and line["steps"] == 5 and line["config"]["parallelism"] == "replicas x2"


def test_argument_validation_returns_before_any_launch(built_lib):
    """every entry point checks its arguments first and reports through the int status (no launch, so this runs
    without a GPU): JDET_E_BADARG -1, JDET_E_UNSUPPORTED -2, JDET_E_WORKSPACE -3"""
    lib = built_lib.lib()
    N = None
    assert lib.jdet_roi_align_forward(9, N, 1, 4, 8, 8, N, 1, 7, 7, 1.0, 2, 1, N, N, N) == -1       # variant
    assert lib.jdet_roi_align_forward(0, N, 1, 4, 8, 8, N, 0, 7, 7, 1.0, 2, 1, N, N, N) == 0        # R = 0: no-op
    assert lib.jdet_roi_align_forward(0, N, 1, 4, 8, 8, N, 3, 7, 7, 1.0, 2, 1, N, N, N) == -1       # null pointers
    assert lib.jdet_roi_align_forward(0, N, 1, 4, 8, 8, N, 0, 17, 17, 1.0, 2, 1, N, N, N) == -2     # > 256 bins
    assert lib.jdet_box_iou_rotated(N, 2, N, 2, 4, 0, 0, N, N) == -1                                # stride < 5
    assert lib.jdet_box_iou_rotated(N, 0, N, 5, 5, 0, 0, N, N) == 0
    assert lib.jdet_nms_rotated(N, 10, 7, N, 0.1, 1, 0, N, N, 0, N) == -1                           # box_len
    assert lib.jdet_nms_rotated(N, 0, 5, N, 0.1, 1, 0, N, N, 0, N) == 0
    assert lib.jdet_assign_max_iou(N, 0, 10, 0.5, 0.0, 0.4, 0.0, 1, 1, N, 0, N, N, N, N, 0, N) == -1  # K = 0
    assert lib.jdet_anchor_targets_rotated(N, N, N, N, 0, 0, N, N, 1.0, N, N, N, N, N, N) == 0      # A = 0
    assert lib.jdet_anchor_targets_rotated(N, N, N, N, 5, 1, N, N, 1.0, N, N, N, N, N, N) == -1
    assert lib.jdet_frozen_bn_act_forward(N, N, 4, 6, N, N, N, N, 1e-5, 1, N, N) == -2              # C % 4
    assert lib.jdet_frozen_bn_act_forward(N, N, 4, 12, N, N, N, N, 1e-5, 1, N, N) == -2             # C/4 !| 256
    assert lib.jdet_frozen_bn_act_forward(N, N, 0, 64, N, N, N, N, 1e-5, 1, N, N) == 0
    assert lib.jdet_frozen_bn_act_forward(N, N, 4, 64, N, N, N, N, 1e-5, 1, N, N) == -1
    assert lib.jdet_frozen_bn_act_backward_workspace(4, 12) == 0
    assert lib.jdet_frozen_bn_act_backward_workspace(1 << 20, 256) == 4 * 512 * 2 * 256
    assert lib.jdet_sigmoid_focal_loss(N, N, N, -1, 15, 0.25, 2.0, N, N, N, 0, N) == -1
    assert lib.jdet_smooth_l1_loss(N, N, N, 10, -1.0, N, N, N, 0, N) == -1
    assert lib.jdet_align_conv_offset(N, 1, 8, 8, 8.0, 2, N, N) == -1                               # even kernel
    assert lib.jdet_align_conv_offset(N, 0, 8, 8, 8.0, 3, N, N) == 0
    assert lib.jdet_deform_im2col_nhwc(N, N, 1, 6, 8, 8, 3, 3, 1, 1, 1, 1, 1, 1, N, N) == -2        # C % 4
    assert lib.jdet_deform_col2im_nhwc_workspace(1, 6, 8, 8, 3, 3, 1, 1, 1, 1, 1, 1) == 0
    assert not hasattr(ctypes.CDLL(built_lib.LIB_PATH), "jdet_set_roi_forward_mode")   # no process-wide mode in the ABI
    assert lib.jdet_roi_align_forward_reference(0, N, 1, 8, 8, 8, N, 0, 7, 7, 1.0, 2, 1, N, N, N) == 0    # R = 0
    assert lib.jdet_roi_align_forward_cl_reference(0, N, 1, 6, 8, 8, N, 0, 7, 7, 1.0, 2, 1, N, N, 0, N) == -2   # C % 4
    # round-2 entry points
    assert lib.jdet_roi_align_forward_cl_roi(2, N, 1, 16, 8, 8, N, 0, 7, 7, 1.0, 2, 3, N, N, N) == -1  # RiRoI: C % nO
    assert lib.jdet_roi_align_forward_cl_roi(2, N, 1, 12, 8, 8, N, 0, 7, 7, 1.0, 2, 2, N, N, N) == -2  # RiRoI nO = 2
    assert lib.jdet_roi_align_forward_cl_roi(0, N, 1, 6, 8, 8, N, 0, 7, 7, 1.0, 2, 1, N, N, N) == -2   # C % 4
    assert lib.jdet_roi_align_forward_cl_roi(0, N, 1, 8, 8, 8, N, 0, 7, 7, 1.0, 2, 1, N, N, N) == 0    # R = 0
    assert lib.jdet_roi_align_forward_cl_roi(0, N, 1, 8, 8, 8, N, 2, 7, 7, 1.0, 2, 1, N, N, N) == -1   # null pointers
    assert lib.jdet_roi_align_backward_workspace(0, 10, 1, 8, 8, 8, 7, 7, 0) == 0                      # adaptive sampling
    assert lib.jdet_roi_align_backward_workspace(0, 10, 1, 6, 8, 8, 7, 7, 2) == 0                      # C % 4
    need = lib.jdet_roi_align_backward_workspace(0, 10, 1, 8, 8, 8, 7, 7, 2)
    clean = lib.jdet_roi_align_backward_clean_bytes(0, 10, 1, 8, 8, 8, 7, 7, 2)
    assert 0 < clean < need and clean == 4 * (16 + 2)                                   # 4x4 patches + ticket + overflow total
    # the workspace grows with R (the header's promise: a buffer sized for the largest R of a map serves smaller calls)
    sizes = [lib.jdet_roi_align_backward_workspace(0, r, 2, 256, 128, 128, 7, 7, 2) for r in (1, 7, 64, 500, 512, 2000, 5000)]
    assert all(b >= a_ > 0 for a_, b in zip(sizes, sizes[1:]))
    assert lib.jdet_roi_align_backward_cl(0, N, N, 10, 1, 8, 8, 8, 7, 7, 1.0, 0, 1, N, N, 0, 0, N) == -2   # adaptive
    assert lib.jdet_roi_align_backward_cl(0, N, N, 10, 1, 8, 8, 8, 7, 7, 1.0, 2, 1, N, N, 0, 0, N) == -3   # workspace
    assert lib.jdet_roi_align_backward_cl(7, N, N, 10, 1, 8, 8, 8, 7, 7, 1.0, 2, 1, N, N, 0, 0, N) == -1   # variant
    assert lib.jdet_nms_labeled(N, 10, 5, N, 0.1, 1, 0, 0, 4, N, N, 0, N) == -1                        # labels need 6 columns
    assert lib.jdet_nms_labeled(N, 10, 6, N, 0.1, 1, 0, 1, 0, N, N, 0, N) == -1                        # n_labels < 1
    assert lib.jdet_nms_labeled(N, 0, 6, N, 0.1, 1, 0, 1, 5, N, N, 0, N) == 0
    assert lib.jdet_bbox_overlaps_hbb(N, 3, N, 5, 3, 0, 0, 1e-6, N, N, N) == -1                        # stride < 4
    assert lib.jdet_bbox_overlaps_hbb(N, 0, N, 5, 4, 0, 0, 1e-6, N, N, N) == 0
    assert lib.jdet_bbox_overlaps_hbb(N, 3, N, 5, 4, 0, 0, 1e-6, N, N, N) == -1                        # null pointers
    six = built_lib.vecn([0.0] * 6, 6)
    assert lib.jdet_midpoint_offset_decode(N, N, 0, N, N, 0.016, N, N) == -1                           # means / stds are host arrays
    assert lib.jdet_midpoint_offset_decode(N, N, 0, six, six, 0.016, N, N) == 0
    assert lib.jdet_midpoint_offset_decode(N, N, 4, six, six, 0.016, N, N) == -1
    assert lib.jdet_oriented_delta_decode(N, N, 4, 0, six, six, 0.016, N, N) == -1                     # ncls < 1
    assert lib.jdet_feature_refine_forward(N, N, 1, 8, 4, 4, 0.125, 3, N, N) == -1                     # points in {1, 5}
    assert lib.jdet_feature_refine_forward(N, N, 1, 6, 4, 4, 0.125, 5, N, N) == -2                     # C % 4
    assert lib.jdet_feature_refine_forward(N, N, 0, 8, 4, 4, 0.125, 5, N, N) == 0
    assert lib.jdet_feature_refine_backward_workspace(1, 6, 4, 4, 5) == 0
    assert lib.jdet_feature_refine_backward(N, N, 1, 8, 4, 4, 0.125, 5, N, N, 0, N) == -1
    assert lib.jdet_poly_iou(N, 2, 7, N, 2, 8, 0, N, N) == -1                                          # stride < 8
    assert lib.jdet_nms_poly(N, 4, 9, N, 0.1, 0, N, N, 0, N) == -1                                     # n_labels < 1


def test_reference_kernel_build_exports_its_entry_points():
    """oracle/_ref/libjdet_ref_hip.so (the reference's GPU kernel text compiled for gfx950, oracle/build_ref_hip.py) loads
    without a device and carries every entry point oracle/ref_hip.py binds; skipped where it was not built"""
    import re

    import pytest
    from oracle import ref_hip as RH
    if not RH.available():
        pytest.skip("oracle/_ref/libjdet_ref_hip.so not built (needs /root/reference)")
    names = set(re.findall(r"refhip_[a-z0-9_]+", open(RH.__file__).read()))
    names |= {n + sfx for n in ("refhip_roi_align_rotated", "refhip_roi_align_rotated_v1", "refhip_roi_align_v0",
                                "refhip_roi_align_v1") for sfx in ("_forward", "_backward")}
    names -= {"refhip_roi_align_rotated", "refhip_roi_align_rotated_v1", "refhip_roi_align_v0", "refhip_roi_align_v1"}
    for fma in (False, True):
        lib = RH.lib(fma)
        missing = [n for n in sorted(names) if not hasattr(lib, n)]
        assert not missing, missing
    assert len(names) >= 20
