#!/usr/bin/env python3
"""Build oracle/_ref/libjdet_ref.so from the reference's OWN CPU sources.  TEST INFRASTRUCTURE.

The reference (Jittor/JDet) keeps its C++ kernels as Python string literals that Jittor JIT-compiles through
`jt.code`.  Jittor itself is not installable here, so this recipe reads the string constants with `ast` from the
files WHERE THEY LIE under /root/reference (nothing is copied into the repo), concatenates the ones that are plain
host C++ in a temp dir with `extern "C"` entry points that call them the way the `cpu_src` snippets do, and compiles
with g++ -O2 -ffp-contract=off into oracle/_ref/ (git-ignored; it travels to the GPU box with the snapshot, where
bench.py's cpu_baseline leg may time it).  Only the .so lands there; the generated .cpp lives and dies in a
TemporaryDirectory.

What is built: the reference's true CPU sources -- rotated IoU (box_iou_rotated.py IOU_ROTATED_CPU_HEADER, v0 and
_v1), rotated NMS (nms_rotated.py ML_NMS_ROTATED_CPU_HEADER, BOX_LENGTH 5 and 6) and the active rotating filter
(orn.py ARF_CPU_HEADER).  The only edit is dropping `#include <executor.h>`, a Jittor header the arithmetic never
uses.

What is NOT built here: RoIAlign (x5 dialects), DeformConv and the other CUDA-only operators (`__global__`, blockIdx,
atomicAdd ...).  Compiling them for the HOST would need stand-ins for the CUDA built-ins, i.e. it would not be "the
reference compiled here" (round 1 did that behind a shim; removed).  Compiling them for the GPU needs none: the
kernel dialect is hipcc's own -- oracle/build_ref_hip.py does that, and tests/test_gpu_reference_kernels.py pins the
restatement (oracle/jdet_oracle.cpp) and the product kernels against those kernels on the device.  On the CPU the
restatement is additionally held to closed forms that involve neither itself nor any reference build:
tests/closed_form.py, tests/test_closed_form_cpu.py.
"""
import ast
import os
import re
import subprocess
import sys
import tempfile

REF = os.environ.get("JDET_REFERENCE", "/root/reference")
OPS = os.path.join(REF, "python", "jdet", "ops")
HERE = os.path.dirname(os.path.abspath(__file__))
OUT_DIR = os.path.join(HERE, "_ref")
OUT_SO = os.path.join(OUT_DIR, "libjdet_ref.so")


def module_strings(path):
    """Evaluate top-level `NAME = <str const | NAME | a + b>` assignments of a module."""
    tree = ast.parse(open(path).read())
    env = {}

    def ev(node):
        if isinstance(node, ast.Constant) and isinstance(node.value, str):
            return node.value
        if isinstance(node, ast.Name) and node.id in env:
            return env[node.id]
        if isinstance(node, ast.BinOp) and isinstance(node.op, ast.Add):
            a, b = ev(node.left), ev(node.right)
            if a is not None and b is not None:
                return a + b
        return None

    for st in tree.body:
        if isinstance(st, ast.Assign) and len(st.targets) == 1 and isinstance(st.targets[0], ast.Name):
            v = ev(st.value)
            if v is not None:
                env[st.targets[0].id] = v
    return env


PRELUDE = r'''
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <climits>
#include <algorithm>
#include <cassert>
#include <vector>
using std::min; using std::max;
'''


def ns(name, body):
    return "namespace %s {\n%s\n}\n" % (name, body)


def build(verbose=True):
    if not os.path.isdir(OPS):
        raise SystemExit("reference not present at %s" % OPS)
    iou = module_strings(os.path.join(OPS, "box_iou_rotated.py"))
    iou1 = module_strings(os.path.join(OPS, "box_iou_rotated_v1.py"))
    nms = module_strings(os.path.join(OPS, "nms_rotated.py"))
    orn = module_strings(os.path.join(OPS, "orn.py"))["ARF_CPU_HEADER"]

    def no_exec(s):
        return re.sub(r"#include\s*<executor.h>", "", s).replace("#undef out", "")

    parts = [PRELUDE]
    parts.append(ns("ref_iou_cpu", no_exec(iou["IOU_ROTATED_CPU_HEADER"])))
    parts.append("#undef HOST_DEVICE\n#undef HOST_DEVICE_INLINE\n#undef CeilDIV\n")
    parts.append(ns("ref_iou_v1_cpu", no_exec(iou1["IOU_ROTATED_CPU_HEADER"])))
    parts.append("#undef HOST_DEVICE\n#undef HOST_DEVICE_INLINE\n#undef CeilDIV\n")
    for bl in (5, 6):
        parts.append("#define BOX_LENGTH %d\n" % bl)
        parts.append(ns("ref_nms%d" % bl, no_exec(nms["ML_NMS_ROTATED_CPU_HEADER"])))
        parts.append("#undef BOX_LENGTH\n#undef HOST_DEVICE\n#undef HOST_DEVICE_INLINE\n#undef CeilDIV\n")
    parts.append(ns("ref_orn", orn))

    # Entry points: argument order copied from the reference's jt.code source snippets.
    parts.append(r'''
#define API extern "C" __attribute__((visibility("default")))
// box_iou_rotated.py:L487-500 (IOU_CPU_SRC)
API void ref_box_iou_rotated(const float* b1, int n1, const float* b2, int n2, int nps, float* ious) {
  for (int i = 0; i < n1; i++) for (int j = 0; j < n2; j++)
    ious[i * n2 + j] = ref_iou_cpu::single_box_iou_rotated(b1 + i * nps, b2 + j * nps);
}
API void ref_box_iou_rotated_v1(const float* b1, int n1, const float* b2, int n2, int nps, float* ious) {
  for (int i = 0; i < n1; i++) for (int j = 0; j < n2; j++)
    ious[i * n2 + j] = ref_iou_v1_cpu::single_box_iou_rotated(b1 + i * nps, b2 + j * nps);
}
// nms_rotated.py:L414-449 (ML_NMS_ROTATED_CPU_SRC), BOX_LENGTH 5 and 6
#define NMS_ENTRY(NS, NAME, BL)                                                               \
API void NAME(const float* dets, int ndets, const int* order, float iou_threshold,            \
              unsigned char* keep) {                                                          \
  std::vector<unsigned char> suppressed(ndets, 0);                                            \
  memset(keep, 0, ndets);                                                                     \
  for (int _i = 0; _i < ndets; _i++) {                                                        \
    auto i = order[_i];                                                                       \
    if (suppressed[i] == 1) continue;                                                         \
    keep[i] = true;                                                                           \
    for (int _j = _i + 1; _j < ndets; _j++) {                                                 \
      auto j = order[_j];                                                                     \
      if (suppressed[j] == 1) continue;                                                       \
      auto ovr = NS::single_box_iou_rotated(dets + i * BL, dets + j * BL);                    \
      if (ovr >= iou_threshold) suppressed[j] = 1;                                            \
    }                                                                                         \
  }                                                                                           \
}
NMS_ENTRY(ref_nms5, ref_nms_rotated5, 5)
NMS_ENTRY(ref_nms6, ref_nms_rotated6, 6)
// orn.py:L213-233, L235-257
API void ref_arf_forward(const float* w, const unsigned char* idx, int nOut, int nIn, int nOri,
                         int kH, int kW, int nRot, float* out) {
  ref_orn::ARF_forward_cpu_kernel<float>(w, idx, nOut, nIn, nOri, kH, kW, nRot, out);
}
API void ref_arf_backward(const unsigned char* idx, const float* go, int nOut, int nIn, int nOri,
                          int kH, int kW, int nRot, float* gw) {
  ref_orn::ARF_backward_cpu_kernel<float>(idx, go, nOut, nIn, nOri, kH, kW, nRot, gw);
}
''')
    os.makedirs(OUT_DIR, exist_ok=True)
    with tempfile.TemporaryDirectory() as td:
        cpp = os.path.join(td, "ref_all.cpp")
        with open(cpp, "w") as f:
            f.write("\n".join(parts))
        cmd = ["g++", "-O2", "-ffp-contract=off", "-std=c++14", "-shared", "-fPIC", "-w",
               "-fvisibility=hidden", cpp, "-o", OUT_SO]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            sys.stderr.write(r.stderr[-6000:])
            raise SystemExit("reference build failed")
    if verbose:
        print("built", OUT_SO)
    return OUT_SO


if __name__ == "__main__":
    build()
