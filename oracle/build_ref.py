#!/usr/bin/env python3
"""Build oracle/_ref/libjdet_ref.so from the reference's OWN kernel text.  TEST INFRASTRUCTURE.

The reference (Jittor/JDet) keeps its C++/CUDA kernels as Python string literals that
Jittor JIT-compiles through `jt.code`.  Jittor itself is not installable here, so this
recipe reads the string constants with `ast` from the files WHERE THEY LIE under
/root/reference (nothing is copied into the repo), concatenates them in a temp dir with

  * a 20-line host prelude that gives the CUDA-only kernels a single-thread meaning
    (`__global__`/`__device__` empty, blockIdx=threadIdx=0, blockDim=gridDim=1,
    atomicAdd -> +=) -- the per-element arithmetic, which is what parity is about,
    is compiled unmodified;
  * `extern "C"` entry points that call the reference's kernel functions the way the
    `jt.code` source snippets do (same argument order as the cuda_src/cpu_src strings),

and compiles with g++ -O2 -ffp-contract=off into oracle/_ref/ (git-ignored).  Only the
.so lands there; the generated .cpp lives and dies in a TemporaryDirectory.

Honest limits (also in DESIGN.md): the IoU / NMS / ARF strings are the reference's true
CPU sources; the RoIAlign / DeformConv strings are CUDA-only in the reference and are
host-shimmed as described, so for those "reference on CPU" means "the reference's kernel
text, executed serially".  `#include <executor.h>` (a Jittor header the arithmetic never
uses) is dropped.  Nothing here runs on the GPU box: /root/reference does not exist
there; tests use the committed tests/golden/*.npz made by tests/golden/gen_golden.py.
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
#define __global__
#define __device__
#define __host__
#define __forceinline__ inline
struct _dim3 { int x, y, z; };
static const _dim3 blockIdx = {0, 0, 0}, threadIdx = {0, 0, 0}, blockDim = {1, 1, 1}, gridDim = {1, 1, 1};
template <typename T> static inline void atomicAdd(T* p, T v) { *p += v; }
'''


def strip_launchers(src):
    """Drop host launcher templates that use <<< >>> (riroi_align.py:L166-181, L360-380)."""
    out, i = [], 0
    pat = re.compile(r"template\s*<typename scalar_t>\s*int\s+\w+Laucher\s*\(")
    while True:
        m = pat.search(src, i)
        if not m:
            out.append(src[i:])
            break
        out.append(src[i:m.start()])
        j = src.index("{", m.end())
        depth = 0
        while True:
            if src[j] == "{":
                depth += 1
            elif src[j] == "}":
                depth -= 1
                if depth == 0:
                    break
            j += 1
        i = j + 1
    return "".join(out)


def ns(name, body):
    return "namespace %s {\n%s\n}\n" % (name, body)


def build(verbose=True):
    if not os.path.isdir(OPS):
        raise SystemExit("reference not present at %s" % OPS)
    rot = module_strings(os.path.join(OPS, "roi_align_rotated.py"))["CUDA_HEADER"]
    rot1 = module_strings(os.path.join(OPS, "roi_align_rotated_v1.py"))["CUDA_HEADER"]
    ri = strip_launchers(module_strings(os.path.join(OPS, "riroi_align.py"))["riroi_cuda_head"])
    hbb = module_strings(os.path.join(OPS, "roi_align.py"))["CUDA_HEADER"]
    iou = module_strings(os.path.join(OPS, "box_iou_rotated.py"))
    iou1 = module_strings(os.path.join(OPS, "box_iou_rotated_v1.py"))
    nms = module_strings(os.path.join(OPS, "nms_rotated.py"))
    dcn = module_strings(os.path.join(OPS, "dcn_v1.py"))["HEADER"]
    orn = module_strings(os.path.join(OPS, "orn.py"))["ARF_CPU_HEADER"]

    def no_exec(s):
        return re.sub(r"#include\s*<executor.h>", "", s).replace("#undef out", "")

    def cuda_sort_header(mod, h1, h2, h3):
        # the CUDA exchange sort (box_iou_rotated.py:L335-351) between HEADER2 and HEADER3,
        # cut off before the __global__ kernels that follow HEADER3 in the CUDA header
        full = mod["IOU_ROTATED_CUDA_HEADER"]
        start = full.index(mod[h2]) + len(mod[h2])
        end = full.index(mod[h3])
        return (no_exec(mod[h1]) + "#define HOST_DEVICE\n#define HOST_DEVICE_INLINE inline\n"
                + mod[h2] + full[start:end] + mod[h3])

    parts = [PRELUDE]
    # RoIAlign family: each header defines the same helper names -> one namespace each
    parts.append(ns("ref_rot", rot))
    parts.append(ns("ref_rot1", rot1))
    parts.append(ns("ref_ri", ri))
    parts.append(ns("ref_hbb0", "#define ROI_ALIGN_VERSION 0\n" + hbb + "\n#undef ROI_ALIGN_VERSION\n"
                    ).replace("using namespace std;", ""))
    parts.append(ns("ref_hbb1", "#define ROI_ALIGN_VERSION 1\n" + hbb + "\n#undef ROI_ALIGN_VERSION\n"
                    ).replace("using namespace std;", ""))
    parts.append("#undef CUDA_1D_KERNEL_LOOP\n#undef THREADS_PER_BLOCK\n#undef PI\n")
    parts.append(ns("ref_iou_cpu", no_exec(iou["IOU_ROTATED_CPU_HEADER"])))
    parts.append("#undef HOST_DEVICE\n#undef HOST_DEVICE_INLINE\n#undef CeilDIV\n")
    parts.append(ns("ref_iou_v1_cpu", no_exec(iou1["IOU_ROTATED_CPU_HEADER"])))
    parts.append("#undef HOST_DEVICE\n#undef HOST_DEVICE_INLINE\n#undef CeilDIV\n")
    parts.append(ns("ref_iou_cudasort", cuda_sort_header(iou, "IOU_ROTATED_HEADER1",
                                                         "IOU_ROTATED_HEADER2", "IOU_ROTATED_HEADER3")))
    parts.append("#undef HOST_DEVICE\n#undef HOST_DEVICE_INLINE\n#undef CeilDIV\n")
    for bl in (5, 6):
        parts.append("#define BOX_LENGTH %d\n" % bl)
        parts.append(ns("ref_nms%d" % bl, no_exec(nms["ML_NMS_ROTATED_CPU_HEADER"])))
        parts.append("#undef BOX_LENGTH\n#undef HOST_DEVICE\n#undef HOST_DEVICE_INLINE\n#undef CeilDIV\n")
    parts.append(ns("ref_dcn", no_exec(dcn)))
    parts.append(ns("ref_orn", orn))

    # Entry points: argument order copied from the reference's jt.code source snippets.
    parts.append(r'''
#define API extern "C" __attribute__((visibility("default")))
// roi_align_rotated.py:L278-282 / L301-306 (and _v1 L321-325/L344-349)
#define ROT_ENTRY(NS, NAME)                                                                   \
API void NAME##_fwd(const float* in, const float* rois, int R, int C, int H, int W, int PH,   \
                    int PW, float scale, int sample, float* out) {                            \
  int n = R * PH * PW * C;                                                                    \
  NS::ROIAlignRotatedForward<float>(n, in, rois, scale, sample, C, H, W, PH, PW, out);        \
}                                                                                             \
API void NAME##_bwd(const float* g, const float* rois, int R, int N, int C, int H, int W,     \
                    int PH, int PW, float scale, int sample, float* gin) {                    \
  memset(gin, 0, sizeof(float) * (size_t)N * C * H * W);                                      \
  int n = R * PH * PW * C;                                                                    \
  NS::ROIAlignBackward<float>(n, g, rois, scale, sample, C, H, W, PH, PW, gin);               \
}
ROT_ENTRY(ref_rot, ref_roi_align_rotated)
ROT_ENTRY(ref_rot1, ref_roi_align_rotated_v1)
// riroi_align.py:L174-179 / L368-378 ; C = channels per orientation
API void ref_riroi_align_fwd(const float* in, const float* rois, int R, int C, int H, int W,
                             int PH, int PW, float scale, int sample, int nO, float* out) {
  int n = R * PH * PW * C * nO;
  ref_ri::RiROIAlignForward<float>(n, in, rois, scale, sample, C, H, W, PH, PW, nO, out);
}
API void ref_riroi_align_bwd(const float* g, const float* rois, int R, int N, int C, int H, int W,
                             int PH, int PW, float scale, int sample, int nO, float* gin) {
  memset(gin, 0, sizeof(float) * (size_t)N * C * nO * H * W);
  int n = R * PH * PW * C * nO;
  ref_ri::RiROIAlignBackward<float>(n, g, rois, scale, sample, C, H, W, PH, PW, nO, gin);
}
// roi_align.py:L232-236 / L257-262 ; sampling_ratio is a float parameter there
#define HBB_ENTRY(NS, NAME)                                                                   \
API void NAME##_fwd(const float* in, const float* rois, int R, int C, int H, int W, int PH,   \
                    int PW, float scale, float sample, float* out) {                          \
  int n = R * PH * PW * C;                                                                    \
  NS::RoIAlignForward(n, in, C, H, W, PH, PW, rois, out, scale, sample);                      \
}                                                                                             \
API void NAME##_bwd(const float* g, const float* rois, int R, int N, int C, int H, int W,     \
                    int PH, int PW, float scale, float sample, float* gin) {                  \
  memset(gin, 0, sizeof(float) * (size_t)N * C * H * W);                                      \
  int n = R * PH * PW * C;                                                                    \
  NS::RoIAlignBackwardFeature(n, g, R, C, H, W, PH, PW, gin, rois, scale, sample);            \
}
HBB_ENTRY(ref_hbb0, ref_roi_align_v0)
HBB_ENTRY(ref_hbb1, ref_roi_align_v1)
// box_iou_rotated.py:L487-500 (IOU_CPU_SRC)
API void ref_box_iou_rotated(const float* b1, int n1, const float* b2, int n2, int nps, float* ious) {
  for (int i = 0; i < n1; i++) for (int j = 0; j < n2; j++)
    ious[i * n2 + j] = ref_iou_cpu::single_box_iou_rotated(b1 + i * nps, b2 + j * nps);
}
API void ref_box_iou_rotated_v1(const float* b1, int n1, const float* b2, int n2, int nps, float* ious) {
  for (int i = 0; i < n1; i++) for (int j = 0; j < n2; j++)
    ious[i * n2 + j] = ref_iou_v1_cpu::single_box_iou_rotated(b1 + i * nps, b2 + j * nps);
}
API void ref_box_iou_rotated_cudasort(const float* b1, int n1, const float* b2, int n2, int nps, float* ious) {
  for (int i = 0; i < n1; i++) for (int j = 0; j < n2; j++)
    ious[i * n2 + j] = ref_iou_cudasort::single_box_iou_rotated(b1 + i * nps, b2 + j * nps);
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
// dcn_v1.py:L327-338, L362-372, L398-409
static inline int out_sz(int in, int pad, int dil, int k, int stride) {
  return (in + 2 * pad - (dil * (k - 1) + 1)) / stride + 1;
}
API void ref_deform_im2col(const float* im, const float* off, int B, int C, int H, int W, int kh,
                           int kw, int ph, int pw, int sh, int sw, int dh, int dw, int dg, float* col) {
  int Ho = out_sz(H, ph, dh, kh, sh), Wo = out_sz(W, pw, dw, kw, sw);
  int n = C * Ho * Wo * B;
  memset(col, 0, sizeof(float) * (size_t)C * kh * kw * B * Ho * Wo);
  ref_dcn::deformable_im2col_gpu_kernel<float>(n, im, off, H, W, kh, kw, ph, pw, sh, sw, dh, dw,
                                               C / dg, B, C, dg, Ho, Wo, col);
}
API void ref_deform_col2im(const float* col, const float* off, int B, int C, int H, int W, int kh,
                           int kw, int ph, int pw, int sh, int sw, int dh, int dw, int dg, float* gim) {
  int Ho = out_sz(H, ph, dh, kh, sh), Wo = out_sz(W, pw, dw, kw, sw);
  int n = C * kh * kw * Ho * Wo * B;
  memset(gim, 0, sizeof(float) * (size_t)B * C * H * W);
  ref_dcn::deformable_col2im_gpu_kernel<float>(n, col, off, C, H, W, kh, kw, ph, pw, sh, sw, dh, dw,
                                               C / dg, B, dg, Ho, Wo, gim, 0);
}
API void ref_deform_col2im_coord(const float* col, const float* im, const float* off, int B, int C,
                                 int H, int W, int kh, int kw, int ph, int pw, int sh, int sw,
                                 int dh, int dw, int dg, float* goff) {
  int Ho = out_sz(H, ph, dh, kh, sh), Wo = out_sz(W, pw, dw, kw, sw);
  int n = Ho * Wo * 2 * kh * kw * dg * B;
  ref_dcn::deformable_col2im_coord_gpu_kernel<float>(n, col, im, off, C, H, W, kh, kw, ph, pw, sh, sw,
                                                     dh, dw, C * kh * kw / dg, B, 2 * kh * kw * dg,
                                                     dg, Ho, Wo, goff);
}
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
