// Two framework operations re-written as PLAIN KERNELS because their framework forms put a hipMemset node into a captured
// train step (round 6; scripts/graph_nodes.py lists the node kinds of a captured step): on this stack a memset node does
// not reliably re-execute on replay (csrc/common.h; profiles/r05_graph_notes.md: garbage gradients out of a multi-workgroup
// reduction whose semaphores are cleared by hipMemsetAsync).
//   jdet_zero_fill    -- what `tensor.zero_()` / `torch.zeros` do with a memset above a size threshold
//   jdet_sum_squares  -- sum of squares of a flat fp32 buffer (the gradient norm of SGD's clip, optims/optimizer.py:L26-36;
//                        the framework's vector_norm is a multi-workgroup reduce with memset semaphores): two stages,
//                        fixed summation order (bitwise reproducible), no atomics
#include <stdlib.h>
#include <string.h>

#include "common.h"

namespace {

typedef float v4f __attribute__((ext_vector_type(4)));
constexpr int kSqBlocks = 1024;

__global__ __launch_bounds__(256) void sumsq_partial_kernel(const float* __restrict__ x, size_t n,
                                                            double* __restrict__ partial) {
  __shared__ double s_red[4];
  // grid-stride over float4 groups (x is 16-byte aligned: checked by the host), tail elements by block 0
  const size_t n4 = n >> 2;
  double acc = 0.0;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) {
    const v4f v = reinterpret_cast<const v4f*>(x)[i];
    acc += (double)(v.x * v.x + v.y * v.y) + (double)(v.z * v.z + v.w * v.w);
  }
  if (blockIdx.x == 0)
    for (size_t i = (n4 << 2) + threadIdx.x; i < n; i += 256) acc += (double)(x[i] * x[i]);
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1) acc += __shfl_xor(acc, m, 64);
  if ((threadIdx.x & 63) == 0) s_red[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) partial[blockIdx.x] = (s_red[0] + s_red[1]) + (s_red[2] + s_red[3]);
}

__global__ __launch_bounds__(1024) void sumsq_finish_kernel(const double* __restrict__ partial, int nblocks,
                                                            float* __restrict__ out, int take_sqrt) {
  __shared__ double s_red[16];
  double acc = threadIdx.x < nblocks ? partial[threadIdx.x] : 0.0;
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1) acc += __shfl_xor(acc, m, 64);
  if ((threadIdx.x & 63) == 0) s_red[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) {
    double t = 0.0;
    for (int i = 0; i < 16; i++) t += s_red[i];
    out[0] = take_sqrt ? (float)sqrt(t) : (float)t;
  }
}

}  // namespace

JDET_API int jdet_zero_fill(void* p, size_t bytes, jdet_stream_t stream) {
  if (bytes == 0) return JDET_OK;
  if (!p || (bytes & 3) || (((uintptr_t)p) & 3)) return JDET_E_BADARG;
  return jdet_zero_async(p, bytes, (hipStream_t)stream);
}

JDET_API size_t jdet_sum_squares_workspace(void) { return sizeof(double) * kSqBlocks; }

JDET_API int jdet_sum_squares(const float* x, size_t n, int take_sqrt, float* out, void* workspace, size_t workspace_bytes,
                              jdet_stream_t stream) {
  if (!out || (n && !x) || (((uintptr_t)x) & 15)) return JDET_E_BADARG;
  if (!workspace || workspace_bytes < jdet_sum_squares_workspace() || (((uintptr_t)workspace) & 7)) return JDET_E_WORKSPACE;
  hipStream_t st = (hipStream_t)stream;
  size_t want = ((n >> 2) + 256 * 8 - 1) / (256 * 8);
  const int blocks = (int)(want < 1 ? 1 : (want > kSqBlocks ? kSqBlocks : want));
  hipLaunchKernelGGL(sumsq_partial_kernel, dim3(blocks), dim3(256), 0, st, x, n, (double*)workspace);
  int e = jdet_launch_status();
  if (e) return e;
  hipLaunchKernelGGL(sumsq_finish_kernel, dim3(1), dim3(1024), 0, st, (const double*)workspace, blocks, out,
                     take_sqrt ? 1 : 0);
  return jdet_launch_status();
}

// ---------------------------------------------------------------------------------------------------------------------
// jdet_graph_replace_memset_nodes: the same cure for the memset nodes this repository does NOT issue -- the framework's
// multi-workgroup reductions clear their semaphores with hipMemsetAsync, the convolution library zero-fills the outputs of
// its atomically adding (split-K weight-gradient, some data-gradient) solvers the same way, and which solver runs is
// decided by its benchmark at capture time (scripts/graph_nodes.py: 4-byte ... 32 MiB memset nodes in the captured
// Oriented R-CNN step, different ones from run to run).  A captured hipGraph is edited BEFORE instantiation: every memset
// node is replaced by a kernel node (a fill kernel with the node's destination, pattern and extent) that inherits the
// node's dependencies and dependents.  After the pass the graph holds kernel (and memcpy) nodes only.
namespace {

// element sizes 1 / 2 / 4 (hipMemsetParams); rows of `width` elements, `pitch` bytes apart
__global__ __launch_bounds__(256) void graph_fill_kernel(unsigned char* __restrict__ dst, unsigned value, unsigned esize,
                                                         size_t width, size_t height, size_t pitch) {
  const size_t total = width * height;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
    const size_t r = i / width, c = i - r * width;
    unsigned char* p = dst + r * pitch + c * esize;
    if (esize == 4) *reinterpret_cast<unsigned*>(p) = value;
    else if (esize == 2) *reinterpret_cast<unsigned short*>(p) = (unsigned short)value;
    else *p = (unsigned char)value;
  }
}

}  // namespace

JDET_API int jdet_graph_replace_memset_nodes(void* graph, int* n_replaced) {
  if (n_replaced) *n_replaced = 0;
  if (!graph) return JDET_E_BADARG;
  hipGraph_t g = (hipGraph_t)graph;
  size_t n = 0;
  hipError_t e = hipGraphGetNodes(g, nullptr, &n);
  if (e != hipSuccess) return (int)e;
  if (n == 0) return JDET_OK;
  hipGraphNode_t* nodes = (hipGraphNode_t*)malloc(sizeof(hipGraphNode_t) * n);
  if (!nodes) return JDET_E_WORKSPACE;
  e = hipGraphGetNodes(g, nodes, &n);
  int status = e == hipSuccess ? JDET_OK : (int)e;
  int replaced = 0;
  for (size_t i = 0; i < n && status == JDET_OK; i++) {
    hipGraphNodeType type;
    if ((e = hipGraphNodeGetType(nodes[i], &type)) != hipSuccess) { status = (int)e; break; }
    if (type != hipGraphNodeTypeMemset) continue;
    hipMemsetParams mp;
    if ((e = hipGraphMemsetNodeGetParams(nodes[i], &mp)) != hipSuccess) { status = (int)e; break; }
    size_t nin = 0, nout = 0;
    (void)hipGraphNodeGetDependencies(nodes[i], nullptr, &nin);
    (void)hipGraphNodeGetDependentNodes(nodes[i], nullptr, &nout);
    hipGraphNode_t* in = (hipGraphNode_t*)malloc(sizeof(hipGraphNode_t) * (nin ? nin : 1));
    hipGraphNode_t* out = (hipGraphNode_t*)malloc(sizeof(hipGraphNode_t) * (nout ? nout : 1));
    if (!in || !out) { free(in); free(out); status = JDET_E_WORKSPACE; break; }
    if (nin) (void)hipGraphNodeGetDependencies(nodes[i], in, &nin);
    if (nout) (void)hipGraphNodeGetDependentNodes(nodes[i], out, &nout);
    unsigned char* dst = (unsigned char*)mp.dst;
    unsigned value = mp.value, esize = mp.elementSize;
    size_t width = mp.width, height = mp.height ? mp.height : 1, pitch = mp.pitch;
    if (esize != 1 && esize != 2 && esize != 4) { free(in); free(out); status = JDET_E_UNSUPPORTED; break; }
    void* args[] = {&dst, &value, &esize, &width, &height, &pitch};
    hipKernelNodeParams kp;
    memset(&kp, 0, sizeof(kp));
    const size_t total = width * height;
    size_t blocks = (total + 255) / 256;
    if (blocks > 16384) blocks = 16384;
    if (blocks < 1) blocks = 1;
    kp.func = (void*)graph_fill_kernel;
    kp.gridDim = dim3((unsigned)blocks);
    kp.blockDim = dim3(256);
    kp.sharedMemBytes = 0;
    kp.kernelParams = args;
    kp.extra = nullptr;
    hipGraphNode_t knode;
    e = hipGraphAddKernelNode(&knode, g, nin ? in : nullptr, nin, &kp);
    if (e == hipSuccess) {
      for (size_t d = 0; d < nout && e == hipSuccess; d++) e = hipGraphAddDependencies(g, &knode, &out[d], 1);
    }
    if (e == hipSuccess) e = hipGraphDestroyNode(nodes[i]);
    free(in);
    free(out);
    if (e != hipSuccess) { status = (int)e; break; }
    replaced++;
  }
  free(nodes);
  if (n_replaced) *n_replaced = replaced;
  return status;
}
