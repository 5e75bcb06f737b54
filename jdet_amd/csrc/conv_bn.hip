// The ResNet bottleneck convolutions as ONE fp32-MFMA implicit-GEMM family with the layer's neighbours in the epilogue,
// channels-last: 1x1 and 3x3, stride 1 and 2, forward AND data gradient.
//
// Reference: Bottleneck.execute (python/jdet/models/backbones/resnet.py:L61-93: conv1x1 -> bn -> relu, conv3x3(stride)
// -> bn -> relu, conv1x1 -> bn -> (+identity | downsample) -> relu), the layers of ResNet._make_layer (L131-154), with
// every BatchNorm in eval mode while training (`norm_eval`, L177-185) -- i.e. a per-channel affine map whose weight / bias
// still train -- and the gradients Jittor's autograd derives for that chain.
//
//   Y[m, n] = epilogue( sum_{tap, c} X[pixel(m) * stride + tap - pad, c] * Wt[n, tap, c] )      m = (image, oy, ox)
//
// The main loop is the tiling of conv_igemm.hip (v_mfma_f32_32x32x2_f32, 2 x 2 waves [x 2 K groups], 128^2 / 64^2 output
// tiles, XOR-swizzled 16-byte LDS chunks so that ONE ds_read_b128 per operand tile feeds four MFMAs, raw buffer loads
// with out-of-range = zero for the halo, double-buffered LDS, XCD-aware tile order) with the tap geometry a run-time
// (R, stride).  What is new is what happens to the accumulators -- a library convolution has no epilogue the caller
// controls, which is why every conv of the backbone used to be followed by an elementwise BatchNorm pass (forward) and
// preceded by one (backward):
//   mode FORWARD : y = [relu]( acc * a[n] + sh[n] [+ residual[m, n]] )        a = gamma * rsqrt(var + eps), sh = beta - mean * a
//   mode ADD     : gx = acc + grad_out[m, n] * [act_out[m, n] > 0]            the data gradient of conv1 of a block PLUS the
//                                                                             identity branch's gradient (the block's true grad_x)
//   mode MASK    : g = acc * [act[m, n] > 0];  partial column sums of g and g * (act - beta[n]);  y = g * a[n]
//                                                                             the data gradient w.r.t. the layer below's
//                                                                             activation act = relu(bn(conv)), turned straight
//                                                                             into the gradient w.r.t. that conv's output, with
//                                                                             the sums its BatchNorm weight / bias gradients need
//                                                                             (dbeta = sum g, dgamma = sum g * xhat,
//                                                                             xhat = (act - beta) / gamma wherever act > 0)
// The data gradient itself is this same kernel run on the flipped / transposed weights (jdet_conv_dgrad_weights: one
// launch per step for the whole backbone).  Column sums leave the workgroup as one partial row per (M tile, wave row)
// -- deterministic, no atomics -- and are finished by jdet_bn_sums_finish (frozen_bn.hip).
// Bound: the matrix pipe at the 3x3 layers (2 * M * N * K flop at 157 TFLOP/s); HBM at the 1x1 layers of the big maps
// (64 <-> 256 channels at 2 x 256^2: ~170-300 MB per layer, where the fused epilogue saves the separate pass's 2-3 tensor
// round trips).
#include <cstdlib>

#include "common.h"

namespace {

typedef float v4f __attribute__((ext_vector_type(4)));
typedef float v16f __attribute__((ext_vector_type(16)));

struct CbArgs {
  const float* x;        // (N, H, W, Cin)
  const float* w;        // (Cout, R, R, Cin)
  float* y;              // (N, Ho, Wo, Cout)
  float* partial;        // cross-workgroup K split: (ksplit, M, Cout) partial sums (no epilogue), else null
  jdet_conv_epilogue_t ep;
  int N, H, W, Cin, Cout, R, stride, Ho, Wo, ksplit;
};

constexpr unsigned kOob = 0xFFFFFFF0u;   // a byte offset past every buffer: the load returns zeros

__device__ __forceinline__ v4f buf_load(__amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff) {
  return __builtin_bit_cast(v4f, __builtin_amdgcn_raw_buffer_load_b128(r, voff, soff, 0));
}

template <int BK>
__device__ __forceinline__ int swz_bytes(int row, int chunk) {
  return (row * BK + ((chunk ^ (BK == 16 ? (row >> 2) & 3 : (row >> 1) & 7)) << 2)) * 4;
}

// a = gamma * rsqrt(var + eps), sh = beta - mean * a (frozen_bn.hip's affine4, the same operation order); without
// statistics (var == null) the map is a = gamma (1), sh = beta (0): a plain bias
__device__ __forceinline__ void bn_affine(const jdet_bn_params_t& p, int n, float& a, float& sh) {
  const float w = p.weight ? p.weight[n] : 1.f, b = p.bias ? p.bias[n] : 0.f;
  if (p.var) {
    const float is = 1.0f / sqrtf(p.var[n] + p.eps);
    a = w * is;
    sh = b - p.mean[n] * (w * is);
  } else {
    a = w;
    sh = b;
  }
}

// DEPTH (round 6; BT = 64, one wave group): the operand tiles of a K step are requested DEPTH steps ahead into DEPTH
// register sets instead of one step ahead into one -- a 64 x 64 tile's K step is 16 MFMAs per wave (0.43 us), less than a
// global round trip under load, so with one step of cover every step ended in a wait for its successor's tiles.
template <int BT, int BK, int KG, int DEPTH = 1, int ABL = 0>
__global__ __launch_bounds__(256 * KG)
__attribute__((amdgpu_waves_per_eu(BT == 128 ? (KG == 2 ? 4 : (BK == 32 ? 2 : 4)) : 4)))
void conv_bn_kernel(CbArgs a) {
  constexpr int NTHR = 256 * KG;
  constexpr int T = BT / 64;             // 32 x 32 tiles per wave and direction
  constexpr int CH = BK / 4;             // 16-byte chunks per LDS row
  constexpr int RPP = NTHR / CH;         // loader: RPP rows x CH chunks per pass
  constexpr int PASSES = BT / RPP;
  constexpr int TILE = BT * BK * 4;      // bytes of one operand tile
  constexpr int QN = BK / 8 / KG;        // 8-deep slices per wave and K step
  static_assert(PASSES >= 1 && QN >= 1, "tile shape");
  __shared__ __attribute__((aligned(16))) char s_raw[4 * TILE];     // [buffer][A | B]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  // ABL & 16: workgroup time stamps (scripts/r6_conv_stamps.py): start, K loop entered / left, end, placement -- written
  // over the first words of the tile's first output row (a profiling build: that row is garbage afterwards)
  long long stamp[4] = {0, 0, 0, 0}, stamp_x[3] = {0, 0, 0};     // _x: requests issued | first tile landed | epilogue operands read
  if constexpr (ABL & 16) stamp[0] = wall_clock64();
  const long M = (long)a.N * a.Ho * a.Wo;
  const long Min = (long)a.N * a.H * a.W;
  const int taps = a.R * a.R, pad = a.R >> 1;
  // ---- XCD-aware tile id (conv_igemm.hip) ----
  const int NT = (a.Cout + BT - 1) / BT;
  const int total = gridDim.x;
  int logical = blockIdx.x;
  if ((total & 7) == 0) logical = (blockIdx.x & 7) * (total >> 3) + (blockIdx.x >> 3);
  // (32-bit unsigned index arithmetic throughout the prologue: the host refuses positions * channels >= 2^30.  Round 6,
  //  workgroup time stamps -- scripts/r6_conv_stamps.py -- showed 7-8.6 us between a workgroup's start and its first operand
  //  request on the layers that fill the chip four workgroups per CU: the 64-bit divisions below, ~200 VALU instructions
  //  each, run by sixteen waves per CU at once)
  const int mtile = (int)((unsigned)logical / (unsigned)NT);
  const long m0 = (long)mtile * BT;
  const int n0 = (logical - mtile * NT) * BT;
  const __amdgpu_buffer_rsrc_t rx =
      __builtin_amdgcn_make_buffer_rsrc((void*)a.x, 0, (unsigned)(Min * a.Cin * 4), 0x00020000);
  const __amdgpu_buffer_rsrc_t rw =
      __builtin_amdgcn_make_buffer_rsrc((void*)a.w, 0, (unsigned)((long)a.Cout * taps * a.Cin * 4), 0x00020000);
  // ---- loader role: pass p covers row p * RPP + tid / CH, chunk tid % CH (4 channels of the BK of a K step) ----
  const int lchunk = tid % CH, lrow = tid / CH;
  int img[PASSES], py[PASSES], px[PASSES];   // image, and the input pixel of tap (0, 0) WITHOUT the padding shift
  bool m_ok[PASSES];
  unsigned wv[PASSES];
  int st_off[PASSES];
#pragma unroll
  for (int p = 0; p < PASSES; p++) {
    const int row = p * RPP + lrow;
    const long lm = m0 + row;
    m_ok[p] = lm < M;
    img[p] = py[p] = px[p] = 0;
    if (m_ok[p]) {
      const unsigned hw = (unsigned)(a.Ho * a.Wo), ulm = (unsigned)lm;
      const unsigned im = ulm / hw;
      const unsigned rem = ulm - im * hw;
      const unsigned oy = rem / (unsigned)a.Wo;
      img[p] = (int)im;
      py[p] = (int)oy * a.stride;
      px[p] = (int)(rem - oy * (unsigned)a.Wo) * a.stride;
    }
    wv[p] = n0 + row < a.Cout ? ((unsigned)((n0 + row) * taps * a.Cin + lchunk * 4)) * 4u : kOob;
    st_off[p] = swz_bytes<BK>(row, lchunk);
  }
  const int spt = a.Cin / BK;               // K steps per tap (host: BK = 32 only when Cin % 32 == 0)
  const int all_steps = taps * spt;
  const int step0 = (int)((unsigned)all_steps * blockIdx.y / (unsigned)a.ksplit);
  const int nsteps = (int)((unsigned)all_steps * (blockIdx.y + 1u) / (unsigned)a.ksplit) - step0;

  unsigned av[PASSES];
  auto set_tap = [&](int tap) {
    // branch-free (round 6: the predicated form was eight exec-mask branches per call, inside the K loop at every tap change;
    // R is 1 or 3: tap / R = (tap * 11) >> 5 for tap < 9)
    const int r = a.R == 1 ? tap : (tap * 11) >> 5, s = tap - r * a.R;
#pragma unroll
    for (int p = 0; p < PASSES; p++) {
      const int yy = py[p] + r - pad, xx = px[p] + s - pad;
      const bool in = m_ok[p] && (unsigned)yy < (unsigned)a.H && (unsigned)xx < (unsigned)a.W;
      const unsigned off = ((unsigned)(((img[p] * a.H + yy) * a.W + xx) * a.Cin + lchunk * 4)) * 4u;
      av[p] = in ? off : kOob;
    }
  };
  constexpr int SETS = DEPTH;
  v4f ra[SETS][PASSES], rb[SETS][PASSES];
  auto load_set = [&](auto setc, int tap, int c) {
    constexpr int S = decltype(setc)::value;
    const unsigned sa = (unsigned)(c * 4), sb = (unsigned)((tap * a.Cin + c) * 4);
#pragma unroll
    for (int p = 0; p < PASSES; p++) {
      ra[S][p] = buf_load(rx, av[p], sa);
      rb[S][p] = buf_load(rw, wv[p], sb);
    }
  };
  auto store_set = [&](auto setc, int buf) {
    constexpr int S = decltype(setc)::value;
#pragma unroll
    for (int p = 0; p < PASSES; p++) {
      *reinterpret_cast<v4f*>(s_raw + buf * 2 * TILE + st_off[p]) = ra[S][p];
      *reinterpret_cast<v4f*>(s_raw + buf * 2 * TILE + TILE + st_off[p]) = rb[S][p];
    }
  };
  using Set0 = std::integral_constant<int, 0>;
  auto load_step = [&](int tap, int c) { load_set(Set0{}, tap, c); };
  auto store_step = [&](int buf) { store_set(Set0{}, buf); };

  // ---- compute role ----
  const int kg = wave >> 2, wm = (wave >> 1) & 1, wn = wave & 1;
  const int frow = lane & 31, fhalf = lane >> 5;
  int fa_off[QN], fb_off[QN];
#pragma unroll
  for (int qq = 0; qq < QN; qq++) {
    const int chunk = (qq * KG + kg) * 2 + fhalf;
    fa_off[qq] = swz_bytes<BK>(wm * (BT / 2) + frow, chunk);
    fb_off[qq] = TILE + swz_bytes<BK>(wn * (BT / 2) + frow, chunk);
  }
  v16f acc[T][T];
#pragma unroll
  for (int i = 0; i < T; i++)
#pragma unroll
    for (int j = 0; j < T; j++)
#pragma unroll
      for (int e = 0; e < 16; e++) acc[i][j][e] = 0.f;
  constexpr bool DUAL = T == 1 && (ABL & 8) != 0;
  v16f acc_b;
#pragma unroll
  for (int e = 0; e < 16; e++) acc_b[e] = 0.f;

  // ---- the neighbour tile(s) of the epilogue (residual | grad_out + act | act): with one 32 x 32 tile per wave (BT = 64)
  // they are requested HERE, ahead of the K loop -- the 1x1 layers of the big maps run two to eight K steps, and a
  // tile fetched only after them costs as much again as the loop (64 -> 256 channels at 2 x 256^2 with a residual:
  // 133 us against 80 us without one, profiles/r05_conv_bn.md)
  const jdet_conv_epilogue_t& ep = a.ep;
  const int mode = ep.mode;
  const float* p0 = mode == JDET_EPI_FORWARD ? ep.residual : (mode == JDET_EPI_ADD ? ep.grad_out : ep.act);
  const float* p1 = mode == JDET_EPI_ADD ? ep.act : nullptr;
  constexpr bool PRE = T == 1;
  float pre0[PRE ? 16 : 1], pre1[PRE ? 16 : 1];
  // Round 6 (workgroup time stamps, scripts/r6_conv_stamps.py): ~900 instructions ran between a workgroup's start and its
  // first operand request -- 6-8 us with sixteen waves per CU issuing them at once -- most of them this block's 32 predicated
  // loads with 64-bit addresses.  A tile that lies inside the map (every tile but the last of a ragged M) now takes raw buffer
  // loads: one 32-bit lane offset, the 16 row offsets in SGPRs, one lane predicate (the column) around the lot.
  const bool full_tile = !a.partial && m0 + BT <= M;        // uniform
  const unsigned ybytes = (unsigned)(M * a.Cout * 4);         // (host: M * Cout < 2^30)
  if (PRE && full_tile && kg == 0) {
    const int n = n0 + wn * (BT / 2) + (lane & 31);
    const unsigned base = ((unsigned)(m0 + wm * (BT / 2) + 4 * (lane >> 5)) * (unsigned)a.Cout + (unsigned)n) * 4u;
#pragma unroll
    for (int e = 0; e < 16; e++) pre0[e] = pre1[e] = 0.f;
    if (n < a.Cout) {
      if (p0) {
        const __amdgpu_buffer_rsrc_t r0 = __builtin_amdgcn_make_buffer_rsrc((void*)p0, 0, ybytes, 0x00020000);
#pragma unroll
        for (int e = 0; e < 16; e++)
          pre0[e] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(
                                                  r0, base, (unsigned)(((e & 3) + 8 * (e >> 2)) * a.Cout * 4), 0));
      }
      if (p1) {
        const __amdgpu_buffer_rsrc_t r1 = __builtin_amdgcn_make_buffer_rsrc((void*)p1, 0, ybytes, 0x00020000);
#pragma unroll
        for (int e = 0; e < 16; e++)
          pre1[e] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(
                                                  r1, base, (unsigned)(((e & 3) + 8 * (e >> 2)) * a.Cout * 4), 0));
      }
    }
  } else if (PRE && !a.partial && kg == 0) {
    const long mrow = m0 + wm * (BT / 2) + 4 * (lane >> 5);
    const int n = n0 + wn * (BT / 2) + (lane & 31);
#pragma unroll
    for (int e = 0; e < 16; e++) {
      const long m = mrow + (e & 3) + 8 * (e >> 2);
      const bool ok = n < a.Cout && m < M;
      pre0[e] = (p0 && ok) ? p0[(size_t)m * a.Cout + n] : 0.f;
      pre1[e] = (p1 && ok) ? p1[(size_t)m * a.Cout + n] : 0.f;
    }
  }

  int tap = step0 / spt, c = (step0 - tap * spt) * BK;
  set_tap(tap);
  auto advance = [&]() {          // the load cursor: next K step (next 32 / 16 channels, then the next tap)
    c += BK;
    if (c == a.Cin) {
      c = 0;
      tap++;
      set_tap(tap);
    }
  };
  auto mfma_step = [&](int buf) {
    const char* sb = s_raw + buf * 2 * TILE;
#pragma unroll
    for (int qq = 0; qq < QN; qq++) {
      v4f fa[T], fb[T];
#pragma unroll
      for (int i = 0; i < T; i++) {
        fa[i] = *reinterpret_cast<const v4f*>(sb + fa_off[qq] + i * 32 * BK * 4);
        fb[i] = *reinterpret_cast<const v4f*>(sb + fb_off[qq] + i * 32 * BK * 4);
      }
      if constexpr (DUAL) {
        // one 32 x 32 tile per wave = ONE chain of dependent MFMAs: the even / odd 2-deep slices go to two accumulators
#pragma unroll
        for (int kk = 0; kk < 4; kk += 2) {
          acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[0][kk], fb[0][kk], acc[0][0], 0, 0, 0);
          acc_b = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[0][kk + 1], fb[0][kk + 1], acc_b, 0, 0, 0);
        }
      } else {
#pragma unroll
        for (int kk = 0; kk < 4; kk++)
#pragma unroll
          for (int i = 0; i < T; i++)
#pragma unroll
            for (int j = 0; j < T; j++)
              acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[i][kk], fb[j][kk], acc[i][j], 0, 0, 0);
      }
    }
  };
  auto fake_step = [&](int buf) {       // ablation: the LDS reads of a step without its MFMAs
    const char* sb = s_raw + buf * 2 * TILE;
#pragma unroll
    for (int qq = 0; qq < QN; qq++) {
      const v4f fa = *reinterpret_cast<const v4f*>(sb + fa_off[qq]);
      const v4f fb = *reinterpret_cast<const v4f*>(sb + fb_off[qq]);
      acc[0][0][qq & 15] += fa[0] + fb[1] + fa[2] + fb[3];
    }
  };
  if constexpr (DEPTH == 1) {
    load_step(tap, c);
    store_step(0);
    __syncthreads();
    for (int step = 0; step < nsteps; step++) {
      const int buf = step & 1;
      const bool more = step + 1 < nsteps;
      if (more) {
        advance();
        load_step(tap, c);
      }
      mfma_step(buf);
      if (more) store_step(buf ^ 1);
      __syncthreads();
    }
  } else {
    // DEPTH == 2.  At the top of turn s: LDS buffer (s & 1) holds step s; register set (s + 1) & 1 holds step s + 1 (in
    // flight); set (s & 1) is free and takes step s + 2.  The steady-state turns request unconditionally, so that hipcc's
    // wait before the LDS stores of step s + 1 leaves the just-issued loads of step s + 2 in flight (vmcnt(7) / (5) / (4)
    // in the ISA; a conditional request makes it wait for everything).  110 VGPRs, no scratch (a generic DEPTH-turn
    // formulation with a switch over the last turns spilled: 128 VGPRs + 76 B of scratch at depth 2, 208 B at depth 3).
    static_assert(DEPTH == 2, "two register sets");
    using Set1 = std::integral_constant<int, 1>;
    if constexpr (ABL & 16) stamp_x[0] = wall_clock64();          // index arithmetic done
    load_set(Set0{}, tap, c);
    if (nsteps > 1) {
      advance();
      load_set(Set1{}, tap, c);
    }
    if constexpr (ABL & 16) {
      asm volatile("s_waitcnt vmcnt(4)" ::: "memory");            // (set 0 = the four oldest of eight loads)
      asm volatile("" : "+v"(ra[0][0]), "+v"(rb[0][0]));
      stamp_x[1] = wall_clock64();                                 // first tile in registers
    }
    store_set(Set0{}, 0);
    __syncthreads();
    if constexpr (ABL & 16) stamp[1] = wall_clock64();
    int step = 0;
    // ABL = ablation builds of the steady-state turns (JDET_CONV_BN_ABL, timing only, wrong results): bit 0 = no operand
    // requests and no LDS stores (the loop runs on whatever LDS holds), bit 1 = no MFMAs (the LDS reads feed one add each),
    // bit 2 = no barriers.  Round 6, measured and removed from the product instantiations: a scheduling fence behind the
    // requests (hipcc sinks the four buffer loads below twelve of the step's sixteen MFMAs; with the fence it serialises
    // the LDS reads instead: + 2 % per layer) and s_setprio 1 around the MFMAs (+ 2 %): profiles/r06_conv_prefetch.md.
    for (; step + 3 < nsteps; step += 2) {
      advance();
      if constexpr (!(ABL & 1)) load_set(Set0{}, tap, c);        // step + 2
      if constexpr (ABL & 2) fake_step(0); else mfma_step(0);
      if constexpr (!(ABL & 1)) store_set(Set1{}, 1);            // step + 1
      if constexpr (!(ABL & 4)) __syncthreads();
      advance();
      if constexpr (!(ABL & 1)) load_set(Set1{}, tap, c);        // step + 3
      if constexpr (ABL & 2) fake_step(1); else mfma_step(1);
      if constexpr (!(ABL & 1)) store_set(Set0{}, 0);            // step + 2
      if constexpr (!(ABL & 4)) __syncthreads();
    }
    while (step < nsteps) {            // the last one to three steps
      if (step + 2 < nsteps) {
        advance();
        load_set(Set0{}, tap, c);
      }
      mfma_step(0);
      if (step + 1 < nsteps) store_set(Set1{}, 1);
      __syncthreads();
      if (++step >= nsteps) break;
      if (step + 2 < nsteps) {
        advance();
        load_set(Set1{}, tap, c);
      }
      mfma_step(1);
      if (step + 1 < nsteps) store_set(Set0{}, 0);
      __syncthreads();
      ++step;
    }
  }

  if constexpr (DUAL) acc[0][0] += acc_b;
  if constexpr (ABL & 16) {
    asm volatile("" : "+v"(acc[0][0]));          // (the stamp stays behind the last MFMA's result)
    stamp[2] = wall_clock64();
  }
  if (KG == 2) {
    float* red = reinterpret_cast<float*>(s_raw);
    static_assert(KG == 1 || 4 * T * T * 16 * 64 * 4 <= 4 * TILE, "reduction buffer");
    if (kg == 1) {
#pragma unroll
      for (int i = 0; i < T; i++)
#pragma unroll
        for (int j = 0; j < T; j++)
#pragma unroll
          for (int e = 0; e < 16; e++) red[(((wave & 3) * T * T + i * T + j) * 16 + e) * 64 + lane] = acc[i][j][e];
    }
    __syncthreads();
    if (kg == 1) return;
  }

  // ---- epilogue.  C/D layout of the 32x32 MFMA: column = lane & 31, row = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5).
  // The neighbour tiles (residual | grad_out + act_out | act) are fetched 8 rows at a time BEFORE the stores of those
  // rows: the loads are in flight together instead of one per dependent store.
  float cs1[T], cs2[T];
#pragma unroll
  for (int j = 0; j < T; j++) cs1[j] = cs2[j] = 0.f;
  bool stored = false;
  if constexpr (T == 1) {
    if (a.partial && m0 + BT <= M) {          // K split over workgroups: the plain sums to this part's plane, same store form
      stored = true;
      const int n = n0 + wn * (BT / 2) + (lane & 31);
      if (n < a.Cout) {
        const __amdgpu_buffer_rsrc_t rp = __builtin_amdgcn_make_buffer_rsrc(
            (void*)(a.partial + (size_t)blockIdx.y * M * a.Cout), 0, ybytes, 0x00020000);
        const unsigned base = ((unsigned)(m0 + wm * (BT / 2) + 4 * (lane >> 5)) * (unsigned)a.Cout + (unsigned)n) * 4u;
        const float* red = reinterpret_cast<const float*>(s_raw) + (size_t)(wave & 3) * 16 * 64 + lane;
#pragma unroll
        for (int e = 0; e < 16; e++) {
          float v = acc[0][0][e];
          if (KG == 2) v += red[e * 64];
          __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), rp, base,
                                                (unsigned)(((e & 3) + 8 * (e >> 2)) * a.Cout * 4), 0);
        }
      }
    }
    // a tile inside the map (round 6): the 16 rows leave by raw buffer stores -- one 32-bit lane offset, the row offsets in
    // SGPRs, the mode decided once -- instead of 16 predicated stores with 64-bit addresses.  Same operations on the same
    // values in the same order as the general path below.
    if (full_tile) {
      stored = true;
      const int n = n0 + wn * (BT / 2) + (lane & 31);
      const bool nok = n < a.Cout;
      float sa = 1.f, sh = 0.f, beta = 0.f;
      if (nok && (mode != JDET_EPI_ADD)) {
        // (measured: read and folded ahead of the K loop instead, riding in pre1 -- 1.3-2.2 us leave the epilogue, 0.6-1.8 us
        //  join the prologue: no gain; requested behind the first barrier and folded here -- epilogue 6.5 -> 5.6 us, nothing
        //  on the layer sums or the step, and the kernel at 128 VGPRs: not kept either)
        bn_affine(ep.bn, n, sa, sh);
        beta = ep.bn.bias ? ep.bn.bias[n] : 0.f;
      }
      if constexpr (ABL & 16) {
        asm volatile("" : "+v"(sa), "+v"(sh));
        stamp_x[2] = wall_clock64();
      }
      if (nok) {
        const __amdgpu_buffer_rsrc_t ry = __builtin_amdgcn_make_buffer_rsrc((void*)a.y, 0, ybytes, 0x00020000);
        const unsigned base = ((unsigned)(m0 + wm * (BT / 2) + 4 * (lane >> 5)) * (unsigned)a.Cout + (unsigned)n) * 4u;
        const float* red = reinterpret_cast<const float*>(s_raw) + (size_t)(wave & 3) * 16 * 64 + lane;
        auto rows = [&](auto modec) {
          constexpr int MODE = decltype(modec)::value;
          float c1 = 0.f, c2 = 0.f;
#pragma unroll
          for (int e = 0; e < 16; e++) {
            float v = acc[0][0][e];
            if (KG == 2) v += red[e * 64];
            if constexpr (MODE == JDET_EPI_FORWARD) {
              if (ep.affine) v = v * sa + sh;
              v += pre0[e];                          // residual (0 without one)
              if (ep.relu) v = fmaxf(v, 0.f);
            } else if constexpr (MODE == JDET_EPI_ADD) {
              v += pre1[e] > 0.f ? pre0[e] : 0.f;
            } else {
              v = pre0[e] > 0.f ? v : 0.f;
              c1 += v;
              c2 += v * (pre0[e] - beta);
              v *= sa;
            }
            __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), ry, base,
                                                  (unsigned)(((e & 3) + 8 * (e >> 2)) * a.Cout * 4), 0);
          }
          cs1[0] = c1;
          cs2[0] = c2;
        };
        if (mode == JDET_EPI_FORWARD) rows(std::integral_constant<int, JDET_EPI_FORWARD>{});
        else if (mode == JDET_EPI_ADD) rows(std::integral_constant<int, JDET_EPI_ADD>{});
        else rows(std::integral_constant<int, JDET_EPI_MASK>{});
      }
    }
  }
  if (!stored)
#pragma unroll
  for (int i = 0; i < T; i++) {
    const long mrow = m0 + wm * (BT / 2) + i * 32 + 4 * (lane >> 5);
#pragma unroll
    for (int j = 0; j < T; j++) {
      const int n = n0 + wn * (BT / 2) + j * 32 + (lane & 31);
      const bool nok = n < a.Cout;
      float sa = 1.f, sh = 0.f, beta = 0.f;
      if (nok && !a.partial && (mode != JDET_EPI_ADD)) {
        bn_affine(ep.bn, n, sa, sh);
        beta = ep.bn.bias ? ep.bn.bias[n] : 0.f;
      }
      if constexpr (ABL & 16) {
        asm volatile("" : "+v"(sa), "+v"(sh));
        stamp_x[2] = wall_clock64();                               // the column's BatchNorm parameters read and folded
      }
#pragma unroll
      for (int h = 0; h < 2; h++) {          // 8 rows at a time: their neighbour loads are in flight together
        float t0[8], t1[8];
        if (PRE) {
#pragma unroll
          for (int u = 0; u < 8; u++) {
            t0[u] = pre0[h * 8 + u];
            t1[u] = pre1[h * 8 + u];
          }
        } else if (!a.partial) {
#pragma unroll
          for (int u = 0; u < 8; u++) {
            const int e = h * 8 + u;
            const long m = mrow + (e & 3) + 8 * (e >> 2);
            const bool ok = nok && m < M;
            t0[u] = (p0 && ok) ? p0[(size_t)m * a.Cout + n] : 0.f;
            t1[u] = (p1 && ok) ? p1[(size_t)m * a.Cout + n] : 0.f;
          }
        }
#pragma unroll
        for (int u = 0; u < 8; u++) {
          const int e = h * 8 + u;
          const long m = mrow + (e & 3) + 8 * (e >> 2);
          if (m < M && nok) {
            float v = acc[i][j][e];
            if (KG == 2)
              v += reinterpret_cast<const float*>(s_raw)[(((wave & 3) * T * T + i * T + j) * 16 + e) * 64 + lane];
            if (a.partial) {
              a.partial[((size_t)blockIdx.y * M + m) * a.Cout + n] = v;
              continue;
            }
            if (mode == JDET_EPI_FORWARD) {
              if (ep.affine) v = v * sa + sh;
              v += t0[u];                          // residual (0 without one)
              if (ep.relu) v = fmaxf(v, 0.f);
            } else if (mode == JDET_EPI_ADD) {
              v += t1[u] > 0.f ? t0[u] : 0.f;
            } else {
              v = t0[u] > 0.f ? v : 0.f;
              cs1[j] += v;
              cs2[j] += v * (t0[u] - beta);
              v *= sa;
            }
            a.y[(size_t)m * a.Cout + n] = v;
          }
        }
      }
    }
  }
  if (mode == JDET_EPI_MASK && ep.sums && !a.partial) {
    // the two half waves hold the same columns (rows 4 apart): combine, then one partial row per (M tile, wave row)
#pragma unroll
    for (int j = 0; j < T; j++) {
      const float s1 = cs1[j] + __shfl_xor(cs1[j], 32);
      const float s2 = cs2[j] + __shfl_xor(cs2[j], 32);
      const int n = n0 + wn * (BT / 2) + j * 32 + (lane & 31);
      if (lane < 32 && n < a.Cout) {
        float* row = ep.sums + (size_t)(mtile * 2 + wm) * 2 * a.Cout;
        row[n] = s1;
        row[a.Cout + n] = s2;
      }
    }
  }
  if constexpr (ABL & 16) {
    __syncthreads();
    if (threadIdx.x == 0 && !a.partial && m0 < M && n0 + 16 <= a.Cout) {
      stamp[3] = wall_clock64();
      int* d = reinterpret_cast<int*>(a.y + (size_t)m0 * a.Cout + n0);
#pragma unroll
      for (int k = 0; k < 4; k++) {
        d[2 * k] = (int)(stamp[k] & 0xffffffff);
        d[2 * k + 1] = (int)(stamp[k] >> 32);
      }
      d[8] = __builtin_amdgcn_s_getreg((31 << 11) | 4);       // HW_ID
      d[9] = __builtin_amdgcn_s_getreg((31 << 11) | 20);      // XCC_ID
      d[10] = (int)blockIdx.x;
      d[11] = 0x5741;
      d[12] = (int)(stamp_x[0] - stamp[0]);
      d[13] = (int)(stamp_x[1] - stamp[0]);
      d[14] = (int)(stamp_x[2] - stamp[2]);
      d[15] = 0;
    }
  }
}

// Second stage of the cross-workgroup K split: sum of the partial planes in a fixed order + the same epilogue.
// Workgroup = 64 rows x 64 columns: thread (ty, tx) owns the column quad tx of rows ty, ty + 16, ty + 32, ty + 48; the
// column sums of mode MASK meet in LDS (fixed order) and leave as one partial row per 64-row block.
__global__ __launch_bounds__(256) void conv_bn_finish_kernel(CbArgs a) {
  __shared__ float s_sum[16][64][2];
  const long M = (long)a.N * a.Ho * a.Wo;
  const jdet_conv_epilogue_t& ep = a.ep;
  const int mode = ep.mode;
  const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;
  const int n = blockIdx.y * 64 + tx * 4;
  const bool nok = n < a.Cout;          // Cout % 4 == 0: the whole quad is in or out
  v4f sa = {1.f, 1.f, 1.f, 1.f}, sh = {0.f, 0.f, 0.f, 0.f}, beta = {0.f, 0.f, 0.f, 0.f};
  if (nok && mode != JDET_EPI_ADD) {
#pragma unroll
    for (int k = 0; k < 4; k++) {
      float x, y;
      bn_affine(ep.bn, n + k, x, y);
      sa[k] = x;
      sh[k] = y;
      beta[k] = ep.bn.bias ? ep.bn.bias[n + k] : 0.f;
    }
  }
  v4f c1 = {0.f, 0.f, 0.f, 0.f}, c2 = {0.f, 0.f, 0.f, 0.f};
  const size_t plane = (size_t)M * a.Cout;
#pragma unroll
  for (int r = 0; r < 4; r++) {
    const long m = (long)blockIdx.x * 64 + ty + 16 * r;
    if (m >= M || !nok) continue;
    const size_t idx = (size_t)m * a.Cout + n;
    v4f v = *reinterpret_cast<const v4f*>(a.partial + idx);
    for (int k = 1; k < a.ksplit; k++) v += *reinterpret_cast<const v4f*>(a.partial + (size_t)k * plane + idx);
    if (mode == JDET_EPI_FORWARD) {
      if (ep.affine) v = v * sa + sh;
      if (ep.residual) v += *reinterpret_cast<const v4f*>(ep.residual + idx);
      if (ep.relu)
#pragma unroll
        for (int k = 0; k < 4; k++) v[k] = fmaxf(v[k], 0.f);
    } else if (mode == JDET_EPI_ADD) {
      const v4f g = *reinterpret_cast<const v4f*>(ep.grad_out + idx), y = *reinterpret_cast<const v4f*>(ep.act + idx);
#pragma unroll
      for (int k = 0; k < 4; k++) v[k] += y[k] > 0.f ? g[k] : 0.f;
    } else {
      const v4f y = *reinterpret_cast<const v4f*>(ep.act + idx);
#pragma unroll
      for (int k = 0; k < 4; k++) v[k] = y[k] > 0.f ? v[k] : 0.f;
      c1 += v;
      c2 += v * (y - beta);
      v *= sa;
    }
    *reinterpret_cast<v4f*>(a.y + idx) = v;
  }
  if (mode == JDET_EPI_MASK && ep.sums) {
#pragma unroll
    for (int k = 0; k < 4; k++) {
      s_sum[ty][tx * 4 + k][0] = c1[k];
      s_sum[ty][tx * 4 + k][1] = c2[k];
    }
    __syncthreads();
    if (threadIdx.x < 128) {
      const int col = threadIdx.x & 63, which = threadIdx.x >> 6;
      float t = 0.f;
#pragma unroll
      for (int r = 0; r < 16; r++) t += s_sum[r][col][which];
      const int nn = blockIdx.y * 64 + col;
      if (nn < a.Cout) ep.sums[((size_t)blockIdx.x * 2 + which) * a.Cout + nn] = t;
    }
  }
}

int cb_env_int(const char* name, int dflt) {
  const char* v = getenv(name);
  return v && *v ? atoi(v) : dflt;
}

template <int BT, int BK, int KG>
int launch(const CbArgs& a, hipStream_t st) {
  const long M = (long)a.N * a.Ho * a.Wo;
  const long tiles = ((M + BT - 1) / BT) * ((a.Cout + BT - 1) / BT);
  if constexpr (BT == 64) {
    static const int deep = cb_env_int("JDET_CONV_BN_DEEP", 1);     // operand tiles TWO K steps ahead (0: one; A/B switch)
    if (deep) {
      if constexpr (KG == 1) {
        static const int abl = cb_env_int("JDET_CONV_BN_ABL", 0);     // ablation builds (timing only): see the kernel
        switch (BK == 32 ? abl : 0) {
          case 1: hipLaunchKernelGGL((conv_bn_kernel<BT, BK, KG, 2, 1>), dim3((unsigned)tiles, a.ksplit), dim3(256), 0, st, a); break;
          case 2: hipLaunchKernelGGL((conv_bn_kernel<BT, BK, KG, 2, 2>), dim3((unsigned)tiles, a.ksplit), dim3(256), 0, st, a); break;
          case 3: hipLaunchKernelGGL((conv_bn_kernel<BT, BK, KG, 2, 3>), dim3((unsigned)tiles, a.ksplit), dim3(256), 0, st, a); break;
          case 4: hipLaunchKernelGGL((conv_bn_kernel<BT, BK, KG, 2, 4>), dim3((unsigned)tiles, a.ksplit), dim3(256), 0, st, a); break;
          case 5: hipLaunchKernelGGL((conv_bn_kernel<BT, BK, KG, 2, 5>), dim3((unsigned)tiles, a.ksplit), dim3(256), 0, st, a); break;
          case 6: hipLaunchKernelGGL((conv_bn_kernel<BT, BK, KG, 2, 6>), dim3((unsigned)tiles, a.ksplit), dim3(256), 0, st, a); break;
          case 16: hipLaunchKernelGGL((conv_bn_kernel<BT, BK, KG, 2, 16>), dim3((unsigned)tiles, a.ksplit), dim3(256), 0, st, a); break;
          case 8: hipLaunchKernelGGL((conv_bn_kernel<BT, BK, KG, 2, 8>), dim3((unsigned)tiles, a.ksplit), dim3(256), 0, st, a); break;
          case 13: hipLaunchKernelGGL((conv_bn_kernel<BT, BK, KG, 2, 13>), dim3((unsigned)tiles, a.ksplit), dim3(256), 0, st, a); break;
          default: break;
        }
        if (BK == 32 && ((abl >= 1 && abl <= 6) || abl == 8 || abl == 13 || abl == 16)) return jdet_launch_status();
      }
      // JDET_CONV_BN_DYN_LDS (measurement switch): unused dynamic LDS per workgroup = fewer workgroups per CU (40960: two
      // instead of four) -- a one-round launch then runs in two rounds whose prologues / epilogues overlap the other
      // resident workgroup's K loop
      static const int dyn = cb_env_int("JDET_CONV_BN_DYN_LDS", 0);
      static const int dyn_min = cb_env_int("JDET_CONV_BN_DYN_MIN_TILES", 0), dyn_max = cb_env_int("JDET_CONV_BN_DYN_MAX_TILES", 1 << 30);
      const unsigned lds = (dyn > 0 && tiles * a.ksplit >= dyn_min && tiles * a.ksplit <= dyn_max) ? (unsigned)dyn : 0u;
      hipLaunchKernelGGL((conv_bn_kernel<BT, BK, KG, 2>), dim3((unsigned)tiles, a.ksplit), dim3(256 * KG), lds, st, a);
      return jdet_launch_status();
    }
  }
  hipLaunchKernelGGL((conv_bn_kernel<BT, BK, KG>), dim3((unsigned)tiles, a.ksplit), dim3(256 * KG), 0, st, a);
  return jdet_launch_status();
}

// ---- the launch plan: one function decides tile / K split for the launch, the workspace query and the sums geometry ----
struct Plan {
  int bt, bk, kg, ksplit;
  long sums_rows;
};

int out_dim(int in, int R, int stride) { return (in + 2 * (R / 2) - R) / stride + 1; }

Plan make_plan(long M, int Cin, int Cout, int taps, int tile, bool workspace) {
  Plan p;
  const int edge = tile & ~3;
  const long tiles128 = ((M + 127) / 128) * ((Cout + 127) / 128);
  const long tiles64 = ((M + 63) / 64) * ((Cout + 63) / 64);
  // 128^2 tiles only for the long reductions (3x3 from 256 input channels up) on maps whose 128-tiles alone fill the
  // chip: measured at the ResNet-50 shapes of a 2 x 1024^2 step (scripts/conv_bn_timing.py tiles, profiles/
  // r05_conv_bn.md), the 64^2 tile without the intra-workgroup K split wins or ties everywhere else (64 -> 256 at
  // 2 x 256^2: 80 vs 103 us; 128 -> 512 at 2 x 128^2: 59 vs 70; 256 -> 256 3x3 at 2 x 64^2: 92 vs 171)
  const bool big = tile ? edge == 128 : (tiles128 >= 512 && Cout > 64 && (long)taps * Cin >= 2304);
  // 16-deep K steps for the 1x1 layers of at most 128 input channels: four to eight short steps instead of two to four
  // (and half the LDS per workgroup): 64 -> 256 + residual at 2 x 256^2 89 vs 97 us, 128 -> 512 + residual at 2 x 128^2
  // 58 vs 65 us; equal elsewhere (scripts/conv_bn_timing.py tiles, profiles/r05_conv_bn.md)
  static const int k16_rule = cb_env_int("JDET_CONV_BN_K16", 0);   // measurement aid: 1 = 16-deep steps wherever no K split over
                                                                     // workgroups follows, 2 = everywhere
  const bool k16_env = tile == 0 && (k16_rule == 2 || (k16_rule == 1 && !(workspace && tiles64 < 384)));
  const bool k32 = Cin % 32 == 0 && !(tile & 1) && !(tile == 0 && taps == 1 && Cin <= 128) && !k16_env;
  const int steps = taps * (Cin / (k32 ? 32 : 16));
  // intra-workgroup K split (8 waves): conv_igemm.hip's rule; not for a K loop of one or two steps (the hand-over
  // through LDS then costs as much as the loop)
  const bool split = k32 && !(tile & 2) && (big || tile || tiles64 < 512) && steps >= 4;
  p.bt = big ? 128 : 64;
  p.bk = k32 ? 32 : 16;
  p.kg = split ? 2 : 1;
  p.ksplit = 1;
  // K steps split over workgroups below 384 tiles, aiming at ~768 workgroups: widening either (below 768 tiles / 1024-2048
  // workgroups) measured equal or 5-15 % slower at the layer3 / layer4 shapes (profiles/r05_conv_bn.md)
  if (tile == 0 && !big && workspace && Cout % 4 == 0 && tiles64 < 384) {
    int k = (int)(768 / tiles64);                 // aim at ~3 workgroups per CU
    if (k > 8) k = 8;
    if (k > steps / 4) k = steps / 4;             // at least 4 K steps per part
    if (k >= 2) {
      p.ksplit = k;
      p.kg = 1;
    }
  }
  p.sums_rows = p.ksplit > 1 ? (M + 63) / 64 : 2 * ((M + p.bt - 1) / p.bt);
  return p;
}

// flipped / transposed weights for the data gradient: dst[ci][R*R-1-tap][co] = src[co][tap][ci]
struct WtJob {
  const float* src;
  float* dst;
  int Cout, Cin, taps, tile_begin;
};

__global__ __launch_bounds__(256) void dgrad_weights_kernel(const WtJob* __restrict__ jobs, int njobs) {
  __shared__ float s[32][33];
  int j = 0;
  while (j + 1 < njobs && (int)blockIdx.x >= jobs[j + 1].tile_begin) j++;
  const WtJob job = jobs[j];
  int t = blockIdx.x - job.tile_begin;
  const int ct = (job.Cin + 31) / 32, ot = (job.Cout + 31) / 32;
  const int tap = t % job.taps;
  t /= job.taps;
  const int ci0 = (t % ct) * 32, co0 = (t / ct) * 32;
  if (t / ct >= ot) return;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
#pragma unroll
  for (int r = 0; r < 4; r++) {
    const int co = co0 + ty + 8 * r, ci = ci0 + tx;
    s[ty + 8 * r][tx] = (co < job.Cout && ci < job.Cin) ? job.src[((size_t)co * job.taps + tap) * job.Cin + ci] : 0.f;
  }
  __syncthreads();
#pragma unroll
  for (int r = 0; r < 4; r++) {
    const int ci = ci0 + ty + 8 * r, co = co0 + tx;
    if (ci < job.Cin && co < job.Cout)
      job.dst[((size_t)ci * job.taps + (job.taps - 1 - tap)) * job.Cout + co] = s[tx][ty + 8 * r];
  }
}

}  // namespace

// Supported: R in {1, 3} (pad R / 2), stride in {1, 2}, Cin % 16 == 0, 16-byte aligned tensors,
// positions * max(Cin, Cout) < 2^30 on both sides.
JDET_API int jdet_conv_bn_supported(int Cin, int Cout, int R, int stride) {
  return Cin > 0 && Cin % 16 == 0 && Cout > 0 && (R == 1 || R == 3) && (stride == 1 || stride == 2);
}

JDET_API size_t jdet_conv_bn_workspace(int N, int H, int W, int Cin, int Cout, int R, int stride) {
  if (N <= 0 || H <= 0 || W <= 0 || !jdet_conv_bn_supported(Cin, Cout, R, stride)) return 0;
  const long M = (long)N * out_dim(H, R, stride) * out_dim(W, R, stride);
  const Plan p = make_plan(M, Cin, Cout, R * R, 0, true);
  return p.ksplit > 1 ? sizeof(float) * (size_t)p.ksplit * M * Cout : 0;
}

JDET_API size_t jdet_conv_bn_sums_rows(int N, int H, int W, int Cin, int Cout, int R, int stride, int tile,
                                     int with_workspace) {
  if (N <= 0 || H <= 0 || W <= 0 || !jdet_conv_bn_supported(Cin, Cout, R, stride)) return 0;
  const long M = (long)N * out_dim(H, R, stride) * out_dim(W, R, stride);
  return (size_t)make_plan(M, Cin, Cout, R * R, tile, with_workspace != 0).sums_rows;
}

JDET_API int jdet_conv_bn_forward(const float* x_nhwc, int N, int H, int W, int Cin, const float* w_krsc, int Cout,
                                  int R, int stride, const jdet_conv_epilogue_t* epilogue, int tile, float* y_nhwc,
                                  void* workspace, size_t workspace_bytes, jdet_stream_t stream) {
  if (N < 0 || H <= 0 || W <= 0 || Cin <= 0 || Cout <= 0 || !epilogue) return JDET_E_BADARG;
  if (!jdet_conv_bn_supported(Cin, Cout, R, stride)) return JDET_E_UNSUPPORTED;
  if (N == 0) return JDET_OK;
  if (!x_nhwc || !w_krsc || !y_nhwc) return JDET_E_BADARG;
  if ((((uintptr_t)x_nhwc) | ((uintptr_t)w_krsc)) & 15) return JDET_E_BADARG;
  const jdet_conv_epilogue_t& ep = *epilogue;
  if (ep.mode != JDET_EPI_FORWARD && ep.mode != JDET_EPI_ADD && ep.mode != JDET_EPI_MASK) return JDET_E_BADARG;
  if (ep.mode == JDET_EPI_ADD && (!ep.grad_out || !ep.act)) return JDET_E_BADARG;
  if (ep.mode == JDET_EPI_MASK && !ep.act) return JDET_E_BADARG;
  if (ep.bn.var && !ep.bn.mean) return JDET_E_BADARG;
  const int Ho = out_dim(H, R, stride), Wo = out_dim(W, R, stride);
  const long M = (long)N * Ho * Wo, Min = (long)N * H * W;
  if (Min * Cin >= (1L << 30) || M * Cout >= (1L << 30) || (long)Cout * R * R * Cin >= (1L << 30))
    return JDET_E_UNSUPPORTED;     // 32-bit byte offsets
  const int edge = tile & ~3;
  if (tile != 0 && edge != 64 && edge != 128) return JDET_E_BADARG;
  const size_t need_ws = jdet_conv_bn_workspace(N, H, W, Cin, Cout, R, stride);
  bool ws_ok = workspace && need_ws && workspace_bytes >= need_ws;
  // the finish kernel moves float4s
  if (ws_ok && ((((uintptr_t)y_nhwc) | ((uintptr_t)workspace) | ((uintptr_t)ep.residual) | ((uintptr_t)ep.grad_out) |
                 ((uintptr_t)ep.act)) & 15)) {
    // dropping the K split here changes the number of partial-sum rows (jdet_conv_bn_sums_rows was asked WITH a
    // workspace): a caller that sized `sums` for the split plan would be written past its end -- refuse instead
    if (ep.sums) return JDET_E_BADARG;
    ws_ok = false;
  }
  const Plan p = make_plan(M, Cin, Cout, R * R, tile, ws_ok);
  CbArgs a{x_nhwc, w_krsc, y_nhwc, nullptr, ep, N, H, W, Cin, Cout, R, stride, Ho, Wo, p.ksplit};
  hipStream_t st = (hipStream_t)stream;
  if (p.ksplit > 1) {
    a.partial = (float*)workspace;
    int e = p.bk == 32 ? launch<64, 32, 1>(a, st) : launch<64, 16, 1>(a, st);
    if (e) return e;
    hipLaunchKernelGGL(conv_bn_finish_kernel, dim3((unsigned)((M + 63) / 64), (unsigned)((Cout + 63) / 64)), dim3(256), 0,
                       st, a);
    return jdet_launch_status();
  }
  if (p.bt == 128)
    return p.bk == 32 ? (p.kg == 2 ? launch<128, 32, 2>(a, st) : launch<128, 32, 1>(a, st)) : launch<128, 16, 1>(a, st);
  return p.bk == 32 ? (p.kg == 2 ? launch<64, 32, 2>(a, st) : launch<64, 32, 1>(a, st)) : launch<64, 16, 1>(a, st);
}

// jobs: DEVICE array of njobs records {src (Cout, R, R, Cin), dst (Cin, R, R, Cout), Cout, Cin, taps, tile_begin} with
// tile_begin the running sum of ceil(Cout / 32) * ceil(Cin / 32) * taps; total_tiles = that sum over all jobs.
JDET_API int jdet_conv_dgrad_weights(const void* jobs_device, int njobs, int total_tiles, jdet_stream_t stream) {
  if (njobs < 0 || total_tiles < 0) return JDET_E_BADARG;
  if (njobs == 0 || total_tiles == 0) return JDET_OK;
  if (!jobs_device) return JDET_E_BADARG;
  static_assert(sizeof(WtJob) == 32, "record layout of the ABI");
  hipLaunchKernelGGL(dgrad_weights_kernel, dim3((unsigned)total_tiles), dim3(256), 0, (hipStream_t)stream,
                     (const WtJob*)jobs_device, njobs);
  return jdet_launch_status();
}
