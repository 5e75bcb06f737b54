// Weight gradient of the 3x3 / stride 1 / pad 1 convolutions of the dense stack (and, with a gathered operand, of the
// deformable convolution of AlignConv) as one fp32 MFMA kernel that ACCUMULATES into the gradient buffer, channels-last.
//
// Reference: the weight gradient Jittor's autograd derives for nn.Conv inside ConvModule
// (python/jdet/models/utils/modules.py:L91-175; towers of models/roi_heads/s2anet_head.py:L127-205, necks/fpn.py:L150-201)
// and, deformable, DeformConvFunction.grad's im2col + matmul into grad_weight (ops/dcn_v1.py:L508-556).
//
//   gW[co, tap, ci] += sum_p gY[p, co] * A[p, (tap, ci)]                 p = (image, y, x)
//   plain conv : A[p, (tap, ci)] = X[p + tap, ci]   (0 outside the image)
//   deformable : A[p, (tap, ci)] = bilinear sample of X[., ci] at p + tap + offset[p, tap]      (dcn_v1.py:L132-166)
//
// GEMM view per tap: M = Cout, N = Cin, K = positions -- the reduction runs over the LONG dimension, so the work is cut
// along K as well (ksplit chunks of positions) and the partial tiles meet in gW by float atomics: that is the "beta = 1"
// the library's split-K kernels cannot offer (they zero-fill a scratch gradient per call, and the framework then adds
// it to p.grad: two extra launches and 3 passes over the weights per layer and pyramid level).  Here the five levels of
// a shared tower accumulate into ONE buffer, which may be p.grad itself.
//
// Tiling (one workgroup = 4 waves as 2 x 2; a wave owns TM x TN tiles of 32 x 32; K step = 16 positions):
//   * both operands are K-major in memory (a position's channels are contiguous), so the tiles go to LDS as they come:
//     [k][channel] rows, 16-byte chunks, one ds_write_b128 per loaded chunk.  v_mfma_f32_32x32x2_f32 wants ONE f32 of
//     each operand per lane (lane l: row / column l & 31, k = l >> 5): a fragment is one ds_read_b32, lanes 0-31 from
//     row 2q, lanes 32-63 from row 2q + 1.  Row stride = tile width + 32 floats: the two rows of a fragment sit on
//     opposite halves of the banks, and the 16 lanes of a write group cover all banks once.
//   * global loads are raw buffer loads with per-thread byte offsets; positions outside the image (the tap's halo),
//     past the last position and channels past the tensor get an out-of-range offset and read as zero.
//   * double-buffered LDS, register-staged (loads of step t + 1 in flight during the MFMAs of step t), one barrier per
//     K step; 32 MFMAs per wave between barriers at the 128 x 128 tile.
//   * workgroup -> (chunk, tile): consecutive workgroup ids go round-robin over the 8 XCDs; XCD x takes the chunks
//     x, x + 8, ... and runs ALL (Cout tile, Cin tile, tap) workgroups of a chunk back to back: the 9 taps x Cout tiles
//     that re-read the same positions of X (and the 9 taps x Cin tiles re-reading gY) find them in that XCD's L2.
// The matrix pipe is the bound: 2 * 9 * Cin * Cout * positions flop at 157 TFLOP/s.
#include <cstdlib>

#include "common.h"

namespace {

typedef float v4f __attribute__((ext_vector_type(4)));
typedef float v16f __attribute__((ext_vector_type(16)));

struct WgradArgs {
  const float* x;        // (N, H, W, Cin)
  const float* gy;       // (N, H, W, Cout)
  const float* offset;   // deformable: (N, 18, H, W), else null
  float* gw;             // (Cout, 3, 3, Cin), accumulated into
  int N, H, W, Cin, Cout, ksplit, steps_per_chunk, skip_epilogue;
  int R, stride, Ho, Wo;   // R x R taps (pad R / 2), stride: gy is (N, Ho, Wo, Cout); 3 / 1 / H / W for the 3x3 form
};

constexpr unsigned kOob = 0xFFFFFFF0u;
constexpr int BK = 16;

__device__ __forceinline__ v4f buf_load(__amdgpu_buffer_rsrc_t r, unsigned voff) {
  return __builtin_bit_cast(v4f, __builtin_amdgcn_raw_buffer_load_b128(r, voff, 0, 0));
}

// S2: the strided form (stride 2, or any case where the x pixel of a position is not at a constant distance from it)
template <int TM, int TN, bool DEFORM, bool WIDE, bool S2, int DEPTH = 1>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(DEFORM ? 3 : 4)))
void conv3x3_wgrad_kernel(WgradArgs a) {
  static_assert(DEPTH == 1 || ((DEPTH == 2 || DEPTH == 4) && !DEFORM), "several register sets: plain form only");
  constexpr int BM = 64 * TM, BN = 64 * TN;
  constexpr int SA = BM + 32, SB = BN + 32;             // LDS row strides in floats
  constexpr int TILE_A = BK * SA * 4, TILE_B = BK * SB * 4;
  constexpr int CA = BM / 4, CB = BN / 4;               // 16-byte chunks per row
  constexpr int PA = BK * CA / 256, PB = BK * CB / 256; // loader passes (rows per pass: 256 / chunks per row)
  constexpr int NC = DEFORM ? 4 : 1;
  static_assert(PA >= 1 && PB >= 1, "tile shape");
  __shared__ __attribute__((aligned(16))) char s_raw[2 * (TILE_A + TILE_B)];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const long M = (long)a.N * a.Ho * a.Wo;            // positions of gy
  const long Mx = (long)a.N * a.H * a.W;             // positions of x
  const int taps = a.R * a.R, pad = a.R >> 1;
  const int mt = (a.Cout + BM - 1) / BM, nt = (a.Cin + BN - 1) / BN;
  const int tiles = mt * nt * taps;
  // grid = tiles * (ksplit rounded up to whole rounds over the 8 XCDs); the surplus workgroups of the last round leave
  const int xcd = blockIdx.x & 7, j = blockIdx.x >> 3;
  const int jq = (int)((unsigned)j / (unsigned)tiles);
  const int chunk = xcd + 8 * jq, tile = j - jq * tiles;
  if (chunk >= a.ksplit) return;
  const int rest = (int)((unsigned)tile / (unsigned)taps), tap = tile - rest * taps;
  const int mq = (int)((unsigned)rest / (unsigned)nt);
  const int n0 = (rest - mq * nt) * BN, m0 = mq * BM;
  const int tr = (int)((unsigned)tap / (unsigned)a.R);
  const int dy = tr - pad, dx = tap - tr * a.R - pad;
  const long p_begin = (long)chunk * a.steps_per_chunk * BK;
  if (p_begin >= M) return;
  long left = (M - p_begin + BK - 1) / BK;
  const int nsteps = left < a.steps_per_chunk ? (int)left : a.steps_per_chunk;

  const __amdgpu_buffer_rsrc_t rx =
      __builtin_amdgcn_make_buffer_rsrc((void*)a.x, 0, (unsigned)(Mx * a.Cin * 4), 0x00020000);
  const __amdgpu_buffer_rsrc_t rg =
      __builtin_amdgcn_make_buffer_rsrc((void*)a.gy, 0, (unsigned)(M * a.Cout * 4), 0x00020000);

  // ---- loader role ----
  // A (gY): pass p covers row p * (256 / CA) + tid / CA, chunk tid % CA; B (X) likewise.  Per K step a row moves BK
  // positions on: the byte offsets advance by a constant, only the B row's (y, x) -- the halo test -- is tracked.
  // The offsets of the NEXT loads are always ready in registers (computed between the MFMAs of the step before), so a
  // step opens with its buffer loads back to back.
  const int a_chunk = tid % CA, a_row = tid / CA;
  const int b_chunk = tid % CB, b_row = tid / CB;
  const bool a_cok = m0 + a_chunk * 4 < a.Cout, b_cok = n0 + b_chunk * 4 < a.Cin;
  unsigned a_off[PA];
  int a_st[PA], b_st[PB];
#pragma unroll
  for (int p = 0; p < PA; p++) {
    const int row = p * (256 / CA) + a_row;
    a_off[p] = a_cok ? (unsigned)(((p_begin + row) * a.Cout + m0 + a_chunk * 4) * 4) : kOob;
    a_st[p] = (row * SA + a_chunk * 4) * 4;
  }
  const unsigned a_step = a_cok ? (unsigned)(BK * a.Cout * 4) : 0u;      // kOob stays kOob
  const unsigned b_step = (unsigned)(BK * a.Cin * 4);
  // WIDE: W >= BK, a step crosses at most one row end
  int bi[PB], by[PB], bx[PB], bp[PB];         // image / y / x / flat index of this thread's B rows
  unsigned b_lin[PB];                         // plain: byte offset of the tap's pixel, valid or not
  unsigned b_off[PB][NC];                     // what the next load uses: b_lin or kOob (deformable: the 4 corners)
  float wt[PB][NC];                           // deformable: bilinear weights of the loads in flight
  float o_h[PB], o_w[PB];                     // deformable: the row's offsets, fetched one step ahead
  auto fetch_offsets = [&](int p) {           // deformable: offsets of row p's current position
    if (bp[p] < M) {
      const size_t ob = ((size_t)bi[p] * 17 + 2 * tap) * a.H * a.W + bp[p];     // ((img*18 + 2 tap)*H + y)*W + x
      o_h[p] = a.offset[ob];
      o_w[p] = a.offset[ob + (size_t)a.H * a.W];
    }
  };
  auto plain_off = [&](int p) {
    if (S2) {
      const int yy = by[p] * a.stride + dy, xx = bx[p] * a.stride + dx;
      const bool in = b_cok && bp[p] < M && (unsigned)yy < (unsigned)a.H && (unsigned)xx < (unsigned)a.W;
      b_off[p][0] = in ? ((unsigned)(((bi[p] * a.H + yy) * a.W + xx) * a.Cin + n0 + b_chunk * 4)) * 4u : kOob;
    } else {
      const bool in = b_cok && bp[p] < M && (unsigned)(by[p] + dy) < (unsigned)a.H && (unsigned)(bx[p] + dx) < (unsigned)a.W;
      b_off[p][0] = in ? b_lin[p] : kOob;
    }
  };
  auto advance = [&](int p) {                 // row p: BK positions on
    bp[p] += BK;
    bx[p] += BK;
    b_lin[p] += b_step;
    if (WIDE) {
      const bool cx = bx[p] >= a.Wo;
      bx[p] -= cx ? a.Wo : 0;
      by[p] += cx ? 1 : 0;
      const bool cy = by[p] >= a.Ho;
      by[p] -= cy ? a.Ho : 0;
      bi[p] += cy ? 1 : 0;
    } else {
      while (bx[p] >= a.Wo) {
        bx[p] -= a.Wo;
        if (++by[p] == a.Ho) {
          by[p] = 0;
          bi[p]++;
        }
      }
    }
    if (!DEFORM) plain_off(p);
  };
#pragma unroll
  for (int p = 0; p < PB; p++) {
    const int row = p * (256 / CB) + b_row;
    const long pp = p_begin + row;
    bp[p] = (int)pp;
    {   // 32-bit unsigned: the host refuses (M + BK) * channels >= 2^30 (a 64-bit division is ~200 VALU instructions, and
        // these workgroups live for a few microseconds: profiles/r06_conv_prefetch.md, the stamps of conv_bn)
      const unsigned hw = (unsigned)(a.Ho * a.Wo), upp = (unsigned)pp;
      const unsigned im = upp / hw, rem = upp - im * hw, yy = rem / (unsigned)a.Wo;
      bi[p] = (int)im;
      by[p] = (int)yy;
      bx[p] = (int)(rem - yy * (unsigned)a.Wo);
    }
    b_lin[p] = (unsigned)(((pp + dy * a.W + dx) * a.Cin + n0 + b_chunk * 4) * 4);
    b_st[p] = TILE_A + (row * SB + b_chunk * 4) * 4;
    o_h[p] = o_w[p] = 0.f;
    if (DEFORM) fetch_offsets(p); else plain_off(p);
  }

  // Round 6: the plain form requests its rows SEVERAL K steps ahead into that many register sets (a 64 x 64 tile's K step is
  // 8 MFMAs per wave, 0.2 us -- far less than a global round trip under load; see conv_bn.hip; a set is 8 VGPRs here):
  // S2ANet step 26.55 (one step ahead) -> 26.48 (two); 26.64 (two) -> 26.52 (four; same-box pairs).  The deformable form keeps one (its
  // bilinear weights belong to the loads in flight).
  constexpr int D = DEPTH;
  v4f ra[D][PA], rb[D][PB][NC];
  auto load_step = [&](auto setc) {
    constexpr int S = decltype(setc)::value;
#pragma unroll
    for (int p = 0; p < PA; p++) {
      ra[S][p] = buf_load(rg, a_off[p]);          // past the last position: out of range -> 0
      a_off[p] += a_step;
    }
#pragma unroll
    for (int p = 0; p < PB; p++) {
      if constexpr (!DEFORM) {
        rb[S][p][0] = buf_load(rx, b_off[p][0]);
      } else {
        // dcn_v1.py:L132-166 (deformable_im2col), the same sampling rule as conv_igemm.hip's gathered operand
        const bool ok = b_cok && bp[p] < M;
        const float h_im = (float)(by[p] + dy) + o_h[p], w_im = (float)(bx[p] + dx) + o_w[p];
        const bool in = ok && h_im > -1.f && w_im > -1.f && h_im < (float)a.H && w_im < (float)a.W;
        const int hl = (int)floorf(h_im), wl = (int)floorf(w_im);
        const float lh = h_im - hl, lw = w_im - wl, hh = 1.f - lh, hw = 1.f - lw;
        const float w4[4] = {hh * hw, hh * lw, lh * hw, lh * lw};
        const int cy[4] = {hl, hl, hl + 1, hl + 1}, cx[4] = {wl, wl + 1, wl, wl + 1};
#pragma unroll
        for (int k = 0; k < NC; k++) {
          const bool kin = in && (unsigned)cy[k] < (unsigned)a.H && (unsigned)cx[k] < (unsigned)a.W;
          wt[p][k] = w4[k];
          rb[S][p][k] = buf_load(rx, kin ? ((unsigned)(((bi[p] * a.H + cy[k]) * a.W + cx[k]) * a.Cin + n0 + b_chunk * 4)) * 4u
                                         : kOob);
        }
        advance(p);
        fetch_offsets(p);
      }
    }
  };
  auto store_step = [&](auto setc, int buf) {
    constexpr int S = decltype(setc)::value;
    char* base = s_raw + buf * (TILE_A + TILE_B);
#pragma unroll
    for (int p = 0; p < PA; p++) *reinterpret_cast<v4f*>(base + a_st[p]) = ra[S][p];
#pragma unroll
    for (int p = 0; p < PB; p++) {
      if constexpr (!DEFORM)
        *reinterpret_cast<v4f*>(base + b_st[p]) = rb[S][p][0];
      else
        *reinterpret_cast<v4f*>(base + b_st[p]) =
            wt[p][0] * rb[S][p][0] + wt[p][1] * rb[S][p][1] + wt[p][2] * rb[S][p][2] + wt[p][3] * rb[S][p][3];
    }
  };

  // ---- compute role ----
  const int wm = wave >> 1, wn = wave & 1;
  const int frow = lane & 31, fhalf = lane >> 5;
  const int fa_off = (fhalf * SA + wm * 32 * TM + frow) * 4;
  const int fb_off = TILE_A + (fhalf * SB + wn * 32 * TN + frow) * 4;
  v16f acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; i++)
#pragma unroll
    for (int j = 0; j < TN; j++)
#pragma unroll
      for (int e = 0; e < 16; e++) acc[i][j][e] = 0.f;

  using Set0 = std::integral_constant<int, 0>;
  load_step(Set0{});
  if (!DEFORM) {
#pragma unroll
    for (int p = 0; p < PB; p++) advance(p);
  }
  if constexpr (D >= 2) {
    load_step(std::integral_constant<int, 1>{});
#pragma unroll
    for (int p = 0; p < PB; p++) advance(p);
  }
  if constexpr (D == 4) {
    load_step(std::integral_constant<int, 2>{});
#pragma unroll
    for (int p = 0; p < PB; p++) advance(p);
    load_step(std::integral_constant<int, 3>{});
#pragma unroll
    for (int p = 0; p < PB; p++) advance(p);
  }
  store_step(Set0{}, 0);
  __syncthreads();
  // Every step loads a later one's rows (the steps past the chunk's end too: their rows are read -- in range or as
  // zeros -- stored and never used, which keeps the loop free of branches).  Before the turn of step s (LDS buffer
  // B = s & 1 holds it): with two sets, set B ^ 1 holds step s + 1 (in flight) and set B is free for step s + 2.
  auto turn = [&](auto kc) {      // turn k of a trip: LDS buffer k & 1 holds its step, set k % D is free, set (k + 1) % D is next
    constexpr int K = decltype(kc)::value;
    constexpr int buf = K & 1;
    using LoadSet = std::integral_constant<int, K % D>;
    using StoreSet = std::integral_constant<int, (K + 1) % D>;
    load_step(LoadSet{});
    const char* sb = s_raw + buf * (TILE_A + TILE_B);
    float fa[2][TM], fb[2][TN];          // fragments of K pair q + 1 are fetched behind the MFMAs of pair q
    auto frags = [&](int q) {
#pragma unroll
      for (int i = 0; i < TM; i++) fa[q & 1][i] = *reinterpret_cast<const float*>(sb + fa_off + (2 * q * SA + i * 32) * 4);
#pragma unroll
      for (int j = 0; j < TN; j++) fb[q & 1][j] = *reinterpret_cast<const float*>(sb + fb_off + (2 * q * SB + j * 32) * 4);
    };
    frags(0);
#pragma unroll
    for (int q = 0; q < BK / 2; q++) {
      if (q + 1 < BK / 2) frags(q + 1);
      __builtin_amdgcn_sched_barrier(0);     // keep the fetch of pair q + 1 ahead of the MFMAs of pair q
#pragma unroll
      for (int i = 0; i < TM; i++)
#pragma unroll
        for (int j = 0; j < TN; j++)
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[q & 1][i], fb[q & 1][j], acc[i][j], 0, 0, 0);
      // behind the MFMAs just issued: the offsets of the step after next (one B row per K pair), then -- late, the
      // loads have had six K pairs to land -- the LDS writes of the next step
      if (!DEFORM && q < PB) advance(q);
      if (q == BK / 2 - 2) store_step(StoreSet{}, buf ^ 1);
      __builtin_amdgcn_sched_barrier(0);
    }
    __syncthreads();
  };
  {
    constexpr int LOOP = D == 4 ? 4 : 2;
    int step = 0;
    for (; step + LOOP - 1 < nsteps; step += LOOP) {
      turn(std::integral_constant<int, 0>{});
      turn(std::integral_constant<int, 1>{});
      if constexpr (LOOP == 4) {
        turn(std::integral_constant<int, 2>{});
        turn(std::integral_constant<int, 3>{});
      }
    }
    if (step < nsteps) turn(std::integral_constant<int, 0>{});
    if constexpr (LOOP == 4) {
      if (step + 1 < nsteps) turn(std::integral_constant<int, 1>{});
      if (step + 2 < nsteps) turn(std::integral_constant<int, 2>{});
    }
  }

  if (a.skip_epilogue) {                 // measurement aid: the cost of the atomics = the difference
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < TM; i++)
#pragma unroll
      for (int j2 = 0; j2 < TN; j2++)
#pragma unroll
        for (int e = 0; e < 16; e++) s += acc[i][j2][e];
    if (s == 12345.678f) a.gw[0] = s;
    return;
  }
  // ---- epilogue: C/D layout of the 32x32 MFMA: column = lane & 31, row = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5);
  // a register's 32 lanes add to 128 contiguous bytes of gW[co, tap, :] ----
  // A tile inside the weight tensor (round 6, conv_bn.hip's stamps: these workgroups live a few microseconds, and 16 x TM x TN
  // predicated atomics with 64-bit addresses were a large part of that): raw buffer atomics, one 32-bit lane offset per
  // 32 x 32 tile, the 16 row offsets in SGPRs.
  const long gw_bytes = (long)a.Cout * taps * a.Cin * 4;
  if (m0 + BM <= a.Cout && n0 + BN <= a.Cin && gw_bytes < (1L << 32)) {
    const __amdgpu_buffer_rsrc_t rgw = __builtin_amdgcn_make_buffer_rsrc((void*)a.gw, 0, (unsigned)gw_bytes, 0x00020000);
    const unsigned row_bytes = (unsigned)(taps * a.Cin * 4);
#pragma unroll
    for (int i = 0; i < TM; i++)
#pragma unroll
      for (int j = 0; j < TN; j++) {
        const unsigned ci = (unsigned)(n0 + wn * 32 * TN + j * 32 + (lane & 31));
        const unsigned co = (unsigned)(m0 + wm * 32 * TM + i * 32 + 4 * (lane >> 5));
        const unsigned base = ((co * (unsigned)taps + (unsigned)tap) * (unsigned)a.Cin + ci) * 4u;
#pragma unroll
        for (int e = 0; e < 16; e++)
          __builtin_amdgcn_raw_ptr_buffer_atomic_fadd_f32(acc[i][j][e], rgw, (int)base,
                                                          (int)(((e & 3) + 8 * (e >> 2)) * row_bytes), 0);
      }
    return;
  }
#pragma unroll
  for (int i = 0; i < TM; i++)
#pragma unroll
    for (int j = 0; j < TN; j++) {
      const int ci = n0 + wn * 32 * TN + j * 32 + (lane & 31);
#pragma unroll
      for (int e = 0; e < 16; e++) {
        const int co = m0 + wm * 32 * TM + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * (lane >> 5);
        if (co < a.Cout && ci < a.Cin) unsafeAtomicAdd(a.gw + ((size_t)co * taps + tap) * a.Cin + ci, acc[i][j][e]);
      }
    }
}

template <int TM, int TN>
int launch(const WgradArgs& a, hipStream_t st) {
  const int mt = (a.Cout + 64 * TM - 1) / (64 * TM), nt = (a.Cin + 64 * TN - 1) / (64 * TN);
  const unsigned grid = (unsigned)(mt * nt * a.R * a.R * ((a.ksplit + 7) & ~7));
  const bool wide = a.Wo >= BK;
  const bool s2 = a.stride != 1 || a.Ho != a.H || a.Wo != a.W;
  static const char* deep_env = getenv("JDET_CONV_WGRAD_DEEP");      // rows 4 (wide stride-1 form) / 2 K steps ahead; 0 / 2 / 4: A/B
  // the 64 x 64 tile only: the wider tiles' K step is long enough for one step of cover, and several register sets on top
  // of their accumulators spill (<2, 2, ..., 4>: 124 VGPRs to scratch -- the head towers' gradients through this kernel
  // ran 3.4 ms per step slower with it)
  static const char* mid_env = getenv("JDET_CONV_WGRAD_DEEP_MID");   // the 128 x 64 / 64 x 128 tiles: 0 (default) / 2
  const int deep_n = (TM == 1 && TN == 1) ? (deep_env ? atoi(deep_env) : 4)
                                          : ((TM + TN == 3 && mid_env) ? (atoi(mid_env) ? 2 : 0) : 0);
  const bool deep = deep_n != 0, deep4 = deep_n == 4;
  if (a.offset) {
    if (wide) hipLaunchKernelGGL((conv3x3_wgrad_kernel<TM, TN, true, true, false>), dim3(grid), dim3(256), 0, st, a);
    else hipLaunchKernelGGL((conv3x3_wgrad_kernel<TM, TN, true, false, false>), dim3(grid), dim3(256), 0, st, a);
  } else if (s2) {
    constexpr int D2 = (TM + TN <= 3) ? 2 : 1;
    if (wide && deep) hipLaunchKernelGGL((conv3x3_wgrad_kernel<TM, TN, false, true, true, D2>), dim3(grid), dim3(256), 0, st, a);
    else if (wide) hipLaunchKernelGGL((conv3x3_wgrad_kernel<TM, TN, false, true, true>), dim3(grid), dim3(256), 0, st, a);
    else if (deep) hipLaunchKernelGGL((conv3x3_wgrad_kernel<TM, TN, false, false, true, D2>), dim3(grid), dim3(256), 0, st, a);
    else hipLaunchKernelGGL((conv3x3_wgrad_kernel<TM, TN, false, false, true>), dim3(grid), dim3(256), 0, st, a);
  } else {
    constexpr int D2 = (TM + TN <= 3) ? 2 : 1, D4 = (TM == 1 && TN == 1) ? 4 : 1;
    if (wide && deep4) hipLaunchKernelGGL((conv3x3_wgrad_kernel<TM, TN, false, true, false, D4>), dim3(grid), dim3(256), 0, st, a);
    else if (wide && deep) hipLaunchKernelGGL((conv3x3_wgrad_kernel<TM, TN, false, true, false, D2>), dim3(grid), dim3(256), 0, st, a);
    else if (wide) hipLaunchKernelGGL((conv3x3_wgrad_kernel<TM, TN, false, true, false>), dim3(grid), dim3(256), 0, st, a);
    else if (deep) hipLaunchKernelGGL((conv3x3_wgrad_kernel<TM, TN, false, false, false, D2>), dim3(grid), dim3(256), 0, st, a);
    else hipLaunchKernelGGL((conv3x3_wgrad_kernel<TM, TN, false, false, false>), dim3(grid), dim3(256), 0, st, a);
  }
  return jdet_launch_status();
}

int wgrad_out_dim(int in, int R, int stride) { return (in + 2 * (R / 2) - R) / stride + 1; }

// general = the backbone's entry point (jdet_conv_wgrad): 64 x 64 tiles and a split aimed at ~2300 workgroups of at
// least 32 K steps -- measured at every ResNet-50 layer shape of a 2 x 1024^2 step (scripts/conv_bn_timing.py wgrad,
// profiles/r05_conv_bn.md): the smaller tile multiplies the workgroups a short tile list offers (a 128 -> 512 1x1 layer
// has FOUR 128^2 tiles) and beats both the 128^2 form (45 vs 59-77 us there) and the library (45 vs 48 us; 3x3
// 128 -> 128: 90 vs 104 us).
int run_wgrad(const float* x_nhwc, const float* gy_nhwc, const float* offset, int N, int H, int W, int Cin, int Cout,
              int R, int stride, float* gw, int ksplit, hipStream_t st, bool general = false) {
  const int Ho = wgrad_out_dim(H, R, stride), Wo = wgrad_out_dim(W, R, stride);
  const long M = (long)N * Ho * Wo, Mx = (long)N * H * W;
  if ((M + BK) * Cout >= (1L << 30) || (Mx + BK) * Cin >= (1L << 30)) return JDET_E_UNSUPPORTED;     // 32-bit byte offsets
  // bit 17: 64 x 64 tiles whatever the channel counts, bit 18: 128-wide tiles where the channels allow (measurement aids)
  const int small = ((ksplit >> 17) & 1) || (general && !((ksplit >> 18) & 1));
  int tm = (Cout > 64 && !small) ? 2 : 1, tn = (Cin > 64 && !small) ? 2 : 1;
  // Round 6: 128 x 64 instead of 128 x 128 for the plain form -- the head towers' shape (2 x 128^2, 256 -> 256): 310 us
  // against 360 (128 x 128: 128 VGPRs, three of them spilled), 332 (64 x 64) and the library's 326 incl. its zero fill
  // (scripts/wgrad_tiles_p3.sh, profiles/r06_conv_prefetch.md).  Bit 21 keeps the 128 x 128 tile (A/B).
  if (tm == 2 && tn == 2 && !offset && !((ksplit >> 21) & 1)) tn = 1;
  if ((ksplit >> 19) & 1) tn = 1;        // bit 19: 128 x 64 tiles, bit 20: 64 x 128 (measurement aids)
  if ((ksplit >> 20) & 1) tm = 1;
  const int mt = (Cout + 64 * tm - 1) / (64 * tm), nt = (Cin + 64 * tn - 1) / (64 * tn);
  const long tiles = (long)mt * nt * R * R;
  const long steps = (M + BK - 1) / BK;
  const int skip = (ksplit >> 16) & 1;   // bit 16: leave the result out (measurement aid)
  long ks = ksplit & 0xFFFF;
  if (ks == 0 && small) {
    // 32 K steps per chunk, in whole rounds over the 8 XCDs (chunk c runs on XCD c % 8: fewer than 8 chunks leave XCDs
    // idle -- 4 chunks of 32 steps ran 82 us where 8 chunks of 16 ran 48, 512 -> 2048 1x1 at 2 x 32^2) -- best or within
    // 2 % of the best forced split at all 17 layer shapes; halved while that makes more than ~9000 workgroups
    // (round 6, re-measured with the rows four steps ahead and the raw-buffer epilogue: still the best forced split at the
    //  layer1-3 shapes -- 64-step chunks ran 7-15 % slower at layer2; profiles/r06_conv_prefetch.md)
    ks = ((steps / 32 + 7) / 8) * 8;
    if (ks < 8) ks = 8;
    while (ks > 8 && tiles * ks > 9216) ks -= 8;
  } else if (ks == 0) {
    // measured on MI355X (scripts/conv_wgrad_timing.py, profiles/r04_conv_wgrad.md): ~2300 workgroups (the chip holds
    // 1024; the staggered later rounds run denser than one lock-step round), at least 16 K steps each -- small maps
    // trade that for parallelism down to 4 steps -- and never more than 64 chunks (each adds a tile of atomics).
    // 1x1 layers have a ninth of the tiles: up to 256 chunks of at least 8 steps there.
    ks = (2304 + tiles - 1) / tiles;
    long cap = steps / (R == 1 ? 8 : 16);
    const long small = steps / 4 < 8 ? steps / 4 : 8;
    if (cap < small) cap = small;
    const long most = R == 1 ? 256 : 64;
    if (cap > most) cap = most;
    if (ks > cap) ks = cap;
  }
  if (ks < 1) ks = 1;
  if (ks > steps) ks = steps;
  WgradArgs a{x_nhwc, gy_nhwc, offset, gw, N, H, W, Cin, Cout, (int)ks, (int)((steps + ks - 1) / ks), skip,
              R, stride, Ho, Wo};
  if (tm == 2) return tn == 2 ? launch<2, 2>(a, st) : launch<2, 1>(a, st);
  return tn == 2 ? launch<1, 2>(a, st) : launch<1, 1>(a, st);
}

}  // namespace

// Supported: Cin % 4 == 0, Cout % 4 == 0, 16-byte aligned x / gy, N*H*W*max(Cin, Cout) < 2^30.
JDET_API int jdet_conv3x3_wgrad_supported(int Cin, int Cout) {
  return Cin > 0 && Cout > 0 && Cin % 4 == 0 && Cout % 4 == 0;
}

// gw_krsc (Cout, 3, 3, Cin) += the weight gradient of y = conv3x3(x, w) [offset: the deformable form] for the output
// gradient gy.  ksplit: 0 = automatic, else the number of position chunks (measurement aid).
JDET_API int jdet_conv3x3_wgrad(const float* x_nhwc, const float* gy_nhwc, const float* offset, int N, int H, int W,
                                int Cin, int Cout, float* gw_krsc, int ksplit, jdet_stream_t stream) {
  if (N < 0 || H <= 0 || W <= 0 || Cin <= 0 || Cout <= 0 || ksplit < 0) return JDET_E_BADARG;
  if (!jdet_conv3x3_wgrad_supported(Cin, Cout)) return JDET_E_UNSUPPORTED;
  if (N == 0) return JDET_OK;
  if (!x_nhwc || !gy_nhwc || !gw_krsc) return JDET_E_BADARG;
  if ((((uintptr_t)x_nhwc) | ((uintptr_t)gy_nhwc)) & 15) return JDET_E_BADARG;
  return run_wgrad(x_nhwc, gy_nhwc, offset, N, H, W, Cin, Cout, 3, 1, gw_krsc, ksplit, (hipStream_t)stream);
}

// The general form of the backbone: R x R taps (R = 1 | 3, pad R / 2), stride 1 | 2: x (N, H, W, Cin),
// gy (N, Ho, Wo, Cout) with Ho = (H + 2 * (R / 2) - R) / stride + 1, gw_krsc (Cout, R, R, Cin) +=.
JDET_API int jdet_conv_wgrad(const float* x_nhwc, const float* gy_nhwc, int N, int H, int W, int Cin, int Cout, int R,
                             int stride, float* gw_krsc, int ksplit, jdet_stream_t stream) {
  if (N < 0 || H <= 0 || W <= 0 || Cin <= 0 || Cout <= 0 || ksplit < 0) return JDET_E_BADARG;
  if ((R != 1 && R != 3) || (stride != 1 && stride != 2)) return JDET_E_UNSUPPORTED;
  if (!jdet_conv3x3_wgrad_supported(Cin, Cout)) return JDET_E_UNSUPPORTED;
  if (N == 0) return JDET_OK;
  if (!x_nhwc || !gy_nhwc || !gw_krsc) return JDET_E_BADARG;
  if ((((uintptr_t)x_nhwc) | ((uintptr_t)gy_nhwc)) & 15) return JDET_E_BADARG;
  return run_wgrad(x_nhwc, gy_nhwc, nullptr, N, H, W, Cin, Cout, R, stride, gw_krsc, ksplit, (hipStream_t)stream, true);
}
