// Rotated / horizontal RoIAlign forward with the taps of TWO neighbouring bins merged (round 6).  A MEASURED ALTERNATIVE, not a
// product path: forward mode 5 of jdet_roi_align_forward_cl_mode (libjdet_experimental.so); included by roi_align_impl.inc
// inside its anonymous namespace under JDET_ROI_EXPERIMENTAL_MODES.
// Measured at the north-star point (profiles/r06_roi_fwd_ring.md section 5): rows 979 452 -> 764 512 (-22 %), vector-L1
// accesses 17.24 M -> 13.80 M, L1 -> L2 requests 5.86 M -> 4.88 M -- and VALU instructions 10.3 M -> 15.05 M: 59.4 us against
// 56.4 us for the rolling-window product kernel.  The launch is co-limited by VALU issue and the L2 -> L1 row path; the rows
// this merge saves cost more in the second accumulator and the wider merge than they return.
//
// Why (profiles/r06_roi_fwd_ring.md): the forward is bound by the rate at which 1 KiB rows come out of the L2 -> L1 path; its
// prologue is hidden (the tap loops alone take what the whole kernel takes).  Only fewer rows move it.  The per-bin merge
// requests 0.979 M rows at the north-star point; the distinct pixels of PAIRS of bins along the denser sample direction
// are 0.70 M (+ padding: 0.78 M).
//
// Item = two neighbouring bins of a line (the last bin of an odd line alone): 8 lanes = 2 bins x 4 samples; a wave holds 8
// items, a workgroup 32 (7 x 7: 28).  Stage 1 is the per-bin merge of the product kernel, unchanged (same exchanges, same
// order: the bins' merged weights are the product kernel's, bit for bit).  Stage 2: a kept tap of bin B whose pixel is a
// kept tap of bin A is dropped and its weight rides on A's entry -- an entry is (row offset, weight for A, weight for B).
// The rolling window folds a row into TWO accumulators; both rows leave when the item's last group has been folded.
// Result = the product kernel's up to the order of the fmaf chain of a bin (<= a few ulp of sum |w v|).
#pragma once

template <int VARIANT, int NO>
__global__ __launch_bounds__(256) void roi_align_fwd_pair_kernel(
    const float* __restrict__ feat, const float* __restrict__ rois, float* __restrict__ out,
    int C, int H, int W, int PH, int PW, float spatial_scale, const int32_t* __restrict__ order) {
  constexpr int NW = 4;
  extern __shared__ __attribute__((aligned(16))) int s_pair[];   // per wave 8 items x 32 entries x 4 words
  __shared__ float s_trig[2];
  const int r = order ? order[blockIdx.x] : blockIdx.x;
  const int c0 = blockIdx.y * kChunkC;
  const int cc = min(kChunkC, C - c0);
  const int nbins = PH * PW;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  constexpr int ROI_COLS = (VARIANT == JDET_ROI_HBB_V0 || VARIANT == JDET_ROI_HBB_V1) ? 5 : 6;
  constexpr bool kRot = ROI_COLS == 6;
  const float* roi = rois + (size_t)r * ROI_COLS;
  if (kRot && wave == 0 && lane == 0) {
    s_trig[0] = (float)cos((double)roi[5]);
    s_trig[1] = (float)sin((double)roi[5]);
  }
  __amdgpu_buffer_rsrc_t rsrc;
  RoiGeom g = vec_prologue<VARIANT, false>(feat, rois, r, C, H, W, PH, PW, spatial_scale, 2, rsrc);
  if (g.batch < 0) return;
  if (kRot) {
    __syncthreads();
    g.cosT = s_trig[0];
    g.sinT = s_trig[1];
  }
  const bool lane_ok = lane * 4 < cc;
  const int voff = (c0 + (lane_ok ? lane * 4 : 0)) * 4;
  const int pix_bytes = C * 4;

  // items: pairs along the direction in which the samples sit closer
  const bool along_w = __builtin_amdgcn_readfirstlane((int)(g.bin_w <= g.bin_h)) != 0;
  const int PD = along_w ? PW : PH, PL = along_w ? PH : PW;
  const int ipl = (PD + 1) >> 1, nitems = ipl * PL;
  const int ki = lane >> 3, l8 = lane & 7, hb = l8 >> 2, q = l8 & 3, qbase = lane & ~3, ibase = lane & ~7;
  const int it = wave + NW * ki;
  const bool item_ok = it < nitems;
  const int line = item_ok ? it / ipl : 0, pos = item_ok ? it - line * ipl : 0;
  const int dd = 2 * pos + hb;
  const bool bin_ok = item_ok && dd < PD;
  const int ph = bin_ok ? (along_w ? line : dd) : 0, pw = bin_ok ? (along_w ? dd : line) : 0;
  const int my_bin = ph * PW + pw;
  Sample s = make_sample<VARIANT>(g, ph, pw, q >> 1, q & 1, H, W);
  if (!bin_ok) s.valid = 0;
  const int o[4] = {s.o1 * pix_bytes, s.o2 * pix_bytes, s.o3 * pix_bytes, s.o4 * pix_bytes};
  const float w[4] = {s.w1, s.w2, s.w3, s.w4};
  // ---- stage 1: the per-bin merge of roi_align_fwd_merged_kernel (same exchanges in the same order)
  float tw[4] = {w[0], w[1], w[2], w[3]};
  bool first[4] = {true, true, true, true};
#pragma unroll
  for (int k = 1; k < 4; k++)
#pragma unroll
    for (int j = 0; j < k; j++)
      if (o[j] == o[k]) {
        tw[j] += w[k];
        first[k] = false;
      }
#pragma unroll
  for (int d = 1; d < 4; d++) {
    const int src = qbase | ((q + d) & 3);
    const bool earlier = ((q + d) & 3) < q;
    const int ov = __shfl(s.valid, src, 64);
#pragma unroll
    for (int j = 0; j < 4; j++) {
      const int oo = __shfl(o[j], src, 64);
      const float ww = __shfl(w[j], src, 64);
#pragma unroll
      for (int k = 0; k < 4; k++) {
        const bool same = ov && oo == o[k];
        tw[k] += same ? ww : 0.f;
        first[k] = first[k] && !(same && earlier);
      }
    }
  }
  int ri_ind = 0;
  float ri_l = 0.f, ri_r = 1.f;
  if (NO) {
    ri_params(roi[5], NO, ri_ind, ri_l, ri_r);
    ri_ind = __builtin_amdgcn_readfirstlane(ri_ind);
  }
  const float inv_count = 1.f / g.count;
  int keep[4];
  float twn[4];
#pragma unroll
  for (int k = 0; k < 4; k++) {
    keep[k] = s.valid && first[k];
    twn[k] = tw[k] * inv_count;
  }
  // ---- stage 2: against the kept taps of the item's other bin
  float wo[4] = {0.f, 0.f, 0.f, 0.f};
  bool matched[4] = {false, false, false, false};
#pragma unroll
  for (int d = 0; d < 4; d++) {
    const int src = ibase | ((hb ^ 1) << 2) | ((q + d) & 3);
#pragma unroll
    for (int j = 0; j < 4; j++) {
      const int oo = __shfl(o[j], src, 64);
      const int kk = __shfl(keep[j], src, 64);
      const float ww = __shfl(twn[j], src, 64);
#pragma unroll
      for (int k = 0; k < 4; k++) {
        const bool same = kk && keep[k] && oo == o[k];   // (kept taps of one bin are distinct: at most one match per tap)
        wo[k] = same ? ww : wo[k];
        matched[k] = matched[k] || same;
      }
    }
  }
  // entries: A's kept taps carry B's weight where B has the pixel; B's kept taps stay only where A has not
  int emit[4];
#pragma unroll
  for (int k = 0; k < 4; k++) emit[k] = hb == 0 ? keep[k] : (keep[k] && !matched[k]);
  const unsigned long long imask = 0xFFull << ibase;
  const unsigned long long lt = (1ull << lane) - 1ull;
  int posn = 0, n_item = 0;
#pragma unroll
  for (int k = 0; k < 4; k++) {
    const unsigned long long mk = __ballot(emit[k]) & imask;
    posn += __builtin_popcountll(mk & lt);
    n_item += __builtin_popcountll(mk);
  }
  int4* list = reinterpret_cast<int4*>(s_pair) + wave * 256 + ki * 32;
#pragma unroll
  for (int k = 0; k < 4; k++)
    if (emit[k]) {
      const float wa = hb == 0 ? twn[k] : 0.f, wb = hb == 0 ? wo[k] : twn[k];
      list[posn++] = make_int4(o[k], __float_as_int(wa), __float_as_int(wb), 0);
    }
  __builtin_amdgcn_wave_barrier();
  // ---- the wave's sequence of groups of 4 rows: lane l8 of an item forms the item's group l8
  const int ngq = item_ok ? (n_item + 3) >> 2 : 0;          // <= 8
  int below_g = 0, gtot = 0;
#pragma unroll
  for (int b = 1; b <= 8; b++) {
    const unsigned long long mk = __ballot(l8 == 0 && ngq >= b);
    below_g += __builtin_amdgcn_mbcnt_hi((unsigned)(mk >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)mk, 0));
    gtot += __builtin_popcountll(mk);
  }
  const int start = below_g - (l8 > 0 ? ngq : 0);
  const int bin_a = __shfl(my_bin, ibase, 64);
  const int bin_b_ok = __shfl((int)bin_ok, ibase | 4, 64);
  const int bin_b = bin_b_ok ? __shfl(my_bin, ibase | 4, 64) : 127;
  // an item without a single valid sample has no group: its rows of zeros are written here
  for (unsigned long long empty = __ballot(item_ok && l8 == 0 && n_item == 0); empty; empty &= empty - 1) {
    const int l0 = __builtin_ctzll(empty);
    const int ba = jdet_readlane_i(bin_a, l0), bb = jdet_readlane_i(bin_b, l0);
    if (lane_ok) {
      __builtin_nontemporal_store(v4f{0.f, 0.f, 0.f, 0.f},
                                  reinterpret_cast<v4f*>(out + ((size_t)r * nbins + ba) * C + c0 + lane * 4));
      if (bb != 127)
        __builtin_nontemporal_store(v4f{0.f, 0.f, 0.f, 0.f},
                                    reinterpret_cast<v4f*>(out + ((size_t)r * nbins + bb) * C + c0 + lane * 4));
    }
  }
  int go[4], gmeta;
  float gwa[4], gwb[4];
  {
    const int4 e0 = list[0];
    const int dst = (l8 < ngq ? start + l8 : 63) << 2;
    const int my_meta = (l8 == ngq - 1 ? 1 : 0) | (bin_a << 1) | (bin_b << 8);
#pragma unroll
    for (int i = 0; i < 4; i++) {
      const int4 e = list[4 * l8 + i];
      const bool live = 4 * l8 + i < n_item;
      go[i] = __builtin_amdgcn_ds_permute(dst, live ? e.x : e0.x);
      gwa[i] = __int_as_float(__builtin_amdgcn_ds_permute(dst, live ? e.y : 0));
      gwb[i] = __int_as_float(__builtin_amdgcn_ds_permute(dst, live ? e.z : 0));
    }
    gmeta = __builtin_amdgcn_ds_permute(dst, my_meta);
  }
  typedef int v4i_ __attribute__((ext_vector_type(4)));
  v4i_ rs;
  {
    const unsigned long long img_bits = (unsigned long long)(feat + (size_t)g.batch * H * W * C);
    rs.x = __builtin_amdgcn_readfirstlane((int)(unsigned)img_bits);
    rs.y = __builtin_amdgcn_readfirstlane((int)((unsigned)(img_bits >> 32) & 0xffffu));
    rs.z = __builtin_amdgcn_readfirstlane((int)((size_t)H * W * C * 4));
    rs.w = 0x00020000;
  }
  constexpr int RING = 4;
  auto ring = [&](auto ind_c) {
    constexpr int IND = decltype(ind_c)::value;
    v4f t[RING][4];
    v4f acc_a = {0.f, 0.f, 0.f, 0.f}, acc_b = {0.f, 0.f, 0.f, 0.f};
    auto issue = [&](auto sc, int gj) {
      constexpr int S = decltype(sc)::value;
      const int o0 = jdet_readlane_i(go[0], gj), o1 = jdet_readlane_i(go[1], gj);
      const int o2 = jdet_readlane_i(go[2], gj), o3 = jdet_readlane_i(go[3], gj);
      v4f &a0 = t[S][0], &a1 = t[S][1], &a2 = t[S][2], &a3 = t[S][3];
      const int vo = voff;
      const v4i_ rd = rs;
      asm volatile(
          "s_nop 4\n\t"
          "buffer_load_dwordx4 %0, %4, %5, %6 offen\n\t"
          "buffer_load_dwordx4 %1, %4, %5, %7 offen\n\t"
          "buffer_load_dwordx4 %2, %4, %5, %8 offen\n\t"
          "buffer_load_dwordx4 %3, %4, %5, %9 offen"
          : "=&v"(a0), "=&v"(a1), "=&v"(a2), "=&v"(a3)
          : "v"(vo), "s"(rd), "s"(o0), "s"(o1), "s"(o2), "s"(o3)
          : "memory");
    };
    auto landed = [&](auto sc, int younger) {
      constexpr int S = decltype(sc)::value;
      v4f &a0 = t[S][0], &a1 = t[S][1], &a2 = t[S][2], &a3 = t[S][3];
      if (younger >= 3)
        asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
      else if (younger == 2)
        asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
      else if (younger == 1)
        asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
      else
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      asm volatile("" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : : "memory");
    };
    auto row_out = [&](const v4f& acc, int bin) {
      v4f o4 = acc;
      if constexpr (NO != 0) {
        const float val[4] = {acc.x, acc.y, acc.z, acc.w};
        float mixed[4];
        ri_mix_static<NO, IND>(mixed, val, ri_r, ri_l);
        o4 = v4f{mixed[0], mixed[1], mixed[2], mixed[3]};
      }
      if (lane_ok)
        __builtin_nontemporal_store(o4, reinterpret_cast<v4f*>(out + ((size_t)r * nbins + bin) * C + c0 + lane * 4));
    };
    auto consume = [&](auto sc, int gi, int younger) {
      constexpr int S = decltype(sc)::value;
      landed(sc, younger);
#pragma unroll
      for (int i = 0; i < 4; i++) {
        const float wa = jdet_readlane_f(gwa[i], gi), wb = jdet_readlane_f(gwb[i], gi);
        acc_a.x = __builtin_fmaf(wa, t[S][i].x, acc_a.x);
        acc_a.y = __builtin_fmaf(wa, t[S][i].y, acc_a.y);
        acc_a.z = __builtin_fmaf(wa, t[S][i].z, acc_a.z);
        acc_a.w = __builtin_fmaf(wa, t[S][i].w, acc_a.w);
        acc_b.x = __builtin_fmaf(wb, t[S][i].x, acc_b.x);
        acc_b.y = __builtin_fmaf(wb, t[S][i].y, acc_b.y);
        acc_b.z = __builtin_fmaf(wb, t[S][i].z, acc_b.z);
        acc_b.w = __builtin_fmaf(wb, t[S][i].w, acc_b.w);
      }
      const int meta = jdet_readlane_i(gmeta, gi);
      if (meta & 1) {
        row_out(acc_a, (meta >> 1) & 127);
        if (((meta >> 8) & 127) != 127) row_out(acc_b, (meta >> 8) & 127);
        acc_a = v4f{0.f, 0.f, 0.f, 0.f};
        acc_b = v4f{0.f, 0.f, 0.f, 0.f};
      }
    };
    using S0 = std::integral_constant<int, 0>;
    using S1 = std::integral_constant<int, 1>;
    using S2 = std::integral_constant<int, 2>;
    using S3 = std::integral_constant<int, 3>;
    if (0 < gtot) issue(S0{}, 0);
    if (1 < gtot) issue(S1{}, 1);
    if (2 < gtot) issue(S2{}, 2);
    if (3 < gtot) issue(S3{}, 3);
    int gq = 0;
    for (; gq + 2 * RING <= gtot; gq += RING) {
      consume(S0{}, gq + 0, 3); issue(S0{}, gq + 0 + RING);
      consume(S1{}, gq + 1, 3); issue(S1{}, gq + 1 + RING);
      consume(S2{}, gq + 2, 3); issue(S2{}, gq + 2 + RING);
      consume(S3{}, gq + 3, 3); issue(S3{}, gq + 3 + RING);
    }
    auto younger_of = [&](int gi) { return min(gtot, gi + RING) - 1 - gi; };
    if (gq + 0 < gtot) { consume(S0{}, gq + 0, younger_of(gq + 0)); if (gq + 0 + RING < gtot) issue(S0{}, gq + 0 + RING); }
    if (gq + 1 < gtot) { consume(S1{}, gq + 1, younger_of(gq + 1)); if (gq + 1 + RING < gtot) issue(S1{}, gq + 1 + RING); }
    if (gq + 2 < gtot) { consume(S2{}, gq + 2, younger_of(gq + 2)); if (gq + 2 + RING < gtot) issue(S2{}, gq + 2 + RING); }
    if (gq + 3 < gtot) { consume(S3{}, gq + 3, younger_of(gq + 3)); if (gq + 3 + RING < gtot) issue(S3{}, gq + 3 + RING); }
    gq += RING;
    if (gq + 0 < gtot) consume(S0{}, gq + 0, younger_of(gq + 0));
    if (gq + 1 < gtot) consume(S1{}, gq + 1, younger_of(gq + 1));
    if (gq + 2 < gtot) consume(S2{}, gq + 2, younger_of(gq + 2));
    if (gq + 3 < gtot) consume(S3{}, gq + 3, younger_of(gq + 3));
  };
  if constexpr (NO == 0) {
    ring(std::integral_constant<int, 0>{});
  } else {
    switch (ri_ind) {   // wave-uniform
      case 0: ring(std::integral_constant<int, 0>{}); break;
      case 1: ring(std::integral_constant<int, 1>{}); break;
      case 2: ring(std::integral_constant<int, 2>{}); break;
      case 3: ring(std::integral_constant<int, 3>{}); break;
      case 4: ring(std::integral_constant<int, 4 % (NO ? NO : 1)>{}); break;
      case 5: ring(std::integral_constant<int, 5 % (NO ? NO : 1)>{}); break;
      case 6: ring(std::integral_constant<int, 6 % (NO ? NO : 1)>{}); break;
      default: ring(std::integral_constant<int, 7 % (NO ? NO : 1)>{}); break;
    }
  }
}

// both pair directions must fit a workgroup's 32 items
inline bool pair_ok(int PH, int PW) {
  return PH * ((PW + 1) / 2) <= 32 && PW * ((PH + 1) / 2) <= 32 && PH * PW <= 127;
}
