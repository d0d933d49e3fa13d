// Dense path on the bf16 matrix cores with a 3-way operand split ("bf16x3").
//
// The f32-input MFMA runs at the fp32 vector rate (157 TFLOP/s); the bf16 MFMA
// (v_mfma_f32_32x32x16_bf16) is 16x faster per k-value.  An fp32 value splits
// EXACTLY into three bf16 pieces (a = a1 + a2 + a3, 8 significant bits each, by
// truncation and exact residuals), every bf16 x bf16 product is exact in the
// MFMA's fp32 accumulator, and the six leading cross terms
//     a1b1 + a1b2 + a2b1 + a2b2 + a1b3 + a3b1
// reproduce the fp32 product to ~2^-23 relative (dropped: a2b3, a3b2, a3b3).  Six
// bf16 MFMAs per k-step cost 6/16 of one f32 MFMA step: 2.7x the fp32 matrix rate
// at fp32-level accuracy (the parity bar is 1e-5 relative; measured error ~1e-7).
//
// The same kernels also run the "fp16x2" scheme (SchemeH2 in pp_bf3_common.h: two fp16
// pieces, three MFMAs per k-step instead of six): the weights are scaled per tensor by a
// power of two into fp16's range (pp_absmax_kernel + the pack kernel), the basis is used
// as is, and a tile that meets a basis value too large for fp16 pieces (far extrapolation)
// is recomputed in plain fp32 by its block.
//
// forward  y[b,o] = sum_{i,k} Phi[b,i,k] * w_t[i,k,o]
//   k-dimension ordered (k, i): one MFMA k-step = 16 consecutive inputs at a fixed
//   weight index k, so a lane evaluates the basis of an input once and uses its K
//   columns in K different k-steps.  Block = 4 waves = 4 column tiles (64 outs each)
//   of one 32-row tile; per step of 16 inputs every wave evaluates and splits the
//   basis of 2 inputs per lane and publishes it in LDS ([plane][k][half][row] x 16 B,
//   conflict-free ds_read_b128); weights come pre-split from a packed buffer
//   P[plane][k][step][out][16 inputs] (coalesced 16-byte loads), written once per
//   call by pp_bf3_pack_w_kernel.
#include "pp_bf3_common.h"

namespace {

template <class SC>
__global__ __launch_bounds__(256) void pp_split_pack_w_kernel(const float* __restrict__ wt,
                                                              void* __restrict__ ws, int I, int K,
                                                              int O, int nstep) {
  pack_w_body<SC>((long)blockIdx.x * 256 + threadIdx.x, wt, ws, I, K, O, nstep);
}

// plain fp32 recomputation of a tile (rows b0..b0+rows, outs o_lo..o_hi) by the whole
// block: the rare way out when a ranged scheme met a basis value it cannot represent
template <int N, int K>
__device__ __forceinline__ void fwd_tile_fp32(const float* __restrict__ x, const float* __restrict__ wt,
                                           float* __restrict__ y, const PPShape& sh,
                                           const BasisTab& tab, const DenseCoord3& dc, int b0,
                                           int rows, int o_lo, int o_hi) {
  const int ncol = o_hi - o_lo;
  for (int idx = threadIdx.x; idx < rows * ncol; idx += blockDim.x) {
    const int row = b0 + idx / ncol, col = o_lo + idx % ncol;
    if (row >= sh.B) continue;
    float sum = 0.0f;
    for (int i = 0; i < sh.I; ++i) {
      float phi[K];
      phi3<N, K>(x[(size_t)row * sh.I + i], true, sh, tab, dc, phi);
#pragma unroll
      for (int k = 0; k < K; ++k)
        if (phi[k] != 0.0f) sum = fmaf(phi[k], wt[((size_t)i * K + k) * sh.O + col], sum);
    }
    y[(size_t)row * sh.O + col] = sum * sh.out_scale;
  }
}

template <class SC, int N, int K, int RM, int NT, int G>
__global__ __launch_bounds__(256 * G, (RM == 1 && NT == 2 && G == 2 && K == 5) ? 4 : 1) void pp_fwd_bf3_kernel(const float* __restrict__ x,
                                                             const float* __restrict__ wt,
                                                             const void* __restrict__ ws,
                                                             float* __restrict__ y, PPShape sh,
                                                             BasisTab tab, DenseCoord3 dc,
                                                             int nstep, float* __restrict__ rstd_out,
                                                             float ln_eps,
                                                             int32_t* __restrict__ amax_out) {
  // rstd_out != nullptr: y is the non-affine LayerNorm of the layer's output over the row
  // (LazyLayerNormLastDim fused into the epilogue) and rstd_out[B] its 1/std.
  // amax_out != nullptr as well: y is MaxAbsNormalizationLast of the row instead, rstd_out[B]
  // receives the denominators max|row| + ln_eps and amax_out[B] the first index attaining the
  // maximum (what pp_max_abs_norm_bwd_y_f32 works from).
  // block = G groups of 4 waves.  A group = 4 column tiles (32*NT outs each) x one tile
  // of 32*RM rows; every W fragment a wave loads feeds RM row tiles.  With G = 2 the two
  // groups take alternate 16-input steps of the SAME output tile (split-k inside the
  // block, two waves per SIMD: one group's waits and basis evaluation sit under the
  // other's MFMAs) and add their accumulators through LDS at the end.  nstep is a
  // multiple of G (the packed weights are zero-padded to it).
  using frag = typename SC::frag;
  constexpr int PL = SC::PL;
  PP_TICK(0)
  extern __shared__ __attribute__((aligned(16))) u32x4 phiL[];  // [G][2][PL][K][2][32*RM] x 16 B
  constexpr int ROWS = 32 * RM;
  constexpr int kBufU = PL * K * 2 * ROWS;
  const int lane = threadIdx.x & 63;
  // wave-uniform ids in SGPRs (a VGPR step offset would turn every buffer load into a
  // readfirstlane loop)
  const int wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int cw = wv & 3;
  const int g = G > 1 ? wv >> 2 : 0;
  const int r = lane & 31, h = lane >> 5;
  const int b0 = blockIdx.x * ROWS;
  const int o0 = (blockIdx.y * 4 + cw) * (32 * NT);
  // producer role: inputs 16*s + 8*hb + ipos, +1 of rows b0 + r + 32*rm
  const int hb = cw >> 1, ipos = 4 * (cw & 1) + 2 * h;
  const rsrc_t rx = make_rsrc3(x, (unsigned)sh.B * sh.I * 4u);
  int vx_off[RM];
#pragma unroll
  for (int rm = 0; rm < RM; ++rm)
    vx_off[rm] = (min(b0 + rm * 32 + r, sh.B - 1) * sh.I + 8 * hb + ipos) * 4;
  u32x4* phiG = phiL + g * 2 * kBufU;
  unsigned* phiW = reinterpret_cast<unsigned*>(phiG);
  // consumer role
  const rsrc_t rp = make_rsrc3(static_cast<const char*>(ws) + kWsHeader,
                               (unsigned)((unsigned)PL * K * nstep * sh.O * 32u));
  int vcol[NT];
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) vcol[nt] = (min(o0 + nt * 32 + r, sh.O - 1) * 2 + h) * 16;
  const int step_bytes = sh.O * 32;        // one (plane, k, step) slab
  const int k_bytes = nstep * step_bytes;  // one (plane, k)
  const int plane_bytes = K * k_bytes;

  f32x16 acc[RM][NT];
#pragma unroll
  for (int rm = 0; rm < RM; ++rm)
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
      for (int q = 0; q < 16; ++q) acc[rm][nt][q] = 0.0f;

  // s: global step index (16 inputs); t: this group's iteration, s = t*G + g
  auto xload = [&](int s, float (&xv)[RM][2]) {
#pragma unroll
    for (int rm = 0; rm < RM; ++rm) {
      xv[rm][0] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rx, vx_off[rm], 64 * s, 0));
      xv[rm][1] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rx, vx_off[rm] + 4, 64 * s, 0));
    }
  };
  float lmax = 0.0f;  // ranged schemes: largest |basis value| this lane produced
  auto produce = [&](int s, int buf, const float (&xv)[RM][2]) {
    const int i0 = 16 * s + 8 * hb + ipos;
#pragma unroll
    for (int rm = 0; rm < RM; ++rm) {
      float pa[K], pb[K];
      // (padded inputs i >= I hold the next row's x: evaluate them at x = 0 so that they
      // cannot trip the range check)
      phi3<N, K>(i0 < sh.I ? xv[rm][0] : 0.0f, i0 < sh.I, sh, tab, dc, pa,
                 SC::kRanged ? &lmax : nullptr);
      phi3<N, K>(i0 + 1 < sh.I ? xv[rm][1] : 0.0f, i0 + 1 < sh.I, sh, tab, dc, pb,
                 SC::kRanged ? &lmax : nullptr);
#pragma unroll
      for (int k = 0; k < K; ++k) {
        unsigned w[PL];
        SC::split_pair(pa[k], pb[k], w);
        const int base = (buf * kBufU + (k * 2 + hb) * ROWS + rm * 32 + r) * 4 + (ipos >> 1);
#pragma unroll
        for (int p = 0; p < PL; ++p) phiW[base + p * K * 2 * ROWS * 4] = w[p];
      }
    }
  };
  auto wload = [&](int s, int k, frag (&w)[PL][NT]) {
#pragma unroll
    for (int p = 0; p < PL; ++p)
#pragma unroll
      for (int nt = 0; nt < NT; ++nt)
        w[p][nt] = __builtin_bit_cast(
            frag, __builtin_amdgcn_raw_buffer_load_b128(
                      rp, vcol[nt],
#ifdef PP_W_SAME  // developer experiment: every W load hits the same lines (L1-resident)
                      p * plane_bytes + 0 * (k + s),
#else
                      p * plane_bytes + k * k_bytes + s * step_bytes,
#endif
                      0));
  };
  auto aload = [&](int buf, int k, frag (&a)[RM][PL]) {
#pragma unroll
    for (int rm = 0; rm < RM; ++rm)
#pragma unroll
      for (int p = 0; p < PL; ++p)
        a[rm][p] = __builtin_bit_cast(
            frag, phiG[buf * kBufU + ((p * K + k) * 2 + h) * ROWS + rm * 32 + r]);
  };
  auto mm = [&](const frag (&a)[RM][PL], const frag (&w)[PL][NT]) {
    // the scheme's cross terms, small ones first, a1*b1 last
#pragma unroll
    for (int c = 0; c < SC::NC; ++c)
#pragma unroll
      for (int rm = 0; rm < RM; ++rm)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
          acc[rm][nt] = SC::mfma(a[rm][SC::pa(c)], w[SC::pb(c)][nt], acc[rm][nt]);
  };

  const int T = nstep / G;
  float xv[RM][2], xn[RM][2];
  xload(g, xv);
  produce(g, 0, xv);
  xload(G + g, xv);
  // one W fragment set per weight index k: as soon as the MFMAs of (step, k) are issued,
  // the set is refilled for (next step, k), so every weight load has K-1 k-steps of MFMAs
  // to arrive under (one k-step is shorter than an L2 round trip).  A fragments (LDS):
  // two sets in ping-pong, one k-step ahead.
  // When K sets do not fit the register budget of two waves per SIMD: two sets in
  // ping-pong over the flattened (step, k) sequence instead (one k-step ahead).
  // (32-row tiles at K = 5: two sets, so that two workgroups fit a CU -- see the launch rule)
  constexpr bool RING = K * PL * NT * 4 <= 96 && !(RM == 1 && NT == 2 && K == 5);
  constexpr int NW = RING ? K : 2;
  frag W[NW][PL][NT], A[2][RM][PL];
  // the packed weights' scale: read here (scalar loads, one register), used by the epilogue
  float oscale = sh.out_scale;
  if constexpr (SC::kRanged) {
    float sc, inv;
    h2_scale(absmax_of_parts(ws), sc, inv);
    oscale *= inv;
  }
  if constexpr (RING) {
#pragma unroll
    for (int k = 0; k < K; ++k) wload(g, k, W[k]);
  } else {
    wload(g, 0, W[0]);
  }
  __builtin_amdgcn_s_waitcnt(0x0F70);
  __syncthreads();
  PP_TICK(1)
  // (Both groups run produce -> MFMA.  The wrong rows once blamed on running them in opposite
  // orders were a packed-fp32 instruction of the SLP vectoriser misbehaving when two workgroups
  // share a CU -- DESIGN.md 3c'; the library is built with -fno-slp-vectorize.)
  auto stepfn = [&](int t, auto par) {
    constexpr int P0 = decltype(par)::value;
    const int buf = t & 1;
    const int s = t * G + g;
    xload(s + 2 * G, xn);
    aload(buf, 0, A[P0 & 1]);
    if (t + 1 < T) produce(s + G, buf ^ 1, xv);
#pragma unroll
    for (int k = 0; k < K; ++k) {
      const int cur = (k + P0) & 1, nxt = cur ^ 1;
      if constexpr (RING) {
        if (k + 1 < K) aload(buf, k + 1, A[nxt]);
        mm(A[cur], W[k]);
        wload(s + G, k, W[k]);
      } else {
        if (k + 1 < K) {
          wload(s, k + 1, W[nxt]);
          aload(buf, k + 1, A[nxt]);
        } else {
          wload(s + G, 0, W[nxt]);
        }
        mm(A[cur], W[cur]);
      }
    }
#pragma unroll
    for (int rm = 0; rm < RM; ++rm) {
      xv[rm][0] = xn[rm][0];
      xv[rm][1] = xn[rm][1];
    }
    __syncthreads();
  };
  for (int t = 0; t < T; t += 2) {
    stepfn(t, std::integral_constant<int, 0>{});
    if (t + 1 < T) stepfn(t + 1, std::integral_constant<int, (K & 1)>{});
  }
  PP_TICK(2)

  // a basis value outside fp16's range anywhere in this tile: the tile is redone in fp32
  // below instead of being stored (decided before any store, so the two never overlap)
  bool bad = false;
  if constexpr (SC::kRanged) bad = __syncthreads_or(!(lmax <= kH2Limit));
  // accumulator tiles of this wave: tix = rm*NT + nt.  With two groups, group g finishes
  // the tiles with tix % 2 == g (all of them in group 0 when there is only one) and hands
  // the others to the partner wave (same cw, other group) through LDS (the basis buffers
  // are free: the loop ended on a barrier).
  constexpr int NTILE = RM * NT;
  constexpr int NX = NTILE > 1 ? NTILE / 2 : 1;  // tiles a wave hands over
  constexpr int kRedFloats = G == 2 ? 2 * 4 * NX * 16 * 64 : 0;
  float* red = reinterpret_cast<float*>(phiL);   // [g][cw][NX][16][64]
  float* statL = red + kRedFloats;               // LayerNorm: [4G waves x 2][ROWS] | mean | rstd
  auto finish = [&](auto gc) {
    constexpr int GS = decltype(gc)::value;  // this wave's group
    auto kept = [](int T) { return G == 1 || (NTILE == 1 ? GS == 0 : (T & 1) == GS); };
    if constexpr (G == 2) {
      float* mine = red + ((g * 4 + cw) * NX * 16) * 64 + lane;
      const float* theirs = red + (((g ^ 1) * 4 + cw) * NX * 16) * 64 + lane;
      static_for<NTILE>([&](auto tc) {
        constexpr int T = decltype(tc)::value;
        if constexpr (!kept(T)) {
#pragma unroll
          for (int q = 0; q < 16; ++q) mine[((T >> 1) * 16 + q) * 64] = acc[T / NT][T % NT][q];
        }
      });
      __syncthreads();
      static_for<NTILE>([&](auto tc) {
        constexpr int T = decltype(tc)::value;
        if constexpr (kept(T)) {
#pragma unroll
          for (int q = 0; q < 16; ++q) acc[T / NT][T % NT][q] += theirs[((T >> 1) * 16 + q) * 64];
        }
      });
    }
    static_for<NTILE>([&](auto tc) {
      constexpr int T = decltype(tc)::value;
      if constexpr (kept(T)) {
#pragma unroll
        for (int q = 0; q < 16; ++q) acc[T / NT][T % NT][q] *= oscale;
      }
    });
    PP_TICK(4)
    if (SC::kRanged && amax_out) {  // (fp16x2 scheme only: folds away in the bf16x3 instantiations)
      // fused MaxAbsNormalizationLast over the row (the block holds whole rows: grid.y == 1).
      // The maximum of a set does not depend on the order it is taken in and the division is
      // the correctly rounded one: bit-identical to this kernel + pp_max_abs_fwd_kernel.
      float* mL = statL + 8 * G * ROWS;
      int* iL = reinterpret_cast<int*>(statL);
#pragma unroll
      for (int rm = 0; rm < RM; ++rm)
#pragma unroll
        for (int q = 0; q < 16; ++q) {
          const int rr = rm * 32 + (q & 3) + 8 * (q >> 2) + 4 * h;
          float m = 0.0f;
          static_for<NT>([&](auto nc) {
            constexpr int nt = decltype(nc)::value;
            if (kept(rm * NT + nt)) {  // (folds after unrolling)
              if (o0 + nt * 32 + r < sh.O) m = fmaxf(m, fabsf(acc[rm][nt][q]));
            }
          });
          m = pp_dpp_max<0x128>(m);
          m = pp_dpp_max<0x124>(m);
          m = pp_dpp_max<0x122>(m);
          m = pp_dpp_max<0x121>(m);
          if ((r & 15) == 0) statL[(wv * 2 + (r >> 4)) * ROWS + rr] = m;
        }
      __syncthreads();
      if (threadIdx.x < ROWS) {
        float tot = 0.0f;
#pragma unroll
        for (int w = 0; w < 8 * G; ++w) tot = fmaxf(tot, statL[w * ROWS + threadIdx.x]);
        mL[threadIdx.x] = tot;
      }
      __syncthreads();
#pragma unroll
      for (int rm = 0; rm < RM; ++rm)
#pragma unroll
        for (int q = 0; q < 16; ++q) {
          const int rr = rm * 32 + (q & 3) + 8 * (q >> 2) + 4 * h;
          const float m = mL[rr];
          int idx = 0x7fffffff;  // first index attaining the maximum
          static_for<NT>([&](auto nc) {
            constexpr int nt = decltype(nc)::value;
            if (kept(rm * NT + nt)) {
              const int col = o0 + nt * 32 + r;
              if (col < sh.O && fabsf(acc[rm][nt][q]) == m) idx = min(idx, col);
            }
          });
          idx = pp_dpp_min_i<0x128>(idx);
          idx = pp_dpp_min_i<0x124>(idx);
          idx = pp_dpp_min_i<0x122>(idx);
          idx = pp_dpp_min_i<0x121>(idx);
          if ((r & 15) == 0) iL[(wv * 2 + (r >> 4)) * ROWS + rr] = idx;
        }
      __syncthreads();
      if (threadIdx.x < ROWS) {
        int best = 0x7fffffff;
#pragma unroll
        for (int w = 0; w < 8 * G; ++w) best = min(best, iL[w * ROWS + threadIdx.x]);
        if (b0 + (int)threadIdx.x < sh.B) {
          amax_out[b0 + threadIdx.x] = best;
          rstd_out[b0 + threadIdx.x] = mL[threadIdx.x] + ln_eps;
        }
      }
      static_for<NTILE>([&](auto tc) {
        constexpr int T = decltype(tc)::value;
        if constexpr (kept(T)) {
#pragma unroll
          for (int q = 0; q < 16; ++q) {
            const int rr = (T / NT) * 32 + (q & 3) + 8 * (q >> 2) + 4 * h;
            acc[T / NT][T % NT][q] = acc[T / NT][T % NT][q] / (mL[rr] + ln_eps);
          }
        }
      });
    } else if (rstd_out) {
      // fused non-affine LayerNorm over the row (the block holds whole rows: grid.y == 1):
      // two passes (mean, then centred second moment), partial row sums per wave in LDS
      float* meanL = statL + 8 * G * ROWS;
      float* rsL = meanL + ROWS;
      auto row_pass = [&](bool centred) {
#pragma unroll
        for (int rm = 0; rm < RM; ++rm)
#pragma unroll
          for (int q = 0; q < 16; ++q) {
            const int rr = rm * 32 + (q & 3) + 8 * (q >> 2) + 4 * h;
            const float mu = centred ? meanL[rr] : 0.0f;
            float sacc = 0.0f;
            static_for<NT>([&](auto nc) {
              constexpr int nt = decltype(nc)::value;
              if (kept(rm * NT + nt)) {  // (folds after unrolling)
                const float d = acc[rm][nt][q] - mu;
                if (o0 + nt * 32 + r < sh.O) sacc += centred ? d * d : d;
              }
            });
            // sum over the 16 lanes of a DPP row (four full-rate adds; a __shfl_xor tree is
            // five dependent LDS round trips per value and made this epilogue 13 us long);
            // the two rows of a 32-column tile go to LDS separately
            sacc = pp_dpp_add<0x128>(sacc);
            sacc = pp_dpp_add<0x124>(sacc);
            sacc = pp_dpp_add<0x122>(sacc);
            sacc = pp_dpp_add<0x121>(sacc);
            if ((r & 15) == 0) statL[(wv * 2 + (r >> 4)) * ROWS + rr] = sacc;
          }
        __syncthreads();
        if (threadIdx.x < ROWS) {
          float tot = 0.0f;
#pragma unroll
          for (int w = 0; w < 8 * G; ++w) tot += statL[w * ROWS + threadIdx.x];
          if (!centred) {
            meanL[threadIdx.x] = tot / (float)sh.O;
          } else {
            const float rs = 1.0f / sqrtf(tot / (float)sh.O + ln_eps);
            rsL[threadIdx.x] = rs;
            if (b0 + (int)threadIdx.x < sh.B) rstd_out[b0 + threadIdx.x] = rs;
          }
        }
        __syncthreads();
      };
      row_pass(false);
      row_pass(true);
      static_for<NTILE>([&](auto tc) {
        constexpr int T = decltype(tc)::value;
        if constexpr (kept(T)) {
#pragma unroll
          for (int q = 0; q < 16; ++q) {
            const int rr = (T / NT) * 32 + (q & 3) + 8 * (q >> 2) + 4 * h;
            acc[T / NT][T % NT][q] = (acc[T / NT][T % NT][q] - meanL[rr]) * rsL[rr];
          }
        }
      });
    }
    PP_TICK(5)
    static_for<NTILE>([&](auto tc) {
      constexpr int T = decltype(tc)::value;
      if constexpr (kept(T)) {
        constexpr int rm = T / NT, nt = T % NT;
        const int col = o0 + nt * 32 + r;
#pragma unroll
        for (int q = 0; q < 16; ++q) {
          const int row = b0 + rm * 32 + (q & 3) + 8 * (q >> 2) + 4 * h;
          if (col < sh.O && row < sh.B) y[(size_t)row * sh.O + col] = acc[rm][nt][q];
        }
      }
    });
  };
  if (!bad) {  // block-uniform
    if (G == 1 || g == 0) finish(std::integral_constant<int, 0>{});
    else finish(std::integral_constant<int, 1>{});
  }
  PP_TICK(3)
  if constexpr (SC::kRanged) {
    if (bad) {
      fwd_tile_fp32<N, K>(x, wt, y, sh, tab, dc, b0, ROWS, blockIdx.y * 128 * NT,
                          min(sh.O, (int)(blockIdx.y + 1) * 128 * NT));
      if (amax_out) {  // ... and normalise its rows in place: wave w takes rows w, w+4G, ...
        __syncthreads();
        for (int rr = wv; rr < ROWS && b0 + rr < sh.B; rr += 4 * G) {
          float* yr = y + (size_t)(b0 + rr) * sh.O;
          float m = 0.0f;
          for (int c = lane; c < sh.O; c += 64) m = fmaxf(m, fabsf(yr[c]));
#pragma unroll
          for (int dd = 32; dd > 0; dd >>= 1) m = fmaxf(m, __shfl_xor(m, dd));
          int idx = 0x7fffffff;
          for (int c = lane; c < sh.O; c += 64)
            if (fabsf(yr[c]) == m) idx = min(idx, c);
#pragma unroll
          for (int dd = 32; dd > 0; dd >>= 1) idx = min(idx, __shfl_xor(idx, dd));
          const float den = m + ln_eps;
          for (int c = lane; c < sh.O; c += 64) yr[c] = yr[c] / den;
          if (lane == 0) {
            rstd_out[b0 + rr] = den;
            amax_out[b0 + rr] = idx;
          }
        }
      } else if (rstd_out) {
        __syncthreads();
        for (int rr = wv; rr < ROWS && b0 + rr < sh.B; rr += 4 * G) {
          float* yr = y + (size_t)(b0 + rr) * sh.O;
          float sacc = 0.0f;
          for (int c = lane; c < sh.O; c += 64) sacc += yr[c];
#pragma unroll
          for (int dd = 32; dd > 0; dd >>= 1) sacc += __shfl_xor(sacc, dd);
          const float mu = sacc / (float)sh.O;
          float vacc = 0.0f;
          for (int c = lane; c < sh.O; c += 64) vacc += (yr[c] - mu) * (yr[c] - mu);
#pragma unroll
          for (int dd = 32; dd > 0; dd >>= 1) vacc += __shfl_xor(vacc, dd);
          const float rs = 1.0f / sqrtf(vacc / (float)sh.O + ln_eps);
          for (int c = lane; c < sh.O; c += 64) yr[c] = (yr[c] - mu) * rs;
          if (lane == 0) rstd_out[b0 + rr] = rs;
        }
      }
    }
  }
}

// 16-input steps, padded to even so the two-group kernel splits them evenly
int nstep_of(const PPShape& sh) { return ((sh.I + 31) / 32) * 2; }

template <class SC, int N, int K, int RM, int NT, int G>
hipError_t launch_bf3(const float* x, const float* wt, const void* ws, float* y, const PPShape& sh,
                      const BasisTab& tab, const DenseCoord3& dc, int nstep, float* rstd,
                      float ln_eps, int32_t* amax, hipStream_t st) {
  size_t lds = (size_t)G * 2 * SC::PL * K * 2 * 32 * RM * 16;
  // epilogue: exchange area of the two groups + LayerNorm row statistics
  const size_t red = G == 2 ? (size_t)2 * 4 * std::max(1, RM * NT / 2) * 16 * 64 * 4 : 0;
  lds = std::max(lds, red + (size_t)(8 * G + 2) * 32 * RM * 4);
  if (lds > 160 * 1024) return hipErrorInvalidValue;
  if (lds > 48 * 1024) {  // per call: the attribute is per device, and setting it is cheap
    hipError_t e = hipFuncSetAttribute((const void*)pp_fwd_bf3_kernel<SC, N, K, RM, NT, G>,
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
  }
  dim3 grid((sh.B + 32 * RM - 1) / (32 * RM), (sh.O + 128 * NT - 1) / (128 * NT));
  if (rstd && grid.y != 1) return hipErrorInvalidValue;  // the fused norm needs whole rows
  hipLaunchKernelGGL((pp_fwd_bf3_kernel<SC, N, K, RM, NT, G>), grid, dim3(256 * G), lds, st, x, wt,
                     ws, y, sh, tab, dc, nstep, rstd, ln_eps, amax);
  return hipGetLastError();
}

template <class SC>
void pack_w_split(const float* wt, void* ws, const PPShape& sh, bool with_absmax, hipStream_t st) {
  const int nstep = nstep_of(sh);
  if (SC::kRanged && with_absmax)
    hipLaunchKernelGGL(pp_absmax_kernel, dim3(kAbsmaxBlocks), dim3(256), 0, st, wt,
                       (long)sh.I * sh.K * sh.O, static_cast<unsigned*>(ws));
  const long pslots = (long)sh.K * nstep * sh.O;
  hipLaunchKernelGGL(pp_split_pack_w_kernel<SC>, dim3((unsigned)((pslots + 255) / 256)), dim3(256),
                     0, st, wt, ws, sh.I, sh.K, sh.O, nstep);
}

template <class SC>
hipError_t launch_fwd_split(const float* x, const float* wt, float* y, int32_t* seg, void* ws,
                            const PPShape& sh, const BasisTab& tab, bool prepacked, float* rstd,
                            float ln_eps, int32_t* amax, hipStream_t st) {
  const int nstep = nstep_of(sh);
  DenseCoord3 dc;
  dc.w2 = 2.0f / (float)sh.S;
  dc.inv_w = (float)sh.S / 2.0f;
  dc.inv_len = 1.0f / sh.length;
  if (!prepacked) pack_w_split<SC>(wt, ws, sh, true, st);
  // variant: 64-row tiles once they fill the chip (K <= 6: two groups' basis buffers fit
  // LDS); waves of 32 outs when O <= 128 so all four column waves have work.  Below 16384
  // rows the 64-row tiles leave CUs idle (8192 rows: 128 workgroups) and the 32-row variant
  // runs two workgroups per CU (for K = 5 that takes the two-set W ping-pong instead of the
  // K-set ring: 90 registers instead of 154) -- 25 vs 34 us at 8192 rows, 18.5 vs 20.3 at
  // 1000 on the cfg2 hidden layer; at 16384 rows and above the 64-row tiles win (39 vs 41).
  int rm = sh.B >= 16384 && sh.K <= 6 ? 2 : 1;
  int nt = sh.O > 128 ? 2 : 1;
#ifdef PP_TUNING  // developer builds only: PP_BF3_VARIANT=<rm><nt>
  if (const char* v = getenv("PP_BF3_VARIANT")) {
    const int code = atoi(v);
    if (code / 10 >= 1 && code / 10 <= 2 && code % 10 >= 1 && code % 10 <= 2) {
      rm = sh.K <= 6 ? code / 10 : 1;
      nt = code % 10;
    }
  }
#endif
#define X(N_, K_)                                                                              \
  if (sh.n == N_ && sh.K == K_) {                                                              \
    hipError_t e = hipErrorInvalidValue;                                                       \
    if constexpr (K_ <= 6) {                                                                   \
      if (rm == 2 && nt == 2)                                                                  \
        e = launch_bf3<SC, N_, K_, 2, 2, 2>(x, wt, ws, y, sh, tab, dc, nstep, rstd, ln_eps, amax, st);             \
      else if (rm == 2)                                                                        \
        e = launch_bf3<SC, N_, K_, 2, 1, 2>(x, wt, ws, y, sh, tab, dc, nstep, rstd, ln_eps, amax, st);             \
    }                                                                                          \
    if (rm == 1 && nt == 2) e = launch_bf3<SC, N_, K_, 1, 2, 2>(x, wt, ws, y, sh, tab, dc, nstep, rstd, ln_eps, amax, st); \
    else if (rm == 1) e = launch_bf3<SC, N_, K_, 1, 1, 2>(x, wt, ws, y, sh, tab, dc, nstep, rstd, ln_eps, amax, st); \
    if (e == hipSuccess && seg) e = pp_launch_segment_ids(x, seg, sh, st);                     \
    return e;                                                                                  \
  }
  PP_BF3_COMBOS(X)
#undef X
  return hipErrorInvalidValue;
}

}  // namespace

bool pp_bf3_supported(const PPShape& sh) {
  if (!pp_dense_supported(sh)) return false;
  return (long)3 * sh.K * nstep_of(sh) * sh.O * 32 < (1L << 31);
}

size_t pp_bf3_fwd_workspace_bytes(const PPShape& sh) {
  return kWsHeader + (size_t)3 * sh.K * nstep_of(sh) * sh.O * 32;  // (fp16x2 uses 2 of 3 planes)
}

hipError_t pp_launch_fwd_bf3(const float* x, const float* wt, float* y, int32_t* seg, void* ws,
                             const PPShape& sh, const BasisTab& tab, hipStream_t st) {
  return launch_fwd_split<SchemeBf3>(x, wt, y, seg, ws, sh, tab, false, nullptr, 0.0f, nullptr, st);
}

hipError_t pp_launch_fwd_h2(const float* x, const float* wt, float* y, int32_t* seg, void* ws,
                            const PPShape& sh, const BasisTab& tab, bool prepacked, float* rstd,
                            float ln_eps, int32_t* amax, hipStream_t st) {
  return launch_fwd_split<SchemeH2>(x, wt, y, seg, ws, sh, tab, prepacked, rstd, ln_eps, amax, st);
}

// whole rows in one block: what the fused LayerNorm epilogue needs
bool pp_h2_fwd_ln_supported(const PPShape& sh) {
  return pp_bf3_supported(sh) && sh.O <= 256;
}

// fp16x2 forward layout only (header | P), absmax included: first half of pp_pack_weights
size_t pp_h2_fwd_packed_bytes(const PPShape& sh) {
  return kWsHeader + (size_t)2 * sh.K * nstep_of(sh) * sh.O * 32;
}
namespace {
// both fp16x2 weight layouts in one launch: blocks [0, nblk_p) write P, the rest write Q
__global__ __launch_bounds__(256) void pp_h2_pack_both_kernel(const float* __restrict__ wt,
                                                              void* __restrict__ ws, unsigned q_off,
                                                              int nblk_p, int I, int K, int O,
                                                              int nstep, int nso, int Ipad) {
  if ((int)blockIdx.x < nblk_p)
    pack_w_body<SchemeH2>((long)blockIdx.x * 256 + threadIdx.x, wt, ws, I, K, O, nstep);
  else
    pack_wo_body<SchemeH2>((long)(blockIdx.x - nblk_p) * 256 + threadIdx.x, wt, ws, q_off, I, K, O,
                           nso, Ipad);
}
}  // namespace

// ---- the same for several layers at once (pp_pack_weights_multi_f32): a training step packs
// every dense-regime layer's weights once; eight launches of 4-8 us each, all latency, become
// two whose blocks run side by side.
struct PackDesc {
  const float* wt;
  void* ws;
  long n;
  unsigned q_off;
  int nblk_p, nblk_q, I, K, O, nstep, nso, Ipad;
};
struct PackDescs {
  PackDesc d[PP_PACK_MAX_LAYERS];
};
namespace {
__global__ __launch_bounds__(256) void pp_absmax_multi_kernel(PackDescs ds) {
  const PackDesc& d = ds.d[blockIdx.y];
  absmax_body(d.wt, d.n, static_cast<unsigned*>(d.ws));
}
__global__ __launch_bounds__(256) void pp_h2_pack_multi_kernel(PackDescs ds) {
  const PackDesc& d = ds.d[blockIdx.y];
  const int b = blockIdx.x;
  if (b < d.nblk_p)
    pack_w_body<SchemeH2>((long)b * 256 + threadIdx.x, d.wt, d.ws, d.I, d.K, d.O, d.nstep);
  else if (b < d.nblk_p + d.nblk_q)
    pack_wo_body<SchemeH2>((long)(b - d.nblk_p) * 256 + threadIdx.x, d.wt, d.ws, d.q_off, d.I, d.K,
                           d.O, d.nso, d.Ipad);
}
}  // namespace

void pp_launch_pack_multi_h2(int count, const float* const* wt, void* const* ws,
                             const PPShape* shapes, hipStream_t st) {
  PackDescs ds;
  int maxblk = 0;
  for (int l = 0; l < count; ++l) {
    const PPShape& sh = shapes[l];
    PackDesc& d = ds.d[l];
    d.wt = wt[l];
    d.ws = ws[l];
    d.n = (long)sh.I * sh.K * sh.O;
    d.q_off = (unsigned)pp_h2_fwd_packed_bytes(sh);
    d.I = sh.I; d.K = sh.K; d.O = sh.O;
    d.nstep = nstep_of(sh);
    d.nso = ((sh.O + 31) / 32) * 2;
    d.Ipad = ((sh.I + 31) / 32) * 32;
    d.nblk_p = (int)(((long)sh.K * d.nstep * sh.O + 255) / 256);
    d.nblk_q = (int)(((long)d.nso * sh.K * d.Ipad + 255) / 256);
    maxblk = std::max(maxblk, d.nblk_p + d.nblk_q);
  }
  hipLaunchKernelGGL(pp_absmax_multi_kernel, dim3(kAbsmaxBlocks, count), dim3(256), 0, st, ds);
  hipLaunchKernelGGL(pp_h2_pack_multi_kernel, dim3(maxblk, count), dim3(256), 0, st, ds);
}

// header (absmax) | P | Q: what pp_pack_weights_f32 writes
void pp_launch_pack_both_h2(const float* wt, void* ws, const PPShape& sh, hipStream_t st) {
  const int nstep = nstep_of(sh);
  const int nso = ((sh.O + 31) / 32) * 2, Ipad = ((sh.I + 31) / 32) * 32;  // as pp_bf3_bwdx.hip
  hipLaunchKernelGGL(pp_absmax_kernel, dim3(kAbsmaxBlocks), dim3(256), 0, st, wt,
                     (long)sh.I * sh.K * sh.O, static_cast<unsigned*>(ws));
  const int nblk_p = (int)(((long)sh.K * nstep * sh.O + 255) / 256);
  const int nblk_q = (int)(((long)nso * sh.K * Ipad + 255) / 256);
  hipLaunchKernelGGL(pp_h2_pack_both_kernel, dim3(nblk_p + nblk_q), dim3(256), 0, st, wt, ws,
                     (unsigned)pp_h2_fwd_packed_bytes(sh), nblk_p, sh.I, sh.K, sh.O, nstep, nso,
                     Ipad);
}
