// Split-operand backward-weight (schemes: pp_bf3_common.h; overview: pp_bf3_fwd.hip).
#include "pp_bf3_common.h"

namespace {

// ---------------------------------------------------------------------------
// backward-weight on the bf16x3 scheme:
//   gw[(i,k), o] = sum_b Phi[b,i,k] * gy[b,o]       (reduction over rows on bf16 MFMA)
// gy is split once per call by pp_bf3_pack_gy_kernel into G[plane][16-row step][Opad]
// [16 rows] (B-operand fragments, 16-byte coalesced loads, two steps in flight).  Block
// = 4 waves = 4 column tiles of 32*NT outs of ONE 32-input tile over one chunk of rows;
// per 16-row step the block evaluates + splits the basis of its 16 x 32 (row, input)
// pairs once (2 pairs per lane) into LDS ([plane][k][half][input] x 16 B, three buffers:
// step t+2 is written and step t+1's fragments are read under the MFMAs of step t) and
// every wave runs 6*K*NT MFMAs into its K x NT accumulators.  Per-chunk partial sums go
// to the workspace and are added in a fixed order by pp_dense_reduce_kernel.
// Ranged scheme (fp16x2): gy is scaled by a power of two per (block of 32 steps = 512
// rows, out column), taken from that block's column maximum inside the pack kernel
// (scales in the workspace header); when the scale of its columns changes from one block
// of rows to the next a wave rescales its accumulators (exact), the store undoes the last
// scale; a row chunk whose basis leaves fp16's range is recomputed in fp32 by its block.
// ---------------------------------------------------------------------------
constexpr int kSB = 32;  // steps per scale block

// ranged pack: block = (scale block, kPackCols out columns); 256/kPackCols step groups x
// kPackCols columns; every thread holds its steps x 16 rows in registers between the max and
// the split pass.  32 columns: 256 blocks at 16384 rows x 256 outs (64 gave 128 blocks, half
// the CUs idle, and twice the registers per thread).
constexpr int kPackCols = 32;

template <class SC>
__global__ __launch_bounds__(256) void pp_split_pack_gy_ranged_kernel(
    const float* __restrict__ gy, float2* __restrict__ gscale, u32x4* __restrict__ G, int B, int O,
    int Opad, int nbs) {
  __shared__ unsigned cmax[kPackCols];
  constexpr int NG = 256 / kPackCols;  // step groups
  const int sb = blockIdx.x;
  const int oc = threadIdx.x & (kPackCols - 1), tg = threadIdx.x / kPackCols;
  const int o = blockIdx.y * kPackCols + oc;
  if (threadIdx.x < kPackCols) cmax[threadIdx.x] = 0u;
  __syncthreads();
  constexpr int PER = kSB / NG;
  float v[PER][16];
  unsigned mx = 0;
#pragma unroll
  for (int m = 0; m < PER; ++m) {
    const int t = sb * kSB + tg + NG * m;
#pragma unroll
    for (int j = 0; j < 16; ++j) {
      const int b = 16 * t + j;
      v[m][j] = (b < B && o < O) ? gy[(size_t)b * O + o] : 0.0f;
      mx = max(mx, __float_as_uint(v[m][j]) & 0x7fffffffu);
    }
  }
  atomicMax(&cmax[oc], mx);
  __syncthreads();
  float sc, inv;
  h2_scale(cmax[oc], sc, inv);
  if (o >= Opad) return;
  if (tg == 0) gscale[(size_t)sb * Opad + o] = make_float2(sc, inv);
#pragma unroll
  for (int m = 0; m < PER; ++m) {
    const int t = sb * kSB + tg + NG * m;
    if (t >= nbs) continue;
#pragma unroll
    for (int j = 0; j < 16; ++j) v[m][j] *= sc;
    u32x4 lo[SC::PL], hi[SC::PL];
    split16<SC>(v[m], lo, hi);
#pragma unroll
    for (int p = 0; p < SC::PL; ++p) {
      G[(((size_t)p * nbs + t) * Opad + o) * 2] = lo[p];
      G[(((size_t)p * nbs + t) * Opad + o) * 2 + 1] = hi[p];
    }
  }
}

// The same pack pass with the LayerNorm backward of the layer's output folded in (fp16x2 only):
// the caller holds the gradient that reached the LayerNorm (gy), the LayerNorm's output (yn),
// its rstd, and the row sums (sum gy, sum gy*yn) the kernel that produced gy left behind in
// tiles (pp_bf3_bwdx.hip); every element becomes g = rstd * (gy - m1 - yn * m2) -- the
// expression of pp_layer_norm_bwd_kernel -- on its way into the registers, is written out in
// fp32 for the layer's backward-input kernel (g_out) and split for the weight gradient.  One
// pass over gy and yn instead of the LayerNorm-backward kernel plus the pack kernel.
template <class SC>
__global__ __launch_bounds__(256) void pp_split_pack_gy_ln_ranged_kernel(
    const float* __restrict__ gy, const float* __restrict__ yn, const float* __restrict__ rstd,
    const float2* __restrict__ rowstats, int ntile, float* __restrict__ g_out,
    float2* __restrict__ gscale, u32x4* __restrict__ G, int B, int O, int Opad, int nbs) {
  __shared__ unsigned cmax[kPackCols];
  __shared__ float4 coef[kSB * 16];  // per row of the scale block: (rstd, m1, m2, -)
  constexpr int NG = 256 / kPackCols;  // step groups
  const int sb = blockIdx.x;
  const int oc = threadIdx.x & (kPackCols - 1), tg = threadIdx.x / kPackCols;
  const int o = blockIdx.y * kPackCols + oc;
  if (threadIdx.x < kPackCols) cmax[threadIdx.x] = 0u;
  // One wave per SIMD (256 blocks of 4 waves on 256 CUs): nothing but this thread's own loads
  // hides their latency, so all 2 x 64 of them are issued before anything waits -- gy and yn of
  // the thread's 64 elements, then the row sums -- and only then the arithmetic runs.  (First
  // version: load pair, compute, store, element by element in 81 registers: 32.7 us against
  // 8.4 + 9.3 for the two kernels it replaces.)
  constexpr int PER = kSB / NG;
  float v[PER][16], u[PER][16];
#pragma unroll
  for (int m = 0; m < PER; ++m) {
    const int tl = tg + NG * m;  // step inside the scale block
#pragma unroll
    for (int j = 0; j < 16; ++j) {
      const int b = sb * (kSB * 16) + 16 * tl + j;
      const bool ok = b < B && o < O;
      const size_t at = ok ? (size_t)b * O + o : 0;
      v[m][j] = gy[at];
      u[m][j] = yn[at];
    }
  }
  for (int lr = threadIdx.x; lr < kSB * 16; lr += 256) {
    const int b = sb * (kSB * 16) + lr;
    float4 c = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    if (b < B) {
      float s1 = 0.0f, s2 = 0.0f;
      for (int tl = 0; tl < ntile; ++tl) {  // fixed order: reproducible
        const float2 s = rowstats[(size_t)tl * B + b];
        s1 += s.x;
        s2 += s.y;
      }
      c = make_float4(rstd[b], s1 / (float)O, s2 / (float)O, 0.0f);
    }
    coef[lr] = c;
  }
  __syncthreads();
  unsigned mx = 0;
#pragma unroll
  for (int m = 0; m < PER; ++m) {
    const int tl = tg + NG * m;
#pragma unroll
    for (int j = 0; j < 16; ++j) {
      const int lr = 16 * tl + j;
      const int b = sb * (kSB * 16) + lr;
      float g = 0.0f;
      if (b < B && o < O) {
        const float4 c = coef[lr];
        g = c.x * (v[m][j] - c.y - u[m][j] * c.z);
        g_out[(size_t)b * O + o] = g;
      }
      v[m][j] = g;
      mx = max(mx, __float_as_uint(g) & 0x7fffffffu);
    }
  }
  atomicMax(&cmax[oc], mx);
  __syncthreads();
  float sc, inv;
  h2_scale(cmax[oc], sc, inv);
  if (o >= Opad) return;
  if (tg == 0) gscale[(size_t)sb * Opad + o] = make_float2(sc, inv);
#pragma unroll
  for (int m = 0; m < PER; ++m) {
    const int t = sb * kSB + tg + NG * m;
    if (t >= nbs) continue;
#pragma unroll
    for (int j = 0; j < 16; ++j) v[m][j] *= sc;
    u32x4 lo[SC::PL], hi[SC::PL];
    split16<SC>(v[m], lo, hi);
#pragma unroll
    for (int p = 0; p < SC::PL; ++p) {
      G[(((size_t)p * nbs + t) * Opad + o) * 2] = lo[p];
      G[(((size_t)p * nbs + t) * Opad + o) * 2 + 1] = hi[p];
    }
  }
}

// plain fp32 recomputation of a block's partial tile (inputs i0..i0+32, outs o_lo..o_hi,
// rows cbeg..cend): the rare way out when the ranged scheme met a basis value beyond fp16
template <int N, int K>
__device__ __forceinline__ void bwd_w_tile_fp32(const float* __restrict__ x,
                                                const float* __restrict__ gy,
                                                float* __restrict__ dst, const PPShape& sh,
                                                const BasisTab& tab, const DenseCoord3& dc, int i0,
                                                int o_lo, int o_hi, int cbeg, int cend,
                                                float scale) {
  const int ncol = o_hi - o_lo;
  for (int idx = threadIdx.x; idx < 32 * ncol; idx += blockDim.x) {
    const int i = i0 + idx / ncol, col = o_lo + idx % ncol;
    if (i >= sh.I) continue;
    float a[K];
#pragma unroll
    for (int k = 0; k < K; ++k) a[k] = 0.0f;
    for (int b = cbeg; b < cend; ++b) {
      float phi[K];
      phi3<N, K>(x[(size_t)b * sh.I + i], true, sh, tab, dc, phi);
      const float g = gy[(size_t)b * sh.O + col];
#pragma unroll
      for (int k = 0; k < K; ++k) a[k] = fmaf(phi[k], g, a[k]);
    }
#pragma unroll
    for (int k = 0; k < K; ++k) dst[((size_t)i * K + k) * sh.O + col] = a[k] * scale;
  }
}

template <class SC>
__global__ __launch_bounds__(256) void pp_split_pack_gy_kernel(const float* __restrict__ gy,
                                                               u32x4* __restrict__ G, int B, int O,
                                                               int Opad, int nbs) {
  const long id = (long)blockIdx.x * 256 + threadIdx.x;
  if (id >= (long)nbs * Opad) return;
  const int o = (int)(id % Opad);
  const int t = (int)(id / Opad);
  float v[16];
#pragma unroll
  for (int j = 0; j < 16; ++j) {
    const int b = 16 * t + j;
    v[j] = (b < B && o < O) ? gy[(size_t)b * O + o] : 0.0f;
  }
  u32x4 lo[SC::PL], hi[SC::PL];
  split16<SC>(v, lo, hi);
#pragma unroll
  for (int p = 0; p < SC::PL; ++p) {
    G[(((size_t)p * nbs + t) * Opad + o) * 2] = lo[p];
    G[(((size_t)p * nbs + t) * Opad + o) * 2 + 1] = hi[p];
  }
}

template <class SC, int N, int K, int NT>
__global__ __launch_bounds__(256) void pp_bwd_w_bf3_kernel(const float* __restrict__ x,
                                                           const float* __restrict__ gy,
                                                           const float2* __restrict__ gscale,
                                                           const void* __restrict__ G,
                                                           float* __restrict__ out, PPShape sh,
                                                           BasisTab tab, DenseCoord3 dc,
                                                           int rows_per_chunk, int nbs, int Opad,
                                                           float scale) {
  using frag = typename SC::frag;
  constexpr int PL = SC::PL;
  extern __shared__ __attribute__((aligned(16))) u32x4 phiL[];  // [3 buf][PL][K][2][32] x 16 B
  constexpr int kBufU = PL * K * 2 * 32;
  const int lane = threadIdx.x & 63;
  const int wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int r = lane & 31, h = lane >> 5;
  // grid = (row chunks, input tiles, out tiles): workgroups go round-robin over the 8 XCDs
  // by linear id, so the input-tile blocks of one row chunk share an XCD and its L2 serves
  // all but the first read of the chunk's gy fragments (with the input tile in x they sat on
  // 8 different XCDs and each fetched the chunk through the fabric: 171 MB vs 34 MB)
  const int i0 = blockIdx.y * 32;
  const int chunk = blockIdx.x;
  const int o0 = (blockIdx.z * 4 + wv) * (32 * NT);
  const int cbeg = chunk * rows_per_chunk;  // multiple of 16
  const int cend = min(sh.B, cbeg + rows_per_chunk);
  const int nst = (cend - cbeg + 15) >> 4;
  const int t0 = cbeg >> 4;

  // producer: input i0 + il, rows cbeg + 16 t + 2 rp, + 1
  const int il = threadIdx.x & 31, rp = threadIdx.x >> 5;
  const int ic = min(i0 + il, sh.I - 1);
  const rsrc_t rx = make_rsrc3(x + (size_t)cbeg * sh.I, (unsigned)(cend - cbeg) * sh.I * 4u);
  const int vx = (2 * rp * sh.I + ic) * 4;
  const int xrow = sh.I * 4, xstep = 16 * sh.I * 4;
  unsigned* phiW = reinterpret_cast<unsigned*>(phiL);
  const int pw_base = ((rp >> 2) * 32 + il) * 4 + (rp & 3);
  auto xload = [&](int t, float (&xv)[2]) {
    xv[0] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rx, vx, t * xstep, 0));
    xv[1] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rx, vx + xrow, t * xstep, 0));
  };
  float lmax = 0.0f;  // ranged schemes: largest |basis value| this lane produced
  auto produce = [&](int t, int buf, const float (&xv)[2]) {
    const int row = cbeg + 16 * t + 2 * rp;
    float pa[K], pb[K];
    // (rows past the chunk: x = 0, so that they cannot trip the range check)
    phi3<N, K>(row < cend ? xv[0] : 0.0f, row < cend, sh, tab, dc, pa, SC::kRanged ? &lmax : nullptr);
    phi3<N, K>(row + 1 < cend ? xv[1] : 0.0f, row + 1 < cend, sh, tab, dc, pb,
               SC::kRanged ? &lmax : nullptr);
#pragma unroll
    for (int k = 0; k < K; ++k) {
      unsigned w[PL];
      SC::split_pair(pa[k], pb[k], w);
      const int base = (buf * kBufU + k * 64) * 4 + pw_base;
#pragma unroll
      for (int p = 0; p < PL; ++p) phiW[base + p * K * 64 * 4] = w[p];
    }
  };
  // consumer
  const rsrc_t rg = make_rsrc3(G, (unsigned)((unsigned)PL * nbs * Opad * 32u));
  int vcol[NT];
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) vcol[nt] = (min(o0 + nt * 32 + r, Opad - 1) * 2 + h) * 16;
  const int gstep = Opad * 32, gplane = nbs * gstep;
  auto bload = [&](int t, frag (&bf)[PL][NT]) {
#pragma unroll
    for (int p = 0; p < PL; ++p)
#pragma unroll
      for (int nt = 0; nt < NT; ++nt)
        bf[p][nt] = __builtin_bit_cast(
            frag, __builtin_amdgcn_raw_buffer_load_b128(rg, vcol[nt],
                                                        p * gplane + (t0 + t) * gstep, 0));
  };
  auto aload = [&](int buf, frag (&a)[K][PL]) {
#pragma unroll
    for (int k = 0; k < K; ++k)
#pragma unroll
      for (int p = 0; p < PL; ++p)
        a[k][p] = __builtin_bit_cast(frag, phiL[buf * kBufU + ((p * K + k) * 2 + h) * 32 + r]);
  };
  f32x16 acc[K][NT];
#pragma unroll
  for (int k = 0; k < K; ++k)
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
      for (int q = 0; q < 16; ++q) acc[k][nt][q] = 0.0f;
  auto mm = [&](const frag (&a)[K][PL], const frag (&bf)[PL][NT]) {
#pragma unroll
    for (int k = 0; k < K; ++k)
#pragma unroll
      for (int c = 0; c < SC::NC; ++c)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
          acc[k][nt] = SC::mfma(a[k][SC::pa(c)], bf[SC::pb(c)][nt], acc[k][nt]);
  };
  // ranged: scale of this lane's columns in the current block of kSB steps
  int ccol[NT];
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) ccol[nt] = min(o0 + nt * 32 + r, Opad - 1);
  float2 csc[NT];
  int sb_cur = t0 / kSB;
  if constexpr (SC::kRanged) {
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) csc[nt] = gscale[(size_t)sb_cur * Opad + ccol[nt]];
  }
  auto rescale = [&](int t) {  // entering step t: a new scale block?
    const int sb = (t0 + t) / kSB;
    if (sb != sb_cur) {  // block-uniform
      sb_cur = sb;
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) {
        const float2 nsc = gscale[(size_t)sb * Opad + ccol[nt]];
        const float f = nsc.x * csc[nt].y;  // new scale / old scale: a power of two
        // blocks of rows whose gradients differ by more than 2^60: the rescaled accumulators
        // could overflow where plain fp32 would not -> this tile takes the fp32 path below
        const int de = (int)((__float_as_uint(nsc.x) >> 23) & 0xffu) -
                       (int)((__float_as_uint(csc[nt].x) >> 23) & 0xffu);
        if (de > 60 || de < -60) lmax = __builtin_inff();
        csc[nt] = nsc;
#pragma unroll
        for (int k = 0; k < K; ++k)
#pragma unroll
          for (int q = 0; q < 16; ++q) acc[k][nt][q] *= f;
      }
    }
  };

  float xr[2][2];
  frag Bf[3][PL][NT], A[2][K][PL];
  {
    float x0[2], x1[2];
    xload(0, x0);
    xload(1, x1);
    xload(2, xr[0]);
    bload(0, Bf[0]);
    bload(1, Bf[1]);
    produce(0, 0, x0);
    produce(1, 1, x1);
  }
  __syncthreads();
  aload(0, A[0]);
  // step t (J = t mod 6 at compile time): x of step t+3 and gy fragments of step t+2 are
  // issued, the basis fragments of step t+1 are read, the basis of step t+2 is written,
  // under the MFMAs of step t.  Steps past the chunk evaluate to zero (valid flags).
  auto stepfn = [&](int t, auto jc) {
    constexpr int J = decltype(jc)::value;
    if constexpr (SC::kRanged) rescale(t);
    // (PP_NO_*: developer ablations of tools/bwdw_timing.hip -- wrong results, timing only)
#ifndef PP_NO_XLOAD
    xload(t + 3, xr[(J + 1) & 1]);
#endif
#ifndef PP_NO_BLOAD
    bload(t + 2, Bf[(J + 2) % 3]);
#endif
#ifndef PP_NO_ALOAD
    aload((J + 1) % 3, A[(J + 1) & 1]);
#endif
#ifndef PP_NO_PRODUCE
    produce(t + 2, (J + 2) % 3, xr[J & 1]);
#endif
#ifndef PP_NO_MFMA
    mm(A[J & 1], Bf[J % 3]);
#endif
    // One wave per SIMD here (256 + 220 registers), so nothing but the wave's own instruction
    // order overlaps the basis evaluation with the MFMAs: left alone the compiler interleaves
    // the first 13 of a step's 30 MFMAs with the VALU work and issues the other 17 back to
    // back behind the barrier.  Groups of one MFMA + four VALU spread them over the whole
    // step: 53.4 -> 50.1 us per call at 16384 rows (tools/bwdw_timing.hip; 2 / 3 / 5 / 6 VALU
    // per group: 54.1 / 51.9 / 51.1 / 52.1).
    // (Two-tile waves only: with one tile there are ten VALU per MFMA and the groups bought
    // nothing -- 27 vs 30 us on the 250 -> 128 layer at 4096 rows.)
    if constexpr (NT == 2) {
#pragma unroll
      for (int i = 0; i < K * SC::NC * NT; ++i) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);  // one MFMA
        __builtin_amdgcn_sched_group_barrier(0x002, 4, 0);  // four VALU under it
      }
    }
#ifndef PP_NO_BARRIER
    __syncthreads();
#endif
  };
  for (int t = 0; t < nst; t += 6) {
    stepfn(t, std::integral_constant<int, 0>{});
    if (t + 1 < nst) stepfn(t + 1, std::integral_constant<int, 1>{});
    if (t + 2 < nst) stepfn(t + 2, std::integral_constant<int, 2>{});
    if (t + 3 < nst) stepfn(t + 3, std::integral_constant<int, 3>{});
    if (t + 4 < nst) stepfn(t + 4, std::integral_constant<int, 4>{});
    if (t + 5 < nst) stepfn(t + 5, std::integral_constant<int, 5>{});
  }
  float* dst = out + (size_t)chunk * sh.I * K * sh.O;
  // a basis value outside fp16's range anywhere in this tile: the tile is redone in fp32
  // below instead of being stored (decided before any store, so the two never overlap)
  bool bad = false;
  if constexpr (SC::kRanged) bad = __syncthreads_or(!(lmax <= kH2Limit));
  if (!bad && o0 < sh.O) {  // wave-uniform: else this wave ran on clamped columns
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
      const int col = o0 + nt * 32 + r;
      float cs = scale;
      if constexpr (SC::kRanged) cs *= csc[nt].y;
      if (col < sh.O) {
#pragma unroll
        for (int k = 0; k < K; ++k)
#pragma unroll
          for (int q = 0; q < 16; ++q) {
            const int row = i0 + (q & 3) + 8 * (q >> 2) + 4 * h;  // input index
#ifdef PP_NO_STORE  // (ablation: one store per tile keeps the accumulators alive)
            if (row < sh.I && acc[k][nt][q] * cs == 12345.678f)
#else
            if (row < sh.I)
#endif
              dst[((size_t)row * K + k) * sh.O + col] = acc[k][nt][q] * cs;
          }
      }
    }
  }
  if (bad)
    bwd_w_tile_fp32<N, K>(x, gy, dst, sh, tab, dc, i0, blockIdx.z * 128 * NT,
                          min(sh.O, (int)(blockIdx.z + 1) * 128 * NT), cbeg, cend, scale);
}

}  // namespace

// ---- backward-weight ----
namespace {
struct WPlan3 {
  int nt, ocols, nchunk, rows_per_chunk, nbs, opad, nsb;
  size_t sc_bytes, g_bytes, part_bytes;
};

WPlan3 wplan3(const PPShape& sh) {
  WPlan3 p;
  // K x NT accumulators of 16 registers per wave; 32-out waves when O <= 128 so that
  // all four column waves have work
  p.nt = (sh.K <= 5 && sh.O > 128) ? 2 : 1;  // K = 6 with two tiles: the accumulators go to scratch
  p.opad = ((sh.O + 63) / 64) * 64;
  p.ocols = (sh.O + 128 * p.nt - 1) / (128 * p.nt);
  const int tiles = ((sh.I + 31) / 32) * p.ocols;
  int target = 256;
#ifdef PP_TUNING  // developer builds only
  if (const char* v = getenv("PP_BF3_BWDW_BLOCKS")) target = std::max(1, atoi(v));
#endif
  int nchunk = std::max(1, target / tiles);
  const int max_chunks = std::max(1, sh.B / 128);  // >= 8 steps of 16 rows per block
  nchunk = std::min(nchunk, max_chunks);
  int rpc = (sh.B + nchunk - 1) / nchunk;
  rpc = (rpc + 15) / 16 * 16;
  p.rows_per_chunk = rpc;
  p.nchunk = (sh.B + rpc - 1) / rpc;
  p.nbs = (sh.B + 15) / 16;  // (steps prefetched past the end read out of range: zeros, unused)
  p.nsb = (p.nbs + kSB - 1) / kSB;
  p.sc_bytes = (((size_t)p.nsb * p.opad * sizeof(float2)) + 255) / 256 * 256;
  p.g_bytes = (size_t)3 * p.nbs * p.opad * 32;  // (fp16x2 uses 2 of the 3 planes)
  p.part_bytes = p.nchunk > 1 ? (size_t)p.nchunk * sh.I * sh.K * sh.O * sizeof(float) : 0;
  return p;
}

template <class SC, int N, int K, int NT>
hipError_t launch_bwd_w_bf3(const float* x, const float* gy, const void* ws, float* out,
                            const PPShape& sh, const BasisTab& tab, const DenseCoord3& dc,
                            const WPlan3& p, float scale, hipStream_t st) {
  const size_t lds = (size_t)3 * SC::PL * K * 2 * 32 * 16;
  if (lds > 48 * 1024) {  // per call: the attribute is per device, and setting it is cheap
    hipError_t e = hipFuncSetAttribute((const void*)pp_bwd_w_bf3_kernel<SC, N, K, NT>,
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
  }
  dim3 grid(p.nchunk, (sh.I + 31) / 32, p.ocols);
  const char* base = static_cast<const char*>(ws);
  hipLaunchKernelGGL((pp_bwd_w_bf3_kernel<SC, N, K, NT>), grid, dim3(256), lds, st, x, gy,
                     reinterpret_cast<const float2*>(base), (const void*)(base + p.sc_bytes), out,
                     sh, tab, dc, p.rows_per_chunk, p.nbs, p.opad, scale);
  return hipGetLastError();
}

// LayerNorm backward folded into the pack pass (ranged scheme only): gy is then the gradient
// at the LayerNorm's output and g_out receives the gradient at the layer's output
struct LnFold {
  const float* yn;
  const float* rstd;
  const float* rowstats;
  int ntile;
  float* g_out;
};

template <class SC>
hipError_t launch_bwd_w_split(const float* x, const float* gy, float* gwt, void* ws,
                              const PPShape& sh, const BasisTab& tab, hipStream_t st,
                              const LnFold* ln = nullptr) {
  DenseCoord3 dc;
  dc.w2 = 2.0f / (float)sh.S;
  dc.inv_w = (float)sh.S / 2.0f;
  dc.inv_len = 1.0f / sh.length;
  const WPlan3 p = wplan3(sh);
  char* base = static_cast<char*>(ws);
  u32x4* G = reinterpret_cast<u32x4*>(base + p.sc_bytes);
  if constexpr (SC::kRanged) {
    if (ln) {
      hipLaunchKernelGGL(pp_split_pack_gy_ln_ranged_kernel<SC>, dim3(p.nsb, p.opad / kPackCols),
                         dim3(256), 0, st, gy, ln->yn, ln->rstd,
                         reinterpret_cast<const float2*>(ln->rowstats), ln->ntile, ln->g_out,
                         reinterpret_cast<float2*>(base), G, sh.B, sh.O, p.opad, p.nbs);
      gy = ln->g_out;  // what the main kernel's fp32 way out reads
    } else {
      hipLaunchKernelGGL(pp_split_pack_gy_ranged_kernel<SC>, dim3(p.nsb, p.opad / kPackCols), dim3(256), 0,
                         st, gy, reinterpret_cast<float2*>(base), G, sh.B, sh.O, p.opad, p.nbs);
    }
  } else {
    if (ln) return hipErrorInvalidValue;
    const long nthr = (long)p.nbs * p.opad;
    hipLaunchKernelGGL(pp_split_pack_gy_kernel<SC>, dim3((unsigned)((nthr + 255) / 256)), dim3(256),
                       0, st, gy, G, sh.B, sh.O, p.opad, p.nbs);
  }
  float* part = reinterpret_cast<float*>(base + p.sc_bytes + p.g_bytes);
  float* out = p.nchunk > 1 ? part : gwt;
  const float scale = p.nchunk > 1 ? 1.0f : sh.out_scale;
#define X(N_, K_)                                                                          \
  if (sh.n == N_ && sh.K == K_) {                                                          \
    hipError_t e = hipErrorInvalidValue;                                                   \
    if constexpr (K_ <= 5) {                                                               \
      if (p.nt == 2)                                                                       \
        e = launch_bwd_w_bf3<SC, N_, K_, 2>(x, gy, ws, out, sh, tab, dc, p, scale, st);    \
    }                                                                                      \
    if (p.nt == 1) e = launch_bwd_w_bf3<SC, N_, K_, 1>(x, gy, ws, out, sh, tab, dc, p, scale, st); \
    if (e != hipSuccess || p.nchunk <= 1) return e;                                        \
    pp_launch_chunk_reduce(part, gwt, (long)sh.I * sh.K * sh.O, p.nchunk, sh.out_scale, st); \
    return hipGetLastError();                                                              \
  }
  PP_BF3_COMBOS(X)
#undef X
  return hipErrorInvalidValue;
}
}  // namespace

bool pp_bf3_bwd_w_supported(const PPShape& sh) {
  if (!pp_dense_supported(sh)) return false;
  const WPlan3 p = wplan3(sh);
  return p.g_bytes < ((size_t)1 << 31);
}

size_t pp_bf3_bwd_w_workspace_bytes(const PPShape& sh) {
  const WPlan3 p = wplan3(sh);
  return p.sc_bytes + p.g_bytes + p.part_bytes;
}

hipError_t pp_launch_bwd_w_bf3(const float* x, const float* gy, float* gwt, void* ws,
                               const PPShape& sh, const BasisTab& tab, hipStream_t st) {
  return launch_bwd_w_split<SchemeBf3>(x, gy, gwt, ws, sh, tab, st);
}

hipError_t pp_launch_bwd_w_h2(const float* x, const float* gy, float* gwt, void* ws,
                              const PPShape& sh, const BasisTab& tab, hipStream_t st) {
  return launch_bwd_w_split<SchemeH2>(x, gy, gwt, ws, sh, tab, st);
}

hipError_t pp_launch_bwd_w_h2_ln(const float* x, const float* gy, const float* yn,
                                 const float* rstd, const float* rowstats, int ntile, float* g_out,
                                 float* gwt, void* ws, const PPShape& sh, const BasisTab& tab,
                                 hipStream_t st) {
  const LnFold ln{yn, rstd, rowstats, ntile, g_out};
  return launch_bwd_w_split<SchemeH2>(x, gy, gwt, ws, sh, tab, st, &ln);
}
