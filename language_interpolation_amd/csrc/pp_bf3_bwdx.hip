// Split-operand backward-input (schemes: pp_bf3_common.h; overview: pp_bf3_fwd.hip).
#include "pp_bf3_common.h"

namespace {

// ---------------------------------------------------------------------------
// backward-input on the same bf16x3 scheme:
//   T[b,(k,i)] = sum_o gy[b,o] * w_t[i,k,o]      (bf16 MFMA, both operands split 3-way)
//   gx[b,i]    = chain * sum_k dPhi[b,i,k] * T[b,(k,i)]          (fp32, in the epilogue)
// Block = 4 waves = 4 adjacent 32-input tiles of one tile of 32*RM rows; a wave keeps
// the K accumulators of its (rows x inputs) tile, so dPhi is evaluated once per pair.
// gy is split on the fly, one chunk of 128 outs at a time, into LDS ([plane][step]
// [half][row] x 16 B, double-buffered) and shared by the four waves; the weights come
// pre-split from Q[plane][ostep][k][Ipad][16 outs], all K fragments of the next
// 16-out step in flight under the 6*K*RM MFMAs of the current one.
// Ranged scheme (fp16x2): the weights carry the tensor's power-of-two scale (pack kernel),
// gy is scaled per (row, chunk) by a power of two taken from the chunk's row maximum
// (LDS max, one extra barrier per chunk); when a row's scale changes between chunks the
// wave rescales that row of its accumulators (exact), and the epilogue undoes both.
// Row statistics (rowstats != nullptr): gx is the gradient that reaches the LayerNorm below
// this layer, whose backward needs sum_i gx[b,i] and sum_i gx[b,i]*x[b,i] of every row (x of
// this layer IS that LayerNorm's output).  Both are at hand in the epilogue: every wave adds
// them up over its 32 inputs (DPP row sums, the two rows of 16 lanes through SGPRs) and
// writes rowstats[tile][b] = (s1, s2), tile = input / 32; the consumer
// (pp_split_pack_gy_ln_ranged_kernel, pp_bf3_bwdw.hip) adds the tiles in order.
// ---------------------------------------------------------------------------
template <class SC>
__global__ __launch_bounds__(256) void pp_split_pack_wo_kernel(const float* __restrict__ wt,
                                                               void* __restrict__ ws,
                                                               unsigned q_off, int I, int K,
                                                               int O, int nso, int Ipad) {
  pack_wo_body<SC>((long)blockIdx.x * 256 + threadIdx.x, wt, ws, q_off, I, K, O, nso, Ipad);
}

constexpr int kOC = 8;  // 16-out steps per gy chunk


template <class SC, int N, int K, int RM, int VW, bool STATS>
__device__ __forceinline__ void bwd_x_bf3_body(const float* __restrict__ x,
                                               const float* __restrict__ gy,
                                               const void* __restrict__ ws, float* __restrict__ gx,
                                               const PPShape& sh, const BasisTab& tab,
                                               const DenseCoord3& dc, int nso, int Ipad,
                                               unsigned q_off, u32x4* gyL,
                                               float* __restrict__ rowstats) {
  PP_TICK(0)
  using frag = typename SC::frag;
  constexpr int PL = SC::PL;
  constexpr int ROWS = 32 * RM;
  constexpr int kBufU = PL * kOC * 2 * ROWS;
  // ranged scheme: per (row, chunk parity) maximum of |gy| and the scale derived from it
  __shared__ unsigned rmaxL[2][ROWS];
  __shared__ float2 rscL[2][ROWS];  // (scale, 1/scale)
  if constexpr (SC::kRanged) {
    if (threadIdx.x < 2 * ROWS) (&rmaxL[0][0])[threadIdx.x] = 0u;
    __syncthreads();
  }
  constexpr int FR = ROWS * 2 * kOC / 256;  // gy fragments (8 outs of one row) per thread per chunk
  const int lane = threadIdx.x & 63;
  const int wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int r = lane & 31, h = lane >> 5;
  const int b0 = blockIdx.x * ROWS;
  const int i0 = (blockIdx.y * 4 + wv) * 32;  // < Ipad by construction of the grid
  const int nchunk = (nso + kOC - 1) / kOC;

  // producer: fragment f = tid + 256 m -> row f % ROWS, out group f / ROWS (8 outs)
  const rsrc_t rg = make_rsrc3(gy, (unsigned)sh.B * sh.O * 4u);
  int prow, pog[FR], pvoff[FR];
  prow = threadIdx.x % ROWS;
#pragma unroll
  for (int m = 0; m < FR; ++m) {
    pog[m] = (threadIdx.x + 256 * m) / ROWS;
    pvoff[m] = (min(b0 + prow, sh.B - 1) * sh.O + pog[m] * 8) * 4;
  }
  float stage[FR][8];
  auto gy_load = [&](int c) {
#pragma unroll
    for (int m = 0; m < FR; ++m) {
      if constexpr (VW == 4) {
#pragma unroll
        for (int e = 0; e < 8; e += 4) {
          const u32x4 t = __builtin_amdgcn_raw_buffer_load_b128(rg, pvoff[m] + 4 * e, c * (kOC * 64), 0);
          stage[m][e] = __uint_as_float(t.x); stage[m][e + 1] = __uint_as_float(t.y);
          stage[m][e + 2] = __uint_as_float(t.z); stage[m][e + 3] = __uint_as_float(t.w);
        }
      } else if constexpr (VW == 2) {
#pragma unroll
        for (int e = 0; e < 8; e += 2) {
          const u32x2 t = __builtin_amdgcn_raw_buffer_load_b64(rg, pvoff[m] + 4 * e, c * (kOC * 64), 0);
          stage[m][e] = __uint_as_float(t.x); stage[m][e + 1] = __uint_as_float(t.y);
        }
      } else {
#pragma unroll
        for (int e = 0; e < 8; ++e)
          stage[m][e] = __uint_as_float(
              __builtin_amdgcn_raw_buffer_load_b32(rg, pvoff[m] + 4 * e, c * (kOC * 64), 0));
      }
    }
  };
  // (every fragment of a thread belongs to the same row: 256 % ROWS == 0)
  auto gy_publish = [&](int c, int buf) {
#pragma unroll
    for (int m = 0; m < FR; ++m) {
      const int obase = c * (kOC * 16) + pog[m] * 8;
#pragma unroll
      for (int e = 0; e < 8; ++e)
        if (obase + e >= sh.O) stage[m][e] = 0.0f;  // outs past O belong to the next row
    }
    float rs = 1.0f;
    if constexpr (SC::kRanged) {
      unsigned mx = 0;
#pragma unroll
      for (int m = 0; m < FR; ++m)
#pragma unroll
        for (int e = 0; e < 8; ++e) mx = max(mx, __float_as_uint(stage[m][e]) & 0x7fffffffu);
      atomicMax(&rmaxL[buf][prow], mx);
      __syncthreads();
      float inv;
      h2_scale(rmaxL[buf][prow], rs, inv);
      if (threadIdx.x < ROWS) {
        rscL[buf][prow] = make_float2(rs, inv);
        // the next chunk starts from this running maximum (its slot was consumed one chunk
        // ago and is next written after a barrier): a row's scale never grows from chunk to
        // chunk, so the accumulator rescale factor below is <= 1 and cannot overflow
        rmaxL[buf ^ 1][prow] = rmaxL[buf][prow];
      }
    }
#pragma unroll
    for (int m = 0; m < FR; ++m) {
      unsigned w[4][PL];
#pragma unroll
      for (int e = 0; e < 4; ++e)
        SC::split_pair(stage[m][2 * e] * rs, stage[m][2 * e + 1] * rs, w[e]);
      const int t = pog[m] >> 1, hh = pog[m] & 1;
#pragma unroll
      for (int p = 0; p < PL; ++p)
        gyL[buf * kBufU + ((p * kOC + t) * 2 + hh) * ROWS + prow] =
            u32x4{w[0][p], w[1][p], w[2][p], w[3][p]};
    }
  };

  // consumer
  const rsrc_t rq = make_rsrc3(static_cast<const char*>(ws) + q_off,
                               (unsigned)((unsigned)PL * nso * K * Ipad * 32u));
  const int vq_off = ((i0 + r) * 2 + h) * 16;
  const int k_bytes = Ipad * 32;          // one (plane, ostep, k) slab
  const int t_bytes = K * k_bytes;        // one (plane, ostep)
  const int plane_bytes = nso * t_bytes;
  auto wload = [&](int t, frag (&w)[K][PL]) {
#pragma unroll
    for (int k = 0; k < K; ++k)
#pragma unroll
      for (int p = 0; p < PL; ++p)
        w[k][p] = __builtin_bit_cast(
            frag, __builtin_amdgcn_raw_buffer_load_b128(
                      rq, vq_off,
#ifdef PP_W_SAME  // developer experiment: every W load hits the same lines (L1-resident)
                      p * plane_bytes + 0 * (t + k),
#else
                      p * plane_bytes + t * t_bytes + k * k_bytes,
#endif
                      0));
  };
  auto aload = [&](int buf, int tt, frag (&a)[RM][PL]) {
#pragma unroll
    for (int rm = 0; rm < RM; ++rm)
#pragma unroll
      for (int p = 0; p < PL; ++p)
        a[rm][p] = __builtin_bit_cast(
            frag, gyL[buf * kBufU + ((p * kOC + tt) * 2 + h) * ROWS + rm * 32 + r]);
  };

  f32x16 acc[K][RM];
#pragma unroll
  for (int k = 0; k < K; ++k)
#pragma unroll
    for (int rm = 0; rm < RM; ++rm)
#pragma unroll
      for (int q = 0; q < 16; ++q) acc[k][rm][q] = 0.0f;
  auto mm = [&](const frag (&a)[RM][PL], const frag (&w)[K][PL]) {
#pragma unroll
    for (int k = 0; k < K; ++k)
#pragma unroll
      for (int c = 0; c < SC::NC; ++c)
#pragma unroll
        for (int rm = 0; rm < RM; ++rm)
          acc[k][rm] = SC::mfma(a[rm][SC::pa(c)], w[k][SC::pb(c)], acc[k][rm]);
  };

  // x of this lane's (row, input) pairs for the epilogue: issued at the start of the
  // LAST chunk (when the gy staging registers are idle), consumed after the loop
  const int i = i0 + r;
  const rsrc_t rx = make_rsrc3(x, (unsigned)sh.B * sh.I * 4u);
  const int icol = min(i, sh.I - 1);
  float xv[RM][16];
  auto x_load = [&]() {
#pragma unroll
    for (int rm = 0; rm < RM; ++rm)
#pragma unroll
      for (int q = 0; q < 16; ++q) {
        const int row = min(b0 + rm * 32 + (q & 3) + 8 * (q >> 2) + 4 * h, sh.B - 1);
        xv[rm][q] = __uint_as_float(
            __builtin_amdgcn_raw_buffer_load_b32(rx, (row * sh.I + icol) * 4, 0, 0));
      }
  };
  // W fragment sets: D steps of prefetch distance.  (D = 2 was tried for the fp16 scheme,
  // whose 16-out step is shorter than an L2 round trip: 4 sets + 160 accumulators spill.)
  constexpr int D = 1;
  constexpr int NWS = 2 * D;
  frag W[NWS][K][PL], A[2][RM][PL];
  // the packed weights' scale: read here (scalar loads, one register), used by the epilogue
  float wsinv = 1.0f;
  if constexpr (SC::kRanged) {
    float sc;
    h2_scale(absmax_of_parts(ws), sc, wsinv);
  }
  gy_load(0);
#pragma unroll
  for (int u = 0; u < D; ++u) wload(u, W[u]);
  PP_TICK(1)
  gy_publish(0, 0);
  __syncthreads();
  PP_TICK(2)
  for (int c = 0; c < nchunk; ++c) {
    const int buf = c & 1;
    const bool more = c + 1 < nchunk;  // block-uniform
    if (more) gy_load(c + 1);
    else x_load();
    const int nt = min(kOC, nso - c * kOC);  // even: nso is
    if constexpr (SC::kRanged) {
      if (c > 0) {  // bring the accumulators to this chunk's row scales (powers of two: exact)
#pragma unroll
        for (int rm = 0; rm < RM; ++rm)
#pragma unroll
          for (int q = 0; q < 16; ++q) {
            const int rr = rm * 32 + (q & 3) + 8 * (q >> 2) + 4 * h;
            const float f = rscL[buf][rr].x * rscL[buf ^ 1][rr].y;
#pragma unroll
            for (int k = 0; k < K; ++k) acc[k][rm][q] *= f;
          }
      }
    }
    aload(buf, 0, A[0]);
    // one step: W of step t+D and A of step t+1 are issued, then the MFMAs of step t.
    // sched_barrier keeps the loads ahead of the MFMAs (the scheduler otherwise sinks them
    // to their first use to save registers).  kOC % NWS == 0: set index = tt % NWS.
    auto step = [&](int tt, auto uc) {
      constexpr int U = decltype(uc)::value;
#ifdef PP_STEP_TICKS
      PP_STEP_TICK(c * kOC + tt)
#endif
#ifndef PP_NO_WLOAD
      wload(c * kOC + tt + D, W[(U + D) % NWS]);
#endif
#ifndef PP_NO_ALOAD
      aload(buf, min(tt + 1, kOC - 1), A[(U + 1) & 1]);  // (past the chunk: a harmless re-read)
#endif
      __builtin_amdgcn_sched_barrier(0);
      mm(A[U & 1], W[U]);
      __builtin_amdgcn_sched_barrier(0);
    };
    for (int tt = 0; tt < nt; tt += NWS) {
      step(tt, std::integral_constant<int, 0>{});
      step(tt + 1, std::integral_constant<int, 1>{});
      if constexpr (NWS == 4) {
        if (tt + 2 < nt) {  // nt is even
          step(tt + 2, std::integral_constant<int, 2>{});
          step(tt + 3, std::integral_constant<int, 3>{});
        }
      }
    }
    PP_TICK(3 + 2 * (c & 1))
    if (more) gy_publish(c + 1, buf ^ 1);
    __syncthreads();
    PP_TICK(4 + 2 * (c & 1))
  }
  const int lastbuf = (nchunk - 1) & 1;

  // epilogue: lane owns input column i0 + r for 16*RM rows (with the row sums every lane
  // goes through it: DPP and readlane want the whole wave, the padded inputs add zeros)
  const bool own = i < sh.I;
  if (STATS || own) {
    const bool stats_on = STATS && i0 < Ipad;  // wave-uniform
    const float chain = (sh.length * dc.inv_w) * sh.out_scale;  // dxi/dx = L/(2/S)
    constexpr int S = Geom3<N, K>::S, stride = Geom3<N, K>::stride;
#pragma unroll
    for (int rm = 0; rm < RM; ++rm) {
      float st1 = 0.0f, st2 = 0.0f;  // lane r < 16: the sums of row slot q = r of this half
#pragma unroll
      for (int q = 0; q < 16; ++q) {
        const int row = b0 + rm * 32 + (q & 3) + 8 * (q >> 2) + 4 * h;
        const float xq = xv[rm][q];
        const float t = ((xq + sh.half) * dc.inv_len) * sh.Sf;
        const float tc = __builtin_amdgcn_fmed3f(t, -1.0f, sh.Sf);
        int si = (int)tc;
        si = min(max(si, 0), sh.S - 1);
        const int s = (__builtin_fabsf(t) < 9.2e18f) ? si : 0;
        const float x_min = (float)s * dc.w2 - 1.0f;
        const float xi = sh.length * ((xq - x_min) * dc.inv_w) - sh.half;
        float dl[N];
        pp_lagrange_dl<N>(xi, tab, dl);
        // sum_j dl_j * T[s*stride + j]: pick the segment's N accumulators, then N fmas
        float gsum = 0.0f;
#pragma unroll
        for (int j = 0; j < N; ++j) {
          float a = acc[j][rm][q];
#pragma unroll
          for (int sq = 1; sq < S; ++sq) a = (s == sq) ? acc[sq * stride + j][rm][q] : a;
          gsum = fmaf(dl[j], a, gsum);
        }
        if constexpr (SC::kRanged)
          gsum *= rscL[lastbuf][rm * 32 + (q & 3) + 8 * (q >> 2) + 4 * h].y * wsinv;
        const float g = gsum * chain;
        if (own && row < sh.B) gx[(size_t)row * sh.I + i] = g;
        if constexpr (STATS) if (stats_on) {  // (all 64 lanes are here: the padded inputs add zeros)
          float a = own ? g : 0.0f, b = own ? g * xq : 0.0f;
          a = pp_dpp_add<0x128>(a); b = pp_dpp_add<0x128>(b);  // row_ror:8, 4, 2, 1
          a = pp_dpp_add<0x124>(a); b = pp_dpp_add<0x124>(b);
          a = pp_dpp_add<0x122>(a); b = pp_dpp_add<0x122>(b);
          a = pp_dpp_add<0x121>(a); b = pp_dpp_add<0x121>(b);
          const float a0 = pp_readlane_f(a, 0) + pp_readlane_f(a, 16);
          const float a1 = pp_readlane_f(a, 32) + pp_readlane_f(a, 48);
          const float b0s = pp_readlane_f(b, 0) + pp_readlane_f(b, 16);
          const float b1s = pp_readlane_f(b, 32) + pp_readlane_f(b, 48);
          if (r == q) {
            st1 = h ? a1 : a0;
            st2 = h ? b1s : b0s;
          }
        }
      }
      if constexpr (STATS) if (stats_on && r < 16) {
        const int row = b0 + rm * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
        if (row < sh.B)
          reinterpret_cast<float2*>(rowstats)[(size_t)(i0 >> 5) * sh.B + row] = make_float2(st1, st2);
      }
    }
  }
  PP_TICK(7)
}

template <class SC, int N, int K, int RM, bool STATS>
__global__ __launch_bounds__(256, (RM == 1 && K <= 5) ? 2 : 1) void pp_bwd_x_bf3_kernel(const float* __restrict__ x,
                                                           const float* __restrict__ gy,
                                                           const void* __restrict__ Q,
                                                           float* __restrict__ gx, PPShape sh,
                                                           BasisTab tab, DenseCoord3 dc, int nso,
                                                           int Ipad, int vw, unsigned q_off,
                                                           float* __restrict__ rowstats) {
  extern __shared__ __attribute__((aligned(16))) u32x4 gyL[];  // [2][PL][kOC][2][32*RM] x 16 B
  // vw: widest aligned vector load of a gy row (block-uniform), fixed per body so the
  // staging loads stay straight-line code
  if (vw == 4) bwd_x_bf3_body<SC, N, K, RM, 4, STATS>(x, gy, Q, gx, sh, tab, dc, nso, Ipad, q_off, gyL, rowstats);
  else if (vw == 2) bwd_x_bf3_body<SC, N, K, RM, 2, STATS>(x, gy, Q, gx, sh, tab, dc, nso, Ipad, q_off, gyL, rowstats);
  else bwd_x_bf3_body<SC, N, K, RM, 1, STATS>(x, gy, Q, gx, sh, tab, dc, nso, Ipad, q_off, gyL, rowstats);
}

}  // namespace

// ---- backward-input ----
namespace {
int nso_of(const PPShape& sh) { return ((sh.O + 31) / 32) * 2; }   // 16-out steps, even
int ipad_of(const PPShape& sh) { return ((sh.I + 31) / 32) * 32; }

template <class SC, int N, int K, int RM, bool STATS>
hipError_t launch_bwd_x_bf3_k(const float* x, const float* gy, const void* ws, float* gx,
                              const PPShape& sh, const BasisTab& tab, const DenseCoord3& dc,
                              unsigned q_off, float* rowstats, hipStream_t st) {
  const size_t lds = (size_t)2 * SC::PL * kOC * 2 * 32 * RM * 16;
  if (lds > 40 * 1024) {  // per call: the attribute is per device, and setting it is cheap
    hipError_t e = hipFuncSetAttribute((const void*)pp_bwd_x_bf3_kernel<SC, N, K, RM, STATS>,
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
  }
  const int vw = sh.O % 4 == 0 ? 4 : (sh.O % 2 == 0 ? 2 : 1);
  dim3 grid((sh.B + 32 * RM - 1) / (32 * RM), (ipad_of(sh) + 127) / 128);
  hipLaunchKernelGGL((pp_bwd_x_bf3_kernel<SC, N, K, RM, STATS>), grid, dim3(256), lds, st, x, gy,
                     ws, gx, sh, tab, dc, nso_of(sh), ipad_of(sh), vw, q_off, rowstats);
  return hipGetLastError();
}

// The row-sum epilogue exists for the kernels PP_METHOD_AUTO runs (fp16x2, 32-row tiles); it is
// a template switch so that the other instances keep their register budget.
template <class SC, int N, int K, int RM>
hipError_t launch_bwd_x_bf3(const float* x, const float* gy, const void* ws, float* gx,
                            const PPShape& sh, const BasisTab& tab, const DenseCoord3& dc,
                            unsigned q_off, float* rowstats, hipStream_t st) {
  if (rowstats) {
    if constexpr (SC::kRanged && RM == 1)
      return launch_bwd_x_bf3_k<SC, N, K, RM, true>(x, gy, ws, gx, sh, tab, dc, q_off, rowstats, st);
    else
      return hipErrorInvalidValue;
  }
  return launch_bwd_x_bf3_k<SC, N, K, RM, false>(x, gy, ws, gx, sh, tab, dc, q_off, nullptr, st);
}

template <class SC>
void pack_wo_split(const float* wt, void* ws, unsigned q_off, const PPShape& sh, bool with_absmax,
                   hipStream_t st) {
  const int nso = nso_of(sh), Ipad = ipad_of(sh);
  if (SC::kRanged && with_absmax)
    hipLaunchKernelGGL(pp_absmax_kernel, dim3(kAbsmaxBlocks), dim3(256), 0, st, wt,
                       (long)sh.I * sh.K * sh.O, static_cast<unsigned*>(ws));
  const long pslots = (long)nso * sh.K * Ipad;
  hipLaunchKernelGGL(pp_split_pack_wo_kernel<SC>, dim3((unsigned)((pslots + 255) / 256)), dim3(256),
                     0, st, wt, ws, q_off, sh.I, sh.K, sh.O, nso, Ipad);
}

// q_off: byte offset of Q in ws (kWsHeader for a per-call workspace; behind the forward
// layout in a buffer written by pp_pack_weights)
template <class SC>
hipError_t launch_bwd_x_split(const float* x, const float* gy, const float* wt, float* gx,
                              void* ws, const PPShape& sh, const BasisTab& tab, bool prepacked,
                              unsigned q_off, float* rowstats, hipStream_t st) {
  DenseCoord3 dc;
  dc.w2 = 2.0f / (float)sh.S;
  dc.inv_w = (float)sh.S / 2.0f;
  dc.inv_len = 1.0f / sh.length;
  if (!prepacked) pack_wo_split<SC>(wt, ws, q_off, sh, true, st);
  // 32-row tiles: with K <= 5 they fit 256 registers (__launch_bounds__(256, 2) on the kernel),
  // so two workgroups share a CU and hide each other's round trips -- 44 vs 53 us at 16384
  // rows, 15 vs 24 at 4096 on the cfg2 hidden layer against the 64-row tiles, which hold
  // 256 + 256 registers and run one workgroup per CU (they remain for the developer override)
  int rm = 1;
#ifdef PP_TUNING  // developer builds only
  if (const char* v = getenv("PP_BF3_BWDX_RM")) {
    const int c = atoi(v);
    if ((c == 1 || c == 2) && (c == 1 || sh.K <= 6)) rm = c;
  }
#endif
#define X(N_, K_)                                                                          \
  if (sh.n == N_ && sh.K == K_) {                                                          \
    if constexpr (K_ <= 5) {                                                               \
      if (rm == 2) return launch_bwd_x_bf3<SC, N_, K_, 2>(x, gy, ws, gx, sh, tab, dc, q_off, rowstats, st); \
    }                                                                                      \
    return launch_bwd_x_bf3<SC, N_, K_, 1>(x, gy, ws, gx, sh, tab, dc, q_off, rowstats, st);                \
  }
  PP_BF3_COMBOS(X)
#undef X
  return hipErrorInvalidValue;
}
}  // namespace

bool pp_bf3_bwd_x_supported(const PPShape& sh) {
  if (!pp_dense_supported(sh)) return false;
  return (long)3 * nso_of(sh) * sh.K * ipad_of(sh) * 32 < (1L << 31);
}

size_t pp_bf3_bwd_x_workspace_bytes(const PPShape& sh) {
  return kWsHeader + (size_t)3 * nso_of(sh) * sh.K * ipad_of(sh) * 32;  // (fp16x2: 2 of 3 planes)
}

hipError_t pp_launch_bwd_x_bf3(const float* x, const float* gy, const float* wt, float* gx,
                               void* ws, const PPShape& sh, const BasisTab& tab, hipStream_t st) {
  return launch_bwd_x_split<SchemeBf3>(x, gy, wt, gx, ws, sh, tab, false, (unsigned)kWsHeader,
                                       nullptr, st);
}

hipError_t pp_launch_bwd_x_h2(const float* x, const float* gy, const float* wt, float* gx,
                              void* ws, const PPShape& sh, const BasisTab& tab, bool prepacked,
                              size_t q_off, float* rowstats, hipStream_t st) {
  return launch_bwd_x_split<SchemeH2>(x, gy, wt, gx, ws, sh, tab, prepacked, (unsigned)q_off,
                                      rowstats, st);
}

// tiles of 32 inputs the row statistics are written in ([tiles][B] float2)
int pp_bwd_x_rowstats_tiles(const PPShape& sh) { return ipad_of(sh) / 32; }

size_t pp_h2_bwd_x_packed_bytes(const PPShape& sh) {
  return (size_t)2 * nso_of(sh) * sh.K * ipad_of(sh) * 32;
}
