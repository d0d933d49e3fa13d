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
// forward  y[b,o] = sum_{i,k} Phi[b,i,k] * w_t[i,k,o]
//   k-dimension ordered (k, i): one MFMA k-step = 16 consecutive inputs at a fixed
//   weight index k, so a lane evaluates the basis of an input once and uses its K
//   columns in K different k-steps.  Block = 4 waves = 4 column tiles (64 outs each)
//   of one 32-row tile; per step of 16 inputs every wave evaluates and splits the
//   basis of 2 inputs per lane and publishes it in LDS ([plane][k][half][row] x 16 B,
//   conflict-free ds_read_b128); weights come pre-split from a packed buffer
//   P[plane][k][step][out][16 inputs] (coalesced 16-byte loads), written once per
//   call by pp_bf3_pack_w_kernel.
#include <algorithm>
#include <type_traits>

#include "pp_device.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
using rsrc_t = __amdgpu_buffer_rsrc_t;

struct DenseCoord3 {
  float w2, inv_w, inv_len;
};

__device__ __forceinline__ rsrc_t make_rsrc3(const void* p, unsigned bytes) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, bytes, 0x00020000);
}

// a = p1 + p2 + p3 exactly (up to 2^-24 |a|); each piece has <= 8 significant bits
// (its low 16 bits are zero), returned as raw fp32 bit patterns.
__device__ __forceinline__ void split3(float a, unsigned& p1, unsigned& p2, unsigned& p3) {
  p1 = __float_as_uint(a) & 0xffff0000u;
  const float r1 = a - __uint_as_float(p1);
  p2 = __float_as_uint(r1) & 0xffff0000u;
  const float r2 = r1 - __uint_as_float(p2);
  p3 = __float_as_uint(r2) & 0xffff0000u;
}

template <int N, int K>
struct Geom3 {
  static constexpr bool disc = (K % N == 0) && (((K / N) & (K / N - 1)) == 0);
  static constexpr int stride = disc ? N : N - 1;
  static constexpr int S = disc ? K / N : (N > 1 ? (K - 1) / (N - 1) : K);
};

template <int N, int K>
__device__ __forceinline__ void phi3(float x, bool valid, const PPShape& sh, const BasisTab& tab,
                                     const DenseCoord3& dc, float (&phi)[K]) {
  const float t = ((x + sh.half) * dc.inv_len) * sh.Sf;
  const float tc = __builtin_amdgcn_fmed3f(t, -1.0f, sh.Sf);
  int si = (int)tc;
  si = min(max(si, 0), sh.S - 1);
  const int s = (__builtin_fabsf(t) < 9.2e18f) ? si : 0;
  const float x_min = (float)s * dc.w2 - 1.0f;
  const float xi = sh.length * ((x - x_min) * dc.inv_w) - sh.half;
  float l[N];
  pp_lagrange<N>(xi, tab, l);
  constexpr int S = Geom3<N, K>::S, stride = Geom3<N, K>::stride;
  bool is[S];
#pragma unroll
  for (int q = 0; q < S; ++q) is[q] = valid && (s == q);
#pragma unroll
  for (int k = 0; k < K; ++k) {
    float v = 0.0f;
#pragma unroll
    for (int q = 0; q < S; ++q) {
      const int j = k - q * stride;
      if (j >= 0 && j < N) v = is[q] ? l[j] : v;
    }
    phi[k] = v;
  }
}

// 8 fp32 pieces (low 16 bits zero) -> one 16-byte fragment of 8 bf16
__device__ __forceinline__ u32x4 pack8(const unsigned* pc) {
  u32x4 w;
  w.x = (pc[0] >> 16) | pc[1];
  w.y = (pc[2] >> 16) | pc[3];
  w.z = (pc[4] >> 16) | pc[5];
  w.w = (pc[6] >> 16) | pc[7];
  return w;
}

// P[plane][k][step][out][16]: bf16 pieces of w_t[i = 16*step + j][k][out].  One thread
// per (k, step, out): 16 reads coalesced over `out`, two 16-byte stores per plane.
__global__ __launch_bounds__(256) void pp_bf3_pack_w_kernel(const float* __restrict__ wt,
                                                            u32x4* __restrict__ P, int I, int K,
                                                            int O, int nstep) {
  const long id = (long)blockIdx.x * 256 + threadIdx.x;
  const long per_plane = (long)K * nstep * O;  // in (k, step, out) slots of 32 bytes
  if (id >= per_plane) return;
  const int o = (int)(id % O);
  const int s = (int)((id / O) % nstep);
  const int k = (int)(id / ((long)O * nstep));
  unsigned pc[3][16];
#pragma unroll
  for (int j = 0; j < 16; ++j) {
    const int i = 16 * s + j;
    const float v = i < I ? wt[((size_t)i * K + k) * O + o] : 0.0f;
    split3(v, pc[0][j], pc[1][j], pc[2][j]);
  }
#pragma unroll
  for (int p = 0; p < 3; ++p) {
    P[(p * per_plane + id) * 2] = pack8(&pc[p][0]);
    P[(p * per_plane + id) * 2 + 1] = pack8(&pc[p][8]);
  }
}

template <int N, int K, int RM, int G>
__global__ __launch_bounds__(256 * G) void pp_fwd_bf3_kernel(const float* __restrict__ x,
                                                             const void* __restrict__ P,
                                                             float* __restrict__ y, PPShape sh,
                                                             BasisTab tab, DenseCoord3 dc,
                                                             int nstep) {
  // block = G groups of 4 waves.  A group = 4 column tiles (64 outs each) x one tile of
  // 32*RM rows; every W fragment a wave loads feeds RM row tiles.  With G = 2 the two
  // groups take alternate 16-input steps of the SAME output tile (split-k inside the
  // block, two waves per SIMD: one group's waits and basis evaluation sit under the
  // other's MFMAs) and add their accumulators through LDS at the end.  nstep is a
  // multiple of G (the packed weights are zero-padded to it).
  extern __shared__ __attribute__((aligned(16))) u32x4 phiL[];  // [G][2][3][K][2][32*RM] x 16 B
  constexpr int ROWS = 32 * RM;
  constexpr int kBufU = 3 * K * 2 * ROWS;
  const int lane = threadIdx.x & 63;
  // wave-uniform ids in SGPRs (a VGPR step offset would turn every buffer load into a
  // readfirstlane loop)
  const int wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int cw = wv & 3;
  const int g = G > 1 ? wv >> 2 : 0;
  const int r = lane & 31, h = lane >> 5;
  const int b0 = blockIdx.x * ROWS;
  const int o0 = (blockIdx.y * 4 + cw) * 64;
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
  const rsrc_t rp = make_rsrc3(P, (unsigned)(3u * K * nstep * sh.O * 32u));
  int vcol[2];
#pragma unroll
  for (int nt = 0; nt < 2; ++nt) vcol[nt] = (min(o0 + nt * 32 + r, sh.O - 1) * 2 + h) * 16;
  const int step_bytes = sh.O * 32;        // one (plane, k, step) slab
  const int k_bytes = nstep * step_bytes;  // one (plane, k)
  const int plane_bytes = K * k_bytes;

  f32x16 acc[RM][2];
#pragma unroll
  for (int rm = 0; rm < RM; ++rm)
#pragma unroll
    for (int nt = 0; nt < 2; ++nt)
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
  auto produce = [&](int s, int buf, const float (&xv)[RM][2]) {
    const int i0 = 16 * s + 8 * hb + ipos;
#pragma unroll
    for (int rm = 0; rm < RM; ++rm) {
      float pa[K], pb[K];
      phi3<N, K>(xv[rm][0], i0 < sh.I, sh, tab, dc, pa);
      phi3<N, K>(xv[rm][1], i0 + 1 < sh.I, sh, tab, dc, pb);
#pragma unroll
      for (int k = 0; k < K; ++k) {
        unsigned a1, a2, a3, c1, c2, c3;
        split3(pa[k], a1, a2, a3);
        split3(pb[k], c1, c2, c3);
        const int base = (buf * kBufU + (k * 2 + hb) * ROWS + rm * 32 + r) * 4 + (ipos >> 1);
        phiW[base] = (a1 >> 16) | c1;
        phiW[base + K * 2 * ROWS * 4] = (a2 >> 16) | c2;
        phiW[base + 2 * K * 2 * ROWS * 4] = (a3 >> 16) | c3;
      }
    }
  };
  auto wload = [&](int s, int k, bf16x8 (&w)[3][2]) {
#pragma unroll
    for (int p = 0; p < 3; ++p)
#pragma unroll
      for (int nt = 0; nt < 2; ++nt)
        w[p][nt] = __builtin_bit_cast(
            bf16x8, __builtin_amdgcn_raw_buffer_load_b128(
                        rp, vcol[nt], p * plane_bytes + k * k_bytes + s * step_bytes, 0));
  };
  auto aload = [&](int buf, int k, bf16x8 (&a)[RM][3]) {
#pragma unroll
    for (int rm = 0; rm < RM; ++rm)
#pragma unroll
      for (int p = 0; p < 3; ++p)
        a[rm][p] = __builtin_bit_cast(
            bf16x8, phiG[buf * kBufU + ((p * K + k) * 2 + h) * ROWS + rm * 32 + r]);
  };
  auto mm = [&](const bf16x8 (&a)[RM][3], const bf16x8 (&w)[3][2]) {
    // the six leading cross terms, small ones first, a1*b1 last
    constexpr int PA[6] = {2, 0, 1, 1, 0, 0};
    constexpr int PB[6] = {0, 2, 1, 0, 1, 0};
#pragma unroll
    for (int c = 0; c < 6; ++c)
#pragma unroll
      for (int rm = 0; rm < RM; ++rm)
#pragma unroll
        for (int nt = 0; nt < 2; ++nt)
          acc[rm][nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[rm][PA[c]], w[PB[c]][nt],
                                                                acc[rm][nt], 0, 0, 0);
  };

  const int T = nstep / G;
  float xv[RM][2], xn[RM][2];
  xload(g, xv);
  produce(g, 0, xv);
  xload(G + g, xv);
  // two fragment sets in ping-pong over the flattened (step, k) sequence: the W loads
  // of k-step q+1 are in flight, and its A fragments are read, under the MFMAs of q
  bf16x8 W[2][3][2], A[2][RM][3];
  wload(g, 0, W[0]);
  __builtin_amdgcn_s_waitcnt(0x0F70);
  __syncthreads();
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
      if (k + 1 < K) {
        wload(s, k + 1, W[nxt]);
        aload(buf, k + 1, A[nxt]);
      } else {
        wload(s + G, 0, W[nxt]);
      }
      mm(A[cur], W[cur]);
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

  auto store_nt = [&](auto ntc, const float* part) {
    constexpr int nt = decltype(ntc)::value;
    const int col = o0 + nt * 32 + r;
#pragma unroll
    for (int rm = 0; rm < RM; ++rm)
#pragma unroll
      for (int q = 0; q < 16; ++q) {
        const int row = b0 + rm * 32 + (q & 3) + 8 * (q >> 2) + 4 * h;
        float v = acc[rm][nt][q];
        if (part) v += part[(rm * 16 + q) * 64];
        if (col < sh.O && row < sh.B) y[(size_t)row * sh.O + col] = v * sh.out_scale;
      }
  };
  if (G == 1) {
    store_nt(std::integral_constant<int, 0>{}, nullptr);
    store_nt(std::integral_constant<int, 1>{}, nullptr);
  } else {
    // group g finishes column half nt = g of its wave's tile: hand the other half to
    // the partner wave (same cw, other group) through LDS (the basis buffers are free:
    // the loop ended on a barrier)
    float* red = reinterpret_cast<float*>(phiL);  // [g][cw][RM*16][64]
    float* mine = red + ((g * 4 + cw) * RM * 16) * 64 + lane;
    const float* theirs = red + (((g ^ 1) * 4 + cw) * RM * 16) * 64 + lane;
    auto give = [&](auto ntc) {
      constexpr int nt = decltype(ntc)::value;
#pragma unroll
      for (int rm = 0; rm < RM; ++rm)
#pragma unroll
        for (int q = 0; q < 16; ++q) mine[(rm * 16 + q) * 64] = acc[rm][nt][q];
    };
    if (g == 0) give(std::integral_constant<int, 1>{});
    else give(std::integral_constant<int, 0>{});
    __syncthreads();
    if (g == 0) store_nt(std::integral_constant<int, 0>{}, theirs);
    else store_nt(std::integral_constant<int, 1>{}, theirs);
  }
}

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
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void pp_bf3_pack_wo_kernel(const float* __restrict__ wt,
                                                             u32x4* __restrict__ Q, int I, int K,
                                                             int O, int nso, int Ipad) {
  // one thread per (i, k, ostep), ostep fastest: consecutive threads read consecutive
  // 64-byte pieces of one weight row
  const long id = (long)blockIdx.x * 256 + threadIdx.x;
  if (id >= (long)Ipad * K * nso) return;
  const int t = (int)(id % nso);
  const int k = (int)((id / nso) % K);
  const int i = (int)(id / ((long)nso * K));
  unsigned pc[3][16];
#pragma unroll
  for (int j = 0; j < 16; ++j) {
    const int o = 16 * t + j;
    const float v = (i < I && o < O) ? wt[((size_t)i * K + k) * O + o] : 0.0f;
    split3(v, pc[0][j], pc[1][j], pc[2][j]);
  }
  const long per_plane = (long)nso * K * Ipad;  // slots of 32 bytes
  const long slot = ((long)t * K + k) * Ipad + i;
#pragma unroll
  for (int p = 0; p < 3; ++p) {
    Q[(p * per_plane + slot) * 2] = pack8(&pc[p][0]);
    Q[(p * per_plane + slot) * 2 + 1] = pack8(&pc[p][8]);
  }
}

constexpr int kOC = 8;  // 16-out steps per gy chunk

#ifdef PP_TIMING
__device__ unsigned long long pp_dbg_times[8 * 8192];
#define PP_TICK(slot)                                                                   \
  if (threadIdx.x == 0 && blockIdx.y == 0 && blockIdx.x < 8192)                         \
    pp_dbg_times[blockIdx.x * 8 + (slot)] = wall_clock64();
#else
#define PP_TICK(slot)
#endif

template <int N, int K, int RM, int VW>
__device__ __forceinline__ void bwd_x_bf3_body(const float* __restrict__ x,
                                               const float* __restrict__ gy,
                                               const void* __restrict__ Q, float* __restrict__ gx,
                                               const PPShape& sh, const BasisTab& tab,
                                               const DenseCoord3& dc, int nso, int Ipad,
                                               u32x4* gyL) {
  PP_TICK(0)
  constexpr int ROWS = 32 * RM;
  constexpr int kBufU = 3 * kOC * 2 * ROWS;
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
  auto gy_publish = [&](int c, int buf) {
#pragma unroll
    for (int m = 0; m < FR; ++m) {
      const int obase = c * (kOC * 16) + pog[m] * 8;
      unsigned pc[3][8];
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const float v = obase + e < sh.O ? stage[m][e] : 0.0f;  // outs past O belong to the next row
        split3(v, pc[0][e], pc[1][e], pc[2][e]);
      }
      const int t = pog[m] >> 1, hh = pog[m] & 1;
#pragma unroll
      for (int p = 0; p < 3; ++p) {
        u32x4 w;
        w.x = (pc[p][0] >> 16) | pc[p][1];
        w.y = (pc[p][2] >> 16) | pc[p][3];
        w.z = (pc[p][4] >> 16) | pc[p][5];
        w.w = (pc[p][6] >> 16) | pc[p][7];
        gyL[buf * kBufU + ((p * kOC + t) * 2 + hh) * ROWS + prow] = w;
      }
    }
  };

  // consumer
  const rsrc_t rq = make_rsrc3(Q, (unsigned)(3u * nso * K * Ipad * 32u));
  const int vq_off = ((i0 + r) * 2 + h) * 16;
  const int k_bytes = Ipad * 32;          // one (plane, ostep, k) slab
  const int t_bytes = K * k_bytes;        // one (plane, ostep)
  const int plane_bytes = nso * t_bytes;
  auto wload = [&](int t, bf16x8 (&w)[K][3]) {
#pragma unroll
    for (int k = 0; k < K; ++k)
#pragma unroll
      for (int p = 0; p < 3; ++p)
        w[k][p] = __builtin_bit_cast(
            bf16x8, __builtin_amdgcn_raw_buffer_load_b128(
                        rq, vq_off, p * plane_bytes + t * t_bytes + k * k_bytes, 0));
  };
  auto aload = [&](int buf, int tt, bf16x8 (&a)[RM][3]) {
#pragma unroll
    for (int rm = 0; rm < RM; ++rm)
#pragma unroll
      for (int p = 0; p < 3; ++p)
        a[rm][p] = __builtin_bit_cast(
            bf16x8, gyL[buf * kBufU + ((p * kOC + tt) * 2 + h) * ROWS + rm * 32 + r]);
  };

  f32x16 acc[K][RM];
#pragma unroll
  for (int k = 0; k < K; ++k)
#pragma unroll
    for (int rm = 0; rm < RM; ++rm)
#pragma unroll
      for (int q = 0; q < 16; ++q) acc[k][rm][q] = 0.0f;
  auto mm = [&](const bf16x8 (&a)[RM][3], const bf16x8 (&w)[K][3]) {
    constexpr int PA[6] = {2, 0, 1, 1, 0, 0};
    constexpr int PB[6] = {0, 2, 1, 0, 1, 0};
#pragma unroll
    for (int k = 0; k < K; ++k)
#pragma unroll
      for (int c = 0; c < 6; ++c)
#pragma unroll
        for (int rm = 0; rm < RM; ++rm)
          acc[k][rm] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[rm][PA[c]], w[k][PB[c]],
                                                               acc[k][rm], 0, 0, 0);
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
  bf16x8 W[2][K][3], A[2][RM][3];
  gy_load(0);
  wload(0, W[0]);
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
    aload(buf, 0, A[0]);
    for (int tt = 0; tt < nt; tt += 2) {
      const int t = c * kOC + tt;
      // sched_barrier: keep the next step's loads ahead of this step's MFMAs (the
      // scheduler otherwise sinks them to their first use to save registers)
      wload(t + 1, W[1]);
      aload(buf, tt + 1, A[1]);
      __builtin_amdgcn_sched_barrier(0);
      mm(A[0], W[0]);
      __builtin_amdgcn_sched_barrier(0);
      wload(t + 2, W[0]);
      aload(buf, min(tt + 2, kOC - 1), A[0]);  // (past the chunk: a harmless re-read)
      __builtin_amdgcn_sched_barrier(0);
      mm(A[1], W[1]);
      __builtin_amdgcn_sched_barrier(0);
    }
    PP_TICK(3 + 2 * (c & 1))
    if (more) gy_publish(c + 1, buf ^ 1);
    __syncthreads();
    PP_TICK(4 + 2 * (c & 1))
  }

  // epilogue: lane owns input column i0 + r for 16*RM rows
  if (i < sh.I) {
    const float chain = (sh.length * dc.inv_w) * sh.out_scale;  // dxi/dx = L/(2/S)
    constexpr int S = Geom3<N, K>::S, stride = Geom3<N, K>::stride;
#pragma unroll
    for (int rm = 0; rm < RM; ++rm)
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
        float l[N], dl[N];
        pp_lagrange_d<N>(xi, tab, l, dl);
        float gsum = 0.0f;
#pragma unroll
        for (int k = 0; k < K; ++k) {
          float d = 0.0f;
#pragma unroll
          for (int sq = 0; sq < S; ++sq) {
            const int j = k - sq * stride;
            if (j >= 0 && j < N) d = (s == sq) ? dl[j] : d;
          }
          gsum = fmaf(d, acc[k][rm][q], gsum);
        }
        if (row < sh.B) gx[(size_t)row * sh.I + i] = gsum * chain;
      }
  }
  PP_TICK(7)
}

template <int N, int K, int RM>
__global__ __launch_bounds__(256) void pp_bwd_x_bf3_kernel(const float* __restrict__ x,
                                                           const float* __restrict__ gy,
                                                           const void* __restrict__ Q,
                                                           float* __restrict__ gx, PPShape sh,
                                                           BasisTab tab, DenseCoord3 dc, int nso,
                                                           int Ipad, int vw) {
  extern __shared__ __attribute__((aligned(16))) u32x4 gyL[];  // [2][3][kOC][2][32*RM] x 16 B
  // vw: widest aligned vector load of a gy row (block-uniform), fixed per body so the
  // staging loads stay straight-line code
  if (vw == 4) bwd_x_bf3_body<N, K, RM, 4>(x, gy, Q, gx, sh, tab, dc, nso, Ipad, gyL);
  else if (vw == 2) bwd_x_bf3_body<N, K, RM, 2>(x, gy, Q, gx, sh, tab, dc, nso, Ipad, gyL);
  else bwd_x_bf3_body<N, K, RM, 1>(x, gy, Q, gx, sh, tab, dc, nso, Ipad, gyL);
}

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
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void pp_bf3_pack_gy_kernel(const float* __restrict__ gy,
                                                             u32x4* __restrict__ G, int B, int O,
                                                             int Opad, int nbs) {
  const long id = (long)blockIdx.x * 256 + threadIdx.x;
  if (id >= (long)nbs * Opad) return;
  const int o = (int)(id % Opad);
  const int t = (int)(id / Opad);
  unsigned pc[3][16];
#pragma unroll
  for (int j = 0; j < 16; ++j) {
    const int b = 16 * t + j;
    const float v = (b < B && o < O) ? gy[(size_t)b * O + o] : 0.0f;
    split3(v, pc[0][j], pc[1][j], pc[2][j]);
  }
#pragma unroll
  for (int p = 0; p < 3; ++p)
#pragma unroll
    for (int hh = 0; hh < 2; ++hh) {
      u32x4 w;
      w.x = (pc[p][8 * hh + 0] >> 16) | pc[p][8 * hh + 1];
      w.y = (pc[p][8 * hh + 2] >> 16) | pc[p][8 * hh + 3];
      w.z = (pc[p][8 * hh + 4] >> 16) | pc[p][8 * hh + 5];
      w.w = (pc[p][8 * hh + 6] >> 16) | pc[p][8 * hh + 7];
      G[(((size_t)p * nbs + t) * Opad + o) * 2 + hh] = w;
    }
}

template <int N, int K, int NT>
__global__ __launch_bounds__(256) void pp_bwd_w_bf3_kernel(const float* __restrict__ x,
                                                           const void* __restrict__ G,
                                                           float* __restrict__ out, PPShape sh,
                                                           BasisTab tab, DenseCoord3 dc,
                                                           int rows_per_chunk, int nbs, int Opad,
                                                           float scale) {
  extern __shared__ __attribute__((aligned(16))) u32x4 phiL[];  // [3 buf][3 plane][K][2][32] x 16 B
  constexpr int kBufU = 3 * K * 2 * 32;
  const int lane = threadIdx.x & 63;
  const int wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int r = lane & 31, h = lane >> 5;
  const int i0 = blockIdx.x * 32;
  const int chunk = blockIdx.y;
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
  auto produce = [&](int t, int buf, const float (&xv)[2]) {
    const int row = cbeg + 16 * t + 2 * rp;
    float pa[K], pb[K];
    phi3<N, K>(xv[0], row < cend, sh, tab, dc, pa);
    phi3<N, K>(xv[1], row + 1 < cend, sh, tab, dc, pb);
#pragma unroll
    for (int k = 0; k < K; ++k) {
      unsigned a1, a2, a3, c1, c2, c3;
      split3(pa[k], a1, a2, a3);
      split3(pb[k], c1, c2, c3);
      const int base = (buf * kBufU + k * 64) * 4 + pw_base;
      phiW[base] = (a1 >> 16) | c1;
      phiW[base + K * 64 * 4] = (a2 >> 16) | c2;
      phiW[base + 2 * K * 64 * 4] = (a3 >> 16) | c3;
    }
  };
  // consumer
  const rsrc_t rg = make_rsrc3(G, (unsigned)(3u * nbs * Opad * 32u));
  int vcol[NT];
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) vcol[nt] = (min(o0 + nt * 32 + r, Opad - 1) * 2 + h) * 16;
  const int gstep = Opad * 32, gplane = nbs * gstep;
  auto bload = [&](int t, bf16x8 (&bf)[3][NT]) {
#pragma unroll
    for (int p = 0; p < 3; ++p)
#pragma unroll
      for (int nt = 0; nt < NT; ++nt)
        bf[p][nt] = __builtin_bit_cast(
            bf16x8, __builtin_amdgcn_raw_buffer_load_b128(rg, vcol[nt],
                                                          p * gplane + (t0 + t) * gstep, 0));
  };
  auto aload = [&](int buf, bf16x8 (&a)[K][3]) {
#pragma unroll
    for (int k = 0; k < K; ++k)
#pragma unroll
      for (int p = 0; p < 3; ++p)
        a[k][p] = __builtin_bit_cast(bf16x8, phiL[buf * kBufU + ((p * K + k) * 2 + h) * 32 + r]);
  };
  f32x16 acc[K][NT];
#pragma unroll
  for (int k = 0; k < K; ++k)
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
      for (int q = 0; q < 16; ++q) acc[k][nt][q] = 0.0f;
  auto mm = [&](const bf16x8 (&a)[K][3], const bf16x8 (&bf)[3][NT]) {
    constexpr int PA[6] = {2, 0, 1, 1, 0, 0};
    constexpr int PB[6] = {0, 2, 1, 0, 1, 0};
#pragma unroll
    for (int k = 0; k < K; ++k)
#pragma unroll
      for (int c = 0; c < 6; ++c)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
          acc[k][nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[k][PA[c]], bf[PB[c]][nt],
                                                               acc[k][nt], 0, 0, 0);
  };

  float xr[2][2];
  bf16x8 Bf[3][3][NT], A[2][K][3];
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
    xload(t + 3, xr[(J + 1) & 1]);
    bload(t + 2, Bf[(J + 2) % 3]);
    aload((J + 1) % 3, A[(J + 1) & 1]);
    produce(t + 2, (J + 2) % 3, xr[J & 1]);
    mm(A[J & 1], Bf[J % 3]);
    __syncthreads();
  };
  for (int t = 0; t < nst; t += 6) {
    stepfn(t, std::integral_constant<int, 0>{});
    if (t + 1 < nst) stepfn(t + 1, std::integral_constant<int, 1>{});
    if (t + 2 < nst) stepfn(t + 2, std::integral_constant<int, 2>{});
    if (t + 3 < nst) stepfn(t + 3, std::integral_constant<int, 3>{});
    if (t + 4 < nst) stepfn(t + 4, std::integral_constant<int, 4>{});
    if (t + 5 < nst) stepfn(t + 5, std::integral_constant<int, 5>{});
  }
  if (o0 >= sh.O) return;  // wave-uniform: this wave ran on clamped columns
  float* dst = out + (size_t)chunk * sh.I * K * sh.O;
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) {
    const int col = o0 + nt * 32 + r;
    if (col < sh.O) {
#pragma unroll
      for (int k = 0; k < K; ++k)
#pragma unroll
        for (int q = 0; q < 16; ++q) {
          const int row = i0 + (q & 3) + 8 * (q >> 2) + 4 * h;  // input index
          if (row < sh.I) dst[((size_t)row * K + k) * sh.O + col] = acc[k][nt][q] * scale;
        }
    }
  }
}

#define PP_BF3_COMBOS(X) \
  X(1, 1) X(1, 2) X(1, 4) X(1, 8) \
  X(2, 2) X(2, 3) X(2, 4) X(2, 5) X(2, 8) \
  X(3, 3) X(3, 5) X(3, 6) X(4, 4) X(4, 7) X(4, 8) \
  X(5, 5) X(6, 6) X(7, 7) X(8, 8)

// 16-input steps, padded to even so the two-group kernel splits them evenly
int nstep_of(const PPShape& sh) { return ((sh.I + 31) / 32) * 2; }

template <int N, int K, int RM, int G>
hipError_t launch_bf3(const float* x, const void* ws, float* y, const PPShape& sh,
                      const BasisTab& tab, const DenseCoord3& dc, int nstep, hipStream_t st) {
  size_t lds = (size_t)G * 2 * 3 * K * 2 * 32 * RM * 16;
  if (G == 2) lds = std::max(lds, (size_t)2 * 4 * RM * 16 * 64 * 4);
  if (lds > 160 * 1024) return hipErrorInvalidValue;
  if (lds > 48 * 1024) {
    static bool once = false;  // per instantiation
    if (!once) {
      hipError_t e = hipFuncSetAttribute((const void*)pp_fwd_bf3_kernel<N, K, RM, G>,
                                         hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
      if (e != hipSuccess) return e;
      once = true;
    }
  }
  dim3 grid((sh.B + 32 * RM - 1) / (32 * RM), (sh.O + 255) / 256);
  hipLaunchKernelGGL((pp_fwd_bf3_kernel<N, K, RM, G>), grid, dim3(256 * G), lds, st, x, ws, y, sh,
                     tab, dc, nstep);
  return hipGetLastError();
}

}  // namespace
bool pp_bf3_supported(const PPShape& sh) {
  if (!pp_dense_supported(sh)) return false;
  return (long)3 * sh.K * nstep_of(sh) * sh.O * 32 < (1L << 31);
}

size_t pp_bf3_fwd_workspace_bytes(const PPShape& sh) {
  return (size_t)3 * sh.K * nstep_of(sh) * sh.O * 32;
}

hipError_t pp_launch_fwd_bf3(const float* x, const float* wt, float* y, int32_t* seg, void* ws,
                             const PPShape& sh, const BasisTab& tab, hipStream_t st) {
  const int nstep = nstep_of(sh);
  DenseCoord3 dc;
  dc.w2 = 2.0f / (float)sh.S;
  dc.inv_w = (float)sh.S / 2.0f;
  dc.inv_len = 1.0f / sh.length;
  const long pslots = (long)sh.K * nstep * sh.O;
  hipLaunchKernelGGL(pp_bf3_pack_w_kernel, dim3((unsigned)((pslots + 255) / 256)), dim3(256), 0, st,
                     wt, static_cast<u32x4*>(ws), sh.I, sh.K, sh.O, nstep);
  // variant: 64-row tiles once they fill the chip; two wave groups (split-k in the
  // block) when their basis buffers fit LDS.  PP_BF3_VARIANT=<rm><g> overrides (tuning).
  int rm = sh.B >= 8192 && sh.K <= 6 ? 2 : 1;
  int grp = (size_t)2 * 2 * 3 * sh.K * 2 * 32 * rm * 16 <= 160 * 1024 ? 2 : 1;
  if (const char* v = getenv("PP_BF3_VARIANT")) {
    const int code = atoi(v);
    if (code / 10 >= 1 && code / 10 <= 2 && code % 10 >= 1 && code % 10 <= 2) {
      rm = code / 10;
      grp = code % 10;
    }
  }
#define X(N_, K_)                                                                          \
  if (sh.n == N_ && sh.K == K_) {                                                          \
    hipError_t e;                                                                          \
    if (rm == 2 && grp == 2) e = launch_bf3<N_, K_, 2, 2>(x, ws, y, sh, tab, dc, nstep, st); \
    else if (rm == 2) e = launch_bf3<N_, K_, 2, 1>(x, ws, y, sh, tab, dc, nstep, st);      \
    else if (grp == 2) e = launch_bf3<N_, K_, 1, 2>(x, ws, y, sh, tab, dc, nstep, st);     \
    else e = launch_bf3<N_, K_, 1, 1>(x, ws, y, sh, tab, dc, nstep, st);                   \
    if (e == hipSuccess && seg) e = pp_launch_segment_ids(x, seg, sh, st);                 \
    return e;                                                                              \
  }
  PP_BF3_COMBOS(X)
#undef X
  return hipErrorInvalidValue;
}

// ---- backward-input ----
namespace {
int nso_of(const PPShape& sh) { return ((sh.O + 31) / 32) * 2; }   // 16-out steps, even
int ipad_of(const PPShape& sh) { return ((sh.I + 31) / 32) * 32; }

template <int N, int K, int RM>
hipError_t launch_bwd_x_bf3(const float* x, const float* gy, const void* ws, float* gx,
                            const PPShape& sh, const BasisTab& tab, const DenseCoord3& dc,
                            hipStream_t st) {
  const size_t lds = (size_t)2 * 3 * kOC * 2 * 32 * RM * 16;
  if (lds > 48 * 1024) {
    static bool once = false;  // per instantiation
    if (!once) {
      hipError_t e = hipFuncSetAttribute((const void*)pp_bwd_x_bf3_kernel<N, K, RM>,
                                         hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
      if (e != hipSuccess) return e;
      once = true;
    }
  }
  const int vw = sh.O % 4 == 0 ? 4 : (sh.O % 2 == 0 ? 2 : 1);
  dim3 grid((sh.B + 32 * RM - 1) / (32 * RM), (ipad_of(sh) + 127) / 128);
  hipLaunchKernelGGL((pp_bwd_x_bf3_kernel<N, K, RM>), grid, dim3(256), lds, st, x, gy, ws, gx, sh,
                     tab, dc, nso_of(sh), ipad_of(sh), vw);
  return hipGetLastError();
}
}  // namespace

bool pp_bf3_bwd_x_supported(const PPShape& sh) {
  if (!pp_dense_supported(sh)) return false;
  return (long)3 * nso_of(sh) * sh.K * ipad_of(sh) * 32 < (1L << 31);
}

size_t pp_bf3_bwd_x_workspace_bytes(const PPShape& sh) {
  return (size_t)3 * nso_of(sh) * sh.K * ipad_of(sh) * 32;
}

hipError_t pp_launch_bwd_x_bf3(const float* x, const float* gy, const float* wt, float* gx,
                               void* ws, const PPShape& sh, const BasisTab& tab, hipStream_t st) {
  DenseCoord3 dc;
  dc.w2 = 2.0f / (float)sh.S;
  dc.inv_w = (float)sh.S / 2.0f;
  dc.inv_len = 1.0f / sh.length;
  const int nso = nso_of(sh), Ipad = ipad_of(sh);
  const long pslots = (long)nso * sh.K * Ipad;
  hipLaunchKernelGGL(pp_bf3_pack_wo_kernel, dim3((unsigned)((pslots + 255) / 256)), dim3(256), 0,
                     st, wt, static_cast<u32x4*>(ws), sh.I, sh.K, sh.O, nso, Ipad);
  // 64-row tiles keep K*2 accumulators per wave: K <= 6 (registers); else 32-row tiles
  int rm = sh.B >= 4096 && sh.K <= 6 ? 2 : 1;
  if (const char* v = getenv("PP_BF3_BWDX_RM")) {
    const int c = atoi(v);
    if ((c == 1 || c == 2) && (c == 1 || sh.K <= 6)) rm = c;
  }
#define X(N_, K_)                                                                          \
  if (sh.n == N_ && sh.K == K_) {                                                          \
    if constexpr (K_ <= 6) {                                                               \
      if (rm == 2) return launch_bwd_x_bf3<N_, K_, 2>(x, gy, ws, gx, sh, tab, dc, st);     \
    }                                                                                      \
    return launch_bwd_x_bf3<N_, K_, 1>(x, gy, ws, gx, sh, tab, dc, st);                    \
  }
  PP_BF3_COMBOS(X)
#undef X
  return hipErrorInvalidValue;
}

// ---- backward-weight ----
namespace {
struct WPlan3 {
  int nt, ocols, nchunk, rows_per_chunk, nbs, opad;
  size_t g_bytes, part_bytes;
};

WPlan3 wplan3(const PPShape& sh) {
  WPlan3 p;
  p.nt = sh.K <= 6 ? 2 : 1;  // K x NT accumulators of 16 registers per wave
  p.opad = ((sh.O + 31) / 32) * 32;
  p.ocols = (sh.O + 128 * p.nt - 1) / (128 * p.nt);
  const int tiles = ((sh.I + 31) / 32) * p.ocols;
  int target = 256;
  if (const char* v = getenv("PP_BF3_BWDW_BLOCKS")) target = std::max(1, atoi(v));
  int nchunk = std::max(1, target / tiles);
  const int max_chunks = std::max(1, sh.B / 128);  // >= 8 steps of 16 rows per block
  nchunk = std::min(nchunk, max_chunks);
  int rpc = (sh.B + nchunk - 1) / nchunk;
  rpc = (rpc + 15) / 16 * 16;
  p.rows_per_chunk = rpc;
  p.nchunk = (sh.B + rpc - 1) / rpc;
  p.nbs = (sh.B + 15) / 16;  // (steps prefetched past the end read out of range: zeros, unused)
  p.g_bytes = (size_t)3 * p.nbs * p.opad * 32;
  p.part_bytes = p.nchunk > 1 ? (size_t)p.nchunk * sh.I * sh.K * sh.O * sizeof(float) : 0;
  return p;
}

template <int N, int K, int NT>
hipError_t launch_bwd_w_bf3(const float* x, const void* G, float* out, const PPShape& sh,
                            const BasisTab& tab, const DenseCoord3& dc, const WPlan3& p,
                            float scale, hipStream_t st) {
  const size_t lds = (size_t)3 * 3 * K * 2 * 32 * 16;
  if (lds > 48 * 1024) {
    static bool once = false;  // per instantiation
    if (!once) {
      hipError_t e = hipFuncSetAttribute((const void*)pp_bwd_w_bf3_kernel<N, K, NT>,
                                         hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
      if (e != hipSuccess) return e;
      once = true;
    }
  }
  dim3 grid((sh.I + 31) / 32, p.nchunk, p.ocols);
  hipLaunchKernelGGL((pp_bwd_w_bf3_kernel<N, K, NT>), grid, dim3(256), lds, st, x, G, out, sh, tab,
                     dc, p.rows_per_chunk, p.nbs, p.opad, scale);
  return hipGetLastError();
}
}  // namespace

bool pp_bf3_bwd_w_supported(const PPShape& sh) {
  if (!pp_dense_supported(sh)) return false;
  const WPlan3 p = wplan3(sh);
  return p.g_bytes < ((size_t)1 << 31);
}

size_t pp_bf3_bwd_w_workspace_bytes(const PPShape& sh) {
  const WPlan3 p = wplan3(sh);
  return p.g_bytes + p.part_bytes;
}

hipError_t pp_launch_bwd_w_bf3(const float* x, const float* gy, float* gwt, void* ws,
                               const PPShape& sh, const BasisTab& tab, hipStream_t st) {
  DenseCoord3 dc;
  dc.w2 = 2.0f / (float)sh.S;
  dc.inv_w = (float)sh.S / 2.0f;
  dc.inv_len = 1.0f / sh.length;
  const WPlan3 p = wplan3(sh);
  const long nthr = (long)p.nbs * p.opad;
  hipLaunchKernelGGL(pp_bf3_pack_gy_kernel, dim3((unsigned)((nthr + 255) / 256)), dim3(256), 0, st,
                     gy, static_cast<u32x4*>(ws), sh.B, sh.O, p.opad, p.nbs);
  float* part = reinterpret_cast<float*>(static_cast<char*>(ws) + p.g_bytes);
  float* out = p.nchunk > 1 ? part : gwt;
  const float scale = p.nchunk > 1 ? 1.0f : sh.out_scale;
#define X(N_, K_)                                                                          \
  if (sh.n == N_ && sh.K == K_) {                                                          \
    hipError_t e;                                                                          \
    if constexpr (K_ <= 6) e = launch_bwd_w_bf3<N_, K_, 2>(x, ws, out, sh, tab, dc, p, scale, st); \
    else e = launch_bwd_w_bf3<N_, K_, 1>(x, ws, out, sh, tab, dc, p, scale, st);           \
    if (e != hipSuccess || p.nchunk <= 1) return e;                                        \
    pp_launch_chunk_reduce(part, gwt, (long)sh.I * sh.K * sh.O, p.nchunk, sh.out_scale, st); \
    return hipGetLastError();                                                              \
  }
  PP_BF3_COMBOS(X)
#undef X
  return hipErrorInvalidValue;
}
