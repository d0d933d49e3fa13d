// Dense path: expanded-basis x weight contraction on the f32-input MFMA
// (v_mfma_f32_32x32x2_f32: an exact fp32 fma chain at the fp32 vector peak), for
// layers whose weight count per (out, in) pair K is small (segments = 2 in every
// hidden layer of the BASELINE configs), i.e. where n of the K basis columns are
// nonzero per sample and the contraction is GEMM-shaped.
//
//   forward    y[b,o]        = sum_{i,k} Phi[b,i,k] * w_t[i,k,o]
//   bwd-input  G[b,(i,k)]    = sum_o gy[b,o] * w_t[i,k,o];  gx[b,i] = chain * sum_k Phi'[b,i,k] G[b,i,k]
//   bwd-weight gw_t[(i,k),o] = sum_b Phi[b,i,k] * gy[b,o]
//
// Phi is never materialised in HBM: lanes evaluate segment id / local coordinate /
// Lagrange basis for "their" (row, input) pair in registers (A1-A3 of SURVEY.md
// 8a) and feed the K basis columns to K MFMAs as the A operand.  The two k-slots
// of the 32x32x2 MFMA are two different inputs (forward) or two different rows
// (bwd-weight), so no lane computes a basis value it does not use.
//
// Only "exact" layers take this path: length and segments powers of two, so that
// the reference's divisions in A1/A2 are exact multiplications and the segment id
// / local coordinate stay bit-identical without a single v_div sequence in the
// MFMA loops.  Everything else goes to the gather path.
#include <algorithm>
#include <cstdlib>

#include "pp_device.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));

struct DenseCoord {
  float w2;       // 2/S
  float inv_w;    // S/2
  float inv_len;  // 1/length
};

// A1 + A2 with the divisions replaced by their exact reciprocal multiplications.
__device__ __forceinline__ void dense_seg_xi(float x, const PPShape& sh, const DenseCoord& dc,
                                             int& s, float& xi) {
  const float t = ((x + sh.half) * dc.inv_len) * sh.Sf;
  // branch-free form of pp_segment(): clamp, truncate, clamp; non-finite / huge t -> 0
  const float tc = __builtin_amdgcn_fmed3f(t, -1.0f, sh.Sf);
  int si = (int)tc;
  si = min(max(si, 0), sh.S - 1);
  s = (__builtin_fabsf(t) < 9.2e18f) ? si : 0;
  const float x_min = (float)s * dc.w2 - 1.0f;  // == (s/S)*2-1 exactly
  xi = sh.length * ((x - x_min) * dc.inv_w) - sh.half;
}

// (N, K) fixes the layer type on this path (S must be a power of two):
// K % N == 0 with K/N a power of two -> discontinuous (stride N), else continuous.
template <int N, int K>
struct DenseGeom {
  static constexpr bool disc = (K % N == 0) && (((K / N) & (K / N - 1)) == 0);
  static constexpr int stride = disc ? N : N - 1;
  static constexpr int S = disc ? K / N : (N > 1 ? (K - 1) / (N - 1) : K);
};

// Phi[k] for k = 0..K-1 from the n active basis values (zeros elsewhere): for each k
// only the segments s' with 0 <= k - s'*stride < N can contribute, so the select
// runs over those compile-time candidates (S compares + <= 2 v_cndmask per k).
template <int N, int K>
__device__ __forceinline__ void expand_basis(const float (&l)[N], int s, float (&phi)[K]) {
  constexpr int S = DenseGeom<N, K>::S, stride = DenseGeom<N, K>::stride;
  bool is[S];
#pragma unroll
  for (int q = 0; q < S; ++q) is[q] = (s == q);
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

template <int N, int K>
__device__ __forceinline__ void phi_of_x(float xv, bool valid, const PPShape& sh,
                                         const BasisTab& tab, const DenseCoord& dc,
                                         float (&phi)[K], int& s) {
  float xi, l[N];
  dense_seg_xi(xv, sh, dc, s, xi);
  pp_lagrange<N>(xi, tab, l);
  if (!valid) {
#pragma unroll
    for (int j = 0; j < N; ++j) l[j] = 0.0f;
  }
  expand_basis<N, K>(l, s, phi);
}

typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
using rsrc_t = __amdgpu_buffer_rsrc_t;

// Raw buffer descriptor over [p, p+bytes): loads take an SGPR base + one 32-bit
// VGPR offset + an SGPR offset, so the MFMA loops carry no 64-bit address
// arithmetic, and reads past `bytes` return 0 instead of faulting (row / column
// tails need no clamping).  p and bytes are kernel arguments (wave-uniform).
__device__ __forceinline__ rsrc_t make_rsrc(const void* p, unsigned bytes) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, bytes, 0x00020000);
}

// W floats at byte offset voff (VGPR) + soff (SGPR), fetched as VW-wide pieces.
template <int W, int VW>
__device__ __forceinline__ void bload(rsrc_t r, int voff, int soff, float (&v)[W]) {
  if constexpr (W >= 4 && VW == 4) {
#pragma unroll
    for (int m = 0; m < W; m += 4) {
      const u32x4 t = __builtin_amdgcn_raw_buffer_load_b128(r, voff + 4 * m, soff, 0);
      v[m] = __uint_as_float(t.x); v[m + 1] = __uint_as_float(t.y);
      v[m + 2] = __uint_as_float(t.z); v[m + 3] = __uint_as_float(t.w);
    }
  } else if constexpr (W >= 2 && VW >= 2) {
#pragma unroll
    for (int m = 0; m < W; m += 2) {
      const u32x2 t = __builtin_amdgcn_raw_buffer_load_b64(r, voff + 4 * m, soff, 0);
      v[m] = __uint_as_float(t.x); v[m + 1] = __uint_as_float(t.y);
    }
  } else {
#pragma unroll
    for (int m = 0; m < W; ++m)
      v[m] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(r, voff + 4 * m, soff, 0));
  }
}

__device__ __forceinline__ float bload1(rsrc_t r, int voff, int soff) {
  return __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(r, voff, soff, 0));
}

// ---------------------------------------------------------------------------
// forward: wave tile = 32 rows x (32*NT) outputs; lane (r,h) owns outputs
// o0 + NT*r .. + NT-1 (one wide load per weight row feeds NT MFMAs) and input
// 2*pair + h.  Software-pipelined: weights of pair p+1 and x of pair p+2 are in
// flight, and Phi of pair p+1 is evaluated, while the MFMAs of pair p issue.
// ---------------------------------------------------------------------------
template <int N, int K, int NT, int VW>
__device__ __forceinline__ void fwd_dense_body(const float* __restrict__ x,
                                               const float* __restrict__ wt,
                                               float* __restrict__ y, const PPShape& sh,
                                               const BasisTab& tab, const DenseCoord& dc) {
  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  const int r = lane & 31, h = lane >> 5;
  const int b0 = (blockIdx.x * 4 + wave) * 32;
  if (b0 >= sh.B) return;
  const int o0 = blockIdx.y * (32 * NT);
  const int b = min(b0 + r, sh.B - 1);
  const rsrc_t rw = make_rsrc(wt, (unsigned)sh.I * sh.K * sh.O * 4u);
  const rsrc_t rx = make_rsrc(x, (unsigned)sh.B * sh.I * 4u);
  const int rowb = sh.O * 4;                              // bytes per weight row
  const int vw_off = (h * K * sh.O + o0 + NT * r) * 4;    // lane-constant part of the W address
  const int vx_off = (b * sh.I + h) * 4;
  const int pair_w = 2 * K * rowb;                        // W bytes per input pair

  f32x16 acc[NT];
#pragma unroll
  for (int nt = 0; nt < NT; ++nt)
#pragma unroll
    for (int q = 0; q < 16; ++q) acc[nt][q] = 0.0f;

  const int npair = (sh.I + 1) >> 1;
  // three register sets (A/B/C) used round-robin, so no copies and no early waits:
  // while the MFMAs of pair p read one set, the other two receive the weights of
  // pairs p+1 (already in flight) and p+2 (issued now); the basis of pair p+1 is
  // evaluated under the MFMAs of pair p.  x runs 3 pairs ahead.
  float wA[K][NT], wB[K][NT], wC[K][NT], phiA[K], phiB[K], phiC[K];
  int s_unused;
  phi_of_x<N, K>(bload1(rx, vx_off, 0), h < sh.I, sh, tab, dc, phiA, s_unused);
#pragma unroll
  for (int k = 0; k < K; ++k) bload<NT, VW>(rw, vw_off, k * rowb, wA[k]);
#pragma unroll
  for (int k = 0; k < K; ++k) bload<NT, VW>(rw, vw_off, pair_w + k * rowb, wB[k]);
  float x0 = bload1(rx, vx_off, 8), x1 = bload1(rx, vx_off, 16), x2 = bload1(rx, vx_off, 24);
  __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0): enter the loop with nothing in flight

  // one pair ip: issue the weights of pair ip+2 into w2, evaluate the basis of pair
  // ip+1 (its x is in xr) into phi1, refill xr with pair ip+4, run the MFMAs of pair
  // ip from (w0, phi0).  Reads past the last input return 0.
  auto step = [&](int ip, const float (&w0)[K][NT], const float (&phi0)[K], float (&phi1)[K],
                  float (&w2)[K][NT], float& xr) {
    const int sw = (ip + 2) * pair_w;
#pragma unroll
    for (int k = 0; k < K; ++k) bload<NT, VW>(rw, vw_off, sw + k * rowb, w2[k]);
    phi_of_x<N, K>(xr, 2 * (ip + 1) + h < sh.I, sh, tab, dc, phi1, s_unused);
    xr = bload1(rx, vx_off, 8 * (ip + 4));
#pragma unroll
    for (int k = 0; k < K; ++k)
#pragma unroll
      for (int nt = 0; nt < NT; ++nt)
        acc[nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(phi0[k], w0[k][nt], acc[nt], 0, 0, 0);
  };
  int ip = 0;
  for (; ip + 2 < npair; ip += 3) {
    step(ip, wA, phiA, phiB, wC, x0);
    step(ip + 1, wB, phiB, phiC, wA, x1);
    step(ip + 2, wC, phiC, phiA, wB, x2);
  }
  if (ip < npair) step(ip, wA, phiA, phiB, wC, x0);
  if (ip + 1 < npair) step(ip + 1, wB, phiB, phiC, wA, x1);
  // epilogue: lane holds NT consecutive columns for 16 rows
  const int oc = o0 + NT * r;
#pragma unroll
  for (int q = 0; q < 16; ++q) {
    const int row = b0 + (q & 3) + 8 * (q >> 2) + 4 * h;
    if (row < sh.B) {
      float* yp = y + (size_t)row * sh.O + oc;
      if constexpr (NT == 4 && VW == 4) {
        if (oc < sh.O)
          *reinterpret_cast<float4*>(yp) =
              make_float4(acc[0][q] * sh.out_scale, acc[1 % NT][q] * sh.out_scale,
                          acc[2 % NT][q] * sh.out_scale, acc[3 % NT][q] * sh.out_scale);
      } else if constexpr (VW >= 2) {
#pragma unroll
        for (int m = 0; m < NT; m += 2)
          if (oc + m < sh.O)
            *reinterpret_cast<float2*>(yp + m) =
                make_float2(acc[m][q] * sh.out_scale, acc[(m + 1) % NT][q] * sh.out_scale);
      } else {
#pragma unroll
        for (int m = 0; m < NT; ++m)
          if (oc + m < sh.O) yp[m] = acc[m][q] * sh.out_scale;
      }
    }
  }
}

template <int N, int K, int NT>
__global__ __launch_bounds__(256) void pp_fwd_dense_kernel(const float* __restrict__ x,
                                                           const float* __restrict__ wt,
                                                           float* __restrict__ y, PPShape sh,
                                                           BasisTab tab, DenseCoord dc, int vw) {
  // vw is wave-uniform: one straight-line body per load width
  if (NT >= 4 && vw == 4) fwd_dense_body<N, K, NT, (NT >= 4 ? 4 : 2)>(x, wt, y, sh, tab, dc);
  else if (vw >= 2) fwd_dense_body<N, K, NT, 2>(x, wt, y, sh, tab, dc);
  else fwd_dense_body<N, K, NT, 1>(x, wt, y, sh, tab, dc);
}

// ---------------------------------------------------------------------------
// forward, shared-basis variant: block = 8 waves = RT row tiles x CW column waves
// (CW*RT = 8), each wave a 32-row x 64-out tile (NT = 2).  The basis of a (row
// tile, input pair) is evaluated ONCE per block: in every group of CW pairs, column
// wave cw evaluates pair g*CW+cw of its row tile and publishes the K columns in
// LDS; after one barrier all CW waves of the row tile run the group's CW*K*2 MFMAs
// from LDS-read Phi and their own weight columns.  VALU work per MFMA falls by CW
// (VALU does not hide under the f32 MFMA on gfx950), and with 8 waves per CU the
// waits of one wave sit under the MFMAs of its SIMD partner.
// ---------------------------------------------------------------------------
template <int N, int K, int CW, int RT, int VW>
__device__ __forceinline__ void fwd_dense_shared_body(const float* __restrict__ x,
                                                      const float* __restrict__ wt,
                                                      float* __restrict__ y, const PPShape& sh,
                                                      const BasisTab& tab, const DenseCoord& dc,
                                                      float* phil) {
  constexpr int NT = 2;
  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  const int rt = wave / CW, cw = wave - rt * CW;
  const int r = lane & 31, h = lane >> 5;
  const int b0 = (blockIdx.x * RT + rt) * 32;
  const int o0 = (blockIdx.y * CW + cw) * (32 * NT);
  const int b = min(b0 + r, sh.B - 1);
  const rsrc_t rw = make_rsrc(wt, (unsigned)sh.I * sh.K * sh.O * 4u);
  const rsrc_t rx = make_rsrc(x, (unsigned)sh.B * sh.I * 4u);
  const int rowb = sh.O * 4;
  const int vw_off = (h * K * sh.O + min(o0 + NT * r, sh.O - VW)) * 4;
  const int vx_off = (b * sh.I + 2 * cw + h) * 4;  // this wave produces pair g*CW + cw
  const int pair_w = 2 * K * rowb;
  const int npair = (sh.I + 1) >> 1;
  const int ngroup = (npair + CW - 1) / CW;
  // LDS: phil[buf][rt][p][k][lane]
  float* my_phi = phil + (size_t)rt * CW * K * 64;
  constexpr int kBufF = RT * CW * K * 64;

  f32x16 acc[NT];
#pragma unroll
  for (int nt = 0; nt < NT; ++nt)
#pragma unroll
    for (int q = 0; q < 16; ++q) acc[nt][q] = 0.0f;

  auto publish = [&](float xv, int g, int buf) {
    float phi[K];
    int s;
    phi_of_x<N, K>(xv, 2 * (g * CW + cw) + h < sh.I, sh, tab, dc, phi, s);
#pragma unroll
    for (int k = 0; k < K; ++k) my_phi[buf * kBufF + (cw * K + k) * 64 + lane] = phi[k];
  };
  auto wload = [&](int g, float (&w)[CW][K][NT]) {
#pragma unroll
    for (int p = 0; p < CW; ++p)
#pragma unroll
      for (int k = 0; k < K; ++k)
        bload<NT, VW>(rw, vw_off, (g * CW + p) * pair_w + k * rowb, w[p][k]);
  };

  float wA[CW][K][NT], wB[CW][K][NT];
  publish(bload1(rx, vx_off, 0), 0, 0);
  float xn = bload1(rx, vx_off, 8 * CW);  // group 1
  wload(0, wA);
  __builtin_amdgcn_s_waitcnt(0x0F70);
  __syncthreads();
  auto group = [&](int g, const float (&w)[CW][K][NT], float (&wn)[CW][K][NT]) {
    const int buf = g & 1;
    wload(g + 1, wn);  // past the last input: reads return 0
    const float xn2 = bload1(rx, vx_off, 8 * CW * (g + 2));
    if (g + 1 < ngroup) publish(xn, g + 1, buf ^ 1);
#pragma unroll
    for (int p = 0; p < CW; ++p) {
      float phi[K];
#pragma unroll
      for (int k = 0; k < K; ++k) phi[k] = my_phi[buf * kBufF + (p * K + k) * 64 + lane];
#pragma unroll
      for (int k = 0; k < K; ++k)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
          acc[nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(phi[k], w[p][k][nt], acc[nt], 0, 0, 0);
    }
    xn = xn2;
    __syncthreads();
  };
  int g = 0;
  for (; g + 1 < ngroup; g += 2) {
    group(g, wA, wB);
    group(g + 1, wB, wA);
  }
  if (g < ngroup) group(g, wA, wB);

  if (b0 >= sh.B) return;
  const int oc = o0 + NT * r;
#pragma unroll
  for (int q = 0; q < 16; ++q) {
    const int row = b0 + (q & 3) + 8 * (q >> 2) + 4 * h;
    if (row < sh.B) {
      float* yp = y + (size_t)row * sh.O + oc;
      if constexpr (VW >= 2) {
        if (oc < sh.O)
          *reinterpret_cast<float2*>(yp) =
              make_float2(acc[0][q] * sh.out_scale, acc[1][q] * sh.out_scale);
      } else {
#pragma unroll
        for (int m = 0; m < NT; ++m)
          if (oc + m < sh.O) yp[m] = acc[m][q] * sh.out_scale;
      }
    }
  }
}

template <int N, int K, int CW, int RT>
__global__ __launch_bounds__(64 * CW * RT) void pp_fwd_dense_shared_kernel(const float* __restrict__ x,
                                                                  const float* __restrict__ wt,
                                                                  float* __restrict__ y, PPShape sh,
                                                                  BasisTab tab, DenseCoord dc,
                                                                  int vw) {
  __shared__ float phil[2 * CW * RT * K * 64];
  if (vw >= 2) fwd_dense_shared_body<N, K, CW, RT, 2>(x, wt, y, sh, tab, dc, phil);
  else fwd_dense_shared_body<N, K, CW, RT, 1>(x, wt, y, sh, tab, dc, phil);
}

// ---------------------------------------------------------------------------
// backward-input: block tile = 128 rows x 32 inputs (wave = 32 rows), K
// accumulator tiles (one per weight index k).  The reduction runs over `out` in
// chunks of 32: the W tile [K][32 inputs][32 outs] of the NEXT chunk is fetched
// into registers while the current chunk's MFMAs run, then written to the other
// LDS buffer (one barrier per chunk).  Rows are padded to 36 floats so that the
// ds_read_b128 of 4 consecutive outs per lane is bank-conflict-free.  Lane (r,h)
// takes outs 8t+4h..+3 of each group of 8: one 16-byte gy load (prefetched one
// group ahead) and K ds_read_b128 feed 4K MFMAs.
// ---------------------------------------------------------------------------
constexpr int kWStride = 36;

template <int N, int K, int VW>
__device__ __forceinline__ void bwd_input_dense_body(const float* __restrict__ x,
                                                     const float* __restrict__ gy,
                                                     const float* __restrict__ wt,
                                                     float* __restrict__ gx, const PPShape& sh,
                                                     const BasisTab& tab, const DenseCoord& dc,
                                                     float* wl) {
  constexpr int kBuf = K * 32 * kWStride;
  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  const int r = lane & 31, h = lane >> 5;
  const int b0 = (blockIdx.x * 4 + wave) * 32;
  const int i0 = blockIdx.y * 32;
  const int b = min(b0 + r, sh.B - 1);
  const bool wave_on = b0 < sh.B;
  const int rowb = sh.O * 4;
  // this tile's (input, k) weight rows are contiguous in w_t: a sub-buffer whose end
  // is the end of the tile's (or the matrix') rows, so rows past the tile read as 0
  const int nrows = min(32, sh.I - i0) * K;
  const rsrc_t rw = make_rsrc(wt + (size_t)i0 * K * sh.O, (unsigned)nrows * rowb);
  const rsrc_t rg = make_rsrc(gy, (unsigned)sh.B * rowb);
  const int vg_off = b * rowb + 16 * h;  // gy: row b, outs 4h.. of each group of 8

  f32x16 acc[K];
#pragma unroll
  for (int k = 0; k < K; ++k)
#pragma unroll
    for (int q = 0; q < 16; ++q) acc[k][q] = 0.0f;

  const int nchunk = (sh.O + 31) >> 5;
  const bool ragged = (sh.O & 31) != 0;

  // staging slot m of this thread: tile row (t + 256 m) / 8, floats 4*((t + 256 m) % 8) .. +3
  int srow[K], slds[K];
#pragma unroll
  for (int m = 0; m < K; ++m) {
    const int slot = threadIdx.x + 256 * m;
    const int rowid = slot >> 3, part = (slot & 7) * 4;
    const int ri = rowid / K, k = rowid - ri * K;
    srow[m] = rowid * rowb + part * 4;
    slds[m] = (k * 32 + ri) * kWStride + part;
  }
  float stage[K][4];
  auto stage_load = [&](int c) {
#pragma unroll
    for (int m = 0; m < K; ++m) bload<4, VW>(rw, srow[m], c * 128, stage[m]);
    if (ragged && c == nchunk - 1) {  // block-uniform: zero the columns past O
#pragma unroll
      for (int m = 0; m < K; ++m) {
        const int o = c * 32 + ((threadIdx.x + 256 * m) & 7) * 4;
#pragma unroll
        for (int e = 0; e < 4; ++e)
          if (o + e >= sh.O) stage[m][e] = 0.0f;
      }
    }
  };
  auto stage_store = [&](int buf) {
#pragma unroll
    for (int m = 0; m < K; ++m)
      *reinterpret_cast<float4*>(&wl[buf * kBuf + slds[m]]) =
          make_float4(stage[m][0], stage[m][1], stage[m][2], stage[m][3]);
  };
  // gy of group q = 4*chunk + t: outs 8q + 4h + 0..3 of row b (zero past O)
  auto gy_load = [&](int q, float (&dst)[4]) {
    bload<4, VW>(rg, vg_off, q * 32, dst);
    if (ragged && q >= 4 * (nchunk - 1)) {  // wave-uniform: only the last chunk has a tail
      const int o = 8 * q + 4 * h;
#pragma unroll
      for (int e = 0; e < 4; ++e)
        if (o + e >= sh.O) dst[e] = 0.0f;
    }
  };

  stage_load(0);
  stage_store(0);
  float gA[4], gB[4];
  gy_load(0, gA);
  __syncthreads();
  auto group = [&](const float* wb, int t, int q, const float (&g)[4], float (&gn)[4]) {
    gy_load(q + 1, gn);  // past the last group: zero-masked / out of range
    float4 wv[K];
#pragma unroll
    for (int k = 0; k < K; ++k)
      wv[k] = *reinterpret_cast<const float4*>(wb + k * 32 * kWStride + 8 * t);
#pragma unroll
    for (int k = 0; k < K; ++k) {
      acc[k] = __builtin_amdgcn_mfma_f32_32x32x2f32(g[0], wv[k].x, acc[k], 0, 0, 0);
      acc[k] = __builtin_amdgcn_mfma_f32_32x32x2f32(g[1], wv[k].y, acc[k], 0, 0, 0);
      acc[k] = __builtin_amdgcn_mfma_f32_32x32x2f32(g[2], wv[k].z, acc[k], 0, 0, 0);
      acc[k] = __builtin_amdgcn_mfma_f32_32x32x2f32(g[3], wv[k].w, acc[k], 0, 0, 0);
    }
  };
  for (int c = 0; c < nchunk; ++c) {
    const int buf = c & 1;
    const bool more = c + 1 < nchunk;  // block-uniform
    if (more) stage_load(c + 1);
    {  // waves past the last row tile run on clamped rows (no divergent control flow here)
      const float* wb = wl + buf * kBuf + r * kWStride + 4 * h;
      group(wb, 0, 4 * c + 0, gA, gB);
      group(wb, 1, 4 * c + 1, gB, gA);
      group(wb, 2, 4 * c + 2, gA, gB);
      group(wb, 3, 4 * c + 3, gB, gA);
    }
    if (more) stage_store(buf ^ 1);
    __syncthreads();
  }
  if (!wave_on) return;
  // epilogue: lane owns input column i0+r for 16 rows
  const int i = i0 + r;
  if (i < sh.I) {
    const float chain = (sh.length * dc.inv_w) * sh.out_scale;  // dxi/dx = L/(2/S)
#pragma unroll
    for (int q = 0; q < 16; ++q) {
      const int row = b0 + (q & 3) + 8 * (q >> 2) + 4 * h;
      if (row < sh.B) {
        const float xv = x[(size_t)row * sh.I + i];
        int s;
        float xi;
        dense_seg_xi(xv, sh, dc, s, xi);
        float l[N], dl[N], dphi[K];
        pp_lagrange_d<N>(xi, tab, l, dl);
        expand_basis<N, K>(dl, s, dphi);
        float gsum = 0.0f;
#pragma unroll
        for (int k = 0; k < K; ++k) gsum = fmaf(dphi[k], acc[k][q], gsum);
        gx[(size_t)row * sh.I + i] = gsum * chain;
      }
    }
  }
}

template <int N, int K>
__global__ __launch_bounds__(256) void pp_bwd_input_dense_kernel(const float* __restrict__ x,
                                                                 const float* __restrict__ gy,
                                                                 const float* __restrict__ wt,
                                                                 float* __restrict__ gx,
                                                                 PPShape sh, BasisTab tab,
                                                                 DenseCoord dc, int vw) {
  extern __shared__ __attribute__((aligned(16))) float wl[];  // 2 x [K][32][kWStride]
  if (vw == 4) bwd_input_dense_body<N, K, 4>(x, gy, wt, gx, sh, tab, dc, wl);
  else if (vw == 2) bwd_input_dense_body<N, K, 2>(x, gy, wt, gx, sh, tab, dc, wl);
  else bwd_input_dense_body<N, K, 1>(x, gy, wt, gx, sh, tab, dc, wl);
}

// ---------------------------------------------------------------------------
// backward-weight: block = NW waves = NW adjacent 32-wide `out` tiles of one
// 32-input tile, over one chunk of rows.  Rows are consumed two per MFMA step
// (k-slots = rows b, b+1).  Phi is shared through LDS: in every group of NW
// steps, wave w evaluates the basis of step w once and publishes its K columns;
// after one barrier all NW waves run the group's NW*K MFMAs, each against its
// own gy column tile.  So the VALU work per MFMA falls by NW and the basis of a
// (row, input) pair is evaluated once per block.  x and gy are prefetched one
// group ahead.  Per-chunk partials go to the workspace and are summed in a fixed
// order by pp_dense_reduce_kernel.
// ---------------------------------------------------------------------------
template <int N, int K, int NW>
__global__ __launch_bounds__(NW * 64) void pp_bwd_weight_dense_kernel(
    const float* __restrict__ x, const float* __restrict__ gy, float* __restrict__ out, PPShape sh,
    BasisTab tab, DenseCoord dc, int rows_per_chunk, float scale) {
  __shared__ float phil[2][NW][K][64];
  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  const int r = lane & 31, h = lane >> 5;
  const int i0 = blockIdx.x * 32;
  const int o0 = (blockIdx.y * NW + wave) * 32;
  const int chunk = blockIdx.z;
  const int cbeg = chunk * rows_per_chunk;
  const int cend = min(sh.B, cbeg + rows_per_chunk);
  const int ic = min(i0 + r, sh.I - 1);
  const int oc = min(o0 + r, sh.O - 1);
  const bool wave_on = o0 < sh.O;  // this wave's out tile exists
  // sub-buffers ending at the chunk's last row: rows past the chunk read as 0
  const rsrc_t rx = make_rsrc(x + (size_t)cbeg * sh.I, (unsigned)(cend - cbeg) * sh.I * 4u);
  const rsrc_t rg = make_rsrc(gy + (size_t)cbeg * sh.O, (unsigned)(cend - cbeg) * sh.O * 4u);
  const int vx_off = ((2 * wave + h) * sh.I + ic) * 4;  // step `wave` of a group, row parity h
  const int vg_off = (h * sh.O + oc) * 4;
  const int grp_x = 2 * NW * sh.I * 4, grp_g = 2 * NW * sh.O * 4, step_g = 2 * sh.O * 4;

  f32x16 acc[K];
#pragma unroll
  for (int k = 0; k < K; ++k)
#pragma unroll
    for (int q = 0; q < 16; ++q) acc[k][q] = 0.0f;

  const int nstep = (cend - cbeg + 1) >> 1;
  const int ngroup = (nstep + NW - 1) / NW;

  auto publish = [&](float xv, int grp, int buf) {  // wave w publishes Phi of step grp*NW + w
    const bool valid = cbeg + 2 * (grp * NW + wave) + h < cend;
    float phi[K];
    int s;
    phi_of_x<N, K>(xv, valid, sh, tab, dc, phi, s);
#pragma unroll
    for (int k = 0; k < K; ++k) phil[buf][wave][k][lane] = phi[k];
  };

  float gvA[NW], gvB[NW];
  publish(bload1(rx, vx_off, 0), 0, 0);
  float xn = bload1(rx, vx_off, grp_x);  // group 1
#pragma unroll
  for (int w = 0; w < NW; ++w) gvA[w] = bload1(rg, vg_off, w * step_g);
  __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0)
  __syncthreads();
  // one group: prefetch gy of group grp+1 (into gvn) and x of group grp+2, publish the
  // basis of group grp+1, run this group's NW*K MFMAs from (LDS Phi, gv).
  auto group = [&](int grp, const float (&gv)[NW], float (&gvn)[NW]) {
    const int buf = grp & 1;
#pragma unroll
    for (int w = 0; w < NW; ++w) gvn[w] = bload1(rg, vg_off, (grp + 1) * grp_g + w * step_g);
    const float xn2 = bload1(rx, vx_off, (grp + 2) * grp_x);
    if (grp + 1 < ngroup) publish(xn, grp + 1, buf ^ 1);
    // (waves whose out tile does not exist run on a clamped column and skip the store)
#pragma unroll
    for (int w = 0; w < NW; ++w) {
      float phi[K];
#pragma unroll
      for (int k = 0; k < K; ++k) phi[k] = phil[buf][w][k][lane];
#pragma unroll
      for (int k = 0; k < K; ++k)
        acc[k] = __builtin_amdgcn_mfma_f32_32x32x2f32(phi[k], gv[w], acc[k], 0, 0, 0);
    }
    xn = xn2;
    __syncthreads();
  };
  int grp = 0;
  for (; grp + 1 < ngroup; grp += 2) {
    group(grp, gvA, gvB);
    group(grp + 1, gvB, gvA);
  }
  if (grp < ngroup) group(grp, gvA, gvB);
  if (!wave_on) return;
  float* dst = out + (size_t)chunk * sh.I * K * sh.O;
  const int col = o0 + r;
  if (col < sh.O) {
#pragma unroll
    for (int k = 0; k < K; ++k)
#pragma unroll
      for (int q = 0; q < 16; ++q) {
        const int row = i0 + (q & 3) + 8 * (q >> 2) + 4 * h;  // input index
        if (row < sh.I) dst[((size_t)row * K + k) * sh.O + col] = acc[k][q] * scale;
      }
  }
}

// Four waves share 64 consecutive elements: wave q sums chunks q, q+4, q+8, ... (loads issued
// eight at a time), the four sums are added in a fixed order through LDS.  The partials were
// written a moment ago and mostly sit in the Infinity Cache: what limits this kernel is how many
// loads are in flight, not bandwidth -- one thread per element walking all chunks ran at
// 3.6 TB/s.
template <typename T>
__global__ __launch_bounds__(256) void pp_dense_reduce_kernel(const T* __restrict__ part,
                                                              T* __restrict__ gwt, long total,
                                                              int nchunk, float scale) {
  __shared__ T sm[3][64];
  const int lane = threadIdx.x & 63, q = threadIdx.x >> 6;
  const long idx = (long)blockIdx.x * 64 + lane;
  const bool ok = idx < total;
  T s = T();
  if (ok) {
    int c = q;
    for (; c + 28 < nchunk; c += 32) {
      T v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) v[u] = part[(size_t)(c + 4 * u) * total + idx];
#pragma unroll
      for (int u = 0; u < 8; ++u) s += v[u];
    }
    for (; c < nchunk; c += 4) s += part[(size_t)c * total + idx];
  }
  if (q > 0) sm[q - 1][lane] = s;
  __syncthreads();
  if (q == 0 && ok) {
    s += sm[0][lane];
    s += sm[1][lane];
    s += sm[2][lane];
    gwt[idx] = s * scale;
  }
}

DenseCoord make_coord(const PPShape& sh) {
  DenseCoord dc;
  dc.w2 = 2.0f / (float)sh.S;
  dc.inv_w = (float)sh.S / 2.0f;
  dc.inv_len = 1.0f / sh.length;
  return dc;
}

bool is_exact(const PPShape& sh) {
  int e;
  const bool len_pow2 = std::frexp(sh.length, &e) == 0.5f;
  const bool s_pow2 = (sh.S & (sh.S - 1)) == 0;
  return len_pow2 && s_pow2;
}

int vec_width(int O) { return O % 4 == 0 ? 4 : (O % 2 == 0 ? 2 : 1); }

// tuning overrides exist in developer builds (-DPP_TUNING) only
int env_int(const char* name, int dflt) {
#ifdef PP_TUNING
  const char* v = std::getenv(name);
  return v ? std::atoi(v) : dflt;
#else
  (void)name;
  return dflt;
#endif
}

struct WeightPlan {
  int nw, ncol, nchunk, rows_per_chunk;
};

WeightPlan weight_plan(const PPShape& sh) {
  WeightPlan p;
  const int otiles = (sh.O + 31) / 32;
  p.nw = otiles > 4 ? 8 : 4;
  p.ncol = (otiles + p.nw - 1) / p.nw;
  const int tiles = ((sh.I + 31) / 32) * p.ncol;
  const int target = env_int("PP_DENSE_BWDW_BLOCKS", 256);
  int nchunk = (target + tiles - 1) / tiles;
  // at least two groups of NW steps per block (8 measured 28 vs 19 us at 256 rows and 36 vs 28
  // at 1000 on the cfg2 hidden layer: the blocks' serial chain is what a small batch waits for;
  // no effect from 4096 rows up, where `target` decides)
  const int min_groups = env_int("PP_DENSE_BWDW_MINGRP", 2);
  const int max_chunks = std::max(1, sh.B / (2 * min_groups * p.nw));
  nchunk = std::max(1, std::min(nchunk, max_chunks));
  int rpc = (sh.B + nchunk - 1) / nchunk;
  rpc = (rpc + 2 * p.nw - 1) / (2 * p.nw) * (2 * p.nw);
  p.rows_per_chunk = rpc;
  p.nchunk = (sh.B + rpc - 1) / rpc;
  return p;
}

// (N, K) pairs with K <= 8 that an exact (S = 1, 2, 4 or 8) layer can produce
#define PP_DENSE_COMBOS(X) \
  X(1, 1) X(1, 2) X(1, 4) X(1, 8) \
  X(2, 2) X(2, 3) X(2, 4) X(2, 5) X(2, 8) \
  X(3, 3) X(3, 5) X(3, 6) X(4, 4) X(4, 7) X(4, 8) \
  X(5, 5) X(6, 6) X(7, 7) X(8, 8)

}  // namespace

bool pp_dense_supported(const PPShape& sh) {
  if (!is_exact(sh)) return false;
  // buffer loads address with 32-bit byte offsets
  const long lim = 1L << 31;
  if ((long)sh.B * sh.I * 4 >= lim || (long)sh.B * sh.O * 4 >= lim ||
      (long)sh.I * sh.K * sh.O * 4 >= lim)
    return false;
#define X(N_, K_)                                            \
  if (sh.n == N_ && sh.K == K_)                               \
    return sh.S == DenseGeom<N_, K_>::S &&                    \
           (sh.S == 1 || sh.stride == DenseGeom<N_, K_>::stride);
  PP_DENSE_COMBOS(X)
#undef X
  return false;
}

size_t pp_dense_bwd_weight_workspace_bytes(const PPShape& sh) {
  const WeightPlan p = weight_plan(sh);
  if (p.nchunk <= 1) return 0;
  return (size_t)p.nchunk * sh.I * sh.K * sh.O * sizeof(float);
}

hipError_t pp_launch_fwd_dense(const float* x, const float* wt, float* y, int32_t* seg,
                               const PPShape& sh, const BasisTab& tab, hipStream_t st) {
  const DenseCoord dc = make_coord(sh);
  const int vw = vec_width(sh.O);
  const int nt = env_int("PP_DENSE_FWD_NT", sh.O > 64 ? 4 : 2);
  const int shared = env_int("PP_DENSE_FWD_SHARED", sh.O > 64 ? 1 : 0);
  if (shared) {
    const int cw = sh.O > 128 ? 4 : 2;
    const int rtl = cw == 4 ? 1 : 2;  // 256-thread blocks: several per CU, barriers decoupled
    dim3 sgrid((sh.B + 32 * rtl - 1) / (32 * rtl), (sh.O + 64 * cw - 1) / (64 * cw));
#define X(N_, K_)                                                                               \
  if (sh.n == N_ && sh.K == K_) {                                                               \
    if (cw == 4)                                                                                \
      hipLaunchKernelGGL((pp_fwd_dense_shared_kernel<N_, K_, 4, 1>), sgrid, dim3(256), 0, st, x, \
                         wt, y, sh, tab, dc, vw);                                                   \
    else                                                                                        \
      hipLaunchKernelGGL((pp_fwd_dense_shared_kernel<N_, K_, 2, 2>), sgrid, dim3(256), 0, st, x, \
                         wt, y, sh, tab, dc, vw);                                                   \
    hipError_t e = hipGetLastError();                                                           \
    if (e == hipSuccess && seg) e = pp_launch_segment_ids(x, seg, sh, st);                      \
    return e;                                                                                   \
  }
    PP_DENSE_COMBOS(X)
#undef X
    return hipErrorInvalidValue;
  }
  dim3 grid((sh.B + 127) / 128, (sh.O + 32 * nt - 1) / (32 * nt));
#define X(N_, K_)                                                                                 \
  if (sh.n == N_ && sh.K == K_) {                                                                 \
    if (nt == 4)                                                                                  \
      hipLaunchKernelGGL((pp_fwd_dense_kernel<N_, K_, 4>), grid, dim3(256), 0, st, x, wt, y, sh, \
                         tab, dc, vw);                                                        \
    else                                                                                          \
      hipLaunchKernelGGL((pp_fwd_dense_kernel<N_, K_, 2>), grid, dim3(256), 0, st, x, wt, y, sh, \
                         tab, dc, std::min(vw, 2));                                           \
    hipError_t e = hipGetLastError();                                                             \
    /* ids: same fp32 expression (the exact reciprocal forms are bit-identical) */                \
    if (e == hipSuccess && seg) e = pp_launch_segment_ids(x, seg, sh, st);                        \
    return e;                                                                                     \
  }
  PP_DENSE_COMBOS(X)
#undef X
  return hipErrorInvalidValue;
}

hipError_t pp_launch_bwd_input_dense(const float* x, const float* gy, const float* wt, float* gx,
                                     const PPShape& sh, const BasisTab& tab, hipStream_t st) {
  if (sh.n == 1) return hipMemsetAsync(gx, 0, (size_t)sh.B * sh.I * sizeof(float), st);
  const DenseCoord dc = make_coord(sh);
  const int vw = vec_width(sh.O);
  dim3 grid((sh.B + 127) / 128, (sh.I + 31) / 32);
#define X(N_, K_)                                                                              \
  if (sh.n == N_ && sh.K == K_) {                                                              \
    const size_t lds = 2 * (size_t)K_ * 32 * kWStride * sizeof(float);                         \
    if (lds > 48 * 1024) { /* per call: the attribute is per device */                         \
      hipError_t e = hipFuncSetAttribute((const void*)pp_bwd_input_dense_kernel<N_, K_>,       \
                                         hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
      if (e != hipSuccess) return e;                                                           \
    }                                                                                          \
    hipLaunchKernelGGL((pp_bwd_input_dense_kernel<N_, K_>), grid, dim3(256), lds, st, x, gy, wt, \
                       gx, sh, tab, dc, vw);                                                   \
    return hipGetLastError();                                                                  \
  }
  PP_DENSE_COMBOS(X)
#undef X
  return hipErrorInvalidValue;
}

// gwt[idx] = scale * sum_c part[c][idx], chunks added in a fixed order
void pp_launch_chunk_reduce(const float* part, float* gwt, long total, int nchunk, float scale,
                            hipStream_t st) {
  if (total % 4 == 0) {
    const long t4 = total / 4;
    const int blocks = (int)((t4 + 63) / 64);
    hipLaunchKernelGGL((pp_dense_reduce_kernel<float4>), dim3(blocks), dim3(256), 0, st,
                       (const float4*)part, (float4*)gwt, t4, nchunk, scale);
  } else {
    const int blocks = (int)((total + 63) / 64);
    hipLaunchKernelGGL((pp_dense_reduce_kernel<float>), dim3(blocks), dim3(256), 0, st, part, gwt,
                       total, nchunk, scale);
  }
}

hipError_t pp_launch_bwd_weight_dense(const float* x, const float* gy, float* gwt, void* ws,
                                      const PPShape& sh, const BasisTab& tab, hipStream_t st) {
  const DenseCoord dc = make_coord(sh);
  const WeightPlan p = weight_plan(sh);
  dim3 grid((sh.I + 31) / 32, p.ncol, p.nchunk);
  float* out = p.nchunk > 1 ? static_cast<float*>(ws) : gwt;
  const float scale = p.nchunk > 1 ? 1.0f : sh.out_scale;
  bool launched = false;
#define X(N_, K_)                                                                              \
  if (!launched && sh.n == N_ && sh.K == K_) {                                                 \
    if (p.nw == 8)                                                                             \
      hipLaunchKernelGGL((pp_bwd_weight_dense_kernel<N_, K_, 8>), grid, dim3(512), 0, st, x, gy, \
                         out, sh, tab, dc, p.rows_per_chunk, scale);                           \
    else                                                                                       \
      hipLaunchKernelGGL((pp_bwd_weight_dense_kernel<N_, K_, 4>), grid, dim3(256), 0, st, x, gy, \
                         out, sh, tab, dc, p.rows_per_chunk, scale);                           \
    launched = true;                                                                           \
  }
  PP_DENSE_COMBOS(X)
#undef X
  if (!launched) return hipErrorInvalidValue;
  hipError_t e = hipGetLastError();
  if (e != hipSuccess || p.nchunk <= 1) return e;
  pp_launch_chunk_reduce(static_cast<const float*>(ws), gwt, (long)sh.I * sh.K * sh.O, p.nchunk,
                         sh.out_scale, st);
  return hipGetLastError();
}
