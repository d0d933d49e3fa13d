// Shared pieces of the split-operand matrix-core kernels (pp_bf3_fwd.hip, pp_bf3_bwdx.hip,
// pp_bf3_bwdw.hip): the exact splits of an fp32 value into 16-bit pieces (3 x bf16 or
// 2 x fp16), the expanded basis, vector types.
#ifndef PP_BF3_COMMON_H
#define PP_BF3_COMMON_H
#include <algorithm>
#include <cstdlib>
#include <type_traits>
#include <utility>

#include "pp_device.h"

namespace {


typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
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

// ---- operand-split schemes ------------------------------------------------------------
// A scheme splits fp32 values into PL planes of 16-bit pieces (two values -> one packed
// dword per plane, first value in the low half) and lists the NC cross terms
// (plane of A, plane of B) whose MFMAs reproduce the fp32 product, smallest first.
//
// SchemeBf3: three bf16 pieces (8 significant bits each, fp32's exponent range, exact by
//   truncation), six cross terms, error ~2^-23 relative.  No range restrictions.
// SchemeH2:  two fp16 pieces (11 significant bits each, round-toward-zero so residuals are
//   exact), three cross terms a2b1 + a1b2 + a1b1, error ~2^-20 relative to the LARGEST
//   values of the operand: fp16 has a 5-bit exponent, so an operand must be scaled by a
//   power of two into fp16's range first (weights / gradients: per tensor or per row, see
//   the kernels), pieces below 2^-24 vanish, and |values| >= 65504 saturate -- the kernels
//   detect that on the basis operand and recompute the tile in fp32.
struct SchemeBf3 {
  static constexpr int PL = 3, NC = 6;
  using frag = bf16x8;
  static constexpr bool kRanged = false;
  __device__ static constexpr int pa(int c) { return c == 0 ? 2 : (c == 2 || c == 3) ? 1 : 0; }
  __device__ static constexpr int pb(int c) { return c == 1 ? 2 : (c == 2 || c == 4) ? 1 : 0; }
  __device__ __forceinline__ static void split_pair(float a, float c, unsigned (&w)[3]) {
    unsigned a1, a2, a3, c1, c2, c3;
    split3(a, a1, a2, a3);
    split3(c, c1, c2, c3);
    w[0] = (a1 >> 16) | c1;
    w[1] = (a2 >> 16) | c2;
    w[2] = (a3 >> 16) | c3;
  }
  __device__ __forceinline__ static f32x16 mfma(frag a, frag b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
  }
};

struct SchemeH2 {
  static constexpr int PL = 2, NC = 3;
  using frag = f16x8;
  static constexpr bool kRanged = true;
  __device__ static constexpr int pa(int c) { return c == 0 ? 1 : 0; }
  __device__ static constexpr int pb(int c) { return c == 1 ? 1 : 0; }
  __device__ __forceinline__ static void split_pair(float a, float c, unsigned (&w)[2]) {
    const auto h1 = __builtin_amdgcn_cvt_pkrtz(a, c);
    const float ra = a - (float)h1[0], rc = c - (float)h1[1];
    const auto h2 = __builtin_amdgcn_cvt_pkrtz(ra, rc);
    w[0] = __builtin_bit_cast(unsigned, h1);
    w[1] = __builtin_bit_cast(unsigned, h2);
  }
  __device__ __forceinline__ static f32x16 mfma(frag a, frag b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
  }
};
constexpr float kH2Limit = 60000.0f;  // basis values beyond this do not fit fp16 pieces

// 16 values -> two 16-byte fragments per plane (values 0..7 and 8..15)
template <class SC>
__device__ __forceinline__ void split16(const float (&v)[16], u32x4 (&lo)[SC::PL],
                                        u32x4 (&hi)[SC::PL]) {
  unsigned w[8][SC::PL];
#pragma unroll
  for (int j = 0; j < 8; ++j) SC::split_pair(v[2 * j], v[2 * j + 1], w[j]);
#pragma unroll
  for (int p = 0; p < SC::PL; ++p) {
    lo[p] = u32x4{w[0][p], w[1][p], w[2][p], w[3][p]};
    hi[p] = u32x4{w[4][p], w[5][p], w[6][p], w[7][p]};
  }
}

// Power-of-two scale that brings a tensor whose largest |value| has the fp32 bit pattern
// `absmax_bits` to [2^11, 2^12) (well inside fp16, 22 bits of pieces above 2^-24 for every
// value within 2^-13 of the largest), and its inverse.  absmax 0 / non-finite -> 1.
__host__ __device__ __forceinline__ void h2_scale(unsigned absmax_bits, float& s, float& inv) {
  const int ex = (int)((absmax_bits >> 23) & 0xffu);
  int se = 138 - ex;  // unbiased exponent of the scale
  if (ex == 0 || ex == 255) se = 0;
  se = se > 120 ? 120 : (se < -120 ? -120 : se);
  union { unsigned u; float f; } a, b;
  a.u = (unsigned)(127 + se) << 23;
  b.u = (unsigned)(127 - se) << 23;
  s = a.f;
  inv = b.f;
}

// workspace header of the ranged (fp16) kernels: 64 partial maxima of the tensor's |values|
constexpr size_t kWsHeader = 256;

template <int N, int K>
struct Geom3 {
  static constexpr bool disc = (K % N == 0) && (((K / N) & (K / N - 1)) == 0);
  static constexpr int stride = disc ? N : N - 1;
  static constexpr int S = disc ? K / N : (N > 1 ? (K - 1) / (N - 1) : K);
};

template <int N, int K>
__device__ __forceinline__ void phi3(float x, bool valid, const PPShape& sh, const BasisTab& tab,
                                     const DenseCoord3& dc, float (&phi)[K], float* lmax = nullptr) {
  const float t = ((x + sh.half) * dc.inv_len) * sh.Sf;
  const float tc = __builtin_amdgcn_fmed3f(t, -1.0f, sh.Sf);
  int si = (int)tc;
  si = min(max(si, 0), sh.S - 1);
  const int s = (__builtin_fabsf(t) < 9.2e18f) ? si : 0;
  const float x_min = (float)s * dc.w2 - 1.0f;
  const float xi = sh.length * ((x - x_min) * dc.inv_w) - sh.half;
  float l[N];
  pp_lagrange<N>(xi, tab, l);
  if (lmax) {  // largest |basis value| seen (NaN counts as too large)
#pragma unroll
    for (int j = 0; j < N; ++j) {
      const float a = __builtin_fabsf(l[j]);
      *lmax = (a <= *lmax) ? *lmax : (a == a ? a : __builtin_inff());
    }
  }
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

// compile-time loop: f(integral_constant<int, 0>) ... f(integral_constant<int, N-1>)
template <int... T, class F>
__device__ __forceinline__ void static_for_impl(std::integer_sequence<int, T...>, F&& f) {
  (f(std::integral_constant<int, T>{}), ...);
}
template <int N, class F>
__device__ __forceinline__ void static_for(F&& f) {
  static_for_impl(std::make_integer_sequence<int, N>{}, f);
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

// largest |v| of a tensor as fp32 bit patterns, one partial per block into part[0..63]
// (always launched with kAbsmaxBlocks blocks: no zero-fill, no atomics; consumers take the
// max of the 64 partials with scalar loads)
constexpr int kAbsmaxBlocks = 64;
__device__ __forceinline__ void absmax_body(const float* __restrict__ v, long n,
                                            unsigned* __restrict__ part) {
  __shared__ unsigned wm[4];
  unsigned m = 0;
  const long n4 = n >> 2;
  const float4* v4 = reinterpret_cast<const float4*>(v);
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long)kAbsmaxBlocks * 256) {
    const float4 t = v4[i];
    m = max(max(m, __float_as_uint(t.x) & 0x7fffffffu), __float_as_uint(t.y) & 0x7fffffffu);
    m = max(max(m, __float_as_uint(t.z) & 0x7fffffffu), __float_as_uint(t.w) & 0x7fffffffu);
  }
  if (blockIdx.x == 0 && threadIdx.x < (n & 3))
    m = max(m, __float_as_uint(v[(n4 << 2) + threadIdx.x]) & 0x7fffffffu);
#pragma unroll
  for (int d = 32; d > 0; d >>= 1) m = max(m, (unsigned)__shfl_xor((int)m, d));
  if ((threadIdx.x & 63) == 0) wm[threadIdx.x >> 6] = m;
  __syncthreads();
  if (threadIdx.x == 0) part[blockIdx.x] = max(max(wm[0], wm[1]), max(wm[2], wm[3]));
}
__global__ __launch_bounds__(256) void pp_absmax_kernel(const float* __restrict__ v, long n,
                                                        unsigned* __restrict__ part) {
  absmax_body(v, n, part);
}

__device__ __forceinline__ unsigned absmax_of_parts(const void* hdr) {
  const unsigned* part = static_cast<const unsigned*>(hdr);
  unsigned m = 0;
#pragma unroll
  for (int i = 0; i < kAbsmaxBlocks; ++i) m = max(m, part[i]);
  return m;
}

// P[plane][k][step][out][16]: 16-bit pieces of w_t[i = 16*step + j][k][out] (times the
// tensor's power-of-two scale for a ranged scheme).  One thread per (k, step, out): 16
// reads coalesced over `out`, two 16-byte stores per plane.  ws = header | P.
template <class SC>
__device__ __forceinline__ void pack_w_body(long id, const float* __restrict__ wt,
                                            void* __restrict__ ws, int I, int K, int O, int nstep) {
  const long per_plane = (long)K * nstep * O;  // in (k, step, out) slots of 32 bytes
  if (id >= per_plane) return;
  float scale = 1.0f, inv;
  if constexpr (SC::kRanged) h2_scale(absmax_of_parts(ws), scale, inv);
  u32x4* P = reinterpret_cast<u32x4*>(static_cast<char*>(ws) + kWsHeader);
  const int o = (int)(id % O);
  const int s = (int)((id / O) % nstep);
  const int k = (int)(id / ((long)O * nstep));
  float v[16];
#pragma unroll
  for (int j = 0; j < 16; ++j) {
    const int i = 16 * s + j;
    v[j] = i < I ? wt[((size_t)i * K + k) * O + o] * scale : 0.0f;
  }
  u32x4 lo[SC::PL], hi[SC::PL];
  split16<SC>(v, lo, hi);
#pragma unroll
  for (int p = 0; p < SC::PL; ++p) {
    P[(p * per_plane + id) * 2] = lo[p];
    P[(p * per_plane + id) * 2 + 1] = hi[p];
  }
}

// Q[plane][ostep][k][Ipad][16 outs] at ws + q_off: the backward-input weight layout.
template <class SC>
__device__ __forceinline__ void pack_wo_body(long id, const float* __restrict__ wt,
                                             void* __restrict__ ws, unsigned q_off, int I, int K,
                                             int O, int nso, int Ipad) {
  // one thread per (i, k, ostep), ostep fastest: consecutive threads read consecutive
  // 64-byte pieces of one weight row.  ws = header | Q.
  if (id >= (long)Ipad * K * nso) return;
  float scale = 1.0f, inv;
  if constexpr (SC::kRanged) h2_scale(absmax_of_parts(ws), scale, inv);
  u32x4* Q = reinterpret_cast<u32x4*>(static_cast<char*>(ws) + q_off);
  const int t = (int)(id % nso);
  const int k = (int)((id / nso) % K);
  const int i = (int)(id / ((long)nso * K));
  float v[16];
#pragma unroll
  for (int j = 0; j < 16; ++j) {
    const int o = 16 * t + j;
    v[j] = (i < I && o < O) ? wt[((size_t)i * K + k) * O + o] * scale : 0.0f;
  }
  u32x4 lo[SC::PL], hi[SC::PL];
  split16<SC>(v, lo, hi);
  const long per_plane = (long)nso * K * Ipad;  // slots of 32 bytes
  const long slot = ((long)t * K + k) * Ipad + i;
#pragma unroll
  for (int p = 0; p < SC::PL; ++p) {
    Q[(p * per_plane + slot) * 2] = lo[p];
    Q[(p * per_plane + slot) * 2 + 1] = hi[p];
  }
}

// developer builds (-DPP_TIMING): wall-clock stamps per block, slot 0..7
#ifdef PP_TIMING
__device__ unsigned long long pp_dbg_times[8 * 8192];
#define PP_TICK(slot)                                                                   \
  if (threadIdx.x == 0 && blockIdx.y == 0 && blockIdx.x < 8192)                         \
    pp_dbg_times[blockIdx.x * 8 + (slot)] = wall_clock64();
__device__ unsigned long long pp_dbg_steps[32 * 8192];
// even slots: wall clock (100 MHz); odd slots: shader clock -- their ratio is the clock the
// SIMD actually ran at between two stamps
#define PP_STEP_TICK(idx)                                                               \
  if (threadIdx.x == 0 && blockIdx.y == 0 && blockIdx.x < 8192 && (idx) < 16) {         \
    pp_dbg_steps[blockIdx.x * 32 + 2 * (idx)] = wall_clock64();                         \
    pp_dbg_steps[blockIdx.x * 32 + 2 * (idx) + 1] = clock64();                          \
  }
#else
#define PP_TICK(slot)
#define PP_STEP_TICK(idx)
#endif

#define PP_BF3_COMBOS(X) \
  X(1, 1) X(1, 2) X(1, 4) X(1, 8) \
  X(2, 2) X(2, 3) X(2, 4) X(2, 5) X(2, 8) \
  X(3, 3) X(3, 5) X(3, 6) X(4, 4) X(4, 7) X(4, 8) \
  X(5, 5) X(6, 6) X(7, 7) X(8, 8)

}  // namespace
#endif
