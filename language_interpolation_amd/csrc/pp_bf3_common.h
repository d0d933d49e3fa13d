// Shared pieces of the bf16x3 kernels (pp_bf3_fwd.hip, pp_bf3_bwdx.hip, pp_bf3_bwdw.hip):
// the exact 3-way bf16 split of an fp32 value, the expanded basis, vector types.
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

#define PP_BF3_COMBOS(X) \
  X(1, 1) X(1, 2) X(1, 4) X(1, 8) \
  X(2, 2) X(2, 3) X(2, 4) X(2, 5) X(2, 8) \
  X(3, 3) X(3, 5) X(3, 6) X(4, 4) X(4, 7) X(4, 8) \
  X(5, 5) X(6, 6) X(7, 7) X(8, 8)

}  // namespace
#endif
