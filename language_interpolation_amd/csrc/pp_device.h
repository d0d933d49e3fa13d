// Device-side helpers: segment id, local coordinate, Lagrange basis and its
// derivative.  Every translation unit that includes this is compiled with
// -ffp-contract=off so that the A1/A2 expressions below keep the reference's
// fp32 op order (no FMA contraction); hot loops use explicit fmaf().
#pragma once
#include <hip/hip_runtime.h>

#include "pp_internal.h"

#define PP_WAVE 64

__device__ __forceinline__ float pp_readlane_f(float v, int lane) {
  return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), lane));
}
__device__ __forceinline__ int pp_readlane_i(int v, int lane) {
  return __builtin_amdgcn_readlane(v, lane);
}

// A1: t = ((x + L/2)/L)*S in fp32 in this order; s = clamp(trunc(t), 0, S-1).
// Non-finite / |t| >= 2^63 follows what `.long()` gives on x86 (cvttss2si
// "integer indefinite" = INT64_MIN, which then clamps to 0).
__device__ __forceinline__ int pp_segment(float x, float length, float half, float Sf, int S) {
  float t = ((x + half) / length) * Sf;
  int s = 0;
  if (t > -9.2e18f && t < 9.2e18f) {
    float tc = fminf(fmaxf(t, -1.0f), Sf);
    s = (int)tc;  // truncation toward zero
    s = s < 0 ? 0 : (s > S - 1 ? S - 1 : s);
  }
  return s;
}

// A2: x_min = (s/S)*2-1, x_max = ((s+1)/S)*2-1, xi = L*((x-x_min)/(x_max-x_min)) - L/2.
// Also returns dxi/dx = L/(x_max-x_min).
__device__ __forceinline__ float pp_local_coord(float x, int s, float length, float half,
                                                float Sf, float* dxi_dx) {
  float sf = (float)s;
  float x_min = (sf / Sf) * 2.0f - 1.0f;
  float x_max = ((sf + 1.0f) / Sf) * 2.0f - 1.0f;
  float width = x_max - x_min;
  if (dxi_dx) *dxi_dx = length / width;
  return length * ((x - x_min) / width) - half;
}

// A3: l_j(xi) = invden_j * prod_{m != j} (xi - X_m), via prefix/suffix products.
template <int N>
__device__ __forceinline__ void pp_lagrange(float xi, const BasisTab& tab, float (&l)[N]) {
  if constexpr (N == 1) {
    l[0] = 1.0f;
  } else {
    float d[N], pre[N], suf[N];
#pragma unroll
    for (int m = 0; m < N; ++m) d[m] = xi - tab.X[m];
    pre[0] = 1.0f;
#pragma unroll
    for (int m = 1; m < N; ++m) pre[m] = pre[m - 1] * d[m - 1];
    suf[N - 1] = 1.0f;
#pragma unroll
    for (int m = N - 2; m >= 0; --m) suf[m] = suf[m + 1] * d[m + 1];
#pragma unroll
    for (int j = 0; j < N; ++j) l[j] = (pre[j] * suf[j]) * tab.invden[j];
  }
}

// l_j and dl_j/dxi together (product rule carried through the prefix/suffix recurrences).
template <int N>
__device__ __forceinline__ void pp_lagrange_d(float xi, const BasisTab& tab, float (&l)[N],
                                              float (&dl)[N]) {
  if constexpr (N == 1) {
    l[0] = 1.0f;
    dl[0] = 0.0f;
  } else {
    float d[N], pre[N], suf[N], dpre[N], dsuf[N];
#pragma unroll
    for (int m = 0; m < N; ++m) d[m] = xi - tab.X[m];
    pre[0] = 1.0f;
    dpre[0] = 0.0f;
#pragma unroll
    for (int m = 1; m < N; ++m) {
      pre[m] = pre[m - 1] * d[m - 1];
      dpre[m] = fmaf(dpre[m - 1], d[m - 1], pre[m - 1]);
    }
    suf[N - 1] = 1.0f;
    dsuf[N - 1] = 0.0f;
#pragma unroll
    for (int m = N - 2; m >= 0; --m) {
      suf[m] = suf[m + 1] * d[m + 1];
      dsuf[m] = fmaf(dsuf[m + 1], d[m + 1], suf[m + 1]);
    }
#pragma unroll
    for (int j = 0; j < N; ++j) {
      l[j] = (pre[j] * suf[j]) * tab.invden[j];
      dl[j] = fmaf(dpre[j], suf[j], pre[j] * dsuf[j]) * tab.invden[j];
    }
  }
}

// dl_j/dxi only (the dense backward-input epilogues need no l_j): closed forms for the
// common low orders, the general recurrence otherwise.
template <int N>
__device__ __forceinline__ void pp_lagrange_dl(float xi, const BasisTab& tab, float (&dl)[N]) {
  if constexpr (N == 1) {
    dl[0] = 0.0f;
  } else if constexpr (N == 2) {
    dl[0] = tab.invden[0];
    dl[1] = tab.invden[1];
  } else if constexpr (N == 3) {
    const float d0 = xi - tab.X[0], d1 = xi - tab.X[1], d2 = xi - tab.X[2];
    dl[0] = (d1 + d2) * tab.invden[0];
    dl[1] = (d0 + d2) * tab.invden[1];
    dl[2] = (d0 + d1) * tab.invden[2];
  } else {
    float l[N];
    pp_lagrange_d<N>(xi, tab, l, dl);
  }
}

// Small fixed-width vector of floats with aligned global load/store.
template <int V>
struct PPVec;
template <>
struct PPVec<1> {
  float v[1];
  __device__ __forceinline__ void load(const float* p) { v[0] = *p; }
  __device__ __forceinline__ void store(float* p) const { *p = v[0]; }
};
template <>
struct PPVec<2> {
  float v[2];
  __device__ __forceinline__ void load(const float* p) {
    float2 t = *reinterpret_cast<const float2*>(p);
    v[0] = t.x; v[1] = t.y;
  }
  __device__ __forceinline__ void store(float* p) const {
    *reinterpret_cast<float2*>(p) = make_float2(v[0], v[1]);
  }
};
template <>
struct PPVec<4> {
  float v[4];
  __device__ __forceinline__ void load(const float* p) {
    float4 t = *reinterpret_cast<const float4*>(p);
    v[0] = t.x; v[1] = t.y; v[2] = t.z; v[3] = t.w;
  }
  __device__ __forceinline__ void store(float* p) const {
    *reinterpret_cast<float4*>(p) = make_float4(v[0], v[1], v[2], v[3]);
  }
};

// v + (v rotated right by CTRL-encoded amount inside each row of 16 lanes), one
// v_add_f32 with a DPP source modifier.
template <int CTRL>
__device__ __forceinline__ float pp_dpp_add(float v) {
  const int t = __builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xf, 0xf, false);
  return v + __int_as_float(t);
}

// max / min with the value CTRL places away inside each row of 16 lanes
template <int CTRL>
__device__ __forceinline__ float pp_dpp_max(float v) {
  const int t = __builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xf, 0xf, false);
  return fmaxf(v, __int_as_float(t));
}
template <int CTRL>
__device__ __forceinline__ int pp_dpp_min_i(int v) {
  const int t = __builtin_amdgcn_update_dpp(0, v, CTRL, 0xf, 0xf, false);
  return min(v, t);
}

// Sum over the 64 lanes, returned in every lane: four row_ror DPP adds give each
// lane its row's (16 lanes) sum; the four row sums are combined through SGPRs.
// No LDS traffic (unlike __shfl_xor, which lowers to ds_bpermute).
__device__ __forceinline__ float pp_wave_sum(float v) {
  v = pp_dpp_add<0x128>(v);  // row_ror:8
  v = pp_dpp_add<0x124>(v);  // row_ror:4
  v = pp_dpp_add<0x122>(v);  // row_ror:2
  v = pp_dpp_add<0x121>(v);  // row_ror:1
  return (pp_readlane_f(v, 0) + pp_readlane_f(v, 16)) + (pp_readlane_f(v, 32) + pp_readlane_f(v, 48));
}
