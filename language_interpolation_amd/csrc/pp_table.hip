// Table form of a layer whose inputs lie on a uniform grid (round 3).
//
// The first layer of every reference network sees characters: x = (c - 63.5)/64 for a byte c
// (single_text_dataset.py:15-16), i.e. one of P = 128 points x_c = (c + 1/2) * 2/P - 1.  On such
// inputs the layer y[b,:] = sum_i sum_j l_j(xi(x)) * w_t[i, s(x)*stride + j, :] is exactly the
// piecewise-CONSTANT layer (discontinuous, n = 1, P segments: the segment id of x_c is c, exact
// in fp32) on the table
//     T[i][c][:] = out_scale * sum_j l_j(xi(x_c)) * w_t[i, s(x_c)*stride + j, :],
// so the forward gathers ONE table row per (row, input) instead of n, and the weight gradient is
// one indexed add per (row, input) into GT[i][c][:] = sum_{b: x[b,i] = x_c} g[b,:], mapped back by
//     gw_t[i][k][:] = out_scale * sum_{c, j: s(x_c)*stride + j = k} l_j(xi(x_c)) * GT[i][c][:].
// The two kernels here build T and map GT back; the table layer itself runs on the library's
// own n = 1 kernels (pp_forward_*, pp_backward_weight_f32 with n = 1, S = P, discontinuous).
// Whether the inputs are on the grid is the CALLER's statement (the library takes float x and
// does not look): language_interpolation_amd.layers.*.assume_input_grid().
#include "pp_device.h"

namespace {

__device__ __forceinline__ float grid_point(int c, int P) {
  return ((float)c + 0.5f) * (2.0f / (float)P) - 1.0f;  // exact for P a power of two <= 2^22
}

template <int N>
__global__ __launch_bounds__(256) void pp_grid_table_kernel(const float* __restrict__ wt,
                                                            float* __restrict__ T, PPShape sh,
                                                            BasisTab tab, int P) {
  const int ic = blockIdx.x;  // i * P + c
  const int i = ic / P, c = ic - i * P;
  const float x = grid_point(c, P);
  const int s = pp_segment(x, sh.length, sh.half, sh.Sf, sh.S);
  const float xi = pp_local_coord(x, s, sh.length, sh.half, sh.Sf, nullptr);
  float l[N];
  pp_lagrange<N>(xi, tab, l);
  const float* w0 = wt + ((size_t)i * sh.K + (size_t)s * sh.stride) * sh.O;
  for (int o = threadIdx.x; o < sh.O; o += 256) {
    float acc = 0.0f;
#pragma unroll
    for (int j = 0; j < N; ++j) acc = fmaf(l[j], w0[(size_t)j * sh.O + o], acc);
    T[(size_t)ic * sh.O + o] = acc * sh.out_scale;
  }
}

// block = one weight row (i, k).  The grid points that touch it (at most a few when S is near
// P, up to P/S per node otherwise) are listed once, in ascending order, by the block's first
// wave; the columns then walk that short list.  (First version: every thread walked all P
// coefficients in LDS for every column -- 70 us for the 16 -> 1000 layer, all of it LDS latency.)
template <int N>
__global__ __launch_bounds__(256) void pp_grid_table_grad_kernel(const float* __restrict__ GT,
                                                                 float* __restrict__ gwt, PPShape sh,
                                                                 BasisTab tab, int P) {
  __shared__ float coef[128];
  __shared__ int cidx[128];
  __shared__ int count;
  const int ik = blockIdx.x;  // i * K + k
  const int i = ik / sh.K, k = ik - i * sh.K;
  if (threadIdx.x < 64) {  // one wave: two grid points a lane, compacted in ascending order
    int base = 0;
#pragma unroll
    for (int half = 0; half < 2; ++half) {
      const int c = half * 64 + (int)threadIdx.x;
      float v = 0.0f;
      if (c < P) {
        const float x = grid_point(c, P);
        const int s = pp_segment(x, sh.length, sh.half, sh.Sf, sh.S);
        const int j = k - s * sh.stride;
        if (j >= 0 && j < N) {
          const float xi = pp_local_coord(x, s, sh.length, sh.half, sh.Sf, nullptr);
          float l[N];
          pp_lagrange<N>(xi, tab, l);
#pragma unroll
          for (int m = 0; m < N; ++m) v = (m == j) ? l[m] : v;
        }
      }
      const unsigned long long hit = __ballot(v != 0.0f);
      if (v != 0.0f) {
        const int pos = base + __popcll(hit & ((1ull << threadIdx.x) - 1ull));
        coef[pos] = v;
        cidx[pos] = c;
      }
      base += __popcll(hit);
    }
    if (threadIdx.x == 0) count = base;
  }
  __syncthreads();
  const int m = count;
  const float* g0 = GT + (size_t)i * P * sh.O;
  for (int o = threadIdx.x; o < sh.O; o += 256) {
    float acc = 0.0f;
    for (int t = 0; t < m; ++t) acc = fmaf(coef[t], g0[(size_t)cidx[t] * sh.O + o], acc);
    gwt[(size_t)ik * sh.O + o] = acc * sh.out_scale;
  }
}

}  // namespace

bool pp_grid_table_supported(const PPShape& sh, int P) {
  // P a power of two (exact grid points), one byte per table row and one part of register
  // accumulators in the n = 1 weight gradient; the local coordinate is tied to [-1, 1]
  return sh.length == 2.0f && P >= 2 && P <= 128 && (P & (P - 1)) == 0;
}

hipError_t pp_launch_grid_table(const float* wt, float* T, const PPShape& sh, const BasisTab& tab,
                                int P, hipStream_t st) {
#define X(N_)                                                                                  \
  case N_:                                                                                     \
    hipLaunchKernelGGL(pp_grid_table_kernel<N_>, dim3(sh.I * P), dim3(256), 0, st, wt, T, sh, tab, P); \
    break;
  switch (sh.n) {
    X(1) X(2) X(3) X(4) X(5) X(6) X(7) X(8)
    default: return hipErrorInvalidValue;
  }
#undef X
  return hipGetLastError();
}

hipError_t pp_launch_grid_table_grad(const float* GT, float* gwt, const PPShape& sh,
                                     const BasisTab& tab, int P, hipStream_t st) {
#define X(N_)                                                                                  \
  case N_:                                                                                     \
    hipLaunchKernelGGL(pp_grid_table_grad_kernel<N_>, dim3(sh.I * sh.K), dim3(256), 0, st, GT, gwt, \
                       sh, tab, P);                                                            \
    break;
  switch (sh.n) {
    X(1) X(2) X(3) X(4) X(5) X(6) X(7) X(8)
    default: return hipErrorInvalidValue;
  }
#undef X
  return hipGetLastError();
}
