// Dense path: expanded-basis x weight contraction on the f32-input MFMA
// (v_mfma_f32_32x32x2_f32, exact fp32 fma chain), for layers whose weight count
// per (out, in) pair K is small (segments = 2 in every hidden layer of the
// BASELINE configs), i.e. where n of the K basis columns are nonzero per sample.
//
//   forward   y[b,o]      = sum_{i,k} Phi[b,i,k] * w_t[i,k,o]
//   bwd-input G[b,(i,k)]  = sum_o gy[b,o] * w_t[i,k,o];  gx[b,i] = chain * sum_k Phi'[b,i,k] G[b,i,k]
//   bwd-weight gw_t[(i,k),o] = sum_b Phi[b,i,k] * gy[b,o]
//
// Phi is never materialised: each lane evaluates segment id / local coordinate /
// Lagrange basis for "its" (row, input) pair in registers (A1-A3 of SURVEY.md 8a)
// and feeds the K basis columns to K MFMAs as the A operand.  The two k-slots of
// the 32x32x2 MFMA are two different inputs (forward) or two different rows
// (bwd-weight), so no lane computes a basis value it does not use.
#include "pp_device.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));

struct DenseCoord {
  float w2;     // 2/S            (exact when S is a power of two)
  float inv_w;  // length / (2/S) is NOT folded: xi = length*((x-x_min)*inv_w) - half
  float inv_len;
  int exact;    // length and S are powers of two: divisions are exact multiplications
};

__device__ __forceinline__ void dense_seg_xi(float x, const PPShape& sh, const DenseCoord& dc,
                                             int& s, float& xi, float& dxi_dx) {
  if (dc.exact) {
    float t = ((x + sh.half) * dc.inv_len) * sh.Sf;
    s = 0;
    if (t > -9.2e18f && t < 9.2e18f) {
      float tc = fminf(fmaxf(t, -1.0f), sh.Sf);
      s = (int)tc;
      s = s < 0 ? 0 : (s > sh.S - 1 ? sh.S - 1 : s);
    }
    const float x_min = (float)s * dc.w2 - 1.0f;  // == (s/S)*2-1 exactly
    xi = sh.length * ((x - x_min) * dc.inv_w) - sh.half;
    dxi_dx = sh.length * dc.inv_w;
  } else {
    s = pp_segment(x, sh.length, sh.half, sh.Sf, sh.S);
    xi = pp_local_coord(x, s, sh.length, sh.half, sh.Sf, &dxi_dx);
  }
}

// Phi[k] for k = 0..K-1 from the n active basis values (zeros elsewhere).
template <int N, int K>
__device__ __forceinline__ void expand_basis(const float (&l)[N], int soff, float (&phi)[K]) {
#pragma unroll
  for (int k = 0; k < K; ++k) {
    const int j = k - soff;
    float v = 0.0f;
#pragma unroll
    for (int jj = 0; jj < N; ++jj) v = (j == jj) ? l[jj] : v;
    phi[k] = v;
  }
}

// ---------------------------------------------------------------------------
// forward: wave tile = 32 rows x (NT*32) outputs, k-slots = two inputs
// ---------------------------------------------------------------------------
template <int N, int K, int NT>
__global__ __launch_bounds__(256) void pp_fwd_dense_kernel(const float* __restrict__ x,
                                                           const float* __restrict__ wt,
                                                           float* __restrict__ y,
                                                           int32_t* __restrict__ seg_out,
                                                           PPShape sh, BasisTab tab,
                                                           DenseCoord dc) {
  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  const int r = lane & 31, h = lane >> 5;
  const int b0 = (blockIdx.x * 4 + wave) * 32;
  if (b0 >= sh.B) return;
  const int o0 = blockIdx.y * (32 * NT);
  const int b = min(b0 + r, sh.B - 1);
  const bool write_seg = seg_out != nullptr && blockIdx.y == 0 && (b0 + r) < sh.B;

  int ocol[NT];
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) ocol[nt] = min(o0 + nt * 32 + r, sh.O - 1);

  f32x16 acc[NT];
#pragma unroll
  for (int nt = 0; nt < NT; ++nt)
#pragma unroll
    for (int q = 0; q < 16; ++q) acc[nt][q] = 0.0f;

  const float* xrow = x + (size_t)b * sh.I;
  for (int ip = 0; ip < sh.I; ip += 2) {
    const int i = ip + h;
    const bool valid = i < sh.I;
    const int ic = valid ? i : sh.I - 1;
    const float xv = xrow[ic];
    int s;
    float xi, dd;
    dense_seg_xi(xv, sh, dc, s, xi, dd);
    float l[N], phi[K];
    pp_lagrange<N>(xi, tab, l);
    if (!valid) {
#pragma unroll
      for (int j = 0; j < N; ++j) l[j] = 0.0f;
    }
    expand_basis<N, K>(l, s * sh.stride, phi);
    if (write_seg && valid) seg_out[(size_t)b * sh.I + i] = s;
    const float* wrow = wt + (size_t)ic * K * sh.O;
#pragma unroll
    for (int k = 0; k < K; ++k) {
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) {
        const float bv = wrow[(size_t)k * sh.O + ocol[nt]];
        acc[nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(phi[k], bv, acc[nt], 0, 0, 0);
      }
    }
  }
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) {
    const int col = o0 + nt * 32 + r;
    if (col < sh.O) {
#pragma unroll
      for (int q = 0; q < 16; ++q) {
        const int row = b0 + (q & 3) + 8 * (q >> 2) + 4 * h;
        if (row < sh.B) y[(size_t)row * sh.O + col] = acc[nt][q] * sh.out_scale;
      }
    }
  }
}

// ---------------------------------------------------------------------------
// backward-input: block tile = 128 rows x 32 inputs, wave = 32 rows; K
// accumulators (one per weight index k); the W tile [K][32 inputs][32 outs] is
// staged through LDS (row pad 33 -> conflict-free ds_read_b32).
// ---------------------------------------------------------------------------
template <int N, int K>
__global__ __launch_bounds__(256) void pp_bwd_input_dense_kernel(const float* __restrict__ x,
                                                                 const float* __restrict__ gy,
                                                                 const float* __restrict__ wt,
                                                                 float* __restrict__ gx,
                                                                 PPShape sh, BasisTab tab,
                                                                 DenseCoord dc) {
  __shared__ float wl[K * 32 * 33];
  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  const int r = lane & 31, h = lane >> 5;
  const int b0 = (blockIdx.x * 4 + wave) * 32;
  const int i0 = blockIdx.y * 32;
  const int b = min(b0 + r, sh.B - 1);
  const int nrows = min(32, sh.I - i0) * K;  // weight rows (i,k) this tile owns, contiguous in w_t

  f32x16 acc[K];
#pragma unroll
  for (int k = 0; k < K; ++k)
#pragma unroll
    for (int q = 0; q < 16; ++q) acc[k][q] = 0.0f;

  const float* grow = gy + (size_t)b * sh.O;
  const float* wbase = wt + (size_t)i0 * K * sh.O;
  for (int oc0 = 0; oc0 < sh.O; oc0 += 32) {
    __syncthreads();
    for (int idx = threadIdx.x; idx < 32 * K * 32; idx += 256) {
      const int rowid = idx >> 5, oo = idx & 31;  // rowid = ri*K + k
      const int ri = rowid / K, k = rowid - ri * K;
      float v = 0.0f;
      if (rowid < nrows && oc0 + oo < sh.O) v = wbase[(size_t)rowid * sh.O + oc0 + oo];
      wl[(k * 32 + ri) * 33 + oo] = v;
    }
    __syncthreads();
    if (b0 < sh.B) {  // wave-uniform
#pragma unroll 4
      for (int t = 0; t < 16; ++t) {
        const int oo = 2 * t + h;
        const float a = (oc0 + oo < sh.O) ? grow[oc0 + oo] : 0.0f;
#pragma unroll
        for (int k = 0; k < K; ++k) {
          const float bv = wl[(k * 32 + r) * 33 + oo];
          acc[k] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, bv, acc[k], 0, 0, 0);
        }
      }
    }
  }
  if (b0 >= sh.B) return;
  // epilogue: lane owns input column i0+r for 16 rows
  const int i = i0 + r;
  if (i < sh.I) {
#pragma unroll
    for (int q = 0; q < 16; ++q) {
      const int row = b0 + (q & 3) + 8 * (q >> 2) + 4 * h;
      if (row < sh.B) {
        const float xv = x[(size_t)row * sh.I + i];
        int s;
        float xi, dxi_dx;
        dense_seg_xi(xv, sh, dc, s, xi, dxi_dx);
        float l[N], dl[N], dphi[K];
        pp_lagrange_d<N>(xi, tab, l, dl);
        expand_basis<N, K>(dl, s * sh.stride, dphi);
        float g = 0.0f;
#pragma unroll
        for (int k = 0; k < K; ++k) g = fmaf(dphi[k], acc[k][q], g);
        gx[(size_t)row * sh.I + i] = g * (dxi_dx * sh.out_scale);
      }
    }
  }
}

// ---------------------------------------------------------------------------
// backward-weight: block = one (32 inputs x 32 outputs) tile x one row chunk;
// its 4 waves split the chunk's rows, accumulate K tiles each, and are summed
// through LDS.  k-slots = two rows.  Partials (one per row chunk) go to the
// workspace and are summed in a fixed order by pp_dense_reduce_kernel.
// ---------------------------------------------------------------------------
template <int N, int K>
__global__ __launch_bounds__(256) void pp_bwd_weight_dense_kernel(const float* __restrict__ x,
                                                                  const float* __restrict__ gy,
                                                                  float* __restrict__ out,
                                                                  PPShape sh, BasisTab tab,
                                                                  DenseCoord dc, int rows_per_chunk,
                                                                  float scale) {
  __shared__ float red[4 * 16 * 64];
  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  const int r = lane & 31, h = lane >> 5;
  const int i0 = blockIdx.x * 32;
  const int o0 = blockIdx.y * 32;
  const int chunk = blockIdx.z;
  const int cbeg = chunk * rows_per_chunk;
  const int cend = min(sh.B, cbeg + rows_per_chunk);
  // wave w takes rows cbeg + 2*(w + 4*t) + h
  const int i = i0 + r;
  const bool ivalid = i < sh.I;
  const int ic = ivalid ? i : sh.I - 1;
  const int oc = min(o0 + r, sh.O - 1);

  f32x16 acc[K];
#pragma unroll
  for (int k = 0; k < K; ++k)
#pragma unroll
    for (int q = 0; q < 16; ++q) acc[k][q] = 0.0f;

  for (int bb = cbeg + 2 * wave; bb < cend; bb += 8) {
    const int b = bb + h;
    const bool bvalid = b < cend;
    const int bc = bvalid ? b : cend - 1;
    const float xv = x[(size_t)bc * sh.I + ic];
    const float gv = bvalid ? gy[(size_t)bc * sh.O + oc] : 0.0f;
    int s;
    float xi, dd;
    dense_seg_xi(xv, sh, dc, s, xi, dd);
    float l[N], phi[K];
    pp_lagrange<N>(xi, tab, l);
    expand_basis<N, K>(l, s * sh.stride, phi);
#pragma unroll
    for (int k = 0; k < K; ++k)
      acc[k] = __builtin_amdgcn_mfma_f32_32x32x2f32(phi[k], gv, acc[k], 0, 0, 0);
  }
  // cross-wave sum through LDS, one k at a time; wave 0 writes
  float* dst = out + (size_t)chunk * sh.I * K * sh.O;
#pragma unroll
  for (int k = 0; k < K; ++k) {
    __syncthreads();
#pragma unroll
    for (int q = 0; q < 16; ++q) red[(wave * 16 + q) * 64 + lane] = acc[k][q];
    __syncthreads();
    if (wave == 0) {
#pragma unroll
      for (int q = 0; q < 16; ++q) {
        const float v = (red[(0 * 16 + q) * 64 + lane] + red[(1 * 16 + q) * 64 + lane]) +
                        (red[(2 * 16 + q) * 64 + lane] + red[(3 * 16 + q) * 64 + lane]);
        const int row = i0 + (q & 3) + 8 * (q >> 2) + 4 * h;  // input index
        const int col = o0 + r;
        if (row < sh.I && col < sh.O) dst[((size_t)row * K + k) * sh.O + col] = v * scale;
      }
    }
  }
}

__global__ __launch_bounds__(256) void pp_dense_reduce_kernel(const float* __restrict__ part,
                                                              float* __restrict__ gwt, long total,
                                                              int nchunk, float scale) {
  long idx = (long)blockIdx.x * 256 + threadIdx.x;
  const long step = (long)gridDim.x * 256;
  for (; idx < total; idx += step) {
    float s = 0.0f;
    for (int c = 0; c < nchunk; ++c) s += part[(size_t)c * total + idx];
    gwt[idx] = s * scale;
  }
}

DenseCoord make_coord(const PPShape& sh) {
  DenseCoord dc;
  int e;
  const bool len_pow2 = std::frexp(sh.length, &e) == 0.5f;
  const bool s_pow2 = (sh.S & (sh.S - 1)) == 0;
  dc.exact = (len_pow2 && s_pow2) ? 1 : 0;
  dc.w2 = 2.0f / (float)sh.S;
  dc.inv_w = (float)sh.S / 2.0f;
  dc.inv_len = 1.0f / sh.length;
  return dc;
}

struct WeightPlan {
  int nchunk, rows_per_chunk;
};

WeightPlan weight_plan(const PPShape& sh) {
  const int tiles = ((sh.I + 31) / 32) * ((sh.O + 31) / 32);
  // aim for >= 2 blocks per CU, at least 64 rows per wave
  int nchunk = (512 + tiles - 1) / tiles;
  const int max_chunks = std::max(1, sh.B / 256);
  nchunk = std::max(1, std::min(nchunk, max_chunks));
  int rpc = (sh.B + nchunk - 1) / nchunk;
  rpc = (rpc + 7) & ~7;
  nchunk = (sh.B + rpc - 1) / rpc;
  return {nchunk, rpc};
}

// (N, K) pairs with K <= 8 that a continuous or discontinuous layer can produce
#define PP_DENSE_COMBOS(X) \
  X(1, 2) X(1, 3) X(1, 4) X(1, 5) X(1, 6) X(1, 7) X(1, 8) \
  X(2, 3) X(2, 4) X(2, 5) X(2, 6) X(2, 7) X(2, 8) \
  X(3, 5) X(3, 6) X(3, 7) X(4, 7) X(4, 8)

}  // namespace

bool pp_dense_supported(const PPShape& sh) {
#define X(N_, K_) if (sh.n == N_ && sh.K == K_) return true;
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
  constexpr int NT = 2;
  dim3 grid((sh.B + 127) / 128, (sh.O + 32 * NT - 1) / (32 * NT));
#define X(N_, K_)                                                                              \
  if (sh.n == N_ && sh.K == K_) {                                                              \
    hipLaunchKernelGGL((pp_fwd_dense_kernel<N_, K_, NT>), grid, dim3(256), 0, st, x, wt, y, seg, \
                       sh, tab, dc);                                                           \
    return hipGetLastError();                                                                  \
  }
  PP_DENSE_COMBOS(X)
#undef X
  return hipErrorInvalidValue;
}

hipError_t pp_launch_bwd_input_dense(const float* x, const float* gy, const float* wt, float* gx,
                                     const PPShape& sh, const BasisTab& tab, hipStream_t st) {
  if (sh.n == 1) return hipMemsetAsync(gx, 0, (size_t)sh.B * sh.I * sizeof(float), st);
  const DenseCoord dc = make_coord(sh);
  dim3 grid((sh.B + 127) / 128, (sh.I + 31) / 32);
#define X(N_, K_)                                                                               \
  if (sh.n == N_ && sh.K == K_) {                                                               \
    hipLaunchKernelGGL((pp_bwd_input_dense_kernel<N_, K_>), grid, dim3(256), 0, st, x, gy, wt, gx, \
                       sh, tab, dc);                                                            \
    return hipGetLastError();                                                                   \
  }
  PP_DENSE_COMBOS(X)
#undef X
  return hipErrorInvalidValue;
}

hipError_t pp_launch_bwd_weight_dense(const float* x, const float* gy, float* gwt, void* ws,
                                      const PPShape& sh, const BasisTab& tab, hipStream_t st) {
  const DenseCoord dc = make_coord(sh);
  const WeightPlan p = weight_plan(sh);
  dim3 grid((sh.I + 31) / 32, (sh.O + 31) / 32, p.nchunk);
  float* out = p.nchunk > 1 ? static_cast<float*>(ws) : gwt;
  const float scale = p.nchunk > 1 ? 1.0f : sh.out_scale;
  bool launched = false;
#define X(N_, K_)                                                                                \
  if (!launched && sh.n == N_ && sh.K == K_) {                                                   \
    hipLaunchKernelGGL((pp_bwd_weight_dense_kernel<N_, K_>), grid, dim3(256), 0, st, x, gy, out, \
                       sh, tab, dc, p.rows_per_chunk, scale);                                    \
    launched = true;                                                                             \
  }
  PP_DENSE_COMBOS(X)
#undef X
  if (!launched) return hipErrorInvalidValue;
  hipError_t e = hipGetLastError();
  if (e != hipSuccess || p.nchunk <= 1) return e;
  const long total = (long)sh.I * sh.K * sh.O;
  const int blocks = (int)std::min<long>((total + 255) / 256, 4096);
  hipLaunchKernelGGL(pp_dense_reduce_kernel, dim3(blocks), dim3(256), 0, st,
                     (const float*)ws, gwt, total, p.nchunk, sh.out_scale);
  return hipGetLastError();
}
