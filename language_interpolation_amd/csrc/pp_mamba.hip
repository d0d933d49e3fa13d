// The pieces of the reference's Mamba block around the projections and the scan
// (/root/reference/language_interpolation/state_space_network.py):
//   causal depthwise conv1d + SiLU   (:315-324  conv1d(x^T)[:, :, :l]^T, then F.silu)
//   y = scan_out * silu(res)         (:326      the gate)
//   RMSNorm                          (:423-433  x * rsqrt(mean(x^2) + eps) * weight)
// each as one streaming kernel per direction (the torch formulations are 8 / 2 / 5 elementwise
// kernels forward and about twice that backward, on [b*l, d] tensors of a few MB: all launch
// latency).  Channels are the contiguous dimension; inputs that are column slices of a wider
// tensor (x and res are the two halves of in_proj's output) come with their row stride.
// Per-channel parameter gradients (conv weight / bias, norm weight) are summed in two stages --
// per-chunk partial sums, then one small kernel -- in a fixed order: no atomics.
#include "pp_device.h"

namespace {

constexpr int kMaxConvK = 8;

__device__ __forceinline__ float silu(float v) { return v / (1.0f + __expf(-v)); }
// d silu / dv = s + v*s*(1-s), s = sigmoid(v)
__device__ __forceinline__ float dsilu(float v) {
  const float s = 1.0f / (1.0f + __expf(-v));
  return s * (1.0f + v * (1.0f - s));
}

// conv[b,t,c] = bias[c] + sum_j w[c][j] * x[b, t-(k-1)+j, c]   (x before t = 0 is 0);  y = silu(conv)
__global__ __launch_bounds__(256) void pp_conv_silu_fwd_kernel(const float* __restrict__ x, int ldx,
                                                               const float* __restrict__ w,
                                                               const float* __restrict__ bias,
                                                               float* __restrict__ conv,
                                                               float* __restrict__ y, int b, int l,
                                                               int d, int k) {
  const long idx = (long)blockIdx.x * 256 + threadIdx.x;
  if (idx >= (long)b * l * d) return;
  const int c = (int)(idx % d);
  const long row = idx / d;
  const int t = (int)(row % l);
  float acc = bias ? bias[c] : 0.0f;
  for (int j = 0; j < k; ++j) {
    const int tt = t - (k - 1) + j;
    if (tt >= 0) acc = fmaf(w[c * k + j], x[(row - (k - 1) + j) * ldx + c], acc);
  }
  conv[idx] = acc;
  y[idx] = silu(acc);
}

// One thread per channel walks the sequences of its chunk: gc = gy * silu'(conv);
// gx[t] = sum_m w[k-1-m] * gc[t+m];  gw[j] += gc[t] * x[t-(k-1)+j];  gb += gc[t].
__global__ __launch_bounds__(64) void pp_conv_silu_bwd_kernel(
    const float* __restrict__ x, int ldx, const float* __restrict__ w,
    const float* __restrict__ conv, const float* __restrict__ gy, float* __restrict__ gx,
    float* __restrict__ part, int b, int l, int d, int k, int seq_per_chunk) {
  const int c = blockIdx.x * 64 + threadIdx.x;
  if (c >= d) return;
  const int chunk = blockIdx.y;
  float wr[kMaxConvK], gw[kMaxConvK], gb = 0.0f;  // wr[m] = w[k-1-m]
#pragma unroll
  for (int j = 0; j < kMaxConvK; ++j) {
    wr[j] = j < k ? w[c * k + (k - 1 - j)] : 0.0f;
    gw[j] = 0.0f;
  }
  const int s0 = chunk * seq_per_chunk, s1 = min(b, s0 + seq_per_chunk);
  for (int s = s0; s < s1; ++s) {
    const long base = (long)s * l;
    float win[kMaxConvK];  // gc[t .. t+k-1]
#pragma unroll
    for (int j = 0; j < kMaxConvK; ++j) win[j] = 0.0f;
    for (int t = l - 1; t >= 0; --t) {
      const long r = base + t;
      const float g = gy[r * d + c] * dsilu(conv[r * d + c]);
#pragma unroll
      for (int j = kMaxConvK - 1; j > 0; --j) win[j] = win[j - 1];
      win[0] = g;
      float a = 0.0f;
#pragma unroll
      for (int m = 0; m < kMaxConvK; ++m) a = fmaf(wr[m], win[m], a);
      gx[r * d + c] = a;
      gb += g;
#pragma unroll
      for (int j = 0; j < kMaxConvK; ++j) {
        const int tt = t - (k - 1) + j;
        if (j < k && tt >= 0) gw[j] = fmaf(g, x[(base + tt) * ldx + c], gw[j]);
      }
    }
  }
  float* p = part + (size_t)chunk * (kMaxConvK + 1) * d;
#pragma unroll
  for (int j = 0; j < kMaxConvK; ++j) p[j * d + c] = gw[j];
  p[kMaxConvK * d + c] = gb;
}

// sum over n partial rows for 64 consecutive entries per block: 4 thread groups take every 4th
// row with 8 loads in flight each, then add up through LDS (fixed order)
__device__ __forceinline__ float partial_sum_64(const float* __restrict__ part, size_t row_stride,
                                                int n, bool live) {
  __shared__ float sm[3][64];
  const int lane = threadIdx.x & 63, q = threadIdx.x >> 6;
  float s = 0.0f;
  if (live) {
    int i = q;
    for (; i + 28 < n; i += 32) {
      float v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) v[u] = part[(size_t)(i + 4 * u) * row_stride];
#pragma unroll
      for (int u = 0; u < 8; ++u) s += v[u];
    }
    for (; i < n; i += 4) s += part[(size_t)i * row_stride];
  }
  if (q > 0) sm[q - 1][lane] = s;
  __syncthreads();
  if (q == 0) s = ((s + sm[0][lane]) + sm[1][lane]) + sm[2][lane];
  return s;  // valid in the first 64 threads
}

__global__ __launch_bounds__(256) void pp_conv_param_reduce_kernel(const float* __restrict__ part,
                                                                   float* __restrict__ gw,
                                                                   float* __restrict__ gb, int d,
                                                                   int k, int nchunk) {
  const int idx = blockIdx.x * 64 + (threadIdx.x & 63);
  const bool live = idx < (k + 1) * d;
  const int j = live ? idx / d : 0, c = live ? idx - j * d : 0;
  const int slot = j < k ? j : kMaxConvK;
  const float s = partial_sum_64(part + (size_t)slot * d + c, (size_t)(kMaxConvK + 1) * d, nchunk,
                                 live);
  if (threadIdx.x < 64 && live) {
    if (j < k) gw[c * k + j] = s;
    else if (gb) gb[c] = s;
  }
}

// y = s * silu(r)
__global__ __launch_bounds__(256) void pp_silu_gate_fwd_kernel(const float* __restrict__ s,
                                                               const float* __restrict__ r, int ldr,
                                                               float* __restrict__ y, long rows,
                                                               int d) {
  const long idx = (long)blockIdx.x * 256 + threadIdx.x;
  if (idx >= rows * d) return;
  const long row = idx / d;
  const int c = (int)(idx - row * d);
  y[idx] = s[idx] * silu(r[row * ldr + c]);
}

__global__ __launch_bounds__(256) void pp_silu_gate_bwd_kernel(const float* __restrict__ s,
                                                               const float* __restrict__ r, int ldr,
                                                               const float* __restrict__ gy,
                                                               float* __restrict__ gs,
                                                               float* __restrict__ gr, long rows,
                                                               int d) {
  const long idx = (long)blockIdx.x * 256 + threadIdx.x;
  if (idx >= rows * d) return;
  const long row = idx / d;
  const int c = (int)(idx - row * d);
  const float rv = r[row * ldr + c], g = gy[idx];
  gs[idx] = g * silu(rv);
  gr[idx] = g * s[idx] * dsilu(rv);
}

// RMSNorm: one wave per row, the row in registers (CH chunks of 64 lanes, D <= 64*CH).
template <int CH>
__global__ __launch_bounds__(256) void pp_rms_norm_fwd_kernel(const float* __restrict__ x,
                                                              const float* __restrict__ w,
                                                              float* __restrict__ y,
                                                              float* __restrict__ rstd, long rows,
                                                              int D, float eps) {
  const int lane = threadIdx.x & 63;
  const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  float v[CH], q = 0.0f;
#pragma unroll
  for (int c = 0; c < CH; ++c) {
    const int e = c * 64 + lane;
    v[c] = e < D ? x[row * D + e] : 0.0f;
    q = fmaf(v[c], v[c], q);
  }
  const float rs = 1.0f / sqrtf(pp_wave_sum(q) / (float)D + eps);
#pragma unroll
  for (int c = 0; c < CH; ++c) {
    const int e = c * 64 + lane;
    if (e < D) y[row * D + e] = v[c] * rs * w[e];
  }
  if (lane == 0) rstd[row] = rs;
}

// gx = rstd * (w*gy - xh * mean(w*gy*xh)), xh = x*rstd;  gw[e] = sum_rows gy*xh (partials per wave)
template <int CH>
__global__ __launch_bounds__(256) void pp_rms_norm_bwd_kernel(
    const float* __restrict__ x, const float* __restrict__ w, const float* __restrict__ rstd,
    const float* __restrict__ gy, float* __restrict__ gx, float* __restrict__ part, long rows,
    int D, int rows_per_wave) {
  const int lane = threadIdx.x & 63;
  const long wave = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  float wv[CH], gwp[CH];
#pragma unroll
  for (int c = 0; c < CH; ++c) {
    const int e = c * 64 + lane;
    wv[c] = e < D ? w[e] : 0.0f;
    gwp[c] = 0.0f;
  }
  const long r0 = wave * rows_per_wave, r1 = min(rows, r0 + rows_per_wave);
  for (long row = r0; row < r1; ++row) {
    const float rs = rstd[row];
    float xh[CH], g[CH], s = 0.0f;
#pragma unroll
    for (int c = 0; c < CH; ++c) {
      const int e = c * 64 + lane;
      const bool in = e < D;
      xh[c] = in ? x[row * D + e] * rs : 0.0f;
      const float gyv = in ? gy[row * D + e] : 0.0f;
      gwp[c] = fmaf(gyv, xh[c], gwp[c]);
      g[c] = gyv * wv[c];
      s = fmaf(g[c], xh[c], s);
    }
    const float m = pp_wave_sum(s) / (float)D;
#pragma unroll
    for (int c = 0; c < CH; ++c) {
      const int e = c * 64 + lane;
      if (e < D) gx[row * D + e] = rs * (g[c] - xh[c] * m);
    }
  }
#pragma unroll
  for (int c = 0; c < CH; ++c) {
    const int e = c * 64 + lane;
    if (e < D) part[wave * D + e] = gwp[c];
  }
}

__global__ __launch_bounds__(256) void pp_col_sum_kernel(const float* __restrict__ part,
                                                         float* __restrict__ out, int D, int n) {
  const int e = blockIdx.x * 64 + (threadIdx.x & 63);
  const bool live = e < D;
  const float s = partial_sum_64(part + (live ? e : 0), (size_t)D, n, live);
  if (threadIdx.x < 64 && live) out[e] = s;
}

int conv_chunks(int b) { return std::min(b, 2048); }
int rms_waves(long rows) { return (int)std::min<long>(rows, 512); }

}  // namespace

size_t pp_conv_silu_bwd_workspace_bytes(int b, int d) {
  return (size_t)conv_chunks(b) * (kMaxConvK + 1) * d * sizeof(float);
}
size_t pp_rms_norm_bwd_workspace_bytes(long rows, int D) {
  const int nw = (rms_waves(rows) + 3) / 4 * 4;
  return (size_t)nw * D * sizeof(float);
}

hipError_t pp_launch_conv_silu_fwd(const float* x, int ldx, const float* w, const float* bias,
                                   float* conv, float* y, int b, int l, int d, int k,
                                   hipStream_t st) {
  const long total = (long)b * l * d;
  hipLaunchKernelGGL(pp_conv_silu_fwd_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0,
                     st, x, ldx, w, bias, conv, y, b, l, d, k);
  return hipGetLastError();
}

hipError_t pp_launch_conv_silu_bwd(const float* x, int ldx, const float* w, const float* conv,
                                   const float* gy, float* gx, float* gw, float* gb, void* ws,
                                   int b, int l, int d, int k, hipStream_t st) {
  const int nchunk = conv_chunks(b);
  const int spc = (b + nchunk - 1) / nchunk;
  const int used = (b + spc - 1) / spc;
  float* part = static_cast<float*>(ws);
  hipLaunchKernelGGL(pp_conv_silu_bwd_kernel, dim3((d + 63) / 64, used), dim3(64), 0, st, x, ldx, w,
                     conv, gy, gx, part, b, l, d, k, spc);
  hipLaunchKernelGGL(pp_conv_param_reduce_kernel, dim3(((k + 1) * d + 63) / 64), dim3(256), 0, st,
                     (const float*)part, gw, gb, d, k, used);
  return hipGetLastError();
}

hipError_t pp_launch_silu_gate_fwd(const float* s, const float* r, int ldr, float* y, long rows,
                                   int d, hipStream_t st) {
  hipLaunchKernelGGL(pp_silu_gate_fwd_kernel, dim3((unsigned)((rows * d + 255) / 256)), dim3(256),
                     0, st, s, r, ldr, y, rows, d);
  return hipGetLastError();
}

hipError_t pp_launch_silu_gate_bwd(const float* s, const float* r, int ldr, const float* gy,
                                   float* gs, float* gr, long rows, int d, hipStream_t st) {
  hipLaunchKernelGGL(pp_silu_gate_bwd_kernel, dim3((unsigned)((rows * d + 255) / 256)), dim3(256),
                     0, st, s, r, ldr, gy, gs, gr, rows, d);
  return hipGetLastError();
}

#define PP_RMS_DISPATCH(KERNEL, GRID, ...)                                                     \
  do {                                                                                         \
    const int need = (D + 63) / 64;                                                            \
    if (need <= 1) hipLaunchKernelGGL((KERNEL<1>), GRID, dim3(256), 0, st, __VA_ARGS__);       \
    else if (need <= 2) hipLaunchKernelGGL((KERNEL<2>), GRID, dim3(256), 0, st, __VA_ARGS__);  \
    else if (need <= 4) hipLaunchKernelGGL((KERNEL<4>), GRID, dim3(256), 0, st, __VA_ARGS__);  \
    else if (need <= 8) hipLaunchKernelGGL((KERNEL<8>), GRID, dim3(256), 0, st, __VA_ARGS__);  \
    else if (need <= 16) hipLaunchKernelGGL((KERNEL<16>), GRID, dim3(256), 0, st, __VA_ARGS__); \
    else return hipErrorInvalidValue;                                                          \
  } while (0)

hipError_t pp_launch_rms_norm_fwd(const float* x, const float* w, float* y, float* rstd, long rows,
                                  int D, float eps, hipStream_t st) {
  const dim3 grid((unsigned)((rows + 3) / 4));
  PP_RMS_DISPATCH(pp_rms_norm_fwd_kernel, grid, x, w, y, rstd, rows, D, eps);
  return hipGetLastError();
}

hipError_t pp_launch_rms_norm_bwd(const float* x, const float* w, const float* rstd,
                                  const float* gy, float* gx, float* gw, void* ws, long rows, int D,
                                  hipStream_t st) {
  const int nw = (rms_waves(rows) + 3) / 4 * 4;
  const int rpw = (int)((rows + nw - 1) / nw);
  float* part = static_cast<float*>(ws);
  const dim3 grid(nw / 4);
  PP_RMS_DISPATCH(pp_rms_norm_bwd_kernel, grid, x, w, rstd, gy, gx, part, rows, D, rpw);
  hipLaunchKernelGGL(pp_col_sum_kernel, dim3((D + 63) / 64), dim3(256), 0, st,
                     (const float*)part, gw, D, nw);
  return hipGetLastError();
}
