// Row-wise normalisations that sit between the polynomial layers
// (/root/reference/language_interpolation/networks.py:48-91):
//   LazyLayerNormLastDim  -> non-affine LayerNorm over the last dim, eps 1e-5
//   MaxAbsNormalizationLast -> x / (max|x| + eps), eps 1e-6                   
// HBM-bound streaming kernels: one wavefront per row, the row held in registers
// (D <= 1024), 16-byte accesses where D allows, DPP wave reductions.  The
// LayerNorm backward works from the OUTPUT y (which the next layer keeps alive
// anyway) and the saved 1/std, so the forward writes only y and rstd[B].
#include "pp_device.h"

namespace {

__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) v = fmaxf(v, __shfl_xor(v, off, 64));
  return v;
}

// Row cache: CH chunks of 64 lanes x V floats.
template <int V, int CH>
struct RowRegs {
  float v[CH][V];
  __device__ __forceinline__ void load(const float* row, int D, int lane, float fill) {
#pragma unroll
    for (int c = 0; c < CH; ++c) {
      const int d = (c * 64 + lane) * V;
      if (d < D) {
        PPVec<V> t;
        t.load(row + d);
#pragma unroll
        for (int e = 0; e < V; ++e) v[c][e] = t.v[e];
      } else {
#pragma unroll
        for (int e = 0; e < V; ++e) v[c][e] = fill;
      }
    }
  }
  __device__ __forceinline__ void store(float* row, int D, int lane) const {
#pragma unroll
    for (int c = 0; c < CH; ++c) {
      const int d = (c * 64 + lane) * V;
      if (d < D) {
        PPVec<V> t;
#pragma unroll
        for (int e = 0; e < V; ++e) t.v[e] = v[c][e];
        t.store(row + d);
      }
    }
  }
};

template <int V, int CH>
__global__ __launch_bounds__(256) void pp_layer_norm_fwd_kernel(const float* __restrict__ x,
                                                                float* __restrict__ y,
                                                                float* __restrict__ rstd, int B,
                                                                int D, float eps) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= B) return;
  RowRegs<V, CH> r;
  r.load(x + (size_t)row * D, D, lane, 0.0f);
  float s = 0.0f;
#pragma unroll
  for (int c = 0; c < CH; ++c)
#pragma unroll
    for (int e = 0; e < V; ++e) s += r.v[c][e];
  const float mean = pp_wave_sum(s) / (float)D;
  float q = 0.0f;
#pragma unroll
  for (int c = 0; c < CH; ++c)
#pragma unroll
    for (int e = 0; e < V; ++e) {
      const bool in = (c * 64 + lane) * V + e < D;
      const float d = in ? r.v[c][e] - mean : 0.0f;
      r.v[c][e] = d;
      q = fmaf(d, d, q);
    }
  const float var = pp_wave_sum(q) / (float)D;
  const float rs = 1.0f / sqrtf(var + eps);
#pragma unroll
  for (int c = 0; c < CH; ++c)
#pragma unroll
    for (int e = 0; e < V; ++e) r.v[c][e] *= rs;
  r.store(y + (size_t)row * D, D, lane);
  if (lane == 0) rstd[row] = rs;
}

// gx = rstd * (gy - mean(gy) - y * mean(gy*y))
template <int V, int CH>
__global__ __launch_bounds__(256) void pp_layer_norm_bwd_kernel(const float* __restrict__ y,
                                                                const float* __restrict__ gy,
                                                                const float* __restrict__ rstd,
                                                                float* __restrict__ gx, int B,
                                                                int D) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= B) return;
  RowRegs<V, CH> ry, rg;
  ry.load(y + (size_t)row * D, D, lane, 0.0f);
  rg.load(gy + (size_t)row * D, D, lane, 0.0f);
  float s1 = 0.0f, s2 = 0.0f;
#pragma unroll
  for (int c = 0; c < CH; ++c)
#pragma unroll
    for (int e = 0; e < V; ++e) {
      s1 += rg.v[c][e];
      s2 = fmaf(rg.v[c][e], ry.v[c][e], s2);
    }
  const float m1 = pp_wave_sum(s1) / (float)D;
  const float m2 = pp_wave_sum(s2) / (float)D;
  const float rs = rstd[row];
#pragma unroll
  for (int c = 0; c < CH; ++c)
#pragma unroll
    for (int e = 0; e < V; ++e) rg.v[c][e] = rs * (rg.v[c][e] - m1 - ry.v[c][e] * m2);
  rg.store(gx + (size_t)row * D, D, lane);
}

template <int V, int CH>
__global__ __launch_bounds__(256) void pp_max_abs_fwd_kernel(const float* __restrict__ x,
                                                             float* __restrict__ y, int B, int D,
                                                             float eps) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= B) return;
  RowRegs<V, CH> r;
  r.load(x + (size_t)row * D, D, lane, 0.0f);
  float m = 0.0f;
#pragma unroll
  for (int c = 0; c < CH; ++c)
#pragma unroll
    for (int e = 0; e < V; ++e) m = fmaxf(m, fabsf(r.v[c][e]));
  const float den = wave_max(m) + eps;
#pragma unroll
  for (int c = 0; c < CH; ++c)
#pragma unroll
    for (int e = 0; e < V; ++e) r.v[c][e] = r.v[c][e] / den;
  r.store(y + (size_t)row * D, D, lane);
}

// y = x/d, d = |x_a| + eps, a = first argmax|x|:
// gx_j = gy_j/d - [j==a] * sign(x_a) * sum_k(gy_k x_k) / d^2
template <int V, int CH>
__global__ __launch_bounds__(256) void pp_max_abs_bwd_kernel(const float* __restrict__ x,
                                                             const float* __restrict__ gy,
                                                             float* __restrict__ gx, int B, int D,
                                                             float eps) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= B) return;
  RowRegs<V, CH> rx, rg;
  rx.load(x + (size_t)row * D, D, lane, 0.0f);
  rg.load(gy + (size_t)row * D, D, lane, 0.0f);
  float m = 0.0f, s = 0.0f;
#pragma unroll
  for (int c = 0; c < CH; ++c)
#pragma unroll
    for (int e = 0; e < V; ++e) {
      m = fmaxf(m, fabsf(rx.v[c][e]));
      s = fmaf(rg.v[c][e], rx.v[c][e], s);
    }
  const float mx = wave_max(m);
  const float dot = pp_wave_sum(s);
  // first index attaining the max
  int idx = 0x7fffffff;
#pragma unroll
  for (int c = 0; c < CH; ++c)
#pragma unroll
    for (int e = 0; e < V; ++e) {
      const int d = (c * 64 + lane) * V + e;
      if (d < D && fabsf(rx.v[c][e]) == mx) idx = min(idx, d);
    }
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) idx = min(idx, __shfl_xor(idx, off, 64));
  const float den = mx + eps;
  const float corr = dot / (den * den);
#pragma unroll
  for (int c = 0; c < CH; ++c)
#pragma unroll
    for (int e = 0; e < V; ++e) {
      const int d = (c * 64 + lane) * V + e;
      float g = rg.v[c][e] / den;
      if (d == idx) g -= (rx.v[c][e] < 0.0f ? -corr : corr);
      rg.v[c][e] = g;
    }
  rg.store(gx + (size_t)row * D, D, lane);
}

// The same backward from what the fused layer+maxabs forward keeps (pp_fwd_gather_ln_kernel,
// NORM = 1): the normalised row y = x/d, d and a.  With x_k = d*y_k:
// gx_j = (gy_j - [j==a] * sign(y_a) * sum_k(gy_k y_k)) / d
template <int V, int CH>
__global__ __launch_bounds__(256) void pp_max_abs_bwd_y_kernel(const float* __restrict__ y,
                                                               const float* __restrict__ gy,
                                                               const float* __restrict__ den,
                                                               const int32_t* __restrict__ amax,
                                                               float* __restrict__ gx, int B,
                                                               int D) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= B) return;
  RowRegs<V, CH> ry, rg;
  ry.load(y + (size_t)row * D, D, lane, 0.0f);
  rg.load(gy + (size_t)row * D, D, lane, 0.0f);
  float s = 0.0f;
#pragma unroll
  for (int c = 0; c < CH; ++c)
#pragma unroll
    for (int e = 0; e < V; ++e) s = fmaf(rg.v[c][e], ry.v[c][e], s);
  const float dot = pp_wave_sum(s);
  const float d = den[row];
  const int idx = amax[row];
#pragma unroll
  for (int c = 0; c < CH; ++c)
#pragma unroll
    for (int e = 0; e < V; ++e) {
      float g = rg.v[c][e];
      if ((c * 64 + lane) * V + e == idx) g -= (ry.v[c][e] < 0.0f ? -dot : dot);
      rg.v[c][e] = g / d;
    }
  rg.store(gx + (size_t)row * D, D, lane);
}

// ---------------------------------------------------------------------------
// Cross entropy over the last dim with the accuracy of the arg-max, for the training harness
// (ASCIIPredictionNet.eval_step, /root/reference/language_interpolation/networks.py:121-132:
// `loss = CrossEntropyLoss()(y_hat, y)`, `accuracy = (argmax(y_hat) == y).mean()`): one pass over
// the logits instead of log-softmax, NLL (a single-block reduction in torch: 27 us at 16384
// rows), arg-max, compare and two means.  A wave holds a row; the block's sums of loss and hits
// go to part[block], a second single-block kernel adds them in a fixed order -> the same bits
// every time.  The backward recomputes the softmax from the logits:
//   glogits = (softmax - onehot) * gout / B.
// ---------------------------------------------------------------------------
constexpr int kCeBlocks = 512;

template <int V, int CH>
__global__ __launch_bounds__(256) void pp_ce_fwd_kernel(const float* __restrict__ logits,
                                                        const long* __restrict__ target,
                                                        float2* __restrict__ part, int B, int D) {
  __shared__ float2 wsum[4];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  float loss_acc = 0.0f, hit_acc = 0.0f;
  for (int row = blockIdx.x * 4 + wave; row < B; row += gridDim.x * 4) {
    RowRegs<V, CH> r;
    r.load(logits + (size_t)row * D, D, lane, -INFINITY);
    const int t = (int)target[row];
    float m = -INFINITY;
#pragma unroll
    for (int c = 0; c < CH; ++c)
#pragma unroll
      for (int e = 0; e < V; ++e) m = fmaxf(m, r.v[c][e]);
    m = wave_max(m);
    float s = 0.0f, xt = 0.0f;
    int idx = 0x7fffffff;  // first index attaining the maximum
#pragma unroll
    for (int c = 0; c < CH; ++c)
#pragma unroll
      for (int e = 0; e < V; ++e) {
        const int d = (c * 64 + lane) * V + e;
        if (d < D) {
          s += expf(r.v[c][e] - m);
          if (d == t) xt = r.v[c][e];
          if (r.v[c][e] == m) idx = min(idx, d);
        }
      }
    s = pp_wave_sum(s);
    xt = pp_wave_sum(xt);
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) idx = min(idx, __shfl_xor(idx, off, 64));
    loss_acc += logf(s) + m - xt;
    hit_acc += idx == t ? 1.0f : 0.0f;
  }
  if (lane == 0) wsum[wave] = make_float2(loss_acc, hit_acc);
  __syncthreads();
  if (threadIdx.x == 0)
    part[blockIdx.x] = make_float2((wsum[0].x + wsum[1].x) + (wsum[2].x + wsum[3].x),
                                   (wsum[0].y + wsum[1].y) + (wsum[2].y + wsum[3].y));
}

__global__ __launch_bounds__(256) void pp_ce_finish_kernel(const float2* __restrict__ part,
                                                           int nblk, float* __restrict__ out,
                                                           float inv_rows) {
  __shared__ float2 sm[256];
  float2 a = make_float2(0.0f, 0.0f);
  for (int i = threadIdx.x; i < nblk; i += 256) {
    a.x += part[i].x;
    a.y += part[i].y;
  }
  sm[threadIdx.x] = a;
  __syncthreads();
  for (int off = 128; off >= 1; off >>= 1) {
    if ((int)threadIdx.x < off) {
      sm[threadIdx.x].x += sm[threadIdx.x + off].x;
      sm[threadIdx.x].y += sm[threadIdx.x + off].y;
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    out[0] = sm[0].x * inv_rows;  // mean loss
    out[1] = sm[0].y * inv_rows;  // accuracy
  }
}

template <int V, int CH>
__global__ __launch_bounds__(256) void pp_ce_bwd_kernel(const float* __restrict__ logits,
                                                        const long* __restrict__ target,
                                                        const float* __restrict__ gout,
                                                        float* __restrict__ glogits, int B, int D) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= B) return;
  RowRegs<V, CH> r;
  r.load(logits + (size_t)row * D, D, lane, -INFINITY);
  const int t = (int)target[row];
  const float scale = gout[0] / (float)B;
  float m = -INFINITY;
#pragma unroll
  for (int c = 0; c < CH; ++c)
#pragma unroll
    for (int e = 0; e < V; ++e) m = fmaxf(m, r.v[c][e]);
  m = wave_max(m);
  float s = 0.0f;
#pragma unroll
  for (int c = 0; c < CH; ++c)
#pragma unroll
    for (int e = 0; e < V; ++e) {
      r.v[c][e] = expf(r.v[c][e] - m);  // (padding: exp(-inf) = 0)
      s += r.v[c][e];
    }
  s = pp_wave_sum(s);
  const float inv = 1.0f / s;
#pragma unroll
  for (int c = 0; c < CH; ++c)
#pragma unroll
    for (int e = 0; e < V; ++e) {
      const int d = (c * 64 + lane) * V + e;
      r.v[c][e] = (r.v[c][e] * inv - (d == t ? 1.0f : 0.0f)) * scale;
    }
  r.store(glogits + (size_t)row * D, D, lane);
}

// ---------------------------------------------------------------------------
// Lion / SparseLion step for up to PP_OPT_MAX_TENSORS parameters in one launch
// (lion_pytorch.Lion as the reference's optimizer configs name it, and high-order-layers-torch's
// SparseLion = Lion that leaves entries with a zero gradient untouched; the Python restatement
// is language_interpolation_amd/train.py:Lion):
//   u = sign(b1*m + (1-b1)*g);  p = p*(1 - lr*wd) - lr*u;  m = b2*m + (1-b2)*g
// One pass over (p, g, m) in storage order: 16 bytes read + 8 written per weight instead of the
// ten elementwise torch kernels per parameter.
// ---------------------------------------------------------------------------
struct LionDescs {
  float* p[PP_OPT_MAX_TENSORS];
  const float* g[PP_OPT_MAX_TENSORS];
  float* m[PP_OPT_MAX_TENSORS];
  long n[PP_OPT_MAX_TENSORS];
};

__device__ __forceinline__ void lion_one(float& p, float g, float& m, float lr, float b1, float b2,
                                         float decay, bool sparse) {
  const float c = m * b1 + g * (1.0f - b1);
  float u = c > 0.0f ? 1.0f : (c < 0.0f ? -1.0f : 0.0f);
  float nm = m * b2 + g * (1.0f - b2);
  if (sparse && g == 0.0f) return;  // an entry no sample touched keeps value (no decay) and momentum
  p = p * decay - lr * u;
  m = nm;
}

__global__ __launch_bounds__(256) void pp_lion_kernel(LionDescs d, float lr, float b1, float b2,
                                                      float decay, int sparse) {
  const int t = blockIdx.y;
  float* __restrict__ p = d.p[t];
  const float* __restrict__ g = d.g[t];
  float* __restrict__ m = d.m[t];
  const long n = d.n[t];
  const long n4 = n >> 2;
  const long stride = (long)gridDim.x * 256;
  const bool vec = ((((size_t)p) | ((size_t)g) | ((size_t)m)) & 15) == 0;
  long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (vec) {
    for (; i < n4; i += stride) {
      float4 pv = reinterpret_cast<float4*>(p)[i];
      const float4 gv = reinterpret_cast<const float4*>(g)[i];
      float4 mv = reinterpret_cast<float4*>(m)[i];
      lion_one(pv.x, gv.x, mv.x, lr, b1, b2, decay, sparse);
      lion_one(pv.y, gv.y, mv.y, lr, b1, b2, decay, sparse);
      lion_one(pv.z, gv.z, mv.z, lr, b1, b2, decay, sparse);
      lion_one(pv.w, gv.w, mv.w, lr, b1, b2, decay, sparse);
      reinterpret_cast<float4*>(p)[i] = pv;
      reinterpret_cast<float4*>(m)[i] = mv;
    }
    for (long e = (n4 << 2) + (long)blockIdx.x * 256 + threadIdx.x; e < n; e += stride) {
      float pv = p[e], mv = m[e];
      lion_one(pv, g[e], mv, lr, b1, b2, decay, sparse);
      p[e] = pv;
      m[e] = mv;
    }
  } else {
    for (; i < n; i += stride) {
      float pv = p[i], mv = m[i];
      lion_one(pv, g[i], mv, lr, b1, b2, decay, sparse);
      p[i] = pv;
      m[i] = mv;
    }
  }
}

}  // namespace

// dispatch on (vector width, chunks): D <= 64*V*CH
#define PP_NORM_DISPATCH(KERNEL, ...)                                                      \
  do {                                                                                     \
    const int V = (D % 4 == 0) ? 4 : (D % 2 == 0 ? 2 : 1);                                 \
    const int need = (D + 64 * V - 1) / (64 * V);                                          \
    dim3 grid((B + 3) / 4), block(256);                                                    \
    if (V == 4 && need <= 1) hipLaunchKernelGGL((KERNEL<4, 1>), grid, block, 0, st, __VA_ARGS__); \
    else if (V == 4 && need <= 2) hipLaunchKernelGGL((KERNEL<4, 2>), grid, block, 0, st, __VA_ARGS__); \
    else if (V == 4 && need <= 4) hipLaunchKernelGGL((KERNEL<4, 4>), grid, block, 0, st, __VA_ARGS__); \
    else if (V == 2 && need <= 2) hipLaunchKernelGGL((KERNEL<2, 2>), grid, block, 0, st, __VA_ARGS__); \
    else if (V == 2 && need <= 4) hipLaunchKernelGGL((KERNEL<2, 4>), grid, block, 0, st, __VA_ARGS__); \
    else if (V == 2 && need <= 8) hipLaunchKernelGGL((KERNEL<2, 8>), grid, block, 0, st, __VA_ARGS__); \
    else if (V == 1 && need <= 4) hipLaunchKernelGGL((KERNEL<1, 4>), grid, block, 0, st, __VA_ARGS__); \
    else if (V == 1 && need <= 16) hipLaunchKernelGGL((KERNEL<1, 16>), grid, block, 0, st, __VA_ARGS__); \
    else return hipErrorInvalidValue;                                                      \
    return hipGetLastError();                                                              \
  } while (0)

hipError_t pp_launch_layer_norm_fwd(const float* x, float* y, float* rstd, int B, int D, float eps,
                                    hipStream_t st) {
  PP_NORM_DISPATCH(pp_layer_norm_fwd_kernel, x, y, rstd, B, D, eps);
}
hipError_t pp_launch_layer_norm_bwd(const float* y, const float* gy, const float* rstd, float* gx,
                                    int B, int D, hipStream_t st) {
  PP_NORM_DISPATCH(pp_layer_norm_bwd_kernel, y, gy, rstd, gx, B, D);
}
hipError_t pp_launch_max_abs_fwd(const float* x, float* y, int B, int D, float eps, hipStream_t st) {
  PP_NORM_DISPATCH(pp_max_abs_fwd_kernel, x, y, B, D, eps);
}
hipError_t pp_launch_max_abs_bwd(const float* x, const float* gy, float* gx, int B, int D, float eps,
                                 hipStream_t st) {
  PP_NORM_DISPATCH(pp_max_abs_bwd_kernel, x, gy, gx, B, D, eps);
}

hipError_t pp_launch_max_abs_bwd_y(const float* y, const float* gy, const float* den,
                                   const int32_t* amax, float* gx, int B, int D, hipStream_t st) {
  PP_NORM_DISPATCH(pp_max_abs_bwd_y_kernel, y, gy, den, amax, gx, B, D);
}

// dispatch with an explicit grid (the forward walks the rows with a fixed number of blocks)
#define PP_NORM_DISPATCH_GRID(KERNEL, GRID, ...)                                            \
  do {                                                                                      \
    const int V = (D % 4 == 0) ? 4 : (D % 2 == 0 ? 2 : 1);                                  \
    const int need = (D + 64 * V - 1) / (64 * V);                                           \
    dim3 grid(GRID), block(256);                                                            \
    if (V == 4 && need <= 1) hipLaunchKernelGGL((KERNEL<4, 1>), grid, block, 0, st, __VA_ARGS__); \
    else if (V == 4 && need <= 2) hipLaunchKernelGGL((KERNEL<4, 2>), grid, block, 0, st, __VA_ARGS__); \
    else if (V == 4 && need <= 4) hipLaunchKernelGGL((KERNEL<4, 4>), grid, block, 0, st, __VA_ARGS__); \
    else if (V == 2 && need <= 2) hipLaunchKernelGGL((KERNEL<2, 2>), grid, block, 0, st, __VA_ARGS__); \
    else if (V == 2 && need <= 4) hipLaunchKernelGGL((KERNEL<2, 4>), grid, block, 0, st, __VA_ARGS__); \
    else if (V == 2 && need <= 8) hipLaunchKernelGGL((KERNEL<2, 8>), grid, block, 0, st, __VA_ARGS__); \
    else if (V == 1 && need <= 4) hipLaunchKernelGGL((KERNEL<1, 4>), grid, block, 0, st, __VA_ARGS__); \
    else if (V == 1 && need <= 16) hipLaunchKernelGGL((KERNEL<1, 16>), grid, block, 0, st, __VA_ARGS__); \
    else return hipErrorInvalidValue;                                                       \
  } while (0)

size_t pp_ce_workspace_bytes() { return (size_t)kCeBlocks * sizeof(float2); }

hipError_t pp_launch_ce_fwd(const float* logits, const long* target, float* out, void* ws, int B,
                            int D, hipStream_t st) {
  const int nblk = std::min(kCeBlocks, (B + 3) / 4);
  float2* part = static_cast<float2*>(ws);
  PP_NORM_DISPATCH_GRID(pp_ce_fwd_kernel, nblk, logits, target, part, B, D);
  hipLaunchKernelGGL(pp_ce_finish_kernel, dim3(1), dim3(256), 0, st, (const float2*)part, nblk, out,
                     1.0f / (float)B);
  return hipGetLastError();
}

hipError_t pp_launch_ce_bwd(const float* logits, const long* target, const float* gout,
                            float* glogits, int B, int D, hipStream_t st) {
  PP_NORM_DISPATCH_GRID(pp_ce_bwd_kernel, (B + 3) / 4, logits, target, gout, glogits, B, D);
  return hipGetLastError();
}

hipError_t pp_launch_lion(int count, float* const* p, const float* const* g, float* const* m,
                          const long* numel, float lr, float b1, float b2, float wd, int sparse,
                          hipStream_t st) {
  LionDescs d;
  long most = 0;
  for (int t = 0; t < count; ++t) {
    d.p[t] = p[t]; d.g[t] = g[t]; d.m[t] = m[t]; d.n[t] = numel[t];
    most = std::max(most, numel[t]);
  }
  const long want = (most / 4 + 255) / 256;
  const int blocks = (int)std::max<long>(1, std::min<long>(want, 2048));
  hipLaunchKernelGGL(pp_lion_kernel, dim3(blocks, count), dim3(256), 0, st, d, lr, b1, b2,
                     1.0f - lr * wd, sparse);
  return hipGetLastError();
}
