// Gather path: wavefront-cooperative gather of the active weight slab.
//
// Layout: w_t[I][K][O].  For one (sample b, input i) pair the n active weight
// rows k = s*stride .. s*stride+n-1 are ONE contiguous n*O slab, so a wavefront
// whose lanes span `out` reads it as full-width coalesced rows.  The per-(b,i)
// scalar work (segment id, local coordinate, Lagrange basis -- A1..A3 of
// SURVEY.md 8a) is done lane-parallel over 64 inputs at a time and then handed
// to the whole wavefront with v_readlane, so it is computed once per pair, not
// once per output.
//
// Replaces: Piecewise.forward of high-order-layers-torch as called through
// HighOrderMLP / high_order_fc_layers (networks.py:399-426, :799-806, :866-882)
// and its autograd backward.
#include "pp_device.h"

namespace {

// ---------------------------------------------------------------------------
// A1 only: segment ids (for the bit-exact index check)
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void pp_segment_ids_kernel(const float* __restrict__ x,
                                                             int32_t* __restrict__ seg, long total,
                                                             float length, float half, float Sf,
                                                             int S) {
  long idx = (long)blockIdx.x * 256 + threadIdx.x;
  const long step = (long)gridDim.x * 256;
  for (; idx < total; idx += step) seg[idx] = pp_segment(x[idx], length, half, Sf, S);
}

// inputs whose weight rows are loaded together by a wave that owns ONE row (the small-batch
// variants): about 32 floats in flight per lane, 8 inputs at most (floats = what one input loads
// per lane).  Waves with several rows already have a load per row in flight and enough
// neighbours; two inputs at a time measured slower there (cfg3 hidden, 16384 rows: 128 vs 110 us).
constexpr int gather_unroll(int floats, int rows_per_wave) {
  return rows_per_wave > 1 || floats >= 32 ? 1 : (32 / floats > 8 ? 8 : 32 / floats);
}

// ---------------------------------------------------------------------------
// forward
//   grid.x: groups of 4*RW rows (4 waves per block, RW rows per wave)
//   grid.y: chunks of 64*V outputs
// ---------------------------------------------------------------------------
template <int N, int V, int RW>
__global__ __launch_bounds__(256) void pp_fwd_gather_kernel(const float* __restrict__ x,
                                                            const float* __restrict__ wt,
                                                            float* __restrict__ y,
                                                            int32_t* __restrict__ seg_out,
                                                            PPShape sh, BasisTab tab) {
  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  const int row0 = (blockIdx.x * 4 + wave) * RW;
  if (row0 >= sh.B) return;  // wave-uniform; no block barrier below
  const int o = (blockIdx.y * 64 + lane) * V;
  const bool o_ok = o < sh.O;
  const int oc = o_ok ? o : sh.O - V;  // clamp: every lane loads valid memory
  const bool write_seg = (seg_out != nullptr) && (blockIdx.y == 0);

  float acc[RW][V];
#pragma unroll
  for (int r = 0; r < RW; ++r)
#pragma unroll
    for (int v = 0; v < V; ++v) acc[r][v] = 0.0f;

  for (int i0 = 0; i0 < sh.I; i0 += 64) {
    const int i = i0 + lane;
    int koff[RW];
    float l[RW][N];
#pragma unroll
    for (int r = 0; r < RW; ++r) {
      const int b = min(row0 + r, sh.B - 1);
      const bool valid = i < sh.I;
      const float xv = valid ? x[(size_t)b * sh.I + i] : 0.0f;
      const int s = pp_segment(xv, sh.length, sh.half, sh.Sf, sh.S);
      const float xi = pp_local_coord(xv, s, sh.length, sh.half, sh.Sf, nullptr);
      pp_lagrange<N>(xi, tab, l[r]);
      koff[r] = (valid ? i : 0) * sh.K + s * sh.stride;
      if (write_seg && valid && row0 + r < sh.B) seg_out[(size_t)b * sh.I + i] = s;
    }
    const int cnt = min(64, sh.I - i0);
    // U inputs at a time, all their loads issued before the first use: one load in flight per
    // wave (what the plain loop compiles to) ran the few waves of a small batch at the L2
    // latency per input.  Same order of additions as the plain loop.
    constexpr int U = gather_unroll(RW * N * V, RW);
    int ii = 0;
    for (; ii + U <= cnt; ii += U) {
      PPVec<V> wv[U][RW][N];
#pragma unroll
      for (int u = 0; u < U; ++u)
#pragma unroll
        for (int r = 0; r < RW; ++r) {
          const int ko = pp_readlane_i(koff[r], ii + u);
          const float* p = wt + (size_t)ko * sh.O + oc;
#pragma unroll
          for (int j = 0; j < N; ++j) wv[u][r][j].load(p + (size_t)j * sh.O);
        }
#pragma unroll
      for (int u = 0; u < U; ++u)
#pragma unroll
        for (int r = 0; r < RW; ++r)
#pragma unroll
          for (int j = 0; j < N; ++j) {
            const float lj = pp_readlane_f(l[r][j], ii + u);
#pragma unroll
            for (int v = 0; v < V; ++v) acc[r][v] = fmaf(lj, wv[u][r][j].v[v], acc[r][v]);
          }
    }
    for (; ii < cnt; ++ii) {
#pragma unroll
      for (int r = 0; r < RW; ++r) {
        const int ko = pp_readlane_i(koff[r], ii);
        const float* p = wt + (size_t)ko * sh.O + oc;
#pragma unroll
        for (int j = 0; j < N; ++j) {
          const float lj = pp_readlane_f(l[r][j], ii);
          PPVec<V> wv;
          wv.load(p + (size_t)j * sh.O);
#pragma unroll
          for (int v = 0; v < V; ++v) acc[r][v] = fmaf(lj, wv.v[v], acc[r][v]);
        }
      }
    }
  }
  if (o_ok) {
#pragma unroll
    for (int r = 0; r < RW; ++r) {
      if (row0 + r < sh.B) {
        PPVec<V> out;
#pragma unroll
        for (int v = 0; v < V; ++v) out.v[v] = acc[r][v] * sh.out_scale;
        out.store(y + (size_t)(row0 + r) * sh.O + o);
      }
    }
  }
}

// ---------------------------------------------------------------------------
// forward with the non-affine LayerNorm over the row fused into the epilogue
// (LazyLayerNormLastDim, /root/reference/language_interpolation/networks.py:72-91): a wave owns
// whole rows (CH chunks of 64*V outs per lane, O <= 1024), so the row it has just summed is
// normalised in registers and the raw layer output never exists in memory.  Same order of the
// sum over inputs as pp_fwd_gather_kernel and the same reductions as pp_layer_norm_fwd_kernel
// (pp_norm.hip) at the same V: bit-identical to running the two kernels one after the other.
// NORM = 1: MaxAbsNormalizationLast instead (networks.py:48-69 of the reference: y / (max|y| +
// eps)); `rstd` then receives the denominator and `amax` the first index attaining the maximum,
// which is all pp_max_abs_bwd_y_kernel (pp_norm.hip) needs besides the normalised row.
// ---------------------------------------------------------------------------
// SPLIT (small batches, RW = 1): the four waves of a block share ONE row, each sums a quarter
// of the inputs, and wave 0 adds the four partial rows in a fixed order (through LDS) before the
// epilogue -- a wave's chain of dependent round trips is a quarter as long, which is what a few
// hundred rows wait for.  NORM = -1: no normalisation (the plain forward at small batches runs
// here too, so fused and unfused results stay bit-identical at every batch size).
template <int N, int V, int CH, int NORM, int RW, bool SPLIT = false>
__global__ __launch_bounds__(256) void pp_fwd_gather_ln_kernel(const float* __restrict__ x,
                                                               const float* __restrict__ wt,
                                                               float* __restrict__ y,
                                                               float* __restrict__ rstd,
                                                               int32_t* __restrict__ amax,
                                                               PPShape sh, BasisTab tab, float eps) {
  static_assert(!SPLIT || RW == 1, "the split variant owns one row per block");
  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  const int row0 = SPLIT ? (int)blockIdx.x : (blockIdx.x * 4 + wave) * RW;
  if (row0 >= sh.B) return;  // wave-uniform (block-uniform when SPLIT: the barrier below is safe)
  // inputs of this wave: a quarter of the row (multiples of 64 so the lane-parallel prologue
  // stays whole), everything otherwise
  const int iq = SPLIT ? ((sh.I + 255) / 256) * 64 : sh.I;
  const int ibeg = SPLIT ? min(sh.I, wave * iq) : 0;
  const int iend = SPLIT ? min(sh.I, ibeg + iq) : sh.I;
  int dcl[CH];
  bool ok[CH];
#pragma unroll
  for (int c = 0; c < CH; ++c) {
    const int d = (c * 64 + lane) * V;
    ok[c] = d < sh.O;
    dcl[c] = ok[c] ? d : sh.O - V;  // clamp: every lane loads valid memory
  }
  float acc[RW][CH][V];
#pragma unroll
  for (int r = 0; r < RW; ++r)
#pragma unroll
    for (int c = 0; c < CH; ++c)
#pragma unroll
      for (int v = 0; v < V; ++v) acc[r][c][v] = 0.0f;

  for (int i0 = ibeg; i0 < iend; i0 += 64) {
    const int i = i0 + lane;
    int koff[RW];
    float l[RW][N];
#pragma unroll
    for (int r = 0; r < RW; ++r) {
      const int b = min(row0 + r, sh.B - 1);
      const bool valid = i < iend;
      const float xv = valid ? x[(size_t)b * sh.I + i] : 0.0f;
      const int s = pp_segment(xv, sh.length, sh.half, sh.Sf, sh.S);
      const float xi = pp_local_coord(xv, s, sh.length, sh.half, sh.Sf, nullptr);
      pp_lagrange<N>(xi, tab, l[r]);
      koff[r] = (valid ? i : 0) * sh.K + s * sh.stride;
    }
    const int cnt = min(64, iend - i0);
    constexpr int U = gather_unroll(RW * N * V * CH, RW);  // as in pp_fwd_gather_kernel
    int ii = 0;
    for (; ii + U <= cnt; ii += U) {
      PPVec<V> wv[U][RW][N][CH];
#pragma unroll
      for (int u = 0; u < U; ++u)
#pragma unroll
        for (int r = 0; r < RW; ++r) {
          const int ko = pp_readlane_i(koff[r], ii + u);
          const float* p = wt + (size_t)ko * sh.O;
#pragma unroll
          for (int j = 0; j < N; ++j)
#pragma unroll
            for (int c = 0; c < CH; ++c) wv[u][r][j][c].load(p + (size_t)j * sh.O + dcl[c]);
        }
#pragma unroll
      for (int u = 0; u < U; ++u)
#pragma unroll
        for (int r = 0; r < RW; ++r)
#pragma unroll
          for (int j = 0; j < N; ++j) {
            const float lj = pp_readlane_f(l[r][j], ii + u);
#pragma unroll
            for (int c = 0; c < CH; ++c)
#pragma unroll
              for (int v = 0; v < V; ++v)
                acc[r][c][v] = fmaf(lj, wv[u][r][j][c].v[v], acc[r][c][v]);
          }
    }
    for (; ii < cnt; ++ii) {
#pragma unroll
      for (int r = 0; r < RW; ++r) {
        const int ko = pp_readlane_i(koff[r], ii);
        const float* p = wt + (size_t)ko * sh.O;
#pragma unroll
        for (int j = 0; j < N; ++j) {
          const float lj = pp_readlane_f(l[r][j], ii);
#pragma unroll
          for (int c = 0; c < CH; ++c) {
            PPVec<V> wv;
            wv.load(p + (size_t)j * sh.O + dcl[c]);
#pragma unroll
            for (int v = 0; v < V; ++v) acc[r][c][v] = fmaf(lj, wv.v[v], acc[r][c][v]);
          }
        }
      }
    }
  }
  if constexpr (SPLIT) {
    __shared__ float part[3][CH * V][64];
    if (wave > 0) {
#pragma unroll
      for (int c = 0; c < CH; ++c)
#pragma unroll
        for (int v = 0; v < V; ++v) part[wave - 1][c * V + v][lane] = acc[0][c][v];
    }
    __syncthreads();
    if (wave > 0) return;
#pragma unroll
    for (int q = 0; q < 3; ++q)
#pragma unroll
      for (int c = 0; c < CH; ++c)
#pragma unroll
        for (int v = 0; v < V; ++v) acc[0][c][v] += part[q][c * V + v][lane];
  }
#pragma unroll
  for (int r = 0; r < RW; ++r) {
    if (row0 + r >= sh.B) break;  // wave-uniform
    if constexpr (NORM == -1) {
#pragma unroll
      for (int c = 0; c < CH; ++c) {
        if (ok[c]) {
          PPVec<V> out;
#pragma unroll
          for (int v = 0; v < V; ++v) out.v[v] = acc[r][c][v] * sh.out_scale;
          out.store(y + (size_t)(row0 + r) * sh.O + dcl[c]);
        }
      }
      continue;
    }
    if constexpr (NORM == 1) {
      float m = 0.0f;
#pragma unroll
      for (int c = 0; c < CH; ++c)
#pragma unroll
        for (int v = 0; v < V; ++v) {
          acc[r][c][v] = ok[c] ? acc[r][c][v] * sh.out_scale : 0.0f;
          m = fmaxf(m, fabsf(acc[r][c][v]));
        }
#pragma unroll
      for (int off = 32; off >= 1; off >>= 1) m = fmaxf(m, __shfl_xor(m, off, 64));
      int idx = 0x7fffffff;  // first index attaining the maximum
#pragma unroll
      for (int c = 0; c < CH; ++c)
#pragma unroll
        for (int v = 0; v < V; ++v)
          if (ok[c] && fabsf(acc[r][c][v]) == m) idx = min(idx, dcl[c] + v);
#pragma unroll
      for (int off = 32; off >= 1; off >>= 1) idx = min(idx, __shfl_xor(idx, off, 64));
      const float den = m + eps;
#pragma unroll
      for (int c = 0; c < CH; ++c) {
        if (ok[c]) {
          PPVec<V> out;
#pragma unroll
          for (int v = 0; v < V; ++v) out.v[v] = acc[r][c][v] / den;
          out.store(y + (size_t)(row0 + r) * sh.O + dcl[c]);
        }
      }
      if (lane == 0) {
        rstd[row0 + r] = den;
        amax[row0 + r] = idx;
      }
      continue;
    }
    float s = 0.0f;
#pragma unroll
    for (int c = 0; c < CH; ++c)
#pragma unroll
      for (int v = 0; v < V; ++v) {
        acc[r][c][v] = ok[c] ? acc[r][c][v] * sh.out_scale : 0.0f;
        s += acc[r][c][v];
      }
    const float mean = pp_wave_sum(s) / (float)sh.O;
    float q = 0.0f;
#pragma unroll
    for (int c = 0; c < CH; ++c)
#pragma unroll
      for (int v = 0; v < V; ++v) {
        const float d = ok[c] ? acc[r][c][v] - mean : 0.0f;
        acc[r][c][v] = d;
        q = fmaf(d, d, q);
      }
    const float var = pp_wave_sum(q) / (float)sh.O;
    const float rs = 1.0f / sqrtf(var + eps);
#pragma unroll
    for (int c = 0; c < CH; ++c) {
      if (ok[c]) {
        PPVec<V> out;
#pragma unroll
        for (int v = 0; v < V; ++v) out.v[v] = acc[r][c][v] * rs;
        out.store(y + (size_t)(row0 + r) * sh.O + dcl[c]);
      }
    }
    if (lane == 0) rstd[row0 + r] = rs;
  }
}

// ---------------------------------------------------------------------------
// backward w.r.t. the input: one wave per row; lanes span `out`, one wave
// reduction per (row, input, out-chunk).  SPLIT (small batches): one row per block, wave w
// takes inputs 16w .. 16w+15 of every 64 -- the inputs are independent, so the chain of a wave
// is a quarter as long and the results are the same bits.
// ---------------------------------------------------------------------------
template <int N, int V, int U, bool SPLIT = false>
__global__ __launch_bounds__(256) void pp_bwd_input_gather_kernel(const float* __restrict__ x,
                                                                  const float* __restrict__ gy,
                                                                  const float* __restrict__ wt,
                                                                  float* __restrict__ gx,
                                                                  PPShape sh, BasisTab tab) {
  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  const int b = SPLIT ? (int)blockIdx.x : blockIdx.x * 4 + wave;
  if (b >= sh.B) return;
  const int nchunk = (sh.O + 64 * V - 1) / (64 * V);

  for (int i0 = 0; i0 < sh.I; i0 += 64) {
    const int i = i0 + lane;
    const bool valid = i < sh.I;
    const float xv = valid ? x[(size_t)b * sh.I + i] : 0.0f;
    const int s = pp_segment(xv, sh.length, sh.half, sh.Sf, sh.S);
    float dxi_dx;
    const float xi = pp_local_coord(xv, s, sh.length, sh.half, sh.Sf, &dxi_dx);
    float l[N], dl[N];
    pp_lagrange_d<N>(xi, tab, l, dl);
    const float chain = dxi_dx * sh.out_scale;
#pragma unroll
    for (int j = 0; j < N; ++j) dl[j] *= chain;
    const int koff = (valid ? i : 0) * sh.K + s * sh.stride;

    float res = 0.0f;
    const int ibeg = SPLIT ? wave * 16 : 0;
    const int cnt = SPLIT ? min(ibeg + 16, min(64, sh.I - i0)) : min(64, sh.I - i0);
    for (int c = 0; c < nchunk; ++c) {
      const int o = (c * 64 + lane) * V;
      const bool o_ok = o < sh.O;
      const int oc = o_ok ? o : sh.O - V;
      PPVec<V> g;
      g.load(gy + (size_t)b * sh.O + oc);
      if (!o_ok) {
#pragma unroll
        for (int v = 0; v < V; ++v) g.v[v] = 0.0f;
      }
      // U inputs at a time: their weight rows are all requested before the first dot product
      // and the U wave reductions interleave (one input at a time ran at one L2 latency plus
      // one dependent reduction chain per input: 78 us for 256 rows of 200 inputs)
      int ii = ibeg;
      for (; ii + U <= cnt; ii += U) {
        PPVec<V> wv[U][N];
#pragma unroll
        for (int u = 0; u < U; ++u) {
          const int ko = pp_readlane_i(koff, ii + u);
          const float* p = wt + (size_t)ko * sh.O + oc;
#pragma unroll
          for (int j = 0; j < N; ++j) wv[u][j].load(p + (size_t)j * sh.O);
        }
        float part[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
          part[u] = 0.0f;
#pragma unroll
          for (int j = 0; j < N; ++j) {
            const float cj = pp_readlane_f(dl[j], ii + u);
            float d = 0.0f;
#pragma unroll
            for (int v = 0; v < V; ++v) d = fmaf(wv[u][j].v[v], g.v[v], d);
            part[u] = fmaf(cj, d, part[u]);
          }
        }
#pragma unroll
        for (int u = 0; u < U; ++u) part[u] = pp_wave_sum(part[u]);
#pragma unroll
        for (int u = 0; u < U; ++u)
          if (lane == ii + u) res += part[u];
      }
      for (; ii < cnt; ++ii) {
        const int ko = pp_readlane_i(koff, ii);
        const float* p = wt + (size_t)ko * sh.O + oc;
        float part = 0.0f;
#pragma unroll
        for (int j = 0; j < N; ++j) {
          const float cj = pp_readlane_f(dl[j], ii);
          PPVec<V> wv;
          wv.load(p + (size_t)j * sh.O);
          float d = 0.0f;
#pragma unroll
          for (int v = 0; v < V; ++v) d = fmaf(wv.v[v], g.v[v], d);
          part = fmaf(cj, d, part);
        }
        const float tot = pp_wave_sum(part);
        if (lane == ii) res += tot;
      }
    }
    if (valid && (!SPLIT || (lane >> 4) == wave)) gx[(size_t)b * sh.I + i] = res;
  }
}

// ---------------------------------------------------------------------------
// backward w.r.t. the weights, atomic variant: gw_t zero-filled by the host
// wrapper, then one 256-byte-contiguous global_atomic_add_f32 wave-instruction
// per (row, input, node, out-chunk).
// ---------------------------------------------------------------------------
template <int N, int RW>
__global__ __launch_bounds__(256) void pp_bwd_weight_atomic_kernel(const float* __restrict__ x,
                                                                   const float* __restrict__ gy,
                                                                   float* __restrict__ gwt,
                                                                   PPShape sh, BasisTab tab) {
  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  const int row0 = (blockIdx.x * 4 + wave) * RW;
  if (row0 >= sh.B) return;
  const int o = blockIdx.y * 64 + lane;
  const bool o_ok = o < sh.O;
  const int oc = o_ok ? o : sh.O - 1;

  float g[RW];
#pragma unroll
  for (int r = 0; r < RW; ++r) {
    const int b = row0 + r;
    g[r] = (b < sh.B && o_ok) ? gy[(size_t)b * sh.O + oc] * sh.out_scale : 0.0f;
  }
  for (int i0 = 0; i0 < sh.I; i0 += 64) {
    const int i = i0 + lane;
    int koff[RW];
    float l[RW][N];
#pragma unroll
    for (int r = 0; r < RW; ++r) {
      const int b = min(row0 + r, sh.B - 1);
      const bool valid = i < sh.I;
      const float xv = valid ? x[(size_t)b * sh.I + i] : 0.0f;
      const int s = pp_segment(xv, sh.length, sh.half, sh.Sf, sh.S);
      const float xi = pp_local_coord(xv, s, sh.length, sh.half, sh.Sf, nullptr);
      pp_lagrange<N>(xi, tab, l[r]);
      koff[r] = (valid ? i : 0) * sh.K + s * sh.stride;
    }
    const int cnt = min(64, sh.I - i0);
    for (int ii = 0; ii < cnt; ++ii) {
#pragma unroll
      for (int r = 0; r < RW; ++r) {
        if (row0 + r >= sh.B) continue;  // wave-uniform
        const int ko = pp_readlane_i(koff[r], ii);
        float* p = gwt + (size_t)ko * sh.O + oc;
#pragma unroll
        for (int j = 0; j < N; ++j) {
          const float lj = pp_readlane_f(l[r][j], ii);
          if (o_ok) atomicAdd(p + (size_t)j * sh.O, lj * g[r]);
        }
      }
    }
  }
}

// ---------------------------------------------------------------------------
// dst[c][r] = src[r][c]  (32x32 LDS tiles, +1 padding)
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void pp_transpose_kernel(const float* __restrict__ src,
                                                           float* __restrict__ dst, int rows,
                                                           int cols) {
  __shared__ float tile[32][33];
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 32 x 8
  const int c0 = blockIdx.x * 32, r0 = blockIdx.y * 32;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int r = r0 + ty + 8 * k, c = c0 + tx;
    if (r < rows && c < cols) tile[ty + 8 * k][tx] = src[(size_t)r * cols + c];
  }
  __syncthreads();
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int c = c0 + ty + 8 * k, r = r0 + tx;
    if (r < rows && c < cols) dst[(size_t)c * rows + r] = tile[tx][ty + 8 * k];
  }
}

// widest vector width V in {4,2,1} with O % V == 0 minimising wave-instructions per row
int pick_vec(int O) {
  int best = 1, best_cost = (O + 63) / 64;
  for (int V : {2, 4}) {
    if (O % V) continue;
    int cost = (O / V + 63) / 64;
    if (cost <= best_cost) { best = V; best_cost = cost; }
  }
  return best;
}

template <int N, int V>
hipError_t fwd_dispatch_rw(const float* x, const float* wt, float* y, int32_t* seg,
                           const PPShape& sh, const BasisTab& tab, hipStream_t st) {
  const int nchunk = (sh.O + 64 * V - 1) / (64 * V);
  // one row per wave with several inputs' loads in flight (gather_unroll) until the waves fill
  // the chip twice over; then four rows per wave (cfg3 hidden, 200 -> 200: 46 vs 81 us at 4096
  // rows, 96 vs 85 at 8192, 109 at 16384 with four rows)
  const long waves1 = (long)sh.B * nchunk;
  if (waves1 >= 8192) {
    constexpr int RW = 4;
    dim3 grid((sh.B + 4 * RW - 1) / (4 * RW), nchunk);
    hipLaunchKernelGGL((pp_fwd_gather_kernel<N, V, RW>), grid, dim3(256), 0, st, x, wt, y, seg, sh,
                       tab);
  } else {
    constexpr int RW = 1;
    dim3 grid((sh.B + 4 * RW - 1) / (4 * RW), nchunk);
    hipLaunchKernelGGL((pp_fwd_gather_kernel<N, V, RW>), grid, dim3(256), 0, st, x, wt, y, seg, sh,
                       tab);
  }
  return hipGetLastError();
}

template <int N>
hipError_t fwd_dispatch_v(const float* x, const float* wt, float* y, int32_t* seg,
                          const PPShape& sh, const BasisTab& tab, hipStream_t st) {
  switch (pick_vec(sh.O)) {
    case 4: return fwd_dispatch_rw<N, 4>(x, wt, y, seg, sh, tab, st);
    case 2: return fwd_dispatch_rw<N, 2>(x, wt, y, seg, sh, tab, st);
    default: return fwd_dispatch_rw<N, 1>(x, wt, y, seg, sh, tab, st);
  }
}

template <int N, int V>
void bwdx_launch(const float* x, const float* gy, const float* wt, float* gx, const PPShape& sh,
                 const BasisTab& tab, hipStream_t st) {
  dim3 grid((sh.B + 3) / 4);
  // few rows: the unrolled loop (38 vs 77 us at 1000 rows of the n = 2, 200 -> 200, 150-segment
  // layer); many rows: the plain one, whose waves hide each other's latency (302 vs 409 us at
  // 16384 rows)
  constexpr int U = gather_unroll(N * V, 1);
  if (sh.B <= 1024)  // one row per block, its inputs spread over the four waves
    hipLaunchKernelGGL((pp_bwd_input_gather_kernel<N, V, U, true>), dim3(sh.B), dim3(256), 0, st, x,
                       gy, wt, gx, sh, tab);
  else if (sh.B <= 4096)
    hipLaunchKernelGGL((pp_bwd_input_gather_kernel<N, V, U>), grid, dim3(256), 0, st, x, gy, wt, gx,
                       sh, tab);
  else
    hipLaunchKernelGGL((pp_bwd_input_gather_kernel<N, V, 1>), grid, dim3(256), 0, st, x, gy, wt, gx,
                       sh, tab);
}

template <int N>
hipError_t bwdx_dispatch_v(const float* x, const float* gy, const float* wt, float* gx,
                           const PPShape& sh, const BasisTab& tab, hipStream_t st) {
  switch (pick_vec(sh.O)) {
    case 4: bwdx_launch<N, 4>(x, gy, wt, gx, sh, tab, st); break;
    case 2: bwdx_launch<N, 2>(x, gy, wt, gx, sh, tab, st); break;
    default: bwdx_launch<N, 1>(x, gy, wt, gx, sh, tab, st);
  }
  return hipGetLastError();
}

template <int N>
hipError_t bwdw_atomic_dispatch(const float* x, const float* gy, float* gwt, const PPShape& sh,
                                const BasisTab& tab, hipStream_t st) {
  const int nchunk = (sh.O + 63) / 64;
  if ((long)sh.B * nchunk >= 4L * 4096) {
    dim3 grid((sh.B + 15) / 16, nchunk);
    hipLaunchKernelGGL((pp_bwd_weight_atomic_kernel<N, 4>), grid, dim3(256), 0, st, x, gy, gwt, sh,
                       tab);
  } else {
    dim3 grid((sh.B + 3) / 4, nchunk);
    hipLaunchKernelGGL((pp_bwd_weight_atomic_kernel<N, 1>), grid, dim3(256), 0, st, x, gy, gwt, sh,
                       tab);
  }
  return hipGetLastError();
}

}  // namespace

#define PP_SWITCH_N(n, CALL)          \
  switch (n) {                        \
    case 1: return CALL(1);           \
    case 2: return CALL(2);           \
    case 3: return CALL(3);           \
    case 4: return CALL(4);           \
    case 5: return CALL(5);           \
    case 6: return CALL(6);           \
    case 7: return CALL(7);           \
    case 8: return CALL(8);           \
    default: return hipErrorInvalidValue; \
  }

hipError_t pp_launch_segment_ids(const float* x, int32_t* seg, const PPShape& sh, hipStream_t st) {
  const long total = (long)sh.B * sh.I;
  const int blocks = (int)std::min<long>((total + 255) / 256, 2048);
  hipLaunchKernelGGL(pp_segment_ids_kernel, dim3(blocks), dim3(256), 0, st, x, seg, total,
                     sh.length, sh.half, sh.Sf, sh.S);
  return hipGetLastError();
}

hipError_t pp_launch_fwd_gather_small(const float* x, const float* wt, float* y, const PPShape& sh,
                                      const BasisTab& tab, hipStream_t st);

// the split variant (one row per block): few rows, a row long enough to be worth four waves
static bool gather_split_rows(const PPShape& sh) { return sh.B <= 1024 && sh.I >= 64; }

hipError_t pp_launch_fwd_gather(const float* x, const float* wt, float* y, int32_t* seg,
                                const PPShape& sh, const BasisTab& tab, hipStream_t st) {
  // few rows of many inputs: one row per block, its inputs split over the four waves
  if (!seg && pp_gather_ln_supported(sh) && gather_split_rows(sh))
    return pp_launch_fwd_gather_small(x, wt, y, sh, tab, st);
#define CALL(N) fwd_dispatch_v<N>(x, wt, y, seg, sh, tab, st)
  PP_SWITCH_N(sh.n, CALL)
#undef CALL
}

// (V, CH) as the row-norm kernels choose them (pp_norm.hip): V by the divisibility of O
static bool gather_ln_combo(int O, int& V, int& CH) {
  V = (O % 4 == 0) ? 4 : (O % 2 == 0 ? 2 : 1);
  const int need = (O + 64 * V - 1) / (64 * V);
  if (V == 4) CH = need <= 1 ? 1 : (need <= 2 ? 2 : (need <= 4 ? 4 : 0));
  else if (V == 2) CH = need <= 2 ? 2 : (need <= 4 ? 4 : (need <= 8 ? 8 : 0));
  else CH = need <= 4 ? 4 : 0;
  return CH != 0;
}

bool pp_gather_ln_supported(const PPShape& sh) {
  int V, CH;
  return sh.n <= 4 && sh.O >= 4 && gather_ln_combo(sh.O, V, CH);
}

namespace {
template <int N, int NORM>
hipError_t fwd_ln_dispatch(const float* x, const float* wt, float* y, float* rstd, int32_t* amax,
                           const PPShape& sh, const BasisTab& tab, float eps, hipStream_t st) {
  int V, CH;
  if (!gather_ln_combo(sh.O, V, CH)) return hipErrorInvalidValue;
  const bool split = gather_split_rows(sh);
  if (NORM == -1 && !split) return hipErrorInvalidValue;  // the plain forward has its own kernels
#define PP_LN_CASE(V_, CH_)                                                                  \
  if (V == V_ && CH == CH_) {                                                                \
    /* two rows per wave once the waves fill the chip and the rows fit the registers; one  */ \
    /* row and several inputs' loads in flight (gather_unroll) for a small batch; one row  */ \
    /* per BLOCK for the smallest                                                          */ \
    if (split)                                                                               \
      hipLaunchKernelGGL((pp_fwd_gather_ln_kernel<N, V_, CH_, NORM, 1, true>), dim3(sh.B),   \
                         dim3(256), 0, st, x, wt, y, rstd, amax, sh, tab, eps);              \
    else if constexpr (NORM >= 0) {                                                          \
      if (CH_ * V_ <= 8 && sh.B >= 8192)                                                     \
        hipLaunchKernelGGL((pp_fwd_gather_ln_kernel<N, V_, CH_, NORM, 2>),                   \
                           dim3((sh.B + 7) / 8), dim3(256), 0, st, x, wt, y, rstd, amax, sh, \
                           tab, eps);                                                        \
      else                                                                                   \
        hipLaunchKernelGGL((pp_fwd_gather_ln_kernel<N, V_, CH_, NORM, 1>),                   \
                           dim3((sh.B + 3) / 4), dim3(256), 0, st, x, wt, y, rstd, amax, sh, \
                           tab, eps);                                                        \
    }                                                                                        \
    return hipGetLastError();                                                                \
  }
  PP_LN_CASE(4, 1) PP_LN_CASE(4, 2) PP_LN_CASE(4, 4)
  PP_LN_CASE(2, 2) PP_LN_CASE(2, 4) PP_LN_CASE(2, 8)
  PP_LN_CASE(1, 4)
#undef PP_LN_CASE
  return hipErrorInvalidValue;
}
}  // namespace

hipError_t pp_launch_fwd_gather_ln(const float* x, const float* wt, float* y, float* rstd,
                                   const PPShape& sh, const BasisTab& tab, float eps,
                                   hipStream_t st) {
  switch (sh.n) {
    case 1: return fwd_ln_dispatch<1, 0>(x, wt, y, rstd, nullptr, sh, tab, eps, st);
    case 2: return fwd_ln_dispatch<2, 0>(x, wt, y, rstd, nullptr, sh, tab, eps, st);
    case 3: return fwd_ln_dispatch<3, 0>(x, wt, y, rstd, nullptr, sh, tab, eps, st);
    case 4: return fwd_ln_dispatch<4, 0>(x, wt, y, rstd, nullptr, sh, tab, eps, st);
    default: return hipErrorInvalidValue;
  }
}

hipError_t pp_launch_fwd_gather_small(const float* x, const float* wt, float* y, const PPShape& sh,
                                      const BasisTab& tab, hipStream_t st) {
  switch (sh.n) {
    case 1: return fwd_ln_dispatch<1, -1>(x, wt, y, nullptr, nullptr, sh, tab, 0.0f, st);
    case 2: return fwd_ln_dispatch<2, -1>(x, wt, y, nullptr, nullptr, sh, tab, 0.0f, st);
    case 3: return fwd_ln_dispatch<3, -1>(x, wt, y, nullptr, nullptr, sh, tab, 0.0f, st);
    case 4: return fwd_ln_dispatch<4, -1>(x, wt, y, nullptr, nullptr, sh, tab, 0.0f, st);
    default: return hipErrorInvalidValue;
  }
}

// the same with MaxAbsNormalizationLast as the epilogue: den[B] and amax[B] for the backward
hipError_t pp_launch_fwd_gather_maxabs(const float* x, const float* wt, float* y, float* den,
                                       int32_t* amax, const PPShape& sh, const BasisTab& tab,
                                       float eps, hipStream_t st) {
  switch (sh.n) {
    case 1: return fwd_ln_dispatch<1, 1>(x, wt, y, den, amax, sh, tab, eps, st);
    case 2: return fwd_ln_dispatch<2, 1>(x, wt, y, den, amax, sh, tab, eps, st);
    case 3: return fwd_ln_dispatch<3, 1>(x, wt, y, den, amax, sh, tab, eps, st);
    case 4: return fwd_ln_dispatch<4, 1>(x, wt, y, den, amax, sh, tab, eps, st);
    default: return hipErrorInvalidValue;
  }
}

hipError_t pp_launch_bwd_input_gather(const float* x, const float* gy, const float* wt, float* gx,
                                      const PPShape& sh, const BasisTab& tab, hipStream_t st) {
  if (sh.n == 1)  // piecewise constant: derivative is identically zero
    return hipMemsetAsync(gx, 0, (size_t)sh.B * sh.I * sizeof(float), st);
#define CALL(N) bwdx_dispatch_v<N>(x, gy, wt, gx, sh, tab, st)
  PP_SWITCH_N(sh.n, CALL)
#undef CALL
}

hipError_t pp_launch_bwd_weight_atomic(const float* x, const float* gy, float* gwt,
                                       const PPShape& sh, const BasisTab& tab, hipStream_t st) {
  hipError_t e = hipMemsetAsync(gwt, 0, (size_t)sh.I * sh.K * sh.O * sizeof(float), st);
  if (e != hipSuccess) return e;
#define CALL(N) bwdw_atomic_dispatch<N>(x, gy, gwt, sh, tab, st)
  PP_SWITCH_N(sh.n, CALL)
#undef CALL
}

hipError_t pp_launch_transpose(const float* src, float* dst, int rows, int cols, hipStream_t st) {
  dim3 grid((cols + 31) / 32, (rows + 31) / 32);
  hipLaunchKernelGGL(pp_transpose_kernel, grid, dim3(256), 0, st, src, dst, rows, cols);
  return hipGetLastError();
}
