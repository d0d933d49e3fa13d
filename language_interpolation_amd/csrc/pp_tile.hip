// Row-tile path: forward of the many-segment layers (cfg3 hidden: discontinuous n = 1, 200 -> 200,
// S = 150; cfg4 input: discontinuous n = 2, 16 -> 1000, S = 128) at large batches.
//
// What bounds the gather kernel on these layers is not HBM but B*I*n*O*4 bytes pulled through
// the vector L1 (2.6 GB per launch for cfg3 hidden at 16384 rows, 64 B/clk per CU); the first
// slab kernel (pp_slab.hip) moved the gathers to LDS but read one float per lane per
// instruction, which is bound by vector-instruction issue.  Here:
//
//  * a workgroup owns R = NW*UW*64 rows x one chunk of 64 outs x a range of inputs, the
//    accumulators of its rows in registers (a wave: UW units of 64 rows = UW*64 VGPRs);
//  * the slab w_t[i, 0..K, chunk] (K x 256 bytes) of input i+1 streams into one LDS buffer by
//    LDS-DMA (global_load_lds, 1 KiB per wave-instruction, no registers) while the waves gather
//    from the other buffer;
//  * one ds_read_b128 serves FOUR rows: lane (slot, q) = (lane / 16, lane % 16) reads floats
//    4q..4q+3 of the weight row that one row of the slot needs, so a wave-instruction moves
//    4 x 256 bytes, conflict-free whatever the four segment ids are (each 16-lane bank group
//    of ds_read_b128 -- {0-3,12-15,20-27}, ... -- holds 16 different q, i.e. all 64 banks);
//  * which row a lane serves rotates with the step: lane (slot, q) computes segment id and
//    basis of ITS row (unit base + lane) once per input, and in step J of 16 it works for the
//    row of the lane J places round its row of 16 lanes -- the byte offset and the basis values
//    come through the DPP row_ror source modifier of the address add / the FMA, not through
//    v_readlane or LDS.  Accumulator J of lane (slot, q) is therefore y[row(slot, ror_J(q))]
//    [4q..4q+3]: a skewed but static layout, undone by the store addresses of the epilogue.
//    Per gather: 1 address add + 4n FMAs for 256 x n multiply-adds (the first slab kernel:
//    ~6 vector instructions per 64 x n).
//
// L2 traffic is (B/R) * I*K*O*4 (0.5 GB for cfg3 hidden at R = 1024) instead of B*I*n*O*4, and
// the gathers run at the LDS rate (256 B/clk per CU).  Tiles are tall and narrow on purpose
// (slab traffic goes with 1/R), so a 16384 x 200 output is only 64 (tile, chunk) pairs: the
// inputs are split over `nsplit` workgroups per pair, whose partial outputs go to the
// caller's workspace and are added in a fixed order by pp_tile_reduce_kernel -- together with the
// following LayerNorm when the caller asked for the fused pair.  Bitwise reproducible.
//
// Replaces: Piecewise.forward of high-order-layers-torch as called from HighOrderMLP
// (networks.py:866-882) and for the input layer of large_standard (networks.py:793-823).
#include <algorithm>
#include <utility>

#include "pp_device.h"

namespace {

typedef __attribute__((address_space(3))) void lds_void;
typedef const __attribute__((address_space(1))) void glb_void;

constexpr int kTileLdsBytes = 160 * 1024;
constexpr int CH = 64;  // outs per workgroup: one 256-byte LDS row

template <int J>
struct IC {
  static constexpr int value = J;
};
template <class F, int... Js>
__device__ __forceinline__ void static_for_impl(F&& f, std::integer_sequence<int, Js...>) {
  (f(IC<Js>{}), ...);
}
template <int CNT, class F>
__device__ __forceinline__ void static_for(F&& f) {
  static_for_impl(f, std::make_integer_sequence<int, CNT>{});
}

// v of the lane J places round this lane's row of 16 lanes (DPP row_ror:J; folds into the
// consuming v_add_u32 / v_fmac_f32 as a source modifier).  Which way "round" is does not
// matter: offsets, basis values and the epilogue's row numbers all come through the same op.
template <int J>
__device__ __forceinline__ int rot_i(int v) {
  if constexpr (J == 0) return v;
  else return __builtin_amdgcn_update_dpp(0, v, 0x120 + J, 0xf, 0xf, false);
}
template <int J>
__device__ __forceinline__ float rot_f(float v) {
  return __int_as_float(rot_i<J>(__float_as_int(v)));
}

struct TilePlan {
  int NW, UW, KP, nchunk, ntile, nsplit, per_split;
  size_t lds;
  bool ok;
};

// N nodes, NW waves per workgroup, UW units of 64 rows per wave, XV4: x rows are 16-byte
// aligned (I % 4 == 0) so four inputs of a row come with one load.
template <int N, int NW, int UW, bool XV4>
__global__ __launch_bounds__(64 * NW) void pp_fwd_tile_kernel(
    const float* __restrict__ x, const float* __restrict__ wt, float* __restrict__ out,
    int32_t* __restrict__ seg_out, PPShape sh, BasisTab tab, int KP, int nchunk, int per_split,
    long part_stride) {
  extern __shared__ __attribute__((aligned(16))) float slab[];  // [2][KP][CH]
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int chunk = blockIdx.x % nchunk, split = blockIdx.x / nchunk;
  const int o0 = chunk * CH;
  const int row0 = (blockIdx.y * NW + wave) * (UW * 64);
  const int i_lo = split * per_split, i_hi = min(sh.I, i_lo + per_split);
  const int bufF = KP * CH;        // floats per buffer
  const int q16 = (lane & 15) * 16;  // this lane's 16-byte piece of a 256-byte slab row

  // ---- LDS-DMA of one slab: instruction q of a wave fills slab rows 4q..4q+3 (lane -> row
  // 4q + lane/16, piece lane%16), source address per lane, clamped into the tensor (rows past K
  // and columns past O land in cells nobody reads).
  const int nq = KP / 4;
  auto slab_load = [&](int i, int buf) {
    const float* src = wt + (size_t)i * sh.K * sh.O;
    const int o = min(o0 + (lane & 15) * 4, sh.O - 4);
    for (int q = wave; q < nq; q += NW) {
      const int row = min(q * 4 + (lane >> 4), sh.K - 1);
      float* dst = slab + buf * bufF + q * (4 * CH);  // wave-uniform; hardware adds lane*16
      __builtin_amdgcn_global_load_lds((glb_void*)(src + (size_t)row * sh.O + o), (lds_void*)dst,
                                       16, 0, 0);
    }
  };

  float acc[UW][16][4];
#pragma unroll
  for (int u = 0; u < UW; ++u)
#pragma unroll
    for (int j = 0; j < 16; ++j)
#pragma unroll
      for (int e = 0; e < 4; ++e) acc[u][j][e] = 0.0f;

  // lane <-> row for the per-(row, input) work
  const float* xrow[UW];
  int myrow[UW];
#pragma unroll
  for (int u = 0; u < UW; ++u) {
    myrow[u] = min(row0 + u * 64 + lane, sh.B - 1);
    xrow[u] = x + (size_t)myrow[u] * sh.I;
  }
  const bool write_seg = seg_out != nullptr && chunk == 0;

  // four inputs of this lane's rows, starting at input g (clamped at the row's end)
  auto load_x = [&](int g, float (&dst)[UW][4]) {
#pragma unroll
    for (int u = 0; u < UW; ++u) {
      if constexpr (XV4) {
        const float4 t = *reinterpret_cast<const float4*>(xrow[u] + g);
        dst[u][0] = t.x; dst[u][1] = t.y; dst[u][2] = t.z; dst[u][3] = t.w;
      } else {
#pragma unroll
        for (int t = 0; t < 4; ++t) dst[u][t] = xrow[u][min(g + t, sh.I - 1)];
      }
    }
  };

  const char* slab_bytes = reinterpret_cast<const char*>(slab);
  float xn[UW][4];
  slab_load(i_lo, 0);
  load_x(i_lo, xn);
  __syncthreads();  // (drains the LDS-DMA: vmcnt(0) + barrier)
  int buf = 0;
  for (int g = i_lo; g < i_hi; g += 4) {
    float xc[UW][4];
#pragma unroll
    for (int u = 0; u < UW; ++u)
#pragma unroll
      for (int t = 0; t < 4; ++t) xc[u][t] = xn[u][t];
    if (g + 4 < i_hi) load_x(g + 4, xn);
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      const int i = g + t;
      if (i < i_hi) {  // workgroup-uniform
        if (i + 1 < i_hi) slab_load(i + 1, buf ^ 1);
#pragma unroll
        for (int u = 0; u < UW; ++u) {
          if (row0 + u * 64 < sh.B) {  // wave-uniform (a wave past the batch still helps with the loads)
            const float xv = xc[u][t];
            const int s = pp_segment(xv, sh.length, sh.half, sh.Sf, sh.S);
            float l[N];
            if constexpr (N > 1) {
              const float xi = pp_local_coord(xv, s, sh.length, sh.half, sh.Sf, nullptr);
              pp_lagrange<N>(xi, tab, l);
            } else {
              l[0] = 1.0f;
            }
            if (write_seg && row0 + u * 64 + lane < sh.B) seg_out[(size_t)myrow[u] * sh.I + i] = s;
            // byte offset of this lane's ROW's first active slab row
            const int koff = (buf * bufF + s * sh.stride * CH) * 4;
            // 16 steps in batches of four: addresses, the 4*N reads in flight together, FMAs
            static_for<4>([&](auto G) {
              float4 w[4][N];
              static_for<4>([&](auto T) {
                constexpr int J = decltype(G)::value * 4 + decltype(T)::value;
                const int a = rot_i<J>(koff) + q16;
#pragma unroll
                for (int m = 0; m < N; ++m)
                  w[decltype(T)::value][m] =
                      *reinterpret_cast<const float4*>(slab_bytes + a + m * (CH * 4));
              });
              static_for<4>([&](auto T) {
                constexpr int TT = decltype(T)::value;
                constexpr int J = decltype(G)::value * 4 + TT;
#pragma unroll
                for (int m = 0; m < N; ++m) {
                  if constexpr (N == 1) {
                    acc[u][J][0] += w[TT][m].x;
                    acc[u][J][1] += w[TT][m].y;
                    acc[u][J][2] += w[TT][m].z;
                    acc[u][J][3] += w[TT][m].w;
                  } else {
                    const float lm = rot_f<J>(l[m]);
                    acc[u][J][0] = fmaf(lm, w[TT][m].x, acc[u][J][0]);
                    acc[u][J][1] = fmaf(lm, w[TT][m].y, acc[u][J][1]);
                    acc[u][J][2] = fmaf(lm, w[TT][m].z, acc[u][J][2]);
                    acc[u][J][3] = fmaf(lm, w[TT][m].w, acc[u][J][3]);
                  }
                }
              });
            });
          }
        }
        __syncthreads();  // slab i+1 has landed; everyone is done with slab i
        buf ^= 1;
      }
    }
  }

  // epilogue: accumulator J of this lane belongs to the row of the lane rot_i<J> names
  const int oc = o0 + (lane & 15) * 4;
  if (oc < sh.O) {
    float* dst = out + (size_t)split * part_stride;
#pragma unroll
    for (int u = 0; u < UW; ++u) {
      static_for<16>([&](auto Jc) {
        constexpr int J = decltype(Jc)::value;
        const int row = row0 + u * 64 + rot_i<J>(lane);
        if (row < sh.B) {
          const float4 v = make_float4(acc[u][J][0] * sh.out_scale, acc[u][J][1] * sh.out_scale,
                                       acc[u][J][2] * sh.out_scale, acc[u][J][3] * sh.out_scale);
          *reinterpret_cast<float4*>(dst + (size_t)row * sh.O + oc) = v;
        }
      });
    }
  }
}

// y = sum of the nsplit partial outputs in split order (float4 per thread).
__global__ __launch_bounds__(256) void pp_tile_reduce_kernel(const float* __restrict__ part,
                                                            float* __restrict__ y, long total4,
                                                            long part_stride4, int nsplit) {
  const long e = (long)blockIdx.x * 256 + threadIdx.x;
  if (e >= total4) return;
  const float4* p = reinterpret_cast<const float4*>(part) + e;
  float4 s = p[0];
  int z = 1;
  for (; z + 4 <= nsplit; z += 4) {
    const float4 a = p[(long)z * part_stride4], b = p[(long)(z + 1) * part_stride4],
                 c = p[(long)(z + 2) * part_stride4], d = p[(long)(z + 3) * part_stride4];
    s.x = (((s.x + a.x) + b.x) + c.x) + d.x;
    s.y = (((s.y + a.y) + b.y) + c.y) + d.y;
    s.z = (((s.z + a.z) + b.z) + c.z) + d.z;
    s.w = (((s.w + a.w) + b.w) + c.w) + d.w;
  }
  for (; z < nsplit; ++z) {
    const float4 a = p[(long)z * part_stride4];
    s.x += a.x; s.y += a.y; s.z += a.z; s.w += a.w;
  }
  reinterpret_cast<float4*>(y)[e] = s;
}

// The same sum with the following non-affine LayerNorm: a wave owns a row (O <= 1024, O % 4 == 0),
// the arithmetic after the sum is pp_layer_norm_fwd_kernel's (mean, centred second moment).
template <int CHN>
__global__ __launch_bounds__(256) void pp_tile_reduce_ln_kernel(const float* __restrict__ part,
                                                               float* __restrict__ y,
                                                               float* __restrict__ rstd, int B,
                                                               int D, long part_stride, int nsplit,
                                                               float eps) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= B) return;
  float v[CHN][4];
#pragma unroll
  for (int c = 0; c < CHN; ++c)
#pragma unroll
    for (int e = 0; e < 4; ++e) v[c][e] = 0.0f;
  const float* p = part + (size_t)row * D;
  for (int z = 0; z < nsplit; ++z) {
#pragma unroll
    for (int c = 0; c < CHN; ++c) {
      const int d = (c * 64 + lane) * 4;
      if (d < D) {
        const float4 t = *reinterpret_cast<const float4*>(p + (size_t)z * part_stride + d);
        v[c][0] += t.x; v[c][1] += t.y; v[c][2] += t.z; v[c][3] += t.w;
      }
    }
  }
  float s = 0.0f;
#pragma unroll
  for (int c = 0; c < CHN; ++c)
#pragma unroll
    for (int e = 0; e < 4; ++e) s += v[c][e];
  const float mean = pp_wave_sum(s) / (float)D;
  float q = 0.0f;
#pragma unroll
  for (int c = 0; c < CHN; ++c)
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const bool in = (c * 64 + lane) * 4 + e < D;
      const float d = in ? v[c][e] - mean : 0.0f;
      v[c][e] = d;
      q = fmaf(d, d, q);
    }
  const float var = pp_wave_sum(q) / (float)D;
  const float rs = 1.0f / sqrtf(var + eps);
#pragma unroll
  for (int c = 0; c < CHN; ++c) {
    const int d = (c * 64 + lane) * 4;
    if (d < D)
      *reinterpret_cast<float4*>(y + (size_t)row * D + d) =
          make_float4(v[c][0] * rs, v[c][1] * rs, v[c][2] * rs, v[c][3] * rs);
  }
  if (lane == 0) rstd[row] = rs;
}

TilePlan tile_plan(const PPShape& sh) {
  TilePlan p{};
  p.ok = false;
  if (sh.n > 4 || sh.O < 4 || sh.O % 4 != 0) return p;
  p.KP = (sh.K + 3) / 4 * 4;
  p.lds = (size_t)2 * p.KP * CH * 4;
  if (p.lds > (size_t)kTileLdsBytes) return p;
  p.nchunk = (sh.O + CH - 1) / CH;
  // rows per workgroup: as tall as the batch allows (slab traffic goes with 1/R)
  if (sh.B > 512) { p.NW = 8; p.UW = 2; }
  else if (sh.B > 256) { p.NW = 4; p.UW = 2; }
  else { p.NW = 4; p.UW = 1; }
  const int R = p.NW * p.UW * 64;
  p.ntile = (sh.B + R - 1) / R;
  // split the inputs until the grid covers the chip (whole groups of four inputs per split)
  const long pairs = (long)p.ntile * p.nchunk;
  int want = (int)std::max(1L, (240 + pairs - 1) / pairs);
  const int groups = (sh.I + 3) / 4;
  want = std::min(want, groups);
  p.per_split = (groups + want - 1) / want * 4;
  p.nsplit = (sh.I + p.per_split - 1) / p.per_split;
  p.ok = true;
  return p;
}

template <int N, int NW, int UW, bool XV4>
hipError_t launch_tile(const float* x, const float* wt, float* out, int32_t* seg, const PPShape& sh,
                       const BasisTab& tab, const TilePlan& p, long part_stride, hipStream_t st) {
  hipError_t e = hipFuncSetAttribute((const void*)pp_fwd_tile_kernel<N, NW, UW, XV4>,
                                     hipFuncAttributeMaxDynamicSharedMemorySize, (int)p.lds);
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL((pp_fwd_tile_kernel<N, NW, UW, XV4>), dim3(p.nchunk * p.nsplit, p.ntile),
                     dim3(64 * NW), p.lds, st, x, wt, out, seg, sh, tab, p.KP, p.nchunk, p.per_split,
                     part_stride);
  return hipGetLastError();
}

template <int N>
hipError_t tile_dispatch(const float* x, const float* wt, float* out, int32_t* seg,
                         const PPShape& sh, const BasisTab& tab, const TilePlan& p,
                         long part_stride, hipStream_t st) {
  const bool xv4 = sh.I % 4 == 0;
#define PP_TILE_CASE(NW_, UW_)                                                               \
  if (p.NW == NW_ && p.UW == UW_)                                                            \
    return xv4 ? launch_tile<N, NW_, UW_, true>(x, wt, out, seg, sh, tab, p, part_stride, st) \
               : launch_tile<N, NW_, UW_, false>(x, wt, out, seg, sh, tab, p, part_stride, st);
  PP_TILE_CASE(8, 2) PP_TILE_CASE(4, 2) PP_TILE_CASE(4, 1)
#undef PP_TILE_CASE
  return hipErrorInvalidValue;
}

}  // namespace

bool pp_tile_fwd_supported(const PPShape& sh) { return tile_plan(sh).ok; }

// Where the row-tile kernel is what PP_METHOD_AUTO should run (given a workspace).
bool pp_tile_fwd_preferred(const PPShape& sh) {
  const TilePlan p = tile_plan(sh);
  if (!p.ok) return false;
  // many-segment layers at large batches: the rows of a tile must reuse the slab's rows
  return sh.K >= 64 && sh.B >= 2048 && sh.O >= 64;
}

size_t pp_tile_fwd_workspace_bytes(const PPShape& sh, bool with_norm) {
  const TilePlan p = tile_plan(sh);
  if (!p.ok) return 0;
  if (p.nsplit == 1 && !with_norm) return 0;
  return (size_t)p.nsplit * sh.B * sh.O * sizeof(float);
}

// y = layer(x); with rstd != nullptr y = LayerNorm(layer(x)) and rstd[B] as pp_layer_norm_fwd_f32
// writes it (O <= 1024).  ws: pp_tile_fwd_workspace_bytes(sh, rstd != nullptr) bytes.
hipError_t pp_launch_fwd_tile(const float* x, const float* wt, float* y, int32_t* seg, void* ws,
                              float* rstd, float ln_eps, const PPShape& sh, const BasisTab& tab,
                              hipStream_t st) {
  const TilePlan p = tile_plan(sh);
  if (!p.ok) return hipErrorInvalidValue;
  if (rstd && sh.O > 1024) return hipErrorInvalidValue;
  const bool direct = p.nsplit == 1 && !rstd;
  float* out = direct ? y : static_cast<float*>(ws);
  const long part_stride = (long)sh.B * sh.O;
  hipError_t e;
  switch (sh.n) {
    case 1: e = tile_dispatch<1>(x, wt, out, seg, sh, tab, p, part_stride, st); break;
    case 2: e = tile_dispatch<2>(x, wt, out, seg, sh, tab, p, part_stride, st); break;
    case 3: e = tile_dispatch<3>(x, wt, out, seg, sh, tab, p, part_stride, st); break;
    case 4: e = tile_dispatch<4>(x, wt, out, seg, sh, tab, p, part_stride, st); break;
    default: return hipErrorInvalidValue;
  }
  if (e != hipSuccess || direct) return e;
  if (rstd) {
    const int need = (sh.O + 255) / 256;
    dim3 grid((sh.B + 3) / 4), block(256);
    if (need <= 1)
      hipLaunchKernelGGL((pp_tile_reduce_ln_kernel<1>), grid, block, 0, st, out, y, rstd, sh.B, sh.O,
                         part_stride, p.nsplit, ln_eps);
    else if (need <= 2)
      hipLaunchKernelGGL((pp_tile_reduce_ln_kernel<2>), grid, block, 0, st, out, y, rstd, sh.B, sh.O,
                         part_stride, p.nsplit, ln_eps);
    else
      hipLaunchKernelGGL((pp_tile_reduce_ln_kernel<4>), grid, block, 0, st, out, y, rstd, sh.B, sh.O,
                         part_stride, p.nsplit, ln_eps);
  } else {
    const long total4 = part_stride / 4;
    hipLaunchKernelGGL(pp_tile_reduce_kernel, dim3((unsigned)((total4 + 255) / 256)), dim3(256), 0,
                       st, out, y, total4, total4, p.nsplit);
  }
  return hipGetLastError();
}
