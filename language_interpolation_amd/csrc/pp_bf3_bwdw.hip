// bf16x3 backward-weight (scheme: pp_bf3_fwd.hip, header comment).
#include "pp_bf3_common.h"

namespace {

// ---------------------------------------------------------------------------
// backward-weight on the bf16x3 scheme:
//   gw[(i,k), o] = sum_b Phi[b,i,k] * gy[b,o]       (reduction over rows on bf16 MFMA)
// gy is split once per call by pp_bf3_pack_gy_kernel into G[plane][16-row step][Opad]
// [16 rows] (B-operand fragments, 16-byte coalesced loads, two steps in flight).  Block
// = 4 waves = 4 column tiles of 32*NT outs of ONE 32-input tile over one chunk of rows;
// per 16-row step the block evaluates + splits the basis of its 16 x 32 (row, input)
// pairs once (2 pairs per lane) into LDS ([plane][k][half][input] x 16 B, three buffers:
// step t+2 is written and step t+1's fragments are read under the MFMAs of step t) and
// every wave runs 6*K*NT MFMAs into its K x NT accumulators.  Per-chunk partial sums go
// to the workspace and are added in a fixed order by pp_dense_reduce_kernel.
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void pp_bf3_pack_gy_kernel(const float* __restrict__ gy,
                                                             u32x4* __restrict__ G, int B, int O,
                                                             int Opad, int nbs) {
  const long id = (long)blockIdx.x * 256 + threadIdx.x;
  if (id >= (long)nbs * Opad) return;
  const int o = (int)(id % Opad);
  const int t = (int)(id / Opad);
  unsigned pc[3][16];
#pragma unroll
  for (int j = 0; j < 16; ++j) {
    const int b = 16 * t + j;
    const float v = (b < B && o < O) ? gy[(size_t)b * O + o] : 0.0f;
    split3(v, pc[0][j], pc[1][j], pc[2][j]);
  }
#pragma unroll
  for (int p = 0; p < 3; ++p)
#pragma unroll
    for (int hh = 0; hh < 2; ++hh) {
      u32x4 w;
      w.x = (pc[p][8 * hh + 0] >> 16) | pc[p][8 * hh + 1];
      w.y = (pc[p][8 * hh + 2] >> 16) | pc[p][8 * hh + 3];
      w.z = (pc[p][8 * hh + 4] >> 16) | pc[p][8 * hh + 5];
      w.w = (pc[p][8 * hh + 6] >> 16) | pc[p][8 * hh + 7];
      G[(((size_t)p * nbs + t) * Opad + o) * 2 + hh] = w;
    }
}

template <int N, int K, int NT>
__global__ __launch_bounds__(256) void pp_bwd_w_bf3_kernel(const float* __restrict__ x,
                                                           const void* __restrict__ G,
                                                           float* __restrict__ out, PPShape sh,
                                                           BasisTab tab, DenseCoord3 dc,
                                                           int rows_per_chunk, int nbs, int Opad,
                                                           float scale) {
  extern __shared__ __attribute__((aligned(16))) u32x4 phiL[];  // [3 buf][3 plane][K][2][32] x 16 B
  constexpr int kBufU = 3 * K * 2 * 32;
  const int lane = threadIdx.x & 63;
  const int wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int r = lane & 31, h = lane >> 5;
  const int i0 = blockIdx.x * 32;
  const int chunk = blockIdx.y;
  const int o0 = (blockIdx.z * 4 + wv) * (32 * NT);
  const int cbeg = chunk * rows_per_chunk;  // multiple of 16
  const int cend = min(sh.B, cbeg + rows_per_chunk);
  const int nst = (cend - cbeg + 15) >> 4;
  const int t0 = cbeg >> 4;

  // producer: input i0 + il, rows cbeg + 16 t + 2 rp, + 1
  const int il = threadIdx.x & 31, rp = threadIdx.x >> 5;
  const int ic = min(i0 + il, sh.I - 1);
  const rsrc_t rx = make_rsrc3(x + (size_t)cbeg * sh.I, (unsigned)(cend - cbeg) * sh.I * 4u);
  const int vx = (2 * rp * sh.I + ic) * 4;
  const int xrow = sh.I * 4, xstep = 16 * sh.I * 4;
  unsigned* phiW = reinterpret_cast<unsigned*>(phiL);
  const int pw_base = ((rp >> 2) * 32 + il) * 4 + (rp & 3);
  auto xload = [&](int t, float (&xv)[2]) {
    xv[0] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rx, vx, t * xstep, 0));
    xv[1] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rx, vx + xrow, t * xstep, 0));
  };
  auto produce = [&](int t, int buf, const float (&xv)[2]) {
    const int row = cbeg + 16 * t + 2 * rp;
    float pa[K], pb[K];
    phi3<N, K>(xv[0], row < cend, sh, tab, dc, pa);
    phi3<N, K>(xv[1], row + 1 < cend, sh, tab, dc, pb);
#pragma unroll
    for (int k = 0; k < K; ++k) {
      unsigned a1, a2, a3, c1, c2, c3;
      split3(pa[k], a1, a2, a3);
      split3(pb[k], c1, c2, c3);
      const int base = (buf * kBufU + k * 64) * 4 + pw_base;
      phiW[base] = (a1 >> 16) | c1;
      phiW[base + K * 64 * 4] = (a2 >> 16) | c2;
      phiW[base + 2 * K * 64 * 4] = (a3 >> 16) | c3;
    }
  };
  // consumer
  const rsrc_t rg = make_rsrc3(G, (unsigned)(3u * nbs * Opad * 32u));
  int vcol[NT];
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) vcol[nt] = (min(o0 + nt * 32 + r, Opad - 1) * 2 + h) * 16;
  const int gstep = Opad * 32, gplane = nbs * gstep;
  auto bload = [&](int t, bf16x8 (&bf)[3][NT]) {
#pragma unroll
    for (int p = 0; p < 3; ++p)
#pragma unroll
      for (int nt = 0; nt < NT; ++nt)
        bf[p][nt] = __builtin_bit_cast(
            bf16x8, __builtin_amdgcn_raw_buffer_load_b128(rg, vcol[nt],
                                                          p * gplane + (t0 + t) * gstep, 0));
  };
  auto aload = [&](int buf, bf16x8 (&a)[K][3]) {
#pragma unroll
    for (int k = 0; k < K; ++k)
#pragma unroll
      for (int p = 0; p < 3; ++p)
        a[k][p] = __builtin_bit_cast(bf16x8, phiL[buf * kBufU + ((p * K + k) * 2 + h) * 32 + r]);
  };
  f32x16 acc[K][NT];
#pragma unroll
  for (int k = 0; k < K; ++k)
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
      for (int q = 0; q < 16; ++q) acc[k][nt][q] = 0.0f;
  auto mm = [&](const bf16x8 (&a)[K][3], const bf16x8 (&bf)[3][NT]) {
    constexpr int PA[6] = {2, 0, 1, 1, 0, 0};
    constexpr int PB[6] = {0, 2, 1, 0, 1, 0};
#pragma unroll
    for (int k = 0; k < K; ++k)
#pragma unroll
      for (int c = 0; c < 6; ++c)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
          acc[k][nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[k][PA[c]], bf[PB[c]][nt],
                                                               acc[k][nt], 0, 0, 0);
  };

  float xr[2][2];
  bf16x8 Bf[3][3][NT], A[2][K][3];
  {
    float x0[2], x1[2];
    xload(0, x0);
    xload(1, x1);
    xload(2, xr[0]);
    bload(0, Bf[0]);
    bload(1, Bf[1]);
    produce(0, 0, x0);
    produce(1, 1, x1);
  }
  __syncthreads();
  aload(0, A[0]);
  // step t (J = t mod 6 at compile time): x of step t+3 and gy fragments of step t+2 are
  // issued, the basis fragments of step t+1 are read, the basis of step t+2 is written,
  // under the MFMAs of step t.  Steps past the chunk evaluate to zero (valid flags).
  auto stepfn = [&](int t, auto jc) {
    constexpr int J = decltype(jc)::value;
    xload(t + 3, xr[(J + 1) & 1]);
    bload(t + 2, Bf[(J + 2) % 3]);
    aload((J + 1) % 3, A[(J + 1) & 1]);
    produce(t + 2, (J + 2) % 3, xr[J & 1]);
    mm(A[J & 1], Bf[J % 3]);
    __syncthreads();
  };
  for (int t = 0; t < nst; t += 6) {
    stepfn(t, std::integral_constant<int, 0>{});
    if (t + 1 < nst) stepfn(t + 1, std::integral_constant<int, 1>{});
    if (t + 2 < nst) stepfn(t + 2, std::integral_constant<int, 2>{});
    if (t + 3 < nst) stepfn(t + 3, std::integral_constant<int, 3>{});
    if (t + 4 < nst) stepfn(t + 4, std::integral_constant<int, 4>{});
    if (t + 5 < nst) stepfn(t + 5, std::integral_constant<int, 5>{});
  }
  if (o0 >= sh.O) return;  // wave-uniform: this wave ran on clamped columns
  float* dst = out + (size_t)chunk * sh.I * K * sh.O;
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) {
    const int col = o0 + nt * 32 + r;
    if (col < sh.O) {
#pragma unroll
      for (int k = 0; k < K; ++k)
#pragma unroll
        for (int q = 0; q < 16; ++q) {
          const int row = i0 + (q & 3) + 8 * (q >> 2) + 4 * h;  // input index
          if (row < sh.I) dst[((size_t)row * K + k) * sh.O + col] = acc[k][nt][q] * scale;
        }
    }
  }
}

}  // namespace

// ---- backward-weight ----
namespace {
struct WPlan3 {
  int nt, ocols, nchunk, rows_per_chunk, nbs, opad;
  size_t g_bytes, part_bytes;
};

WPlan3 wplan3(const PPShape& sh) {
  WPlan3 p;
  // K x NT accumulators of 16 registers per wave; 32-out waves when O <= 128 so that
  // all four column waves have work
  p.nt = (sh.K <= 6 && sh.O > 128) ? 2 : 1;
  p.opad = ((sh.O + 31) / 32) * 32;
  p.ocols = (sh.O + 128 * p.nt - 1) / (128 * p.nt);
  const int tiles = ((sh.I + 31) / 32) * p.ocols;
  int target = 256;
  if (const char* v = getenv("PP_BF3_BWDW_BLOCKS")) target = std::max(1, atoi(v));
  int nchunk = std::max(1, target / tiles);
  const int max_chunks = std::max(1, sh.B / 128);  // >= 8 steps of 16 rows per block
  nchunk = std::min(nchunk, max_chunks);
  int rpc = (sh.B + nchunk - 1) / nchunk;
  rpc = (rpc + 15) / 16 * 16;
  p.rows_per_chunk = rpc;
  p.nchunk = (sh.B + rpc - 1) / rpc;
  p.nbs = (sh.B + 15) / 16;  // (steps prefetched past the end read out of range: zeros, unused)
  p.g_bytes = (size_t)3 * p.nbs * p.opad * 32;
  p.part_bytes = p.nchunk > 1 ? (size_t)p.nchunk * sh.I * sh.K * sh.O * sizeof(float) : 0;
  return p;
}

template <int N, int K, int NT>
hipError_t launch_bwd_w_bf3(const float* x, const void* G, float* out, const PPShape& sh,
                            const BasisTab& tab, const DenseCoord3& dc, const WPlan3& p,
                            float scale, hipStream_t st) {
  const size_t lds = (size_t)3 * 3 * K * 2 * 32 * 16;
  if (lds > 48 * 1024) {
    static bool once = false;  // per instantiation
    if (!once) {
      hipError_t e = hipFuncSetAttribute((const void*)pp_bwd_w_bf3_kernel<N, K, NT>,
                                         hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
      if (e != hipSuccess) return e;
      once = true;
    }
  }
  dim3 grid((sh.I + 31) / 32, p.nchunk, p.ocols);
  hipLaunchKernelGGL((pp_bwd_w_bf3_kernel<N, K, NT>), grid, dim3(256), lds, st, x, G, out, sh, tab,
                     dc, p.rows_per_chunk, p.nbs, p.opad, scale);
  return hipGetLastError();
}
}  // namespace

bool pp_bf3_bwd_w_supported(const PPShape& sh) {
  if (!pp_dense_supported(sh)) return false;
  const WPlan3 p = wplan3(sh);
  return p.g_bytes < ((size_t)1 << 31);
}

size_t pp_bf3_bwd_w_workspace_bytes(const PPShape& sh) {
  const WPlan3 p = wplan3(sh);
  return p.g_bytes + p.part_bytes;
}

hipError_t pp_launch_bwd_w_bf3(const float* x, const float* gy, float* gwt, void* ws,
                               const PPShape& sh, const BasisTab& tab, hipStream_t st) {
  DenseCoord3 dc;
  dc.w2 = 2.0f / (float)sh.S;
  dc.inv_w = (float)sh.S / 2.0f;
  dc.inv_len = 1.0f / sh.length;
  const WPlan3 p = wplan3(sh);
  const long nthr = (long)p.nbs * p.opad;
  hipLaunchKernelGGL(pp_bf3_pack_gy_kernel, dim3((unsigned)((nthr + 255) / 256)), dim3(256), 0, st,
                     gy, static_cast<u32x4*>(ws), sh.B, sh.O, p.opad, p.nbs);
  float* part = reinterpret_cast<float*>(static_cast<char*>(ws) + p.g_bytes);
  float* out = p.nchunk > 1 ? part : gwt;
  const float scale = p.nchunk > 1 ? 1.0f : sh.out_scale;
#define X(N_, K_)                                                                          \
  if (sh.n == N_ && sh.K == K_) {                                                          \
    hipError_t e;                                                                          \
    e = hipErrorInvalidValue;                                                              \
    if constexpr (K_ <= 6) {                                                               \
      if (p.nt == 2) e = launch_bwd_w_bf3<N_, K_, 2>(x, ws, out, sh, tab, dc, p, scale, st); \
    }                                                                                      \
    if (p.nt == 1) e = launch_bwd_w_bf3<N_, K_, 1>(x, ws, out, sh, tab, dc, p, scale, st); \
    if (e != hipSuccess || p.nchunk <= 1) return e;                                        \
    pp_launch_chunk_reduce(part, gwt, (long)sh.I * sh.K * sh.O, p.nchunk, sh.out_scale, st); \
    return hipGetLastError();                                                              \
  }
  PP_BF3_COMBOS(X)
#undef X
  return hipErrorInvalidValue;
}
