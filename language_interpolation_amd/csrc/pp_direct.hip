// Weight gradient for SMALL batches of many-segment layers (PP_BWD_W_DIRECT): every weight row
// is produced by the one wave that owns it.
//
//   gw_t[i][k][o] = out_scale * sum_{(b,j): s[b,i]*stride + j == k} l_j(xi[b,i]) * gy[b][o]
//
// At the reference's own batch sizes (256 rows, 1000 rows) the gradient of a many-segment layer
// is a table of I*K rows of which at most B*n per input are non-zero.  The other formulations
// pay for that with machinery sized for large batches: a zero fill of the whole table plus one
// float atomic per (row, input, node, out) (PP_BWD_W_ATOMIC), a zero fill plus five kernels of
// counting sort and segmented reduce (PP_BWD_W_SORT: 44 us for the 16 -> 1000 input layer at 256
// rows), or a pass over the batch per (input, 64 outs) (PP_BWD_W_REGACC).  Here:
//
//   block = (input i, chunk of 64*V outs, group of kKB weight rows); the block reads column i of x
//   once, keeps each row's segment id and basis values in LDS (B <= kMaxRows), and wave w then
//   walks its weight rows k: for every node j that can land on k it finds the batch rows whose
//   segment matches with a ballot over the ids, in row order, loads their gy chunks (up to eight
//   requests in flight) and accumulates; the finished row -- zeros when nothing matched -- is
//   stored once.  No zero fill, no atomics, no workspace, bitwise reproducible (the order of
//   the sum is j, then b).
//
// Replaces autograd's index_put_(accumulate=True) scatter into w.grad of Piecewise.forward
// (high-order-layers-torch, called at /root/reference/language_interpolation/networks.py:399-426).
#include "pp_device.h"

namespace {

constexpr int kMaxRows = 1024;  // batch rows whose ids fit the block's LDS
constexpr int kKB = 8;          // weight rows per block (16: 17.8 us on the cfg4 input layer at 256 rows, 8: 16.2, 4: 16.6)
constexpr int kDU = 8;          // gy rows requested before the first is used

template <int N, int V>
__global__ __launch_bounds__(256) void pp_bwd_weight_direct_kernel(const float* __restrict__ x,
                                                                   const float* __restrict__ gy,
                                                                   float* __restrict__ gwt,
                                                                   PPShape sh, BasisTab tab) {
  __shared__ int sid[kMaxRows];
  __shared__ float lag[N][kMaxRows];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int i = blockIdx.x;
  const int o = (blockIdx.y * 64 + lane) * V;
  const bool o_ok = o < sh.O;
  const int oc = o_ok ? o : sh.O - V;  // clamp: every lane loads valid memory
  const int k0 = blockIdx.z * kKB;
  const int k1 = min(sh.K, k0 + kKB);
  const int rows64 = (sh.B + 63) & ~63;

  for (int b = threadIdx.x; b < rows64; b += 256) {
    int s = -1;
    float l[N];
#pragma unroll
    for (int j = 0; j < N; ++j) l[j] = 0.0f;
    if (b < sh.B) {
      const float xv = x[(size_t)b * sh.I + i];
      s = pp_segment(xv, sh.length, sh.half, sh.Sf, sh.S);
      const float xi = pp_local_coord(xv, s, sh.length, sh.half, sh.Sf, nullptr);
      pp_lagrange<N>(xi, tab, l);
    }
    sid[b] = s;
#pragma unroll
    for (int j = 0; j < N; ++j) lag[j][b] = l[j];
  }
  __syncthreads();

  // (Sharing one weight row's batch rows over the four waves, partial rows added through LDS,
  // one barrier per weight row, was measured slower: 22 vs 17 us on the 16 -> 1000 layer at 256
  // rows, 56 vs 32 us on the 200 -> 200, K = 300 layer.)
  for (int k = k0 + wave; k < k1; k += 4) {
    float acc[V];
#pragma unroll
    for (int v = 0; v < V; ++v) acc[v] = 0.0f;
#pragma unroll
    for (int j = 0; j < N; ++j) {
      // node j of segment seg lands on row k when seg*stride + j == k
      const int kj = k - j;
      if (kj < 0 || kj % sh.stride != 0) continue;  // wave-uniform
      const int seg = kj / sh.stride;
      if (seg >= sh.S) continue;
      for (int rb = 0; rb < rows64; rb += 64) {
        unsigned long long m = __ballot(sid[rb + lane] == seg);
        while (m) {  // wave-uniform: up to kDU matching rows at a time, in row order
          int bs[kDU];
#pragma unroll
          for (int u = 0; u < kDU; ++u) {
            bs[u] = -1;
            if (m) {
              bs[u] = rb + __builtin_ctzll(m);
              m &= m - 1;
            }
          }
          PPVec<V> g[kDU];
#pragma unroll
          for (int u = 0; u < kDU; ++u)
            if (bs[u] >= 0) g[u].load(gy + (size_t)bs[u] * sh.O + oc);
#pragma unroll
          for (int u = 0; u < kDU; ++u)
            if (bs[u] >= 0) {
              const float lj = lag[j][bs[u]];
#pragma unroll
              for (int v = 0; v < V; ++v) acc[v] = fmaf(lj, g[u].v[v], acc[v]);
            }
        }
      }
    }
    if (o_ok) {
      PPVec<V> out;
#pragma unroll
      for (int v = 0; v < V; ++v) out.v[v] = acc[v] * sh.out_scale;
      out.store(gwt + ((size_t)i * sh.K + k) * sh.O + o);
    }
  }
}

template <int N>
hipError_t direct_dispatch(const float* x, const float* gy, float* gwt, const PPShape& sh,
                           const BasisTab& tab, hipStream_t st) {
  const int V = (sh.O % 4 == 0) ? 4 : (sh.O % 2 == 0 ? 2 : 1);
  dim3 grid(sh.I, (sh.O + 64 * V - 1) / (64 * V), (sh.K + kKB - 1) / kKB);
  switch (V) {
    case 4:
      hipLaunchKernelGGL((pp_bwd_weight_direct_kernel<N, 4>), grid, dim3(256), 0, st, x, gy, gwt, sh,
                         tab);
      break;
    case 2:
      hipLaunchKernelGGL((pp_bwd_weight_direct_kernel<N, 2>), grid, dim3(256), 0, st, x, gy, gwt, sh,
                         tab);
      break;
    default:
      hipLaunchKernelGGL((pp_bwd_weight_direct_kernel<N, 1>), grid, dim3(256), 0, st, x, gy, gwt, sh,
                         tab);
  }
  return hipGetLastError();
}

}  // namespace

bool pp_direct_supported(const PPShape& sh) {
  const long chunks = (sh.O + 63) / 64;  // (grid.y <= 65535, grid.z <= 65535)
  return sh.B <= kMaxRows && sh.n <= 4 && chunks <= 65535 && (sh.K + kKB - 1) / kKB <= 65535;
}

hipError_t pp_launch_bwd_weight_direct(const float* x, const float* gy, float* gwt,
                                       const PPShape& sh, const BasisTab& tab, hipStream_t st) {
  switch (sh.n) {
    case 1: return direct_dispatch<1>(x, gy, gwt, sh, tab, st);
    case 2: return direct_dispatch<2>(x, gy, gwt, sh, tab, st);
    case 3: return direct_dispatch<3>(x, gy, gwt, sh, tab, st);
    case 4: return direct_dispatch<4>(x, gy, gwt, sh, tab, st);
    default: return hipErrorInvalidValue;
  }
}
