// Weight gradient by counting sort + segmented reduce (float atomics on gw only for buckets
// that are shared between segments or span reduce units; see below).
//
//   gw_t[i,k,:] = sum over rows b whose active range covers k of  l_j(xi[b,i]) * gy[b,:]
//
// Rows are bucketed per (input i, segment s) with a two-level counting sort
// (LDS histograms per block -> per-bucket chunk prefix -> scan of bucket totals
// -> placement), which leaves the (row id, xi, bucket id) records sorted by bucket.
// The sorted record array is then cut into equal units of E records; one
// wavefront owns one unit x one 64*V-wide out chunk, so every wave does the same
// amount of gather work however skewed the buckets are (text: the space
// character's bucket holds ~18 % of all rows).  Walking its records, a wave keeps
// the n node accumulators of the current bucket in registers (each gy row is read
// ONCE for all n nodes) and flushes them when the bucket id changes: with a plain
// store when the bucket lies wholly inside the unit and its rows are exclusive
// (discontinuous layers), else with float atomics into the zero-filled gw
// (continuous layers share a node between neighbouring segments; long buckets
// span units).  Atomic traffic is therefore O(units + buckets) rows, not O(B*I).
//
// Replaces autograd's index_put_(accumulate=True) scatter into w.grad
// (loss.backward() at networks.py:164-165 of the reference).
#include "pp_device.h"

namespace {

constexpr int kMaxBins = 12288;  // LDS histogram ints per block (48 KiB)

struct SortPlan {
  int CG;      // columns per block group (power of two)
  int ncg;     // column groups
  int RC;      // rows per chunk
  int nrc;     // row chunks
  int NB;      // buckets = I*S
  int E;       // records per reduce unit
  int nunit;   // reduce units
  size_t off_counts, off_totals, off_bucket, off_perm_b, off_perm_xi, off_perm_bkt, bytes;
};

int pick_vec(int O);

size_t align16(size_t v) { return (v + 15) & ~(size_t)15; }

SortPlan make_plan(const PPShape& sh) {
  SortPlan p;
  int cg = 64;
  while (cg > 1 && (long)cg * sh.S > kMaxBins) cg >>= 1;  // histogram must fit LDS
  while (cg > 1 && cg / 2 >= sh.I) cg >>= 1;              // no wider than the row
  p.CG = cg;
  p.ncg = (sh.I + cg - 1) / cg;
  const int nrc_target = std::max(1, 256 / p.ncg);
  p.RC = std::max(128, (((sh.B + nrc_target - 1) / nrc_target) + 63) & ~63);
  p.nrc = (sh.B + p.RC - 1) / p.RC;
  p.NB = sh.I * sh.S;
  {
    const int V = pick_vec(sh.O);
    const long nchunk = (sh.O + 64 * V - 1) / (64 * V);
    const long total = (long)sh.B * sh.I;
    long e = (total * nchunk / 8192 + 63) & ~63L;  // aim for ~8192 waves
    p.E = (int)std::max<long>(64, std::min<long>(512, e));
    p.nunit = (int)((total + p.E - 1) / p.E);
  }
  size_t o = 0;
  p.off_counts = o;  o += align16((size_t)p.nrc * p.NB * sizeof(int));
  p.off_totals = o;  o += align16((size_t)p.NB * sizeof(int));
  p.off_bucket = o;  o += align16((size_t)(p.NB + 1) * sizeof(int));
  p.off_perm_b = o;  o += align16((size_t)sh.B * sh.I * sizeof(int));
  p.off_perm_xi = o; o += align16((size_t)sh.B * sh.I * sizeof(float));
  p.off_perm_bkt = o; o += align16((size_t)sh.B * sh.I * sizeof(int));
  p.bytes = o;
  return p;
}

// K1 / K4 share the traversal: block (cg, rc) visits rows [rc*RC, ...) x columns
// [cg*CG, ...), thread t -> column t % CG, row phase t / CG (coalesced along a row).
template <bool PLACE>
__global__ __launch_bounds__(256) void pp_sort_pass_kernel(const float* __restrict__ x,
                                                           int* __restrict__ counts,
                                                           const int* __restrict__ bucket_off,
                                                           int* __restrict__ perm_b,
                                                           float* __restrict__ perm_xi,
                                                           int* __restrict__ perm_bkt, PPShape sh,
                                                           int CG, int RC, int NB) {
  extern __shared__ int hist[];  // CG*S
  const int t = threadIdx.x;
  const int col0 = blockIdx.x * CG;
  const int ncol = min(CG, sh.I - col0);
  const int rc = blockIdx.y;
  const int nbin = ncol * sh.S;
  const size_t cbase = (size_t)rc * NB + (size_t)col0 * sh.S;
  if (PLACE) {
    for (int idx = t; idx < nbin; idx += 256)
      hist[idx] = bucket_off[col0 * sh.S + idx] + counts[cbase + idx];
  } else {
    for (int idx = t; idx < nbin; idx += 256) hist[idx] = 0;
  }
  __syncthreads();
  const int c = t & (CG - 1);
  const int rstep = 256 / CG;
  const int row_end = min(sh.B, (rc + 1) * RC);
  if (c < ncol) {
    for (int b = rc * RC + t / CG; b < row_end; b += rstep) {
      const float xv = x[(size_t)b * sh.I + col0 + c];
      const int s = pp_segment(xv, sh.length, sh.half, sh.Sf, sh.S);
      if (PLACE) {
        const int pos = atomicAdd(&hist[c * sh.S + s], 1);
        perm_b[pos] = b;
        perm_xi[pos] = pp_local_coord(xv, s, sh.length, sh.half, sh.Sf, nullptr);
        perm_bkt[pos] = (col0 + c) * sh.S + s;
      } else {
        atomicAdd(&hist[c * sh.S + s], 1);
      }
    }
  }
  if (!PLACE) {
    __syncthreads();
    for (int idx = t; idx < nbin; idx += 256) counts[cbase + idx] = hist[idx];
  }
}

// K2: per bucket, exclusive prefix over the row chunks (in place) and the total.
// Thread per bucket (coalesced across buckets); the chunk loads are issued eight
// at a time so the loop is not one dependent L2 round trip per chunk.
__global__ __launch_bounds__(256) void pp_sort_prefix_kernel(int* __restrict__ counts,
                                                             int* __restrict__ totals, int NB,
                                                             int nrc) {
  const int bkt = blockIdx.x * 256 + threadIdx.x;
  if (bkt >= NB) return;
  int run = 0;
  int rc = 0;
  for (; rc + 8 <= nrc; rc += 8) {
    int v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) v[u] = counts[(size_t)(rc + u) * NB + bkt];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      counts[(size_t)(rc + u) * NB + bkt] = run;
      run += v[u];
    }
  }
  for (; rc < nrc; ++rc) {
    const size_t idx = (size_t)rc * NB + bkt;
    const int v = counts[idx];
    counts[idx] = run;
    run += v;
  }
  totals[bkt] = run;
}

// K3: exclusive scan of the bucket totals (single block).
__global__ __launch_bounds__(1024) void pp_sort_scan_kernel(const int* __restrict__ totals,
                                                            int* __restrict__ bucket_off, int NB) {
  __shared__ int sums[1024];
  const int t = threadIdx.x;
  const int per = (NB + 1023) / 1024;
  const int beg = min(NB, t * per), end = min(NB, beg + per);
  int s = 0;
  for (int e = beg; e < end; ++e) s += totals[e];
  sums[t] = s;
  __syncthreads();
  for (int off = 1; off < 1024; off <<= 1) {
    const int v = t >= off ? sums[t - off] : 0;
    __syncthreads();
    sums[t] += v;
    __syncthreads();
  }
  int run = sums[t] - s;
  for (int e = beg; e < end; ++e) {
    bucket_off[e] = run;
    run += totals[e];
  }
  if (t == 1023) bucket_off[NB] = sums[1023];
}

// K5: balanced segmented reduce.  One wave per (unit of E sorted records, out chunk).
template <int N, int V>
__global__ __launch_bounds__(256) void pp_bwd_weight_segreduce_kernel(
    const float* __restrict__ gy, float* __restrict__ gwt, const int* __restrict__ bucket_off,
    const int* __restrict__ perm_b, const float* __restrict__ perm_xi,
    const int* __restrict__ perm_bkt, PPShape sh, BasisTab tab, int E, int total) {
  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  const int unit = blockIdx.x * 4 + wave;
  const int e_beg = unit * E;
  if (e_beg >= total) return;
  const int e_end = min(total, e_beg + E);
  const int o = (blockIdx.y * 64 + lane) * V;
  const bool o_ok = o < sh.O;
  const int oc = o_ok ? o : sh.O - V;

  float acc[N][V];
#pragma unroll
  for (int j = 0; j < N; ++j)
#pragma unroll
    for (int v = 0; v < V; ++v) acc[j][v] = 0.0f;
  int cur = -1;

  auto flush = [&]() {
    if (cur < 0) return;
    const int i = cur / sh.S;
    const int s = cur - i * sh.S;
    const size_t row0 = (size_t)i * sh.K + (size_t)s * sh.stride;
    // exclusive rows + bucket wholly inside this unit -> nobody else writes them
    const bool excl = sh.disc && bucket_off[cur] >= e_beg && bucket_off[cur + 1] <= e_end;
    if (o_ok) {
#pragma unroll
      for (int j = 0; j < N; ++j) {
        float* p = gwt + (row0 + j) * sh.O + o;
        if (excl) {
          PPVec<V> out;
#pragma unroll
          for (int v = 0; v < V; ++v) out.v[v] = acc[j][v] * sh.out_scale;
          out.store(p);
        } else {
#pragma unroll
          for (int v = 0; v < V; ++v) atomicAdd(p + v, acc[j][v] * sh.out_scale);
        }
      }
    }
#pragma unroll
    for (int j = 0; j < N; ++j)
#pragma unroll
      for (int v = 0; v < V; ++v) acc[j][v] = 0.0f;
  };

  for (int e0 = e_beg; e0 < e_end; e0 += 64) {
    const int e = e0 + lane;
    const bool valid = e < e_end;
    const int bb = valid ? perm_b[e] : 0;
    const int bk = valid ? perm_bkt[e] : 0;
    const float xi = valid ? perm_xi[e] : 0.0f;
    float l[N];
    pp_lagrange<N>(xi, tab, l);
    const int cnt = min(64, e_end - e0);
    for (int g = 0; g < cnt; g += 8) {
      PPVec<V> rows[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int ee = min(g + u, cnt - 1);
        const int b = pp_readlane_i(bb, ee);
        rows[u].load(gy + (size_t)b * sh.O + oc);
      }
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        if (g + u < cnt) {  // wave-uniform
          const int ee = g + u;
          const int bke = pp_readlane_i(bk, ee);
          if (bke != cur) {
            flush();
            cur = bke;
          }
#pragma unroll
          for (int j = 0; j < N; ++j) {
            const float c = pp_readlane_f(l[j], ee);
#pragma unroll
            for (int v = 0; v < V; ++v) acc[j][v] = fmaf(c, rows[u].v[v], acc[j][v]);
          }
        }
      }
    }
  }
  flush();
}

int pick_vec(int O) {
  int best = 1, best_cost = (O + 63) / 64;
  for (int V : {2, 4}) {
    if (O % V) continue;
    int cost = (O / V + 63) / 64;
    if (cost <= best_cost) { best = V; best_cost = cost; }
  }
  return best;
}

template <int N>
hipError_t reduce_dispatch(const float* gy, float* gwt, const int* bucket_off, const int* perm_b,
                           const float* perm_xi, const int* perm_bkt, const SortPlan& pl,
                           const PPShape& sh, const BasisTab& tab, hipStream_t st) {
  const int V = pick_vec(sh.O);
  const int nchunk = (sh.O + 64 * V - 1) / (64 * V);
  const int total = sh.B * sh.I;
  dim3 grid((pl.nunit + 3) / 4, nchunk);
  switch (V) {
    case 4:
      hipLaunchKernelGGL((pp_bwd_weight_segreduce_kernel<N, 4>), grid, dim3(256), 0, st, gy, gwt,
                         bucket_off, perm_b, perm_xi, perm_bkt, sh, tab, pl.E, total);
      break;
    case 2:
      hipLaunchKernelGGL((pp_bwd_weight_segreduce_kernel<N, 2>), grid, dim3(256), 0, st, gy, gwt,
                         bucket_off, perm_b, perm_xi, perm_bkt, sh, tab, pl.E, total);
      break;
    default:
      hipLaunchKernelGGL((pp_bwd_weight_segreduce_kernel<N, 1>), grid, dim3(256), 0, st, gy, gwt,
                         bucket_off, perm_b, perm_xi, perm_bkt, sh, tab, pl.E, total);
  }
  return hipGetLastError();
}

}  // namespace

bool pp_sort_supported(const PPShape& sh) {
  return sh.S <= kMaxBins && sh.stride >= 1 && (long)sh.I * sh.S < (1L << 30);
}

size_t pp_sort_workspace_bytes(const PPShape& sh) { return make_plan(sh).bytes; }

hipError_t pp_launch_bwd_weight_sort(const float* x, const float* gy, float* gwt, void* ws,
                                     const PPShape& sh, const BasisTab& tab, hipStream_t st) {
  const SortPlan p = make_plan(sh);
  char* base = static_cast<char*>(ws);
  int* counts = reinterpret_cast<int*>(base + p.off_counts);
  int* totals = reinterpret_cast<int*>(base + p.off_totals);
  int* bucket_off = reinterpret_cast<int*>(base + p.off_bucket);
  int* perm_b = reinterpret_cast<int*>(base + p.off_perm_b);
  float* perm_xi = reinterpret_cast<float*>(base + p.off_perm_xi);
  int* perm_bkt = reinterpret_cast<int*>(base + p.off_perm_bkt);
  hipError_t e = hipMemsetAsync(gwt, 0, (size_t)sh.I * sh.K * sh.O * sizeof(float), st);
  if (e != hipSuccess) return e;

  const dim3 sgrid(p.ncg, p.nrc);
  const size_t lds = (size_t)p.CG * sh.S * sizeof(int);
  hipLaunchKernelGGL((pp_sort_pass_kernel<false>), sgrid, dim3(256), lds, st, x, counts,
                     (const int*)nullptr, (int*)nullptr, (float*)nullptr, (int*)nullptr, sh, p.CG, p.RC,
                     p.NB);
  hipLaunchKernelGGL(pp_sort_prefix_kernel, dim3((p.NB + 255) / 256), dim3(256), 0, st, counts,
                     totals, p.NB, p.nrc);
  hipLaunchKernelGGL(pp_sort_scan_kernel, dim3(1), dim3(1024), 0, st, totals, bucket_off, p.NB);
  hipLaunchKernelGGL((pp_sort_pass_kernel<true>), sgrid, dim3(256), lds, st, x, counts,
                     (const int*)bucket_off, perm_b, perm_xi, perm_bkt, sh, p.CG, p.RC, p.NB);
  e = hipGetLastError();
  if (e != hipSuccess) return e;
  switch (sh.n) {
    case 1: return reduce_dispatch<1>(gy, gwt, bucket_off, perm_b, perm_xi, perm_bkt, p, sh, tab, st);
    case 2: return reduce_dispatch<2>(gy, gwt, bucket_off, perm_b, perm_xi, perm_bkt, p, sh, tab, st);
    case 3: return reduce_dispatch<3>(gy, gwt, bucket_off, perm_b, perm_xi, perm_bkt, p, sh, tab, st);
    case 4: return reduce_dispatch<4>(gy, gwt, bucket_off, perm_b, perm_xi, perm_bkt, p, sh, tab, st);
    case 5: return reduce_dispatch<5>(gy, gwt, bucket_off, perm_b, perm_xi, perm_bkt, p, sh, tab, st);
    case 6: return reduce_dispatch<6>(gy, gwt, bucket_off, perm_b, perm_xi, perm_bkt, p, sh, tab, st);
    case 7: return reduce_dispatch<7>(gy, gwt, bucket_off, perm_b, perm_xi, perm_bkt, p, sh, tab, st);
    case 8: return reduce_dispatch<8>(gy, gwt, bucket_off, perm_b, perm_xi, perm_bkt, p, sh, tab, st);
    default: return hipErrorInvalidValue;
  }
}
