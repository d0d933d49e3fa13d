// Slab path: forward of the many-segment layers with few inputs (the character-grid first
// layers: I = 16, S = 128) with the weight slab of one input staged in LDS.
//
// The gather kernel (pp_gather.hip) pulls B*I*n*O*4 bytes through the vector L1/L2 -- 2.1 GB for
// the cfg4 input layer at 16384 rows, which is what bounds it (91 us = 23 TB/s of L2 traffic
// against 83 MB of algorithmic bytes).  Here a workgroup owns a tile of R rows x one chunk of
// CH = 64*V outs and walks the inputs: the slab w_t[i, 0..K, chunk] (K*CH*4 bytes, <= 80 KiB)
// of input i+1 streams into one LDS buffer by LDS-DMA (global_load_lds, whole 1-KiB
// wave-instructions, no registers) while the waves gather their rows' n active weight rows of
// input i from the other buffer: lanes span the chunk, so every gather is one conflict-free
// ds_read of 256*V contiguous bytes, and the accumulators of a wave's RW rows stay in
// registers.  L2 traffic drops to (B/R) * I*K*O*4 (16 x the table at R = 1024), the gather
// itself runs at LDS rate.  Segment id / local coordinate / basis are evaluated lane-parallel
// (lane <-> row) once per (row, input) and handed to the wave by v_readlane, bit-identical to
// the gather kernel (same pp_device.h expressions).
//
// Replaces: Piecewise.forward of high-order-layers-torch as called for the input layers of
// select_network (networks.py:793-823 high_order_input, :862-882 high_order).
#include <algorithm>

#include "pp_device.h"

namespace {

typedef __attribute__((address_space(3))) void lds_void;
typedef const __attribute__((address_space(1))) void glb_void;

constexpr int kSlabLdsBytes = 160 * 1024;

struct SlabPlan {
  int V, NW, RW, KP, nchunk, ntile, lv;
  size_t lds;
  bool ok;
};

// N nodes, V floats per lane (chunk = 64*V outs), LV = bytes per lane of one LDS-DMA
// instruction (16 when rows of w_t are 16-byte aligned, else 4), NW waves, RW rows per wave.
template <int N, int V, int LV, int NW, int RW>
__global__ __launch_bounds__(64 * NW) void pp_fwd_slab_kernel(const float* __restrict__ x,
                                                               const float* __restrict__ wt,
                                                               float* __restrict__ y,
                                                               int32_t* __restrict__ seg_out,
                                                               PPShape sh, BasisTab tab, int KP) {
  constexpr int CH = 64 * V;
  extern __shared__ __attribute__((aligned(16))) float slab[];  // [2][KP][CH]
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int o0 = blockIdx.y * CH;
  const int row0 = (blockIdx.x * NW + wave) * RW;
  const int bufF = KP * CH;  // floats per buffer

  // ---- LDS-DMA of one slab: instruction q of a wave covers 64*LV contiguous bytes of the
  // buffer; the per-lane SOURCE address picks (row, column piece), clamped into the tensor
  // (rows past K and columns past O land in LDS cells nobody reads).
  constexpr int PF = LV / 4;                  // floats per lane per instruction
  constexpr int PPR = CH / PF;                // pieces per slab row
  const int nq = (KP * CH) / (64 * PF);       // instructions per slab (KP makes it whole)
  auto slab_load = [&](int i, int buf) {
    const float* src = wt + (size_t)i * sh.K * sh.O;
    for (int q = wave; q < nq; q += NW) {
      const int p = q * 64 + lane;
      const int row = min(p / PPR, sh.K - 1);
      const int o = min(o0 + (p % PPR) * PF, sh.O - PF);
      float* dst = slab + buf * bufF + q * (64 * PF);  // wave-uniform; hardware adds lane*LV
      glb_void* g = (glb_void*)(src + (size_t)row * sh.O + o);
      if constexpr (LV == 16) __builtin_amdgcn_global_load_lds(g, (lds_void*)dst, 16, 0, 0);
      else __builtin_amdgcn_global_load_lds(g, (lds_void*)dst, 4, 0, 0);
    }
  };

  float acc[RW][V];
#pragma unroll
  for (int r = 0; r < RW; ++r)
#pragma unroll
    for (int v = 0; v < V; ++v) acc[r][v] = 0.0f;

  // lane <-> row for the per-(row, input) scalar work
  const int myrow = min(row0 + (lane < RW ? lane : 0), sh.B - 1);
  const float* xrow = x + (size_t)myrow * sh.I;
  const bool write_seg = seg_out != nullptr && blockIdx.y == 0 && lane < RW && row0 + lane < sh.B;

  slab_load(0, 0);
  float xn = xrow[0];
  __syncthreads();  // (drains the LDS-DMA: vmcnt(0) + barrier)
  for (int i = 0; i < sh.I; ++i) {
    const int buf = i & 1;
    if (i + 1 < sh.I) slab_load(i + 1, buf ^ 1);
    const float xv = xn;
    if (i + 1 < sh.I) xn = xrow[i + 1];
    const int s = pp_segment(xv, sh.length, sh.half, sh.Sf, sh.S);
    const float xi = pp_local_coord(xv, s, sh.length, sh.half, sh.Sf, nullptr);
    float l[N];
    pp_lagrange<N>(xi, tab, l);
    if (write_seg) seg_out[(size_t)myrow * sh.I + i] = s;
    const int kbase = buf * bufF + s * sh.stride * CH;  // float index of this ROW's first slab row
    if (row0 < sh.B) {  // wave-uniform (a wave past the batch still helps with the loads)
      const float* mine = slab + lane * V;
      // batches of 8 rows: all broadcasts first (independent v_readlanes need no wait states
      // between them), then the 8*N LDS reads in flight together, then the FMAs
#pragma unroll
      for (int r0 = 0; r0 < RW; r0 += 8) {
        int kb[8];
        float lj[8][N];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          kb[u] = pp_readlane_i(kbase, r0 + u);
#pragma unroll
          for (int j = 0; j < N; ++j) lj[u][j] = pp_readlane_f(l[j], r0 + u);
        }
        PPVec<V> w[8][N];
#pragma unroll
        for (int u = 0; u < 8; ++u)
#pragma unroll
          for (int j = 0; j < N; ++j) w[u][j].load(mine + kb[u] + j * CH);
#pragma unroll
        for (int u = 0; u < 8; ++u)
#pragma unroll
          for (int j = 0; j < N; ++j)
#pragma unroll
            for (int v = 0; v < V; ++v) acc[r0 + u][v] = fmaf(lj[u][j], w[u][j].v[v], acc[r0 + u][v]);
      }
    }
    __syncthreads();  // slab i+1 has landed; everyone is done with slab i
  }
  const int o = o0 + lane * V;
  if (o < sh.O) {
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

// prefer = false: any plan that fits LDS (explicit PP_FWD_SLAB).  prefer = true: only where the
// kernel measured faster than the gather kernel (what PP_METHOD_AUTO asks).
SlabPlan slab_plan(const PPShape& sh, bool prefer) {
  SlabPlan best{};
  best.ok = false;
  if (sh.O < 4) return best;
  double best_t = 1e30;
  const int lv = (sh.O % 4 == 0) ? 16 : 4;
  for (int V : {1, 2}) {
    if (V == 2 && sh.O % 2) continue;  // lanes own V consecutive outs of a row: 8-byte alignment
    const int CH = 64 * V, RW = 64 / V;
    const int rows_per_q = std::max(1, (64 * (lv / 4)) / CH);  // slab rows per DMA instruction
    const int KP = (sh.K + rows_per_q - 1) / rows_per_q * rows_per_q;
    const size_t lds = (size_t)2 * KP * CH * 4;
    if (lds > (size_t)kSlabLdsBytes) continue;
    for (int NW : {4, 16}) {
      const int R = NW * RW;
      const long ntile = (sh.B + R - 1) / R, nchunk = (sh.O + CH - 1) / CH;
      const long blocks = ntile * nchunk;
      // rough cost: vector instructions of the gather loop (what bounds it) and L2 slab
      // traffic, whichever is larger, x rounds of workgroups
      const double valu_cyc = (double)R * sh.I * (2.0 + sh.n * (1.0 + V)) * 4.0 / 4.0;
      const double l2_us = (double)ntile * sh.I * KP * (double)nchunk * CH * 4 / 16e6;  // 16 TB/s
      const double rounds = std::max(1.0, (double)blocks / 256.0) * (blocks < 192 ? 256.0 / blocks : 1.0);
      const double t = std::max(valu_cyc / 2400.0 * rounds, l2_us);
      if (t < best_t) {
        best_t = t;
        best = SlabPlan{V, NW, RW, KP, (int)nchunk, (int)ntile, lv, lds, true};
      }
    }
  }
  if (!best.ok || !prefer) return best;
  // Measured on MI355X at 16384 rows (profiles/r02_e_layer_bench.txt): cfg4 input layer
  // (K = 256, O = 1000) 80 us vs 92 us for the gather kernel; cfg1/cfg2 input layers
  // (K = 129, O = 200/250) 36/46 us vs 25/37 us.  The gather loop is bound by vector
  // instructions per (row, input, 64 outs) -- 6 here with one float per lane against ~3.3 in
  // the gather kernel's four-floats-per-lane form -- so the slab only pays where the table is
  // too large for the gather kernel's L2 traffic to stay cheap, and only for few inputs: at 200
  // inputs (cfg3 with n = 2, K = 300) it lost, 431 vs 394 us, the per-input barrier showing.
  // Round 3: on character batches drawn from the reference's text (24 % spaces, 90 distinct
  // characters: the gather kernel's table rows repeat and stay in L2) the one shape it was
  // picked for went the other way -- cfg4 input at 4096 rows 42.8 us vs 24.2 us for the gather
  // kernel, at 16384 rows 82.1 vs 77.7 (and 56.1 for the row-tile kernel, which is what the
  // planner takes there; profiles/r03_e_regacc_lds.txt).  The planner no longer picks it; it
  // stays selectable by name.
  best.ok = false;
  return best;
}

template <int N, int V, int LV, int NW, int RW>
hipError_t launch_slab(const float* x, const float* wt, float* y, int32_t* seg, const PPShape& sh,
                       const BasisTab& tab, const SlabPlan& p, hipStream_t st) {
  hipError_t e = hipFuncSetAttribute((const void*)pp_fwd_slab_kernel<N, V, LV, NW, RW>,
                                     hipFuncAttributeMaxDynamicSharedMemorySize, (int)p.lds);
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL((pp_fwd_slab_kernel<N, V, LV, NW, RW>), dim3(p.ntile, p.nchunk), dim3(64 * NW),
                     p.lds, st, x, wt, y, seg, sh, tab, p.KP);
  return hipGetLastError();
}

template <int N>
hipError_t slab_dispatch(const float* x, const float* wt, float* y, int32_t* seg, const PPShape& sh,
                         const BasisTab& tab, const SlabPlan& p, hipStream_t st) {
#define PP_SLAB_CASE(V_, LV_, NW_)                                                         \
  if (p.V == V_ && p.lv == LV_ && p.NW == NW_)                                             \
    return launch_slab<N, V_, LV_, NW_, 64 / V_>(x, wt, y, seg, sh, tab, p, st);
  PP_SLAB_CASE(1, 16, 4) PP_SLAB_CASE(1, 16, 16) PP_SLAB_CASE(2, 16, 4) PP_SLAB_CASE(2, 16, 16)
  PP_SLAB_CASE(1, 4, 4) PP_SLAB_CASE(1, 4, 16) PP_SLAB_CASE(2, 4, 4) PP_SLAB_CASE(2, 4, 16)
#undef PP_SLAB_CASE
  return hipErrorInvalidValue;
}

}  // namespace

bool pp_slab_fwd_supported(const PPShape& sh) { return sh.n <= 4 && slab_plan(sh, false).ok; }
bool pp_slab_fwd_preferred(const PPShape& sh) { return sh.n <= 4 && slab_plan(sh, true).ok; }

hipError_t pp_launch_fwd_slab(const float* x, const float* wt, float* y, int32_t* seg,
                              const PPShape& sh, const BasisTab& tab, hipStream_t st) {
  const SlabPlan p = slab_plan(sh, false);
  if (!p.ok) return hipErrorInvalidValue;
  switch (sh.n) {
    case 1: return slab_dispatch<1>(x, wt, y, seg, sh, tab, p, st);
    case 2: return slab_dispatch<2>(x, wt, y, seg, sh, tab, p, st);
    case 3: return slab_dispatch<3>(x, wt, y, seg, sh, tab, p, st);
    case 4: return slab_dispatch<4>(x, wt, y, seg, sh, tab, p, st);
    default: return hipErrorInvalidValue;
  }
}
