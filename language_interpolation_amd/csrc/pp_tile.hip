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
//  * which row a lane serves rotates with the step: lane (slot, q) holds the slab offset and
//    the basis values of ITS row (unit base + lane; computed once per call by
//    pp_tile_prep_kernel and read back with one coalesced load per input), and in step J of 16 it works for the
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

// One LDS-DMA instruction: 16 bytes per lane from each lane's own global address to LDS bytes
// lds_dst + lane*16 (lds_dst wave-uniform).  Inline asm on purpose: through the builtin the
// compiler waits (vmcnt(0)) for the slab of input i+1 before the first ds_read of input i -- it
// cannot see that the two halves of the buffer do not alias -- which serialises the stream it
// is meant to overlap.  The asm form is invisible to that bookkeeping; the kernel waits for
// its DMA itself (dma_wait) before the barrier that hands the buffer over.
__device__ __forceinline__ void glds16(const void* gsrc, unsigned lds_dst) {
  unsigned keep;
  asm volatile(
      "s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\t"
      "s_mov_b32 m0, %0"
      : "=&s"(keep)
      : "v"(gsrc), "s"(lds_dst)
      : "memory");
}
// Wait until at most `newer` of this wave's vector-memory operations are outstanding (they
// complete in issue order): everything but the last `newer` DMA instructions has landed.
__device__ __forceinline__ void dma_wait(int newer) {
  switch (newer) {  // wave-uniform
    case 0: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
    case 1: asm volatile("s_waitcnt vmcnt(1)" ::: "memory"); break;
    case 2: asm volatile("s_waitcnt vmcnt(2)" ::: "memory"); break;
    case 3: asm volatile("s_waitcnt vmcnt(3)" ::: "memory"); break;
    case 4: asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); break;
    case 5: asm volatile("s_waitcnt vmcnt(5)" ::: "memory"); break;
    case 6: asm volatile("s_waitcnt vmcnt(6)" ::: "memory"); break;
    case 7: asm volatile("s_waitcnt vmcnt(7)" ::: "memory"); break;
    case 8: asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); break;
    default: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
  }
}

// A1 with the division by `length` as an exact multiplication when length is a power of two
// (inv_len != 0; every reference call site has length = 2) and without the divergent branch of
// pp_segment: same bits.  t >= 9.2e18, +-inf and NaN give 0 as `.long()` + clamps do on x86.
__device__ __forceinline__ int tile_segment(float x, const PPShape& sh, float inv_len) {
  const float a = x + sh.half;
  const float t = (inv_len != 0.0f ? a * inv_len : a / sh.length) * sh.Sf;
  const float tc = fminf(fmaxf(t, -1.0f), sh.Sf);
  int s = (int)tc;  // truncation toward zero
  s = min(max(s, 0), sh.S - 1);
  return t < 9.2e18f ? s : 0;
}

// Developer builds (tools/tile_timing.hip, -DPP_TILE_DBG): parts of the kernel switched off by
// a bit mask, to see what each costs.  Release builds: the mask is the constant 0.
#ifdef PP_TILE_DBG
#define PP_TILE_DBG_PARAM , int dbg
#define PP_TILE_DBG_ARG , g_tile_dbg
#define DBG(bit) (dbg & (bit))
static int g_tile_dbg = 0, g_tile_nsplit = 0, g_tile_nw = 0, g_tile_uw = 0, g_tile_nbuf = 0;
#else
#define PP_TILE_DBG_PARAM
#define PP_TILE_DBG_ARG
#define DBG(bit) 0
#endif

struct TilePlan {
  int NW, UW, KP, nchunk, ntile, nsplit, per_split, Bpad, planes, nbuf;
  size_t lds, rec_bytes;
  bool ok;
};

// Records: what the main kernel needs per (row, input), computed once per call (the main
// kernel would repeat it for every out chunk) and laid out so that a wave reads the 64 rows of
// a unit with one coalesced load: rec[i][plane][Bpad], plane 0 = byte offset of the row's first
// active slab row (s * stride * 256), planes 1..N = the basis values (N > 1 only).  x is read
// in 64 x 64 tiles and transposed through LDS.  Rows past the batch get offset 0 / basis 0.
// DERIV (backward-input): planes 1..N hold dl_m/dx * out_scale instead of the basis values.
template <int N, bool DERIV = false>
__global__ __launch_bounds__(256) void pp_tile_prep_kernel(const float* __restrict__ x,
                                                          int32_t* __restrict__ rec,
                                                          int32_t* __restrict__ seg_out, PPShape sh,
                                                          BasisTab tab, int Bpad) {
  constexpr int P = N > 1 ? N + 1 : 1;
  __shared__ float tile[64][65];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int row0 = blockIdx.x * 64, i0 = blockIdx.y * 64;
  // (all 16 loads of a wave in flight before the first LDS store: written as one loop the
  // compiler waited for each load in turn, 8 us for a 1 MB input)
  float xr[16];
#pragma unroll
  for (int t = 0; t < 16; ++t) {
    const int row = row0 + wave + 4 * t, i = i0 + lane;
    xr[t] = (row < sh.B && i < sh.I) ? x[(size_t)row * sh.I + i] : 0.0f;
  }
#pragma unroll
  for (int t = 0; t < 16; ++t) tile[wave + 4 * t][lane] = xr[t];
  __syncthreads();
  // length a power of two (mantissa bits zero): x / length == x * (1 / length) bit for bit
  const float inv_len =
      (__float_as_uint(sh.length) & 0x7fffffu) == 0u ? 1.0f / sh.length : 0.0f;
  const int row = row0 + lane;
  for (int ii = wave; ii < 64; ii += 4) {
    const int i = i0 + ii;
    if (i >= sh.I) break;
    const float xv = tile[lane][ii];
    const int s = tile_segment(xv, sh, inv_len);
    const bool live = row < sh.B;
    int32_t* r = rec + (size_t)i * P * Bpad + row;
    r[0] = live ? s * sh.stride * (CH * 4) : 0;
    if constexpr (N > 1 && DERIV) {
      float dxi_dx;
      const float xi = pp_local_coord(xv, s, sh.length, sh.half, sh.Sf, &dxi_dx);
      float l[N], dl[N];
      pp_lagrange_d<N>(xi, tab, l, dl);
      const float chain = dxi_dx * sh.out_scale;  // (the gather kernel's order: dl * chain)
#pragma unroll
      for (int m = 0; m < N; ++m)
        r[(size_t)(1 + m) * Bpad] = live ? __float_as_int(dl[m] * chain) : 0;
    } else if constexpr (N > 1) {
      const float xi = pp_local_coord(xv, s, sh.length, sh.half, sh.Sf, nullptr);
      float l[N];
      pp_lagrange<N>(xi, tab, l);
#pragma unroll
      for (int m = 0; m < N; ++m) r[(size_t)(1 + m) * Bpad] = live ? __float_as_int(l[m]) : 0;
    }
    if (seg_out != nullptr && live) seg_out[(size_t)row * sh.I + i] = s;
  }
}

// N nodes, NW waves per workgroup, UW units of 64 rows per wave.
template <int N, int NW, int UW>
__global__ __launch_bounds__(64 * NW) void pp_fwd_tile_kernel(
    const int32_t* __restrict__ rec, const float* __restrict__ wt, float* __restrict__ out,
    PPShape sh, int KP, int Bpad, int nchunk, int per_split, long part_stride,
    int nbuf PP_TILE_DBG_PARAM) {
  constexpr int P = N > 1 ? N + 1 : 1;
  extern __shared__ __attribute__((aligned(16))) float slab[];  // [nbuf][KP][CH]; epilogue: [8][64][CH]
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int chunk = blockIdx.x % nchunk, split = blockIdx.x / nchunk;
  const int o0 = chunk * CH;
  const int row0 = (blockIdx.y * NW + wave) * (UW * 64);
  const int i_lo = split * per_split, i_hi = min(sh.I, i_lo + per_split);
  const int bufB = KP * CH * 4;      // bytes per buffer
  const int q16 = (lane & 15) * 16;  // this lane's 16-byte piece of a 256-byte slab row

  // ---- LDS-DMA of one slab: instruction q of a wave fills slab rows 4q..4q+3 (lane -> row
  // 4q + lane/16, piece lane%16), source address per lane, clamped into the tensor (rows past K
  // and columns past O land in cells nobody reads).
  const int nq = KP / 4;
  const unsigned slab_base = (unsigned)(__UINTPTR_TYPE__)(lds_void*)slab;  // LDS byte address
  const int ocol = min(o0 + (lane & 15) * 4, sh.O - 4);
  // Every wave issues the same number of instructions per slab (the last ones repeat piece
  // nq-1), so that "all but the newest `cnt`" means the same slab in every wave.
  const int cnt = (nq + NW - 1) / NW;
  auto slab_load = [&](int i, int buf) {
    const float* src = wt + (size_t)i * sh.K * sh.O + ocol;
    for (int t = 0; t < cnt; ++t) {
      const int q = min(wave + t * NW, nq - 1);
      const int row = min(q * 4 + (lane >> 4), sh.K - 1);
      glds16(src + (size_t)row * sh.O, slab_base + (unsigned)(buf * bufB + q * (4 * CH * 4)));
    }
  };

  float acc[UW][16][4];
#pragma unroll
  for (int u = 0; u < UW; ++u)
#pragma unroll
    for (int j = 0; j < 16; ++j)
#pragma unroll
      for (int e = 0; e < 4; ++e) acc[u][j][e] = 0.0f;

  // this lane's rows' records of input i: one coalesced 256-byte load per plane and unit
  const int32_t* rbase = rec + row0 + lane;
  auto load_rec = [&](int i, int (&dst)[UW][P]) {
#pragma unroll
    for (int u = 0; u < UW; ++u)
      if (row0 + u * 64 < sh.B) {  // wave-uniform
#pragma unroll
        for (int pl = 0; pl < P; ++pl) dst[u][pl] = rbase[((size_t)i * P + pl) * Bpad + u * 64];
      }
  };

  const char* slab_bytes = reinterpret_cast<const char*>(slab);
  int rn[UW][P];
  load_rec(i_lo, rn);
  // a ring of nbuf slab buffers: the slabs of the next nbuf-1 inputs are in flight or landed
  for (int a = 0; a < nbuf - 1; ++a)
    if (i_lo + a < i_hi) slab_load(i_lo + a, a);
  dma_wait((nbuf - 2) * cnt);
  __syncthreads();
  int buf = 0, bnew = nbuf - 1;  // buffer of input i, buffer that input i+nbuf-1 goes to
  for (int i = i_lo; i < i_hi; ++i) {
    int rc[UW][P];
#pragma unroll
    for (int u = 0; u < UW; ++u)
#pragma unroll
      for (int pl = 0; pl < P; ++pl) rc[u][pl] = rn[u][pl];
    // next input: records first (loads the compiler counts), then the slab (asm, uncounted)
    if (i + 1 < i_hi && !DBG(16)) load_rec(i + 1, rn);
    if (i + nbuf - 1 < i_hi && !DBG(1)) slab_load(i + nbuf - 1, bnew);
#pragma unroll
    for (int u = 0; u < UW; ++u) {
      if (row0 + u * 64 < sh.B) {  // wave-uniform (a wave past the batch still helps with the loads)
        // byte offset of this lane's ROW's first active slab row
        const int koff = rc[u][0] + buf * bufB;
        // 16 steps in batches of four: addresses, the 4*N reads in flight together, FMAs
        if (!DBG(2)) static_for<4>([&](auto G) {
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
                const float lm = rot_f<J>(__int_as_float(rc[u][1 + m]));
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
    if (!DBG(4)) {
      // this wave's pieces of slab i+1 have landed (newer ones may still be in flight; near the
      // end of the range nothing newer was issued) ...
      dma_wait(i + nbuf - 1 < i_hi ? (nbuf - 2) * cnt : 0);
      __syncthreads();  // ... everyone's have, and everyone is done with slab i
    }
    buf = buf + 1 == nbuf ? 0 : buf + 1;
    bnew = bnew + 1 == nbuf ? 0 : bnew + 1;
  }

  // ---- epilogue.  Accumulator J of this lane belongs to the row of the lane rot_i<J> names:
  // stored from here, every wave-instruction would write 64 pieces of 16 bytes into 64
  // different rows (measured: 40 us of a 118 us kernel).  The slabs are done with (the loop
  // ends with a barrier), so each wave turns its units round through 16 KiB of LDS of its own
  // and stores whole 256-byte row pieces, four rows per instruction.  The DPP moves run with
  // every lane active (a source lane switched off by EXEC reads as 0).
  // (workgroups of more than 8 waves take turns, 8 waves -- 128 KiB -- at a time)
  char* stage = reinterpret_cast<char*>(slab) + (wave & 7) * (64 * CH * 4);
  const int oc = o0 + (lane & 15) * 4;
  float* dst = out + (size_t)split * part_stride;
#pragma unroll
  for (int turn = 0; turn < (NW + 7) / 8; ++turn) {
    if (NW > 8 && turn > 0) __syncthreads();
    if (NW > 8 && (wave >> 3) != turn) continue;
#pragma unroll
    for (int u = 0; u < UW; ++u) {
      if (row0 + u * 64 < sh.B) {  // wave-uniform
        static_for<16>([&](auto Jc) {
          constexpr int J = decltype(Jc)::value;
          const int rl = rot_i<J>(lane);  // row of the unit this accumulator belongs to
          *reinterpret_cast<float4*>(stage + rl * (CH * 4) + q16) =
              make_float4(acc[u][J][0] * sh.out_scale, acc[u][J][1] * sh.out_scale,
                          acc[u][J][2] * sh.out_scale, acc[u][J][3] * sh.out_scale);
        });
#pragma unroll
        for (int g = 0; g < 16; ++g) {
          const int rl = g * 4 + (lane >> 4);
          const float4 v = *reinterpret_cast<const float4*>(stage + rl * (CH * 4) + q16);
          const int row = row0 + u * 64 + rl;
          if (row < sh.B && oc < sh.O) *reinterpret_cast<float4*>(dst + (size_t)row * sh.O + oc) = v;
        }
      }
    }
  }
}

// Backward-input on the same tiles (round 3): gx[b,i] = sum_m dl_m(b,i) * <gy[b,:], w_t[i, k0+m, :]>.
// The workgroup keeps the gy values of its rows x 64 outs in registers -- in the gathers' skewed
// layout: g[J] of lane (slot, q) is gy[row of the lane rot_i<J> names][4q..4q+3] -- streams the
// slabs exactly as the forward does, and at step J every lane has a 4-float piece of the dot
// product of ANOTHER lane's row; the piece goes to the row's owner through the opposite rotation
// (DPP row_ror:16-J on the add).  16 steps later the owner has its row's dot product over the 64
// outs, multiplies by its record's derivative values and stores part[chunk][row][i]; the out
// chunks are summed by pp_tile_reduce_kernel.  Inputs are independent: an input split only adds
// workgroups, no partial sums.  n >= 2 (n = 1 has no derivative), I % 4 == 0 (the chunk sum).
template <int N, int NW, int UW>
__global__ __launch_bounds__(64 * NW) void pp_bwdx_tile_kernel(
    const int32_t* __restrict__ rec, const float* __restrict__ wt, const float* __restrict__ gy,
    float* __restrict__ out, PPShape sh, int KP, int Bpad, int nchunk, int per_split,
    long part_stride, int nbuf) {
  constexpr int P = N + 1;
  extern __shared__ __attribute__((aligned(16))) float slab[];  // [nbuf][KP][CH]
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int chunk = blockIdx.x % nchunk, split = blockIdx.x / nchunk;
  const int o0 = chunk * CH;
  const int row0 = (blockIdx.y * NW + wave) * (UW * 64);
  const int i_lo = split * per_split, i_hi = min(sh.I, i_lo + per_split);
  const int bufB = KP * CH * 4;
  const int q16 = (lane & 15) * 16;
  const int nq = KP / 4;
  const unsigned slab_base = (unsigned)(__UINTPTR_TYPE__)(lds_void*)slab;
  const int ocol = min(o0 + (lane & 15) * 4, sh.O - 4);
  const int cnt = (nq + NW - 1) / NW;
  auto slab_load = [&](int i, int buf) {
    const float* src = wt + (size_t)i * sh.K * sh.O + ocol;
    for (int t = 0; t < cnt; ++t) {
      const int q = min(wave + t * NW, nq - 1);
      const int row = min(q * 4 + (lane >> 4), sh.K - 1);
      glds16(src + (size_t)row * sh.O, slab_base + (unsigned)(buf * bufB + q * (4 * CH * 4)));
    }
  };

  // gy of the rows this lane works for, step by step (columns past O: 0)
  float g[UW][16][4];
  const int oc = o0 + (lane & 15) * 4;
#pragma unroll
  for (int u = 0; u < UW; ++u)
    static_for<16>([&](auto Jc) {
      constexpr int J = decltype(Jc)::value;
      const int row = row0 + u * 64 + rot_i<J>(lane);
      float4 v = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
      if (row < sh.B && oc < sh.O) v = *reinterpret_cast<const float4*>(gy + (size_t)row * sh.O + oc);
      g[u][J][0] = v.x; g[u][J][1] = v.y; g[u][J][2] = v.z; g[u][J][3] = v.w;
    });

  const int32_t* rbase = rec + row0 + lane;
  auto load_rec = [&](int i, int (&dst)[UW][P]) {
#pragma unroll
    for (int u = 0; u < UW; ++u)
      if (row0 + u * 64 < sh.B) {  // wave-uniform
#pragma unroll
        for (int pl = 0; pl < P; ++pl) dst[u][pl] = rbase[((size_t)i * P + pl) * Bpad + u * 64];
      }
  };

  const char* slab_bytes = reinterpret_cast<const char*>(slab);
  float* dst = out + (size_t)chunk * part_stride;
  int rn[UW][P];
  load_rec(i_lo, rn);
  for (int a = 0; a < nbuf - 1; ++a)
    if (i_lo + a < i_hi) slab_load(i_lo + a, a);
  dma_wait((nbuf - 2) * cnt);
  __syncthreads();
  int buf = 0, bnew = nbuf - 1;
  float pend[UW][8];
  for (int i = i_lo; i < i_hi; ++i) {
    int rc[UW][P];
#pragma unroll
    for (int u = 0; u < UW; ++u)
#pragma unroll
      for (int pl = 0; pl < P; ++pl) rc[u][pl] = rn[u][pl];
    if (i + 1 < i_hi) load_rec(i + 1, rn);
    if (i + nbuf - 1 < i_hi) slab_load(i + nbuf - 1, bnew);
#pragma unroll
    for (int u = 0; u < UW; ++u) {
      if (row0 + u * 64 < sh.B) {  // wave-uniform: every lane takes part in the rotations
        const int koff = rc[u][0] + buf * bufB;
        // BS steps' reads in flight together (two for n >= 3: four spilled at 128 registers);
        // BS partial sums a node, so that the 16 rotated adds are not one dependent chain
        constexpr int BS = N >= 3 ? 2 : 4;
        float res[N][BS];
#pragma unroll
        for (int m = 0; m < N; ++m)
#pragma unroll
          for (int t = 0; t < BS; ++t) res[m][t] = 0.0f;
        static_for<16 / BS>([&](auto G) {
          float4 w[BS][N];
          static_for<BS>([&](auto T) {
            constexpr int J = decltype(G)::value * BS + decltype(T)::value;
            const int a = rot_i<J>(koff) + q16;
#pragma unroll
            for (int m = 0; m < N; ++m)
              w[decltype(T)::value][m] =
                  *reinterpret_cast<const float4*>(slab_bytes + a + m * (CH * 4));
          });
          static_for<BS>([&](auto T) {
            constexpr int TT = decltype(T)::value;
            constexpr int J = decltype(G)::value * BS + TT;
#pragma unroll
            for (int m = 0; m < N; ++m) {
              float d = w[TT][m].x * g[u][J][0];
              d = fmaf(w[TT][m].y, g[u][J][1], d);
              d = fmaf(w[TT][m].z, g[u][J][2], d);
              d = fmaf(w[TT][m].w, g[u][J][3], d);
              res[m][TT] += rot_f<(16 - J) & 15>(d);  // to the lane that owns the row
            }
          });
        });
        float v = 0.0f;
#pragma unroll
        for (int m = 0; m < N; ++m) {
          float r = res[m][0] + res[m][1];
          if constexpr (BS == 4) r += res[m][2] + res[m][3];
          v = fmaf(__int_as_float(rc[u][1 + m]), r, v);
        }
        // eight inputs' values go out together, as two 16-byte stores side by side: a whole
        // 32-byte sector per lane (groups by absolute input index; i_lo, i_hi and I are multiples
        // of 4, so a split may begin or end on half a group, which goes out as one store).  With
        // one 16-byte store per four inputs every sector went to memory half-written --
        // WRITE_SIZE 104.7 MB for 52.4 MB of partial gx (profiles/r03_k_pmc_cfg3_quirk_16384.json).
        const int slot = i & 7;
        pend[u][slot] = v;
        const int row = row0 + u * 64 + lane;
        if (row < sh.B) {
          if (slot == 7) {
            float4* q = reinterpret_cast<float4*>(dst + (size_t)row * sh.I + (i - 7));
            if (i - i_lo >= 7) q[0] = make_float4(pend[u][0], pend[u][1], pend[u][2], pend[u][3]);
            q[1] = make_float4(pend[u][4], pend[u][5], pend[u][6], pend[u][7]);
          } else if (slot == 3 && i + 1 == i_hi) {  // the split ends on half a group
            *reinterpret_cast<float4*>(dst + (size_t)row * sh.I + (i - 3)) =
                make_float4(pend[u][0], pend[u][1], pend[u][2], pend[u][3]);
          }
        }
      }
    }
    dma_wait(i + nbuf - 1 < i_hi ? (nbuf - 2) * cnt : 0);
    __syncthreads();
    buf = buf + 1 == nbuf ? 0 : buf + 1;
    bnew = bnew + 1 == nbuf ? 0 : bnew + 1;
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
  // rows per workgroup: as tall as the batch allows (slab traffic goes with 1/R)
  // (1024 rows as 16 waves x 64 rows, four waves per SIMD, measured against 8 waves x 128 rows:
  // 80 vs 92 us on cfg3 hidden, 54 vs 64 us on cfg4 input at 16384 rows)
  if (sh.B > 512) { p.NW = 16; p.UW = 1; }
  else if (sh.B > 256) { p.NW = 4; p.UW = 2; }
  else { p.NW = 4; p.UW = 1; }
#ifdef PP_TILE_DBG
  if (g_tile_nw) { p.NW = g_tile_nw; p.UW = g_tile_uw; }
#endif
  // two slab buffers (a ring of three, two slabs in flight, measured the same: 83 vs 84 us on
  // cfg3 hidden -- the stream is bound by its rate, not its latency); the epilogue turns each
  // wave's units round through 16 KiB of its own
  p.nbuf = 2;
#ifdef PP_TILE_DBG
  if (g_tile_nbuf) p.nbuf = g_tile_nbuf;
#endif
  p.lds = std::max((size_t)p.nbuf * p.KP * CH * 4, (size_t)std::min(p.NW, 8) * 64 * CH * 4);
  if (p.lds > (size_t)kTileLdsBytes) return p;
  p.nchunk = (sh.O + CH - 1) / CH;
  const int R = p.NW * p.UW * 64;
  p.ntile = (sh.B + R - 1) / R;
  p.Bpad = (sh.B + 63) / 64 * 64;
  p.planes = sh.n > 1 ? sh.n + 1 : 1;
  // split the inputs until the grid covers the chip
  const long pairs = (long)p.ntile * p.nchunk;
  int want = (int)std::max(1L, (240 + pairs - 1) / pairs);
#ifdef PP_TILE_DBG
  if (g_tile_nsplit) want = g_tile_nsplit;
#endif
  want = std::min(want, sh.I);
  p.per_split = (sh.I + want - 1) / want;
  p.nsplit = (sh.I + p.per_split - 1) / p.per_split;
  p.rec_bytes = ((size_t)sh.I * p.planes * p.Bpad * sizeof(int32_t) + 255) / 256 * 256;
  p.ok = true;
  return p;
}

template <int N, int NW, int UW>
hipError_t launch_tile(const int32_t* rec, const float* wt, float* out, const PPShape& sh,
                       const TilePlan& p, long part_stride, hipStream_t st) {
  hipError_t e = hipFuncSetAttribute((const void*)pp_fwd_tile_kernel<N, NW, UW>,
                                     hipFuncAttributeMaxDynamicSharedMemorySize, (int)p.lds);
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL((pp_fwd_tile_kernel<N, NW, UW>), dim3(p.nchunk * p.nsplit, p.ntile),
                     dim3(64 * NW), p.lds, st, rec, wt, out, sh, p.KP, p.Bpad, p.nchunk, p.per_split,
                     part_stride, p.nbuf PP_TILE_DBG_ARG);
  return hipGetLastError();
}

template <int N>
hipError_t tile_dispatch(const float* x, const float* wt, float* out, int32_t* seg, int32_t* rec,
                         const PPShape& sh, const BasisTab& tab, const TilePlan& p,
                         long part_stride, hipStream_t st) {
  hipLaunchKernelGGL((pp_tile_prep_kernel<N>), dim3((sh.B + 63) / 64, (sh.I + 63) / 64), dim3(256),
                     0, st, x, rec, seg, sh, tab, p.Bpad);
#define PP_TILE_CASE(NW_, UW_) \
  if (p.NW == NW_ && p.UW == UW_) return launch_tile<N, NW_, UW_>(rec, wt, out, sh, p, part_stride, st);
  PP_TILE_CASE(8, 2) PP_TILE_CASE(16, 1) PP_TILE_CASE(4, 2) PP_TILE_CASE(4, 1)
#undef PP_TILE_CASE
  return hipErrorInvalidValue;
}

template <int N, int NW, int UW>
hipError_t launch_bwdx_tile(const int32_t* rec, const float* wt, const float* gy, float* out,
                            const PPShape& sh, const TilePlan& p, long part_stride, hipStream_t st) {
  hipError_t e = hipFuncSetAttribute((const void*)pp_bwdx_tile_kernel<N, NW, UW>,
                                     hipFuncAttributeMaxDynamicSharedMemorySize, (int)p.lds);
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL((pp_bwdx_tile_kernel<N, NW, UW>), dim3(p.nchunk * p.nsplit, p.ntile),
                     dim3(64 * NW), p.lds, st, rec, wt, gy, out, sh, p.KP, p.Bpad, p.nchunk,
                     p.per_split, part_stride, p.nbuf);
  return hipGetLastError();
}

template <int N>
hipError_t bwdx_tile_dispatch(const float* x, const float* wt, const float* gy, float* out,
                              int32_t* rec, const PPShape& sh, const BasisTab& tab,
                              const TilePlan& p, long part_stride, hipStream_t st) {
  hipLaunchKernelGGL((pp_tile_prep_kernel<N, true>), dim3((sh.B + 63) / 64, (sh.I + 63) / 64),
                     dim3(256), 0, st, x, rec, nullptr, sh, tab, p.Bpad);
#define PP_TILE_CASE(NW_, UW_) \
  if (p.NW == NW_ && p.UW == UW_)  \
    return launch_bwdx_tile<N, NW_, UW_>(rec, wt, gy, out, sh, p, part_stride, st);
  PP_TILE_CASE(16, 1) PP_TILE_CASE(4, 2) PP_TILE_CASE(4, 1)
#undef PP_TILE_CASE
  return hipErrorInvalidValue;
}

}  // namespace

bool pp_tile_bwd_x_supported(const PPShape& sh) {
  return sh.n >= 2 && sh.I % 4 == 0 && tile_plan(sh).ok;
}

// Where PP_METHOD_AUTO takes it (given a workspace).  Against the gather kernel on the 200 -> 200,
// n = 2, K = 300 layer (HIP events, us): 1000 rows 38.6 vs 31.7, 2048: 53 vs 46, 4096: 71 vs 90,
// 8192: 105 vs 135, 16384: 168 vs 295 -- ahead once the gather kernel's table-row traffic
// B*I*n*O*4 passes ~1.2 GB; few-input layers are first layers and have no backward-input.
bool pp_tile_bwd_x_preferred(const PPShape& sh) {
  if (!pp_tile_bwd_x_supported(sh) || sh.K < 64 || sh.O < 64) return false;
  const double gather_bytes = 4.0 * sh.B * sh.I * sh.n * sh.O;
  return sh.I >= 64 && gather_bytes >= 1.2e9;
}

// records | one partial gx per out chunk (when there is more than one)
size_t pp_tile_bwd_x_workspace_bytes(const PPShape& sh) {
  const TilePlan p = tile_plan(sh);
  if (!p.ok) return 0;
  return p.rec_bytes + (p.nchunk > 1 ? (size_t)p.nchunk * sh.B * sh.I * sizeof(float) : 0);
}

hipError_t pp_launch_bwd_x_tile(const float* x, const float* gy, const float* wt, float* gx, void* ws,
                                const PPShape& sh, const BasisTab& tab, hipStream_t st) {
  TilePlan p = tile_plan(sh);
  if (!p.ok || sh.n < 2 || sh.I % 4 != 0) return hipErrorInvalidValue;
  p.per_split = (p.per_split + 3) & ~3;  // the kernel stores four or eight inputs' values at a time
  p.nsplit = (sh.I + p.per_split - 1) / p.per_split;
  int32_t* rec = static_cast<int32_t*>(ws);
  const bool direct = p.nchunk == 1;
  float* out = direct ? gx : reinterpret_cast<float*>(static_cast<char*>(ws) + p.rec_bytes);
  const long part_stride = (long)sh.B * sh.I;
  hipError_t e;
  switch (sh.n) {
    case 2: e = bwdx_tile_dispatch<2>(x, wt, gy, out, rec, sh, tab, p, part_stride, st); break;
    case 3: e = bwdx_tile_dispatch<3>(x, wt, gy, out, rec, sh, tab, p, part_stride, st); break;
    case 4: e = bwdx_tile_dispatch<4>(x, wt, gy, out, rec, sh, tab, p, part_stride, st); break;
    default: return hipErrorInvalidValue;
  }
  if (e != hipSuccess || direct) return e;
  const long total4 = part_stride / 4;
  hipLaunchKernelGGL(pp_tile_reduce_kernel, dim3((unsigned)((total4 + 255) / 256)), dim3(256), 0, st,
                     out, gx, total4, total4, p.nchunk);
  return hipGetLastError();
}

bool pp_tile_fwd_supported(const PPShape& sh) { return tile_plan(sh).ok; }

// Where the row-tile kernel is what PP_METHOD_AUTO should run (given a workspace).  Measured
// against the gather kernel (profiles/r03_a_tile_fwd_layers.txt, kernel + host launch time):
//   cfg3 hidden (n=1 200->200 K=150)  4096 rows 45 vs 46 us, 8192: 58 vs 85, 16384: 88 vs 109
//   cfg3 output (200->128)            4096: 38 vs 30 (loses), 8192: 46 vs 82, 16384: 60 vs 104
//   cfg3 with n=2 (K=300)             2048: 51 vs 40 (loses), 4096: 64 vs 88, 16384: 147 vs 248
//   cfg4 input (n=2 16->1000 K=256)   8192: 53 vs 47 (loses), 16384: 57 vs 88
//   cfg1 input (n=2 16->200 K=129)    loses at every batch (39 vs 21 at 16384)
//   cfg4 input as a table (n=1 16->1000 K=128, pp_table.hip)  16384: 36 vs 44, 8192: 43 vs 26 (loses)
// i.e. it pays once the gather kernel's L1 traffic B*I*n*O*4 passes ~0.8 GB, and for the
// few-input layers (three launches and the output tile's write-out over 16 inputs) ~2 GB --
// or ~1 GB where the (tile, chunk) pairs fill the chip without an input split, so that the
// tiles go straight to y (no partial outputs to write and add up).
// (with a normalisation behind the layer the tiles go through the partial buffer either way)
bool pp_tile_fwd_preferred(const PPShape& sh, bool with_norm) {
  const TilePlan p = tile_plan(sh);
  if (!p.ok || sh.K < 64 || sh.O < 64) return false;
  const double gather_bytes = 4.0 * sh.B * sh.I * sh.n * sh.O;
  if (sh.I >= 64) return gather_bytes >= 0.8e9;
  return gather_bytes >= (p.nsplit == 1 && !with_norm ? 1.0e9 : 2.0e9);
}

// records | partial outputs (when the inputs are split, or a normalisation follows)
size_t pp_tile_fwd_workspace_bytes(const PPShape& sh, bool with_norm) {
  const TilePlan p = tile_plan(sh);
  if (!p.ok) return 0;
  const bool direct = p.nsplit == 1 && !with_norm;
  return p.rec_bytes + (direct ? 0 : (size_t)p.nsplit * sh.B * sh.O * sizeof(float));
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
  int32_t* rec = static_cast<int32_t*>(ws);
  float* out = direct ? y : reinterpret_cast<float*>(static_cast<char*>(ws) + p.rec_bytes);
  const long part_stride = (long)sh.B * sh.O;
  hipError_t e;
  switch (sh.n) {
    case 1: e = tile_dispatch<1>(x, wt, out, seg, rec, sh, tab, p, part_stride, st); break;
    case 2: e = tile_dispatch<2>(x, wt, out, seg, rec, sh, tab, p, part_stride, st); break;
    case 3: e = tile_dispatch<3>(x, wt, out, seg, rec, sh, tab, p, part_stride, st); break;
    case 4: e = tile_dispatch<4>(x, wt, out, seg, rec, sh, tab, p, part_stride, st); break;
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
