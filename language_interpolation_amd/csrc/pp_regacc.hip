// Weight gradient of the many-segment layers with the accumulators in registers.
//
//   gw_t[i, k, o] = sum over rows b of  l_j(xi[b,i]) * gy[b, o]   for k = k0(b,i) + j, j < n
//
// A wave owns one input i and 64 consecutive outs (lane <-> out) and keeps ONE VGPR PER TABLE ROW k
// (K <= 192 rows, v64..v255, or v96..v255 for K <= 160).  It walks the batch rows in their natural order: gy[b, 64 outs] is
// one coalesced load, and the row is added into the registers k0..k0+n-1 picked by the row's
// segment through the VGPR index mode of gfx9 (s_set_gpr_idx_idx + v_fma_f32, two or three
// instructions per row).  Nothing is sorted, nothing is zero-filled and no atomics are used: every
// table row is produced by exactly one wave and stored once (rows no batch row hits store 0).
// When inputs x out chunks are too few waves to fill the GPU (the 16-input first layers), the
// batch is cut into slices whose tables go to the workspace and are summed by the partial-sum
// kernel of pp_dense.hip.
//
// The segment/basis evaluation is hoisted into a prepass that writes, per input and per 64 rows,
// the scalars the loop needs (k0 and, for n >= 2, the n basis values) transposed, so that the
// main loop fetches them with one coalesced load per 64 rows (lane <-> row) and broadcasts them
// with v_readlane: pp_regacc_prep_kernel.  The loop itself is generated assembly
// (tools/gen_regacc_asm.py -> pp_regacc_body.inc; the register map is documented there): the
// compiler has no way to express a register-indexed accumulate (an indexed array of this size is
// placed in scratch memory).
//
// Round 3 (profiles/r03_e_regacc_lds.txt):
//  * gy through LDS (pp_bwd_weight_regacc_lds_kernel): the four waves of a workgroup -- four
//    inputs, the same rows and outs -- used to load every gy row themselves, and that load path
//    was what the kernel ran at.  Now a block of 64 rows x 256 bytes is copied ONCE per workgroup
//    into one of two LDS buffers by LDS-DMA while the waves accumulate out of the other
//    (ds_read2st64_b32: two rows an instruction).  For n = 1 the records are one byte a row and
//    the index mode stays on for 64 rows; for discontinuous n = 2 (k0 even) the two accumulators
//    of a row are an aligned register pair and take ONE indexed v_pk_fma_f32.  Used where
//    rows x workgroups >= 500 k; the per-wave-load kernel keeps the small layers and batches.
//  * K > 192 (discontinuous, n >= 2): the table rows of an input go to 2-3 waves by whole
//    segments ("parts"); the prepass gives a batch row k0 = 0 and zero basis values in the
//    record sets of the parts its segment is not in, so every wave still walks all rows with the
//    same loop and adds 0 for the rows of the others.
//
// Replaces autograd's index_put_(accumulate=True) scatter into w.grad (loss.backward() at
// networks.py:164-165 of the reference), like pp_sort.hip; measured against it in
// profiles/r02_e_layer_bench.txt.
#include <algorithm>
#include <cstdlib>

#include "pp_device.h"
#ifndef PP_REGACC_INC
#define PP_REGACC_INC "pp_regacc_body.inc"
#endif
#include PP_REGACC_INC

namespace {

constexpr int kRB = 16;   // rows per block of gy loads
constexpr int kSB = 64;   // rows per set of record registers (lane <-> row)
constexpr int planes_of(int n) { return n == 1 ? 1 : 1 + n; }

struct RegaccPlan {
  int nblk;        // blocks of kRB rows covering the batch
  int nsb;         // super-blocks of kSB rows covering the batch
  int nchunk;      // out chunks of 64
  int nquad;       // groups of 4 inputs (one workgroup = 4 waves = 4 inputs, same rows and outs)
  int sbps;        // super-blocks per slice
  int nslice;      // row slices (1 = tables go straight to gw)
  int H;           // parts the table rows of an input are cut into (K > 192: one wave a part)
  int KH;          // table rows a part
  bool lds;        // gy shared through LDS by the workgroup's four waves
  bool packed;     // lds, n = 1: records are one byte a row
  size_t rec_ints_per_input;
  size_t off_rec, off_part, bytes;
};

RegaccPlan make_plan(const PPShape& sh) {
  RegaccPlan p;
  p.nblk = (sh.B + kRB - 1) / kRB;
  p.nsb = (sh.B + kSB - 1) / kSB;
  p.nchunk = (sh.O + 63) / 64;
  // More table rows than accumulator registers (discontinuous, n >= 2 only): H waves an input,
  // each with the rows of S/H segments; a batch row outside a wave's segments is given k0 = 0
  // and zero basis values by the prepass (it adds 0), so the loop does not change.
  p.H = (sh.K + PP_REGACC_KMAX - 1) / PP_REGACC_KMAX;
  p.KH = p.H == 1 ? sh.K : (sh.S + p.H - 1) / p.H * sh.stride;
  const int IV = sh.I * p.H;
  p.nquad = (IV + 3) / 4;
  const long waves_full = (long)IV * p.nchunk;
  // gy through LDS (tools/regacc_sweep.py, profiles/r03_e_regacc_lds.txt): n = 1 from 512 rows
  // (byte records, index mode on for 64 rows: 213 -> 111 us on the cfg3 hidden layer at 16384
  // rows, 24.9 -> 21.3 at 1000); n >= 2 where rows x workgroups-per-slice >= 250 k (16 -> 1000
  // with K = 128: 94 -> 82 us at 16384 rows, level at 2048; behind on the 16-input layers from
  // 1000 to 4096 rows, 25-31 vs 22-27 us)
  p.lds = sh.n == 1 ? sh.B >= 512 : (long)sh.B * p.nquad * p.nchunk >= 250000L;
#ifdef PP_TUNING  // developer builds only
  if (const char* e = getenv("PP_REGACC_LDS")) p.lds = atoi(e) != 0;
#endif
  long ns = 1;
  if (p.lds) {
    // two workgroups a CU (256 registers a wave): 512 resident at a time; one alone on a CU
    // idles at its barriers (212 us unsliced on the cfg3 hidden layer, 106 + 12 in two slices)
    // ... and at least 1024 rows a slice: below that the per-workgroup fixed work and the
    // slice sum grow faster than the loop shrinks (16 -> 250 at 16384 rows: 37.8 us in 16
    // slices, 47 in 32; tools/regacc_ns.py)
    ns = std::max<long>(1, std::min<long>(512 / ((long)p.nquad * p.nchunk), p.nsb / 16));
  } else if (waves_full < 768) {
    ns = (1024 + waves_full - 1) / waves_full;
  }
  ns = std::min<long>(ns, std::max<long>(1, p.nsb / 4));  // at least 256 rows per slice
  const size_t table = (size_t)sh.I * sh.K * sh.O * sizeof(float);
  ns = std::min<long>(ns, std::max<long>(1, (long)((96u << 20) / table)));
#ifdef PP_TUNING  // developer builds only
  if (const char* e = getenv("PP_REGACC_NSLICE")) ns = std::max(1, atoi(e));
#endif
  p.packed = p.lds && sh.n == 1;
  auto slices = [&](long want) {
    p.sbps = (int)((p.nsb + want - 1) / want);
    if (p.packed) p.sbps = (p.sbps + 3) & ~3;  // slices start on a group of 256 rows
    p.nslice = (p.nsb + p.sbps - 1) / p.sbps;
  };
  // (chunk, slice) pairs a multiple of 8 where a nearby slice count gives that: regacc_block
  slices(ns);
  if ((p.nchunk * p.nslice) % 8 != 0) {
    for (long want = ns - 1; want >= 1 && 2 * want > ns; --want) {
      slices(want);
      if ((p.nchunk * p.nslice) % 8 == 0) break;
    }
    if ((p.nchunk * p.nslice) % 8 != 0) slices(ns);
  }
  p.rec_ints_per_input = p.packed
                             ? (size_t)((p.nsb + 3) / 4 + 1) * 64  // a byte a row, one group of slack
                             : (size_t)(p.nsb + 1) * planes_of(sh.n) * kSB;  // one super-block of slack
  size_t o = 0;
  p.off_rec = o;
  o += ((size_t)sh.I * p.H * p.rec_ints_per_input * sizeof(int) + 255) & ~(size_t)255;
  p.off_part = o;
  if (p.nslice > 1) o += (size_t)p.nslice * table;
  p.bytes = o;
  return p;
}

// Prepass: rec[i][sb][0][r] = k0 of row sb*64+r (first table row of its segment) and, for
// n >= 2, rec[i][sb][1+j][r] = l_j(xi).  Rows past the batch get k0 = 0 and zero basis values.
// One workgroup = 64 rows x 32 inputs, transposed through LDS (x is read along a row, the
// records are written along the batch).
template <int N>
__global__ __launch_bounds__(256) void pp_regacc_prep_kernel(const float* __restrict__ x,
                                                             int* __restrict__ rec, PPShape sh,
                                                             BasisTab tab,
                                                             size_t rec_ints_per_input, int H,
                                                             int KH) {
  constexpr int P = planes_of(N);
  __shared__ int tile[P][32][65];
  const int t = threadIdx.x;
  const int b0 = blockIdx.x * 64, i0 = blockIdx.y * 32;
  {
    const int col = t & 31, rp = t >> 5;
    float xr[8];  // all eight loads before the first use (a chain of 8 round trips otherwise)
#pragma unroll
    for (int rr = 0; rr < 8; ++rr) {
      const int b = b0 + rr * 8 + rp, i = i0 + col;
      xr[rr] = (b < sh.B && i < sh.I) ? x[(size_t)b * sh.I + i] : 0.0f;
    }
#pragma unroll
    for (int rr = 0; rr < 8; ++rr) {
      const int rl = rr * 8 + rp;
      const int b = b0 + rl, i = i0 + col;
      int k0 = 0;
      float l[N];
#pragma unroll
      for (int j = 0; j < N; ++j) l[j] = 0.0f;
      if (b < sh.B && i < sh.I) {
        const float xv = xr[rr];
        const int s = pp_segment(xv, sh.length, sh.half, sh.Sf, sh.S);
        if constexpr (N > 1) {
          const float xi = pp_local_coord(xv, s, sh.length, sh.half, sh.Sf, nullptr);
          pp_lagrange<N>(xi, tab, l);
        }
        k0 = s * sh.stride;
      }
      tile[0][col][rl] = k0;
      if constexpr (N > 1) {
#pragma unroll
        for (int j = 0; j < N; ++j) tile[1 + j][col][rl] = __float_as_int(l[j]);
      }
    }
  }
  __syncthreads();
  {
    const int rl = t & 63, cp = t >> 6;
#pragma unroll
    for (int cc = 0; cc < 8; ++cc) {
      const int col = cc * 4 + cp;
      const int i = i0 + col;
      if (i < sh.I) {
        int* dst = rec + (size_t)i * H * rec_ints_per_input + (size_t)blockIdx.x * P * kSB + rl;
        if (H == 1) {
#pragma unroll
          for (int j = 0; j < P; ++j) dst[j * kSB] = tile[j][col][rl];
        } else {  // one record set a part: the row counts in the part its segment lies in only
          const int k0 = tile[0][col][rl];
          const int part = k0 / KH;
          for (int h = 0; h < H; ++h, dst += rec_ints_per_input) {
            dst[0] = h == part ? k0 - part * KH : 0;
#pragma unroll
            for (int j = 1; j < P; ++j) dst[j * kSB] = h == part ? tile[j][col][rl] : 0;
          }
        }
      }
    }
  }
}

// Packed records of the n = 1 LDS variant: rec8[i][row] = k0 of the row (< 256), 0 past the
// batch.  One workgroup = 256 rows x 16 inputs, transposed through LDS.
__global__ __launch_bounds__(256) void pp_regacc_prep_packed_kernel(const float* __restrict__ x,
                                                                    int* __restrict__ rec,
                                                                    PPShape sh,
                                                                    size_t rec_ints_per_input) {
  __shared__ __attribute__((aligned(16))) unsigned char tile[16][260];
  const int t = threadIdx.x;
  const int b0 = blockIdx.x * 256, i0 = blockIdx.y * 16;
  {
    const int col = t & 15, rp = t >> 4;
    const int i = i0 + col;
    float xv[16];
#pragma unroll
    for (int rr = 0; rr < 16; ++rr) {
      const int b = b0 + rr * 16 + rp;
      xv[rr] = (b < sh.B && i < sh.I) ? x[(size_t)b * sh.I + i] : 0.0f;
    }
#pragma unroll
    for (int rr = 0; rr < 16; ++rr) {
      const int b = b0 + rr * 16 + rp;
      int k0 = 0;
      if (b < sh.B && i < sh.I) k0 = pp_segment(xv[rr], sh.length, sh.half, sh.Sf, sh.S) * sh.stride;
      tile[col][rr * 16 + rp] = (unsigned char)k0;
    }
  }
  __syncthreads();
#pragma unroll
  for (int pp = 0; pp < 4; ++pp) {
    const int idx = t + 256 * pp;
    const int col = idx >> 6, dw = idx & 63;
    const int i = i0 + col;
    if (i < sh.I)
      rec[(size_t)i * rec_ints_per_input + (size_t)blockIdx.x * 64 + dw] =
          *reinterpret_cast<const int*>(&tile[col][dw * 4]);
  }
}

struct RegaccArgs {
  int B, I, O, K, nchunk, nquad, sbps, nblk, H, KH, xcd;
  size_t rec_ints_per_input;
  size_t slice_stride;  // floats between the tables of two slices
  float scale;
};

#define PP_REGACC_ASM(BODY)                                                                       \
  asm volatile(BODY                                                                               \
               :                                                                                  \
               : [r0] "s"(r0), [r1] "s"(r1), [r2] "s"(r2), [r3] "s"(r3), [dst] "s"(d),            \
                 [rec] "s"(recp), [nblk] "s"(nblk), [rowb] "s"(rowb), [K] "s"(K),                 \
                 [scale] "s"(scale), [laneb] "v"(laneb), [vok] "v"(vok), [lane4] "v"(lane4)       \
               : PP_REGACC_CLOBBERS)

#define PP_REGACC_LDS_ASM(BODY)                                                                   \
  asm volatile(BODY                                                                               \
               :                                                                                  \
               : [r0] "s"(r0), [r1] "s"(r1), [r2] "s"(r2), [r3] "s"(r3), [dst] "s"(d),            \
                 [rec] "s"(recp), [nblk] "s"(nblk), [rowb] "s"(rowb), [K] "s"(K),                 \
                 [scale] "s"(scale), [ldsw] "s"(ldsw), [laneb] "v"(laneb), [vok] "v"(vok),        \
                 [lane4] "v"(lane4), [ldsr] "v"(ldsr), [dma0] "v"(dma0)                           \
               : PP_REGACC_CLOBBERS)

// Which (group of 4 inputs, out chunk, row slice) a workgroup takes.  Workgroups are handed to the
// 8 XCDs round-robin by block id, each XCD has its own L2, and all nquad workgroups of one (chunk,
// slice) read the same piece of gy: with the plain numbering they land on all 8 XCDs and every
// one of them fetches that piece from HBM again (PMC, cfg4 input at 16384 rows: 262 k KB fetched
// per launch for a 65 MB gy).  When the number of (chunk, slice) pairs is a multiple of 8 the
// pair is picked by the XCD (block id % 8) and the quads of a pair follow each other on it.
__device__ __forceinline__ void regacc_block(const RegaccArgs& a, int& q, int& c, int& slice) {
  int bid = blockIdx.x;
  if (a.xcd) {
    const int x = bid & 7, t = bid >> 3;
    q = t % a.nquad;
    const int pair = (t / a.nquad) * 8 + x;
    c = pair % a.nchunk;
    slice = pair / a.nchunk;
  } else {
    q = bid % a.nquad;
    bid /= a.nquad;
    c = bid % a.nchunk;
    slice = bid / a.nchunk;
  }
}

// DEEP: the variant for K <= PP_REGACC_KDEEP (64 staging registers: gy loads run 3 blocks ahead).
template <int N, bool DEEP>
__global__ __launch_bounds__(256) void pp_bwd_weight_regacc_kernel(const float* __restrict__ gy,
                                                                   const int* __restrict__ rec,
                                                                   float* __restrict__ dst,
                                                                   RegaccArgs a) {
  constexpr int P = planes_of(N);
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int lane = threadIdx.x & 63;
  int q, c, slice;
  regacc_block(a, q, c, slice);
  const int iv = q * 4 + wave;  // input i, part h of its table rows
  if (iv >= a.I * a.H) return;  // whole wave
  const int i = iv / a.H, h = iv - i * a.H;
  const int sb0 = slice * a.sbps;
  const int blk0 = sb0 * (kSB / kRB);
  const int nblk = min(a.sbps * (kSB / kRB), a.nblk - blk0);  // >= 1 by the plan
  const long row0 = (long)blk0 * kRB;
  const unsigned long long gaddr = (unsigned long long)(gy + row0 * a.O);
  // buffer resource: base, stride 0, num_records = bytes from base to the end of gy, raw dword
  const int r0 = (int)(unsigned)(gaddr & 0xffffffffull);
  const int r1 = (int)(unsigned)((gaddr >> 32) & 0xffffull);
  const int r2 = (int)(unsigned)(((long)a.B - row0) * a.O * 4);
  const int r3 = 0x00020000;
  const int* recp = rec + (size_t)iv * a.rec_ints_per_input + (size_t)sb0 * P * kSB;
  float* d = dst + (size_t)slice * a.slice_stride + ((size_t)i * a.K + (size_t)h * a.KH) * a.O;
  const int o = c * 64 + lane;
  const int laneb = o * 4;
  const int lane4 = lane * 4;
  const int vok = o < a.O ? 1 : 0;
  const int rowb = a.O * 4;
  const int K = min(a.KH, a.K - h * a.KH);
  const float scale = a.scale;
  if constexpr (N == 1 && DEEP) {
    PP_REGACC_ASM(PP_REGACC_BODY_N1_K160);
  } else if constexpr (N == 1) {
    PP_REGACC_ASM(PP_REGACC_BODY_N1_K192);
  } else if constexpr (N == 2 && DEEP) {
    PP_REGACC_ASM(PP_REGACC_BODY_N2_K160);
  } else if constexpr (N == 2) {
    PP_REGACC_ASM(PP_REGACC_BODY_N2_K192);
  } else if constexpr (DEEP) {
    PP_REGACC_ASM(PP_REGACC_BODY_N3_K160);
  } else {
    PP_REGACC_ASM(PP_REGACC_BODY_N3_K192);
  }
}

// gy shared through LDS by the four waves of a workgroup (tools/gen_regacc_asm.py, body_lds):
// O % 4 == 0 (16-byte DMA pieces).  A wave without an input of its own (I % 4 != 0) repeats the
// last input and stores nothing: it still has to bring in its share of every block and meet the
// barriers.
template <int N, bool DEEP, bool PK = false>
__global__ __launch_bounds__(256) void pp_bwd_weight_regacc_lds_kernel(const float* __restrict__ gy,
                                                                       const int* __restrict__ rec,
                                                                       float* __restrict__ dst,
                                                                       RegaccArgs a) {
  constexpr int P = planes_of(N);
  __shared__ __attribute__((aligned(1024))) float stage[2 * kSB * 64];  // two blocks of 64 rows x 64 outs
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int lane = threadIdx.x & 63;
  int q, c, slice;
  regacc_block(a, q, c, slice);
  const bool own = q * 4 + wave < a.I * a.H;
  const int iv = own ? q * 4 + wave : a.I * a.H - 1;  // input i, part h of its table rows
  const int i = iv / a.H, h = iv - i * a.H;
  const int sb0 = slice * a.sbps;
  const int blk0 = sb0 * (kSB / kRB);
  // n = 1 counts blocks of 64 rows (packed records), the others blocks of 16; >= 1 by the plan
  const int nblk = N == 1 ? min(a.sbps, (a.nblk + 3) / 4 - sb0)
                          : min(a.sbps * (kSB / kRB), a.nblk - blk0);
  const long row0 = (long)blk0 * kRB;
  const unsigned long long gaddr = (unsigned long long)(gy + row0 * a.O);
  const int r0 = (int)(unsigned)(gaddr & 0xffffffffull);
  const int r1 = (int)(unsigned)((gaddr >> 32) & 0xffffull);
  const int r2 = (int)(unsigned)(((long)a.B - row0) * a.O * 4);
  const int r3 = 0x00020000;
  const int* recp = rec + (size_t)iv * a.rec_ints_per_input +
                    (N == 1 ? (size_t)(sb0 / 4) * 64 : (size_t)sb0 * P * kSB);
  float* d = dst + (size_t)slice * a.slice_stride + ((size_t)i * a.K + (size_t)h * a.KH) * a.O;
  const int o = c * 64 + lane;
  const int laneb = o * 4;
  const int lane4 = lane * 4;
  const int vok = (own && o < a.O) ? 1 : 0;
  const int rowb = a.O * 4;
  const int K = min(a.KH, a.K - h * a.KH);
  const float scale = a.scale;
  typedef __attribute__((address_space(3))) void lds_void;
  const unsigned ldsbase = (unsigned)(__UINTPTR_TYPE__)(lds_void*)stage;
  const int ldsw = (int)(ldsbase + (unsigned)wave * 1024u);
  const int ldsr = (int)(ldsbase + (unsigned)lane * 4u);
  // source of this lane's 16 bytes of the wave's first DMA piece: row 4*wave + lane/16 of the
  // block, outs c*64 + 4*(lane%16) .. +3 (past the end of a row: the next row's values, or the
  // 0 of the range check, in cells no lane with an out reads)
  const int dma0 = ((lane >> 4) + 4 * wave) * rowb + c * 256 + (lane & 15) * 16;
  if constexpr (N == 1 && DEEP) {
    PP_REGACC_LDS_ASM(PP_REGACC_LDS1_BODY_K160);
  } else if constexpr (N == 1) {
    PP_REGACC_LDS_ASM(PP_REGACC_LDS1_BODY_K192);
  } else if constexpr (N == 2 && PK && DEEP) {
    PP_REGACC_LDS_ASM(PP_REGACC_LDS_BODY_N2PK_K160);
  } else if constexpr (N == 2 && PK) {
    PP_REGACC_LDS_ASM(PP_REGACC_LDS_BODY_N2PK_K192);
  } else if constexpr (N == 2 && DEEP) {
    PP_REGACC_LDS_ASM(PP_REGACC_LDS_BODY_N2_K160);
  } else if constexpr (N == 2) {
    PP_REGACC_LDS_ASM(PP_REGACC_LDS_BODY_N2_K192);
  } else if constexpr (DEEP) {
    PP_REGACC_LDS_ASM(PP_REGACC_LDS_BODY_N3_K160);
  } else {
    PP_REGACC_LDS_ASM(PP_REGACC_LDS_BODY_N3_K192);
  }
}

template <int N>
hipError_t launch(const float* x, const float* gy, float* gwt, void* ws, const PPShape& sh,
                  const BasisTab& tab, hipStream_t st) {
  const RegaccPlan p = make_plan(sh);
  char* base = static_cast<char*>(ws);
  int* rec = reinterpret_cast<int*>(base + p.off_rec);
  float* part = reinterpret_cast<float*>(base + p.off_part);
  if (p.packed)
    hipLaunchKernelGGL(pp_regacc_prep_packed_kernel, dim3((p.nsb + 3) / 4, (sh.I + 15) / 16),
                       dim3(256), 0, st, x, rec, sh, p.rec_ints_per_input);
  else  // the slack super-block is written too (zeros)
    hipLaunchKernelGGL((pp_regacc_prep_kernel<N>), dim3(p.nsb + 1, (sh.I + 31) / 32), dim3(256), 0,
                       st, x, rec, sh, tab, p.rec_ints_per_input, p.H, p.KH);
  RegaccArgs a;
  a.B = sh.B; a.I = sh.I; a.O = sh.O; a.K = sh.K;
  a.nchunk = p.nchunk; a.nquad = p.nquad; a.sbps = p.sbps; a.nblk = p.nblk;
  a.H = p.H; a.KH = p.KH;
  a.xcd = (p.nchunk * p.nslice) % 8 == 0;
#ifdef PP_TUNING  // developer builds only
  if (const char* e = getenv("PP_REGACC_XCD")) a.xcd = a.xcd && atoi(e) != 0;
#endif
  a.rec_ints_per_input = p.rec_ints_per_input;
  a.slice_stride = (size_t)sh.I * sh.K * sh.O;
  a.scale = p.nslice > 1 ? 1.0f : sh.out_scale;
  float* dst = p.nslice > 1 ? part : gwt;
  const dim3 grid(p.nquad * p.nchunk * p.nslice);
  bool pk = N == 2 && sh.disc;  // k0 even: one packed FMA a row
#ifdef PP_TUNING  // developer builds only
  if (const char* e = getenv("PP_REGACC_PK")) pk = pk && atoi(e) != 0;
#endif
  if constexpr (N == 2) {
    if (p.lds && pk) {
      if (p.KH <= PP_REGACC_KDEEP)
        hipLaunchKernelGGL((pp_bwd_weight_regacc_lds_kernel<2, true, true>), grid, dim3(256), 0, st,
                           gy, (const int*)rec, dst, a);
      else
        hipLaunchKernelGGL((pp_bwd_weight_regacc_lds_kernel<2, false, true>), grid, dim3(256), 0,
                           st, gy, (const int*)rec, dst, a);
      if (p.nslice > 1)
        pp_launch_chunk_reduce(part, gwt, (long)sh.I * sh.K * sh.O, p.nslice, sh.out_scale, st);
      return hipGetLastError();
    }
  }
  if (p.lds) {
    if (p.KH <= PP_REGACC_KDEEP)
      hipLaunchKernelGGL((pp_bwd_weight_regacc_lds_kernel<N, true>), grid, dim3(256), 0, st, gy,
                         (const int*)rec, dst, a);
    else
      hipLaunchKernelGGL((pp_bwd_weight_regacc_lds_kernel<N, false>), grid, dim3(256), 0, st, gy,
                         (const int*)rec, dst, a);
  } else if (p.KH <= PP_REGACC_KDEEP)
    hipLaunchKernelGGL((pp_bwd_weight_regacc_kernel<N, true>), grid, dim3(256), 0, st, gy,
                       (const int*)rec, dst, a);
  else
    hipLaunchKernelGGL((pp_bwd_weight_regacc_kernel<N, false>), grid, dim3(256), 0, st, gy,
                       (const int*)rec, dst, a);
  if (p.nslice > 1)
    pp_launch_chunk_reduce(part, gwt, (long)sh.I * sh.K * sh.O, p.nslice, sh.out_scale, st);
  return hipGetLastError();
}

}  // namespace

int pp_regacc_parts(const PPShape& sh) { return (sh.K + PP_REGACC_KMAX - 1) / PP_REGACC_KMAX; }

bool pp_regacc_supported(const PPShape& sh) {
  const int H = pp_regacc_parts(sh);
  // more rows than registers: whole segments to a part, which a shared node (continuous) or an
  // all-ones basis (n = 1: the row outside its part has nothing to multiply by 0) rules out
  if (H > 1 && (!sh.disc || sh.n < 2 || H > 3 ||
                (sh.S + H - 1) / H * sh.stride > PP_REGACC_KMAX))
    return false;
  return sh.n <= 3 && (long)sh.B * sh.O * 4 < (1L << 31) && (long)sh.O * 4 * 16 < (1L << 24);
}

// Where it measured fastest (profiles/r02_e_layer_bench.txt): every supported shape, except the
// very small ones where a few million float atomics are cheaper than its three launches.  In
// parts (K > 192) every wave walks all rows for the table rows of its part: still ahead of the
// sort at every batch (cfg4 input 16 -> 1000, K = 256: 35 vs 50 us at 1000 rows, 110 vs 209 at
// 16384; 200 -> 200 with K = 300: 48 vs 146 and 332 vs 531), and of the row-owner kernel except
// on few-input layers up to 512 rows (16 -> 1000: 19.2 vs 17.8 us at 256 rows, 32 vs 28 at 512;
// 200 -> 200, K = 300: 27 vs 36 and 34 vs 67; profiles/r03_e_regacc_lds.txt).
bool pp_regacc_preferred(const PPShape& sh) {
  if (!pp_regacc_supported(sh)) return false;
  if ((long)sh.B * sh.I * sh.n * sh.O < 2500000L) return false;
  return pp_regacc_parts(sh) == 1 || sh.B > 512 || sh.I > 32;  // (else PP_BWD_W_DIRECT, pp_api.hip)
}

size_t pp_regacc_workspace_bytes(const PPShape& sh) { return make_plan(sh).bytes; }

hipError_t pp_launch_bwd_weight_regacc(const float* x, const float* gy, float* gwt, void* ws,
                                       const PPShape& sh, const BasisTab& tab, hipStream_t st) {
  switch (sh.n) {
    case 1: return launch<1>(x, gy, gwt, ws, sh, tab, st);
    case 2: return launch<2>(x, gy, gwt, ws, sh, tab, st);
    case 3: return launch<3>(x, gy, gwt, ws, sh, tab, st);
    default: return hipErrorInvalidValue;
  }
}
