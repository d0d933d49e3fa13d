// C ABI of libpphip.so (declared in include/pp_hip.h): argument validation,
// node table construction and method dispatch.  No allocation, no
// synchronisation.  The only state kept between calls belongs to the allocation-free entry
// point (pp_forward_noalloc_f32): the per-device address of its scratch array, looked up once
// per device under a mutex; every other entry point is stateless.
#include <cmath>
#include <cstdint>
#include <mutex>

#include "pp_internal.h"

// Scratch of the allocation-free entry point (pp_forward_noalloc_f32): lives in the code
// object, one copy per GPU the library runs on, held until the process ends.  Host code must
// not use this name as an address -- it is resolved per device with hipGetSymbolAddress below.
static __device__ __align__(256) unsigned char scratch[PP_NOALLOC_SCRATCH_BYTES];

namespace {

bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

int make_shape(PPShape& sh, int B, int I, int O, int n, int S, int disc, float length,
               float out_scale) {
  if (B < 1 || I < 1 || O < 1 || S < 1 || n < 1 || n > PP_MAX_N) return PP_ERR_SHAPE;
  if (!disc && n < 2) return PP_ERR_SHAPE;  // continuous needs >= 2 nodes per segment
  if (!(length > 0.0f)) return PP_ERR_SHAPE;
  if (S > (1 << 24)) return PP_ERR_SHAPE;  // float(S) must be exact
  const long K = disc ? (long)n * S : (long)(n - 1) * S + 1;
  if ((long)B * I >= (1L << 31) || (long)B * O >= (1L << 31) || (long)I * K >= (1L << 31) ||
      K >= (1L << 31))
    return PP_ERR_RANGE;
  sh.B = B; sh.I = I; sh.O = O; sh.n = n; sh.S = S; sh.K = (int)K;
  sh.stride = disc ? n : n - 1;
  sh.disc = disc ? 1 : 0;
  sh.length = length;
  sh.half = 0.5f * length;
  sh.Sf = (float)S;
  sh.out_scale = out_scale;
  return PP_OK;
}

void make_basis(BasisTab& tab, int n, float length, const float* nodes) {
  for (int j = 0; j < PP_MAX_N; ++j) { tab.X[j] = 0.0f; tab.invden[j] = 1.0f; }
  if (nodes) {
    for (int j = 0; j < n; ++j) tab.X[j] = nodes[j];
  } else if (n > 1) {
    for (int j = 0; j < n; ++j) {
      float a = (float)j * (float)M_PI / (float)(n - 1);
      tab.X[j] = (0.5f * length) * (-cosf(a));
    }
  }
  for (int j = 0; j < n; ++j) {
    double den = 1.0;
    for (int m = 0; m < n; ++m)
      if (m != j) den *= (double)tab.X[j] - (double)tab.X[m];
    tab.invden[j] = (float)(1.0 / den);
  }
}

int hip_rc(hipError_t e) { return e == hipSuccess ? PP_OK : (int)e; }

// Device address of `scratch` on the current device, resolved on the first call per device
// (each GPU holds its own copy at its own address).
constexpr int kMaxDevices = 64;
std::mutex g_scratch_mutex;
void* g_scratch_ptr[kMaxDevices] = {};

hipError_t scratch_on_current_device(void** out) {
  int dev = -1;
  hipError_t e = hipGetDevice(&dev);
  if (e != hipSuccess) return e;
  if (dev < 0 || dev >= kMaxDevices) return hipErrorInvalidDevice;
  std::lock_guard<std::mutex> lock(g_scratch_mutex);
  if (!g_scratch_ptr[dev]) {
    void* ptr = nullptr;
    e = hipGetSymbolAddress(&ptr, HIP_SYMBOL(scratch));
    if (e != hipSuccess) return e;
    if (!ptr) return hipErrorInvalidSymbol;
    g_scratch_ptr[dev] = ptr;
  }
  *out = g_scratch_ptr[dev];
  return hipSuccess;
}

// K small enough that the expanded basis Phi[B, I*K] is "genuinely dense"
// (n/K nonzero per input) and the contraction wide enough to feed MFMA tiles.
bool dense_preferred(const PPShape& sh) {
  return pp_dense_supported(sh) && sh.K <= 8 && sh.I >= 32 && sh.O >= 32;
}

// Under 256 rows (the Mamba projections at the reference's b = 16: 240 rows) the matrix-core
// kernels are a chain of k-steps in a handful of workgroups, and the gather kernels in their
// one-row-per-block form (inputs spread over the four waves, several loads in flight) are
// faster on every projection shape: forward 256 -> 256 10-14 us against 56 f32-MFMA / 20 fp16x2,
// backward-input 14-24 us against 19-27 (kernel times, rocprofv3 trace of tools/layer_bench.py
// at 240 rows; profiles/r02_p_small_rows_trace.txt).
// The forward stays on the gather kernel a little longer (cfg2 hidden, 250 -> 250 at 256 rows:
// 14 us against 20 for the fp16x2 kernel's 8 workgroups; level at ~400 rows, 25 vs 21 at 512).
bool matrix_core_fwd(const PPShape& sh) { return dense_preferred(sh) && sh.B >= 384; }

// The row-tile forward where it is the faster kernel and its scratch (per-row records, partial
// outputs) is at hand.
bool tile_fwd(const PPShape& sh, bool have_ws) { return pp_tile_fwd_preferred(sh) && have_ws; }
// ... with a normalisation fused behind the layer
bool tile_fwd_norm(const PPShape& sh) { return pp_tile_fwd_preferred(sh, true); }
bool matrix_core_bwd_x(const PPShape& sh) { return dense_preferred(sh) && sh.B >= 256; }

}  // namespace

extern "C" {

int pp_version(void) { return 302; }  // round 3: PP_FWD_TILE, PP_BWD_X_TILE, pp_forward_max_abs_ws_f32, regacc in parts, LayerNorm fold, grid tables

const char* pp_error_string(int code) {
  switch (code) {
    case PP_OK: return "ok";
    case PP_ERR_NULL: return "required pointer is NULL";
    case PP_ERR_SHAPE: return "shape out of range (need B,I,O,S >= 1, 1 <= n <= 8, continuous n >= 2, length > 0)";
    case PP_ERR_ALIGN: return "pointer is not 16-byte aligned";
    case PP_ERR_WORKSPACE: return "workspace missing or too small (see pp_workspace_bytes)";
    case PP_ERR_METHOD: return "unknown or inapplicable method for this shape";
    case PP_ERR_RANGE: return "index range exceeds int32";
    case PP_ERR_NOALLOC_SIZE:
      return "this shape and method need more scratch than the allocation-free entry point holds "
             "(PP_NOALLOC_SCRATCH_BYTES); use pp_forward_ws_f32 with a caller-owned workspace, or a "
             "method that needs none";
    default: return code > 0 ? hipGetErrorString((hipError_t)code) : "unknown error";
  }
}

static int bwd_x_auto(const PPShape& sh, bool have_ws);

int pp_auto_method(int pass, int B, int I, int O, int n, int S, int discontinuous, float length) {
  PPShape sh;
  int rc = make_shape(sh, B, I, O, n, S, discontinuous, length, 1.0f);
  if (rc != PP_OK) return rc;
  const bool dense = dense_preferred(sh);
  switch (pass) {
    case 0: return matrix_core_fwd(sh) ? (pp_bf3_supported(sh) ? PP_FWD_FP16X2 : PP_FWD_DENSE)
                   : tile_fwd(sh, true) ? PP_FWD_TILE
                                        : (pp_slab_fwd_preferred(sh) ? PP_FWD_SLAB : PP_FWD_GATHER);
    case 1: return bwd_x_auto(sh, true);
    case 2:
      // (fp16x2 from 4096 rows for layers wider than 128 outs: 38.5 vs 42 us f32-MFMA on 250 -> 250;
      // the 128-out layers stay on the f32-MFMA kernel up to 8192: 27 vs 30 us at 4096)
      if (dense)
        return pp_bf3_bwd_w_supported(sh) && B >= (O > 128 ? 4096 : 8192) ? PP_BWD_W_FP16X2
                                                                         : PP_BWD_W_DENSE;
      if (pp_regacc_preferred(sh)) return PP_BWD_W_REGACC;
      if (pp_regacc_supported(sh) && pp_regacc_parts(sh) == 1)
        return PP_BWD_W_ATOMIC;  // too small for anything but atomics
      // more weight rows per input than the register accumulators hold, at the reference's
      // batch sizes: the row-owner kernel (16 -> 1000, K = 256: 17 us against 42-45 at 256 rows,
      // level with the sort at ~700; profiles/r02_q_small_batch_bwd_w.txt)
      if (sh.B <= 512 && pp_direct_supported(sh)) return PP_BWD_W_DIRECT;
      return pp_sort_supported(sh) ? PP_BWD_W_SORT : PP_BWD_W_ATOMIC;
    default: return PP_ERR_METHOD;
  }
}

int pp_segment_ids_f32(const float* x, int32_t* seg, int B, int I, int S, float length,
                       void* stream) {
  if (!x || !seg) return PP_ERR_NULL;
  PPShape sh;
  int rc = make_shape(sh, B, I, 1, 1, S, 1, length, 1.0f);
  if (rc != PP_OK) return rc;
  return hip_rc(pp_launch_segment_ids(x, seg, sh, (hipStream_t)stream));
}

// header | forward layout | backward-input layout (fp16x2 pieces)
static size_t packed_weights_bytes(const PPShape& sh) {
  return pp_h2_fwd_packed_bytes(sh) + pp_h2_bwd_x_packed_bytes(sh);
}

size_t pp_packed_weights_bytes(int I, int O, int n, int S, int discontinuous, float length) {
  PPShape sh;
  if (make_shape(sh, 256, I, O, n, S, discontinuous, length, 1.0f) != PP_OK) return 0;
  if (!pp_bf3_supported(sh) || !pp_bf3_bwd_x_supported(sh)) return 0;
  return packed_weights_bytes(sh);
}

int pp_pack_weights_f32(const float* w_t, void* packed, size_t packed_bytes, int I, int O, int n,
                        int S, int discontinuous, float length, void* stream) {
  if (!w_t || !packed) return PP_ERR_NULL;
  if (!aligned16(w_t) || !aligned16(packed)) return PP_ERR_ALIGN;
  PPShape sh;
  int rc = make_shape(sh, 256, I, O, n, S, discontinuous, length, 1.0f);
  if (rc != PP_OK) return rc;
  if (!pp_bf3_supported(sh) || !pp_bf3_bwd_x_supported(sh)) return PP_ERR_METHOD;
  if (packed_bytes < packed_weights_bytes(sh)) return PP_ERR_WORKSPACE;
  hipStream_t st = (hipStream_t)stream;
  pp_launch_pack_both_h2(w_t, packed, sh, st);
  return hip_rc(hipGetLastError());
}

int pp_pack_weights_multi_f32(int count, const float* const* w_t, void* const* packed,
                              const size_t* packed_bytes, const int* I, const int* O, const int* n,
                              const int* S, const int* discontinuous, const float* length,
                              void* stream) {
  if (count < 1 || count > PP_PACK_MAX_LAYERS) return PP_ERR_SHAPE;
  if (!w_t || !packed || !packed_bytes || !I || !O || !n || !S || !discontinuous || !length)
    return PP_ERR_NULL;
  PPShape shapes[PP_PACK_MAX_LAYERS];
  for (int l = 0; l < count; ++l) {
    if (!w_t[l] || !packed[l]) return PP_ERR_NULL;
    if (!aligned16(w_t[l]) || !aligned16(packed[l])) return PP_ERR_ALIGN;
    int rc = make_shape(shapes[l], 256, I[l], O[l], n[l], S[l], discontinuous[l], length[l], 1.0f);
    if (rc != PP_OK) return rc;
    if (!pp_bf3_supported(shapes[l]) || !pp_bf3_bwd_x_supported(shapes[l])) return PP_ERR_METHOD;
    if (packed_bytes[l] < packed_weights_bytes(shapes[l])) return PP_ERR_WORKSPACE;
  }
  pp_launch_pack_multi_h2(count, w_t, packed, shapes, (hipStream_t)stream);
  return hip_rc(hipGetLastError());
}

static int fwd_auto(const PPShape& sh, bool have_ws) {
  if (!matrix_core_fwd(sh)) {
    if (tile_fwd(sh, have_ws)) return PP_FWD_TILE;
    return pp_slab_fwd_preferred(sh) ? PP_FWD_SLAB : PP_FWD_GATHER;
  }
  if (have_ws && pp_bf3_supported(sh)) return PP_FWD_FP16X2;
  return PP_FWD_DENSE;
}

size_t pp_forward_workspace_bytes(int B, int I, int O, int n, int S, int discontinuous,
                                  float length, int method) {
  PPShape sh;
  if (make_shape(sh, B, I, O, n, S, discontinuous, length, 1.0f) != PP_OK) return 0;
  if (method == PP_METHOD_AUTO) method = fwd_auto(sh, true);
  if ((method == PP_FWD_BF16X3 || method == PP_FWD_FP16X2) && pp_bf3_supported(sh))
    return pp_bf3_fwd_workspace_bytes(sh);
  if (method == PP_FWD_TILE && pp_tile_fwd_supported(sh)) return pp_tile_fwd_workspace_bytes(sh, false);
  return 0;
}

int pp_forward_ws_f32(const float* x, const float* w_t, float* y, int32_t* seg_out,
                      void* workspace, size_t workspace_bytes, int B, int I, int O, int n, int S,
                      int discontinuous, float length, const float* nodes, float out_scale,
                      int method, void* stream) {
  if (!x || !w_t || !y) return PP_ERR_NULL;
  if (!aligned16(x) || !aligned16(w_t) || !aligned16(y) || !aligned16(workspace))
    return PP_ERR_ALIGN;
  PPShape sh;
  int rc = make_shape(sh, B, I, O, n, S, discontinuous, length, out_scale);
  if (rc != PP_OK) return rc;
  BasisTab tab;
  make_basis(tab, n, length, nodes);
  if (method == (PP_FWD_FP16X2 | PP_WS_PREPACKED)) {
    if (!pp_bf3_supported(sh)) return PP_ERR_METHOD;
    if (!workspace || workspace_bytes < packed_weights_bytes(sh)) return PP_ERR_WORKSPACE;
    return hip_rc(pp_launch_fwd_h2(x, w_t, y, seg_out, workspace, sh, tab, true, nullptr, 0.0f, nullptr,
                                   (hipStream_t)stream));
  }
  if (method == PP_METHOD_AUTO) {
    method = fwd_auto(sh, workspace != nullptr);
    if (method == PP_FWD_FP16X2 && workspace_bytes < pp_bf3_fwd_workspace_bytes(sh))
      method = PP_FWD_DENSE;
    if (method == PP_FWD_TILE && workspace_bytes < pp_tile_fwd_workspace_bytes(sh, false))
      method = fwd_auto(sh, false);
  }
  hipStream_t st = (hipStream_t)stream;
  if (method == PP_FWD_GATHER) return hip_rc(pp_launch_fwd_gather(x, w_t, y, seg_out, sh, tab, st));
  if (method == PP_FWD_TILE) {
    if (!pp_tile_fwd_supported(sh)) return PP_ERR_METHOD;
    const size_t need = pp_tile_fwd_workspace_bytes(sh, false);
    if (need && (!workspace || workspace_bytes < need)) return PP_ERR_WORKSPACE;
    return hip_rc(pp_launch_fwd_tile(x, w_t, y, seg_out, workspace, nullptr, 0.0f, sh, tab, st));
  }
  if (method == PP_FWD_SLAB) {
    if (!pp_slab_fwd_supported(sh)) return PP_ERR_METHOD;
    return hip_rc(pp_launch_fwd_slab(x, w_t, y, seg_out, sh, tab, st));
  }
  if (method == PP_FWD_DENSE) {
    if (!pp_dense_supported(sh)) return PP_ERR_METHOD;
    return hip_rc(pp_launch_fwd_dense(x, w_t, y, seg_out, sh, tab, st));
  }
  if (method == PP_FWD_BF16X3 || method == PP_FWD_FP16X2) {
    if (!pp_bf3_supported(sh)) return PP_ERR_METHOD;
    if (!workspace || workspace_bytes < pp_bf3_fwd_workspace_bytes(sh)) return PP_ERR_WORKSPACE;
    if (method == PP_FWD_FP16X2)
      return hip_rc(pp_launch_fwd_h2(x, w_t, y, seg_out, workspace, sh, tab, false, nullptr, 0.0f, nullptr,
                                     st));
    return hip_rc(pp_launch_fwd_bf3(x, w_t, y, seg_out, workspace, sh, tab, st));
  }
  return PP_ERR_METHOD;
}

int pp_forward_f32(const float* x, const float* w_t, float* y, int32_t* seg_out, int B, int I,
                   int O, int n, int S, int discontinuous, float length, const float* nodes,
                   float out_scale, int method, void* stream) {
  return pp_forward_ws_f32(x, w_t, y, seg_out, nullptr, 0, B, I, O, n, S, discontinuous, length,
                           nodes, out_scale, method, stream);
}

int pp_forward_noalloc_f32(const float* x, const float* w_t, float* y, int32_t* seg_out, int B, int I,
                           int O, int n, int S, int discontinuous, float length, const float* nodes,
                           float out_scale, int method, void* stream) {
  if (method & PP_WS_PREPACKED) return PP_ERR_METHOD;  // that form reads a caller-owned buffer
  PPShape sh;
  int rc = make_shape(sh, B, I, O, n, S, discontinuous, length, out_scale);
  if (rc != PP_OK) return rc;
  size_t need = pp_forward_workspace_bytes(B, I, O, n, S, discontinuous, length, method);
  if (need > (size_t)PP_NOALLOC_SCRATCH_BYTES) {
    // an explicitly requested method is refused; the automatic choice falls back to what
    // pp_forward_f32 would run for the shape (a method without scratch), which always fits
    if (method != PP_METHOD_AUTO) return PP_ERR_NOALLOC_SIZE;
    method = fwd_auto(sh, false);
    need = 0;
  }
  void* ptr = nullptr;
  if (need) {
    hipError_t e = scratch_on_current_device(&ptr);
    if (e != hipSuccess) return hip_rc(e);
  }
  // Nothing in the forward scratch has to start at zero: the weight pack writes every byte it
  // later reads (64 absmax partials + all pieces).  Calls that overlap on one GPU share it.
  return pp_forward_ws_f32(x, w_t, y, seg_out, ptr, need ? (size_t)PP_NOALLOC_SCRATCH_BYTES : 0, B, I,
                           O, n, S, discontinuous, length, nodes, out_scale, method, stream);
}

static int bwd_x_auto(const PPShape& sh, bool have_ws) {
  if (!matrix_core_bwd_x(sh))
    return have_ws && pp_tile_bwd_x_preferred(sh) ? PP_BWD_X_TILE : PP_BWD_X_GATHER;
  if (have_ws && pp_bf3_bwd_x_supported(sh) && sh.B >= 256) return PP_BWD_X_FP16X2;
  return PP_BWD_X_DENSE;
}

size_t pp_backward_input_workspace_bytes(int B, int I, int O, int n, int S, int discontinuous,
                                         float length, int method) {
  PPShape sh;
  if (make_shape(sh, B, I, O, n, S, discontinuous, length, 1.0f) != PP_OK) return 0;
  if (method == PP_METHOD_AUTO) method = bwd_x_auto(sh, true);
  if ((method == PP_BWD_X_BF16X3 || method == PP_BWD_X_FP16X2) && pp_bf3_bwd_x_supported(sh))
    return pp_bf3_bwd_x_workspace_bytes(sh);
  if (method == PP_BWD_X_TILE && pp_tile_bwd_x_supported(sh)) return pp_tile_bwd_x_workspace_bytes(sh);
  return 0;
}

static int backward_input_impl(const float* x, const float* gy, const float* w_t, float* gx,
                               float* row_stats, void* workspace, size_t workspace_bytes, int B,
                               int I, int O, int n, int S, int discontinuous, float length,
                               const float* nodes, float out_scale, int method, void* stream);

int pp_backward_input_ws_f32(const float* x, const float* gy, const float* w_t, float* gx,
                             void* workspace, size_t workspace_bytes, int B, int I, int O, int n,
                             int S, int discontinuous, float length, const float* nodes,
                             float out_scale, int method, void* stream) {
  return backward_input_impl(x, gy, w_t, gx, nullptr, workspace, workspace_bytes, B, I, O, n, S,
                             discontinuous, length, nodes, out_scale, method, stream);
}

int pp_row_stats_tiles(int I) { return I > 0 ? (I + 31) / 32 : 0; }

int pp_backward_input_stats_supported(int B, int I, int O, int n, int S, int discontinuous,
                                      float length) {
  PPShape sh;
  if (make_shape(sh, B, I, O, n, S, discontinuous, length, 1.0f) != PP_OK) return 0;
  return bwd_x_auto(sh, true) == PP_BWD_X_FP16X2 ? 1 : 0;
}

int pp_backward_input_stats_ws_f32(const float* x, const float* gy, const float* w_t, float* gx,
                                   float* row_stats, void* workspace, size_t workspace_bytes, int B,
                                   int I, int O, int n, int S, int discontinuous, float length,
                                   const float* nodes, float out_scale, int method, void* stream) {
  if (!row_stats) return PP_ERR_NULL;
  if (!aligned16(row_stats)) return PP_ERR_ALIGN;
  return backward_input_impl(x, gy, w_t, gx, row_stats, workspace, workspace_bytes, B, I, O, n, S,
                             discontinuous, length, nodes, out_scale, method, stream);
}

// row_stats != nullptr: only the kernels with the row-sum epilogue (fp16x2) may run
static int backward_input_impl(const float* x, const float* gy, const float* w_t, float* gx,
                               float* row_stats, void* workspace, size_t workspace_bytes, int B,
                               int I, int O, int n, int S, int discontinuous, float length,
                               const float* nodes, float out_scale, int method, void* stream) {
  if (!x || !gy || !w_t || !gx) return PP_ERR_NULL;
  if (!aligned16(x) || !aligned16(gy) || !aligned16(w_t) || !aligned16(gx) ||
      !aligned16(workspace))
    return PP_ERR_ALIGN;
  PPShape sh;
  int rc = make_shape(sh, B, I, O, n, S, discontinuous, length, out_scale);
  if (rc != PP_OK) return rc;
  BasisTab tab;
  make_basis(tab, n, length, nodes);
  if (method == (PP_BWD_X_FP16X2 | PP_WS_PREPACKED)) {
    if (!pp_bf3_bwd_x_supported(sh) || !pp_bf3_supported(sh)) return PP_ERR_METHOD;
    if (!workspace || workspace_bytes < packed_weights_bytes(sh)) return PP_ERR_WORKSPACE;
    return hip_rc(pp_launch_bwd_x_h2(x, gy, w_t, gx, workspace, sh, tab, true,
                                     pp_h2_fwd_packed_bytes(sh), row_stats, (hipStream_t)stream));
  }
  if (method == PP_METHOD_AUTO) {
    method = bwd_x_auto(sh, workspace != nullptr);
    if (method == PP_BWD_X_FP16X2 && workspace_bytes < pp_bf3_bwd_x_workspace_bytes(sh))
      method = PP_BWD_X_DENSE;
    if (method == PP_BWD_X_TILE && workspace_bytes < pp_tile_bwd_x_workspace_bytes(sh))
      method = PP_BWD_X_GATHER;
  }
  if (row_stats && method != PP_BWD_X_FP16X2) return PP_ERR_METHOD;
  hipStream_t st = (hipStream_t)stream;
  if (method == PP_BWD_X_GATHER)
    return hip_rc(pp_launch_bwd_input_gather(x, gy, w_t, gx, sh, tab, st));
  if (method == PP_BWD_X_DENSE) {
    if (!pp_dense_supported(sh)) return PP_ERR_METHOD;
    return hip_rc(pp_launch_bwd_input_dense(x, gy, w_t, gx, sh, tab, st));
  }
  if (method == PP_BWD_X_TILE) {
    if (!pp_tile_bwd_x_supported(sh)) return PP_ERR_METHOD;
    if (!workspace || workspace_bytes < pp_tile_bwd_x_workspace_bytes(sh)) return PP_ERR_WORKSPACE;
    return hip_rc(pp_launch_bwd_x_tile(x, gy, w_t, gx, workspace, sh, tab, st));
  }
  if (method == PP_BWD_X_BF16X3 || method == PP_BWD_X_FP16X2) {
    if (!pp_bf3_bwd_x_supported(sh)) return PP_ERR_METHOD;
    if (!workspace || workspace_bytes < pp_bf3_bwd_x_workspace_bytes(sh)) return PP_ERR_WORKSPACE;
    if (method == PP_BWD_X_FP16X2)
      return hip_rc(pp_launch_bwd_x_h2(x, gy, w_t, gx, workspace, sh, tab, false, 256, row_stats, st));
    return hip_rc(pp_launch_bwd_x_bf3(x, gy, w_t, gx, workspace, sh, tab, st));
  }
  return PP_ERR_METHOD;
}

int pp_backward_input_f32(const float* x, const float* gy, const float* w_t, float* gx, int B,
                          int I, int O, int n, int S, int discontinuous, float length,
                          const float* nodes, float out_scale, int method, void* stream) {
  return pp_backward_input_ws_f32(x, gy, w_t, gx, nullptr, 0, B, I, O, n, S, discontinuous, length,
                                  nodes, out_scale, method, stream);
}

size_t pp_workspace_bytes(int B, int I, int O, int n, int S, int discontinuous, float length,
                          int method) {
  PPShape sh;
  if (make_shape(sh, B, I, O, n, S, discontinuous, length, 1.0f) != PP_OK) return 0;
  if (method == PP_METHOD_AUTO) method = pp_auto_method(2, B, I, O, n, S, discontinuous, length);
  if (method == PP_BWD_W_SORT) return pp_sort_supported(sh) ? pp_sort_workspace_bytes(sh) : 0;
  if (method == PP_BWD_W_REGACC)
    return pp_regacc_supported(sh) ? pp_regacc_workspace_bytes(sh) : 0;
  if (method == PP_BWD_W_DENSE)
    return pp_dense_supported(sh) ? pp_dense_bwd_weight_workspace_bytes(sh) : 0;
  if (method == PP_BWD_W_BF16X3 || method == PP_BWD_W_FP16X2)
    return pp_bf3_bwd_w_supported(sh) ? pp_bf3_bwd_w_workspace_bytes(sh) : 0;
  return 0;
}

int pp_backward_weight_f32(const float* x, const float* gy, float* gw_t, void* workspace,
                           size_t workspace_bytes, int B, int I, int O, int n, int S,
                           int discontinuous, float length, const float* nodes, float out_scale,
                           int method, void* stream) {
  if (!x || !gy || !gw_t) return PP_ERR_NULL;
  if (!aligned16(x) || !aligned16(gy) || !aligned16(gw_t) || !aligned16(workspace))
    return PP_ERR_ALIGN;
  PPShape sh;
  int rc = make_shape(sh, B, I, O, n, S, discontinuous, length, out_scale);
  if (rc != PP_OK) return rc;
  BasisTab tab;
  make_basis(tab, n, length, nodes);
  if (method == PP_METHOD_AUTO) method = pp_auto_method(2, B, I, O, n, S, discontinuous, length);
  hipStream_t st = (hipStream_t)stream;
  switch (method) {
    case PP_BWD_W_ATOMIC:
      return hip_rc(pp_launch_bwd_weight_atomic(x, gy, gw_t, sh, tab, st));
    case PP_BWD_W_SORT: {
      if (!pp_sort_supported(sh)) return PP_ERR_METHOD;
      const size_t need = pp_sort_workspace_bytes(sh);
      if (!workspace || workspace_bytes < need) return PP_ERR_WORKSPACE;
      return hip_rc(pp_launch_bwd_weight_sort(x, gy, gw_t, workspace, sh, tab, st));
    }
    case PP_BWD_W_DIRECT:
      if (!pp_direct_supported(sh)) return PP_ERR_METHOD;
      return hip_rc(pp_launch_bwd_weight_direct(x, gy, gw_t, sh, tab, st));
    case PP_BWD_W_REGACC: {
      if (!pp_regacc_supported(sh)) return PP_ERR_METHOD;
      const size_t need = pp_regacc_workspace_bytes(sh);
      if (!workspace || workspace_bytes < need) return PP_ERR_WORKSPACE;
      return hip_rc(pp_launch_bwd_weight_regacc(x, gy, gw_t, workspace, sh, tab, st));
    }
    case PP_BWD_W_DENSE: {
      if (!pp_dense_supported(sh)) return PP_ERR_METHOD;
      const size_t need = pp_dense_bwd_weight_workspace_bytes(sh);
      if (need && (!workspace || workspace_bytes < need)) return PP_ERR_WORKSPACE;
      return hip_rc(pp_launch_bwd_weight_dense(x, gy, gw_t, workspace, sh, tab, st));
    }
    case PP_BWD_W_BF16X3:
    case PP_BWD_W_FP16X2: {
      if (!pp_bf3_bwd_w_supported(sh)) return PP_ERR_METHOD;
      if (!workspace || workspace_bytes < pp_bf3_bwd_w_workspace_bytes(sh))
        return PP_ERR_WORKSPACE;
      if (method == PP_BWD_W_FP16X2)
        return hip_rc(pp_launch_bwd_w_h2(x, gy, gw_t, workspace, sh, tab, st));
      return hip_rc(pp_launch_bwd_w_bf3(x, gy, gw_t, workspace, sh, tab, st));
    }
    default:
      return PP_ERR_METHOD;
  }
}

int pp_backward_weight_layer_norm_supported(int B, int I, int O, int n, int S, int discontinuous,
                                            float length) {
  return pp_auto_method(2, B, I, O, n, S, discontinuous, length) == PP_BWD_W_FP16X2 ? 1 : 0;
}

int pp_backward_weight_layer_norm_f32(const float* x, const float* gy, const float* y_norm,
                                      const float* rstd, const float* row_stats, int stats_tiles,
                                      float* g_out, float* gw_t, void* workspace,
                                      size_t workspace_bytes, int B, int I, int O, int n, int S,
                                      int discontinuous, float length, const float* nodes,
                                      float out_scale, int method, void* stream) {
  if (!x || !gy || !y_norm || !rstd || !row_stats || !g_out || !gw_t) return PP_ERR_NULL;
  if (!aligned16(x) || !aligned16(gy) || !aligned16(y_norm) || !aligned16(row_stats) ||
      !aligned16(g_out) || !aligned16(gw_t) || !aligned16(workspace))
    return PP_ERR_ALIGN;
  PPShape sh;
  int rc = make_shape(sh, B, I, O, n, S, discontinuous, length, out_scale);
  if (rc != PP_OK) return rc;
  if (stats_tiles != pp_row_stats_tiles(O)) return PP_ERR_SHAPE;
  if (method == PP_METHOD_AUTO) method = pp_auto_method(2, B, I, O, n, S, discontinuous, length);
  if (method != PP_BWD_W_FP16X2 || !pp_bf3_bwd_w_supported(sh)) return PP_ERR_METHOD;
  if (!workspace || workspace_bytes < pp_bf3_bwd_w_workspace_bytes(sh)) return PP_ERR_WORKSPACE;
  BasisTab tab;
  make_basis(tab, n, length, nodes);
  return hip_rc(pp_launch_bwd_w_h2_ln(x, gy, y_norm, rstd, row_stats, stats_tiles, g_out, gw_t,
                                      workspace, sh, tab, (hipStream_t)stream));
}

int pp_grid_table_supported(int I, int O, int n, int S, int discontinuous, float length,
                            int points) {
  PPShape sh;
  if (make_shape(sh, 1, I, O, n, S, discontinuous, length, 1.0f) != PP_OK) return 0;
  PPShape tsh;  // the table layer must itself be a valid layer
  if (make_shape(tsh, 1, I, O, 1, points, 1, length, 1.0f) != PP_OK) return 0;
  return pp_grid_table_supported(sh, points) ? 1 : 0;
}

int pp_grid_table_f32(const float* w_t, float* table, int I, int O, int n, int S, int discontinuous,
                      float length, const float* nodes, float out_scale, int points,
                      void* stream) {
  if (!w_t || !table) return PP_ERR_NULL;
  if (!aligned16(w_t) || !aligned16(table)) return PP_ERR_ALIGN;
  PPShape sh;
  int rc = make_shape(sh, 1, I, O, n, S, discontinuous, length, out_scale);
  if (rc != PP_OK) return rc;
  if (!pp_grid_table_supported(sh, points)) return PP_ERR_METHOD;
  if ((long)I * points >= (1L << 31)) return PP_ERR_RANGE;
  BasisTab tab;
  make_basis(tab, n, length, nodes);
  return hip_rc(pp_launch_grid_table(w_t, table, sh, tab, points, (hipStream_t)stream));
}

int pp_grid_table_grad_f32(const float* g_table, float* gw_t, int I, int O, int n, int S,
                           int discontinuous, float length, const float* nodes, float out_scale,
                           int points, void* stream) {
  if (!g_table || !gw_t) return PP_ERR_NULL;
  if (!aligned16(g_table) || !aligned16(gw_t)) return PP_ERR_ALIGN;
  PPShape sh;
  int rc = make_shape(sh, 1, I, O, n, S, discontinuous, length, out_scale);
  if (rc != PP_OK) return rc;
  if (!pp_grid_table_supported(sh, points)) return PP_ERR_METHOD;
  BasisTab tab;
  make_basis(tab, n, length, nodes);
  return hip_rc(pp_launch_grid_table_grad(g_table, gw_t, sh, tab, points, (hipStream_t)stream));
}

int pp_transpose_w_f32(const float* w, float* w_t, int O, int I, int K, void* stream) {
  if (!w || !w_t) return PP_ERR_NULL;
  if (O < 1 || I < 1 || K < 1 || (long)I * K >= (1L << 31)) return PP_ERR_SHAPE;
  return hip_rc(pp_launch_transpose(w, w_t, O, I * K, (hipStream_t)stream));
}

int pp_transpose_wt_f32(const float* w_t, float* w, int O, int I, int K, void* stream) {
  if (!w || !w_t) return PP_ERR_NULL;
  if (O < 1 || I < 1 || K < 1 || (long)I * K >= (1L << 31)) return PP_ERR_SHAPE;
  return hip_rc(pp_launch_transpose(w_t, w, I * K, O, (hipStream_t)stream));
}

static int norm_args_ok(int B, int D) {
  if (B < 1 || D < 1 || D > 1024) return PP_ERR_SHAPE;
  if ((long)B * D >= (1L << 31)) return PP_ERR_RANGE;
  return PP_OK;
}

int pp_forward_layer_norm_f32(const float* x, const float* w_t, float* y, float* rstd,
                              void* packed, size_t packed_bytes, int B, int I, int O, int n, int S,
                              int discontinuous, float length, const float* nodes, float out_scale,
                              float eps, void* stream) {
  if (!x || !w_t || !y || !rstd) return PP_ERR_NULL;
  if (!aligned16(x) || !aligned16(w_t) || !aligned16(y) || !aligned16(packed)) return PP_ERR_ALIGN;
  PPShape sh;
  int rc = make_shape(sh, B, I, O, n, S, discontinuous, length, out_scale);
  if (rc != PP_OK) return rc;
  if (packed && !matrix_core_fwd(sh) && tile_fwd_norm(sh) && sh.O <= 1024) {
    // mode 3: `packed` is the row-tile kernel's scratch; the kernel that sums the partial
    // outputs normalises the rows
    if (packed_bytes < pp_tile_fwd_workspace_bytes(sh, true)) return PP_ERR_WORKSPACE;
    BasisTab ttab;
    make_basis(ttab, n, length, nodes);
    return hip_rc(pp_launch_fwd_tile(x, w_t, y, nullptr, packed, rstd, eps, sh, ttab,
                                     (hipStream_t)stream));
  }
  if (!packed) {  // gather path: the wave that sums a row normalises it
    if (!pp_gather_ln_supported(sh)) return PP_ERR_METHOD;
    BasisTab gtab;
    make_basis(gtab, n, length, nodes);
    return hip_rc(pp_launch_fwd_gather_ln(x, w_t, y, rstd, sh, gtab, eps, (hipStream_t)stream));
  }
  if (!pp_h2_fwd_ln_supported(sh) || !pp_bf3_bwd_x_supported(sh)) return PP_ERR_METHOD;
  if (packed_bytes < packed_weights_bytes(sh)) return PP_ERR_WORKSPACE;
  BasisTab tab;
  make_basis(tab, n, length, nodes);
  return hip_rc(pp_launch_fwd_h2(x, w_t, y, nullptr, packed, sh, tab, true, rstd, eps, nullptr,
                                 (hipStream_t)stream));
}

int pp_forward_layer_norm_supported(int B, int I, int O, int n, int S, int discontinuous,
                                    float length) {
  PPShape sh;
  if (make_shape(sh, B, I, O, n, S, discontinuous, length, 1.0f) != PP_OK) return 0;
  if (pp_h2_fwd_ln_supported(sh) && pp_bf3_bwd_x_supported(sh) && matrix_core_fwd(sh))
    return 1;
  if (!matrix_core_fwd(sh) && tile_fwd_norm(sh) && sh.O <= 1024) return 3;
  // layers the gather kernels serve (many segments, first layers, lengths off the matrix-core
  // path): fused there too
  if (!matrix_core_fwd(sh) && pp_gather_ln_supported(sh)) return 2;
  return 0;
}

size_t pp_forward_layer_norm_workspace_bytes(int B, int I, int O, int n, int S, int discontinuous,
                                             float length) {
  if (pp_forward_layer_norm_supported(B, I, O, n, S, discontinuous, length) != 3) return 0;
  PPShape sh;
  make_shape(sh, B, I, O, n, S, discontinuous, length, 1.0f);
  return pp_tile_fwd_workspace_bytes(sh, true);
}

int pp_forward_max_abs_f32(const float* x, const float* w_t, float* y, float* den, int32_t* amax,
                           int B, int I, int O, int n, int S, int discontinuous, float length,
                           const float* nodes, float out_scale, float eps, void* stream) {
  if (!x || !w_t || !y || !den || !amax) return PP_ERR_NULL;
  if (!aligned16(x) || !aligned16(w_t) || !aligned16(y)) return PP_ERR_ALIGN;
  PPShape sh;
  int rc = make_shape(sh, B, I, O, n, S, discontinuous, length, out_scale);
  if (rc != PP_OK) return rc;
  if (!pp_gather_ln_supported(sh)) return PP_ERR_METHOD;
  BasisTab tab;
  make_basis(tab, n, length, nodes);
  return hip_rc(pp_launch_fwd_gather_maxabs(x, w_t, y, den, amax, sh, tab, eps,
                                            (hipStream_t)stream));
}

int pp_forward_max_abs_ws_f32(const float* x, const float* w_t, float* y, float* den, int32_t* amax,
                              void* packed, size_t packed_bytes, int B, int I, int O, int n, int S,
                              int discontinuous, float length, const float* nodes, float out_scale,
                              float eps, void* stream) {
  if (!packed)
    return pp_forward_max_abs_f32(x, w_t, y, den, amax, B, I, O, n, S, discontinuous, length, nodes,
                                  out_scale, eps, stream);
  if (!x || !w_t || !y || !den || !amax) return PP_ERR_NULL;
  if (!aligned16(x) || !aligned16(w_t) || !aligned16(y) || !aligned16(packed)) return PP_ERR_ALIGN;
  PPShape sh;
  int rc = make_shape(sh, B, I, O, n, S, discontinuous, length, out_scale);
  if (rc != PP_OK) return rc;
  if (!pp_h2_fwd_ln_supported(sh) || !pp_bf3_bwd_x_supported(sh)) return PP_ERR_METHOD;
  if (packed_bytes < packed_weights_bytes(sh)) return PP_ERR_WORKSPACE;
  BasisTab tab;
  make_basis(tab, n, length, nodes);
  return hip_rc(pp_launch_fwd_h2(x, w_t, y, nullptr, packed, sh, tab, true, den, eps, amax,
                                 (hipStream_t)stream));
}

int pp_forward_max_abs_supported(int B, int I, int O, int n, int S, int discontinuous,
                                 float length) {
  PPShape sh;
  if (make_shape(sh, B, I, O, n, S, discontinuous, length, 1.0f) != PP_OK) return 0;
  // 1: the fp16x2 matrix-core kernel (whole rows per block, O <= 256), 2: the gather kernel
  if (pp_h2_fwd_ln_supported(sh) && pp_bf3_bwd_x_supported(sh) && matrix_core_fwd(sh)) return 1;
  return (!matrix_core_fwd(sh) && pp_gather_ln_supported(sh)) ? 2 : 0;
}

static int scan_args_ok(int b, int l, int d, int n) {
  if (b < 1 || l < 1 || d < 1 || n < 1 || b > 65535) return PP_ERR_SHAPE;
  if ((long)b * l * d >= (1L << 31) || (long)b * l * n >= (1L << 31)) return PP_ERR_RANGE;
  return PP_OK;
}

int pp_selective_scan_fwd_f32(const float* u, const float* delta, const float* A, const float* Bm,
                              const float* Cm, const float* D, float* y, float* xhist, int b, int l,
                              int d, int n, void* stream) {
  if (!u || !delta || !A || !Bm || !Cm || !D || !y) return PP_ERR_NULL;
  int rc = scan_args_ok(b, l, d, n);
  if (rc != PP_OK) return rc;
  return hip_rc(pp_launch_scan_fwd(u, delta, A, Bm, Cm, D, y, xhist, b, l, d, n,
                                   (hipStream_t)stream));
}

int pp_selective_scan_bwd_f32(const float* u, const float* delta, const float* A, const float* Bm,
                              const float* Cm, const float* D, const float* gy, const float* xhist,
                              float* gu, float* gdelta, float* gA_t, float* gB, float* gC, float* gD,
                              int b, int l, int d, int n, void* stream) {
  if (!u || !delta || !A || !Bm || !Cm || !D || !gy || !xhist || !gu || !gdelta || !gA_t || !gB ||
      !gC || !gD)
    return PP_ERR_NULL;
  int rc = scan_args_ok(b, l, d, n);
  if (rc != PP_OK) return rc;
  return hip_rc(pp_launch_scan_bwd(u, delta, A, Bm, Cm, D, gy, xhist, gu, gdelta, gA_t, gB, gC, gD,
                                   b, l, d, n, (hipStream_t)stream));
}

int pp_layer_norm_fwd_f32(const float* x, float* y, float* rstd, int B, int D, float eps,
                          void* stream) {
  if (!x || !y || !rstd) return PP_ERR_NULL;
  if (!aligned16(x) || !aligned16(y)) return PP_ERR_ALIGN;
  int rc = norm_args_ok(B, D);
  if (rc != PP_OK) return rc;
  return hip_rc(pp_launch_layer_norm_fwd(x, y, rstd, B, D, eps, (hipStream_t)stream));
}

int pp_layer_norm_bwd_f32(const float* y, const float* gy, const float* rstd, float* gx, int B,
                          int D, void* stream) {
  if (!y || !gy || !rstd || !gx) return PP_ERR_NULL;
  if (!aligned16(y) || !aligned16(gy) || !aligned16(gx)) return PP_ERR_ALIGN;
  int rc = norm_args_ok(B, D);
  if (rc != PP_OK) return rc;
  return hip_rc(pp_launch_layer_norm_bwd(y, gy, rstd, gx, B, D, (hipStream_t)stream));
}

int pp_max_abs_norm_fwd_f32(const float* x, float* y, int B, int D, float eps, void* stream) {
  if (!x || !y) return PP_ERR_NULL;
  if (!aligned16(x) || !aligned16(y)) return PP_ERR_ALIGN;
  int rc = norm_args_ok(B, D);
  if (rc != PP_OK) return rc;
  return hip_rc(pp_launch_max_abs_fwd(x, y, B, D, eps, (hipStream_t)stream));
}

int pp_max_abs_norm_bwd_f32(const float* x, const float* gy, float* gx, int B, int D, float eps,
                            void* stream) {
  if (!x || !gy || !gx) return PP_ERR_NULL;
  if (!aligned16(x) || !aligned16(gy) || !aligned16(gx)) return PP_ERR_ALIGN;
  int rc = norm_args_ok(B, D);
  if (rc != PP_OK) return rc;
  return hip_rc(pp_launch_max_abs_bwd(x, gy, gx, B, D, eps, (hipStream_t)stream));
}

int pp_max_abs_norm_bwd_y_f32(const float* y, const float* gy, const float* den,
                              const int32_t* amax, float* gx, int B, int D, void* stream) {
  if (!y || !gy || !den || !amax || !gx) return PP_ERR_NULL;
  if (!aligned16(y) || !aligned16(gy) || !aligned16(gx)) return PP_ERR_ALIGN;
  int rc = norm_args_ok(B, D);
  if (rc != PP_OK) return rc;
  return hip_rc(pp_launch_max_abs_bwd_y(y, gy, den, amax, gx, B, D, (hipStream_t)stream));
}

int pp_char_windows(const uint8_t* text, int64_t text_len, const int64_t* idx, int B, int features,
                    float* x, int64_t* tokens, int64_t* y, void* stream) {
  if (!text || !idx || (!x && !tokens && !y)) return PP_ERR_NULL;
  if (B < 1 || features < 1 || text_len < (int64_t)features + 1) return PP_ERR_SHAPE;
  if ((int64_t)B * (features + 1) >= (1LL << 40)) return PP_ERR_RANGE;
  return hip_rc(pp_launch_char_windows(text, (long)text_len, reinterpret_cast<const long*>(idx), B,
                                       features, x, reinterpret_cast<long*>(tokens),
                                       reinterpret_cast<long*>(y), (hipStream_t)stream));
}

size_t pp_cross_entropy_workspace_bytes(void) { return pp_ce_workspace_bytes(); }

int pp_cross_entropy_fwd_f32(const float* logits, const int64_t* target, float* loss_acc,
                             void* workspace, size_t workspace_bytes, int B, int C, void* stream) {
  if (!logits || !target || !loss_acc) return PP_ERR_NULL;
  if (!aligned16(logits)) return PP_ERR_ALIGN;
  int rc = norm_args_ok(B, C);
  if (rc != PP_OK) return rc;
  if (!workspace || workspace_bytes < pp_ce_workspace_bytes()) return PP_ERR_WORKSPACE;
  return hip_rc(pp_launch_ce_fwd(logits, reinterpret_cast<const long*>(target), loss_acc, workspace,
                                 B, C, (hipStream_t)stream));
}

int pp_cross_entropy_bwd_f32(const float* logits, const int64_t* target, const float* gout,
                             float* glogits, int B, int C, void* stream) {
  if (!logits || !target || !gout || !glogits) return PP_ERR_NULL;
  if (!aligned16(logits) || !aligned16(glogits)) return PP_ERR_ALIGN;
  int rc = norm_args_ok(B, C);
  if (rc != PP_OK) return rc;
  return hip_rc(pp_launch_ce_bwd(logits, reinterpret_cast<const long*>(target), gout, glogits, B, C,
                                 (hipStream_t)stream));
}

static int mamba_dims_ok(long rows, int d) {
  if (rows < 1 || d < 1) return PP_ERR_SHAPE;
  if (rows * d >= (1L << 31)) return PP_ERR_RANGE;
  return PP_OK;
}

size_t pp_causal_conv_silu_bwd_workspace_bytes(int b, int d) {
  return b > 0 && d > 0 ? pp_conv_silu_bwd_workspace_bytes(b, d) : 0;
}

int pp_causal_conv_silu_fwd_f32(const float* x, int ldx, const float* w, const float* bias,
                                float* conv, float* y, int b, int l, int d, int k, void* stream) {
  if (!x || !w || !conv || !y) return PP_ERR_NULL;
  if (b < 1 || l < 1 || k < 1 || k > 8 || ldx < d) return PP_ERR_SHAPE;
  int rc = mamba_dims_ok((long)b * l, std::max(d, ldx));
  if (rc != PP_OK) return rc;
  return hip_rc(pp_launch_conv_silu_fwd(x, ldx, w, bias, conv, y, b, l, d, k, (hipStream_t)stream));
}

int pp_causal_conv_silu_bwd_f32(const float* x, int ldx, const float* w, const float* conv,
                                const float* gy, float* gx, float* gw, float* gbias,
                                void* workspace, size_t workspace_bytes, int b, int l, int d, int k,
                                void* stream) {
  if (!x || !w || !conv || !gy || !gx || !gw) return PP_ERR_NULL;
  if (b < 1 || l < 1 || k < 1 || k > 8 || ldx < d) return PP_ERR_SHAPE;
  int rc = mamba_dims_ok((long)b * l, std::max(d, ldx));
  if (rc != PP_OK) return rc;
  if (!workspace || workspace_bytes < pp_conv_silu_bwd_workspace_bytes(b, d)) return PP_ERR_WORKSPACE;
  return hip_rc(pp_launch_conv_silu_bwd(x, ldx, w, conv, gy, gx, gw, gbias, workspace, b, l, d, k,
                                        (hipStream_t)stream));
}

int pp_silu_gate_fwd_f32(const float* s, const float* r, int ldr, float* y, long rows, int d,
                         void* stream) {
  if (!s || !r || !y) return PP_ERR_NULL;
  if (ldr < d) return PP_ERR_SHAPE;
  int rc = mamba_dims_ok(rows, std::max(d, ldr));
  if (rc != PP_OK) return rc;
  return hip_rc(pp_launch_silu_gate_fwd(s, r, ldr, y, rows, d, (hipStream_t)stream));
}

int pp_silu_gate_bwd_f32(const float* s, const float* r, int ldr, const float* gy, float* gs,
                         float* gr, long rows, int d, void* stream) {
  if (!s || !r || !gy || !gs || !gr) return PP_ERR_NULL;
  if (ldr < d) return PP_ERR_SHAPE;
  int rc = mamba_dims_ok(rows, std::max(d, ldr));
  if (rc != PP_OK) return rc;
  return hip_rc(pp_launch_silu_gate_bwd(s, r, ldr, gy, gs, gr, rows, d, (hipStream_t)stream));
}

size_t pp_rms_norm_bwd_workspace_bytes_f32(long rows, int D) {
  return rows > 0 && D > 0 ? pp_rms_norm_bwd_workspace_bytes(rows, D) : 0;
}

int pp_rms_norm_fwd_f32(const float* x, const float* weight, float* y, float* rstd, long rows,
                        int D, float eps, void* stream) {
  if (!x || !weight || !y || !rstd) return PP_ERR_NULL;
  if (D > 1024) return PP_ERR_SHAPE;
  int rc = mamba_dims_ok(rows, D);
  if (rc != PP_OK) return rc;
  return hip_rc(pp_launch_rms_norm_fwd(x, weight, y, rstd, rows, D, eps, (hipStream_t)stream));
}

int pp_rms_norm_bwd_f32(const float* x, const float* weight, const float* rstd, const float* gy,
                        float* gx, float* gweight, void* workspace, size_t workspace_bytes,
                        long rows, int D, void* stream) {
  if (!x || !weight || !rstd || !gy || !gx || !gweight) return PP_ERR_NULL;
  if (D > 1024) return PP_ERR_SHAPE;
  int rc = mamba_dims_ok(rows, D);
  if (rc != PP_OK) return rc;
  if (!workspace || workspace_bytes < pp_rms_norm_bwd_workspace_bytes(rows, D))
    return PP_ERR_WORKSPACE;
  return hip_rc(pp_launch_rms_norm_bwd(x, weight, rstd, gy, gx, gweight, workspace, rows, D,
                                       (hipStream_t)stream));
}

int pp_lion_step_f32(int count, float* const* p, const float* const* g, float* const* m,
                     const long* numel, float lr, float beta1, float beta2, float weight_decay,
                     int sparse, void* stream) {
  if (count < 1 || count > PP_OPT_MAX_TENSORS) return PP_ERR_SHAPE;
  if (!p || !g || !m || !numel) return PP_ERR_NULL;
  for (int t = 0; t < count; ++t) {
    if (!p[t] || !g[t] || !m[t]) return PP_ERR_NULL;
    if (numel[t] < 1) return PP_ERR_SHAPE;
    if ((((size_t)p[t]) | ((size_t)g[t]) | ((size_t)m[t])) & 3) return PP_ERR_ALIGN;
  }
  return hip_rc(pp_launch_lion(count, p, g, m, numel, lr, beta1, beta2, weight_decay, sparse,
                               (hipStream_t)stream));
}

}  // extern "C"
