// Internal declarations shared by the translation units of libpphip.so.
// Not part of the C ABI (that is include/pp_hip.h).
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

#include "pp_hip.h"

struct PPShape {
  int B, I, O, n, S, K, stride, disc;
  float length, half, Sf, out_scale;
};

// Node table handed to kernels by value (lands in SGPRs / kernarg).
struct BasisTab {
  float X[PP_MAX_N];       // Chebyshev-Lobatto nodes (fp32 values of the caller)
  float invden[PP_MAX_N];  // 1 / prod_{m!=j} (X_j - X_m)
};

// ---- gather path (pp_gather.hip) ----
hipError_t pp_launch_segment_ids(const float* x, int32_t* seg, const PPShape& sh, hipStream_t st);
hipError_t pp_launch_fwd_gather(const float* x, const float* wt, float* y, int32_t* seg,
                                const PPShape& sh, const BasisTab& tab, hipStream_t st);
bool pp_gather_ln_supported(const PPShape& sh);  // fused LayerNorm epilogue: n <= 4, O <= 1024
hipError_t pp_launch_fwd_gather_ln(const float* x, const float* wt, float* y, float* rstd,
                                   const PPShape& sh, const BasisTab& tab, float eps,
                                   hipStream_t st);
hipError_t pp_launch_fwd_gather_maxabs(const float* x, const float* wt, float* y, float* den,
                                       int32_t* amax, const PPShape& sh, const BasisTab& tab,
                                       float eps, hipStream_t st);
hipError_t pp_launch_bwd_input_gather(const float* x, const float* gy, const float* wt, float* gx,
                                      const PPShape& sh, const BasisTab& tab, hipStream_t st);
hipError_t pp_launch_bwd_weight_atomic(const float* x, const float* gy, float* gwt,
                                       const PPShape& sh, const BasisTab& tab, hipStream_t st);
hipError_t pp_launch_transpose(const float* src, float* dst, int rows, int cols, hipStream_t st);

// ---- table form of a layer whose inputs lie on a uniform grid of P points (pp_table.hip) ----
bool pp_grid_table_supported(const PPShape& sh, int P);
hipError_t pp_launch_grid_table(const float* wt, float* T, const PPShape& sh, const BasisTab& tab,
                                int P, hipStream_t st);
hipError_t pp_launch_grid_table_grad(const float* GT, float* gwt, const PPShape& sh,
                                     const BasisTab& tab, int P, hipStream_t st);

// ---- LDS-staged slab path for many-segment layers with few inputs (pp_slab.hip) ----
bool pp_slab_fwd_supported(const PPShape& sh);  // the kernel can run this shape
bool pp_slab_fwd_preferred(const PPShape& sh);  // ... and measured faster than the gather kernel
hipError_t pp_launch_fwd_slab(const float* x, const float* wt, float* y, int32_t* seg,
                              const PPShape& sh, const BasisTab& tab, hipStream_t st);

// ---- row-tile path: accumulators of a tall row tile in registers, slab in LDS (pp_tile.hip) ----
bool pp_tile_fwd_supported(const PPShape& sh);
bool pp_tile_fwd_preferred(const PPShape& sh, bool with_norm = false);
size_t pp_tile_fwd_workspace_bytes(const PPShape& sh, bool with_norm);
// backward-input on the same tiles (n >= 2, I % 4 == 0)
bool pp_tile_bwd_x_supported(const PPShape& sh);
bool pp_tile_bwd_x_preferred(const PPShape& sh);
size_t pp_tile_bwd_x_workspace_bytes(const PPShape& sh);
hipError_t pp_launch_bwd_x_tile(const float* x, const float* gy, const float* wt, float* gx, void* ws,
                                const PPShape& sh, const BasisTab& tab, hipStream_t st);
hipError_t pp_launch_fwd_tile(const float* x, const float* wt, float* y, int32_t* seg, void* ws,
                              float* rstd, float ln_eps, const PPShape& sh, const BasisTab& tab,
                              hipStream_t st);

// ---- sort + segmented reduce path (pp_sort.hip) ----
size_t pp_sort_workspace_bytes(const PPShape& sh);
bool pp_sort_supported(const PPShape& sh);
hipError_t pp_launch_bwd_weight_sort(const float* x, const float* gy, float* gwt, void* ws,
                                     const PPShape& sh, const BasisTab& tab, hipStream_t st);

// ---- register-resident accumulator tables (pp_regacc.hip) ----
bool pp_direct_supported(const PPShape& sh);  // B <= 1024, n <= 4
hipError_t pp_launch_bwd_weight_direct(const float* x, const float* gy, float* gwt,
                                       const PPShape& sh, const BasisTab& tab, hipStream_t st);
int pp_regacc_parts(const PPShape& sh);  // waves an input (1: K <= 192)
bool pp_regacc_supported(const PPShape& sh);
bool pp_regacc_preferred(const PPShape& sh);
size_t pp_regacc_workspace_bytes(const PPShape& sh);
hipError_t pp_launch_bwd_weight_regacc(const float* x, const float* gy, float* gwt, void* ws,
                                       const PPShape& sh, const BasisTab& tab, hipStream_t st);

// ---- dense f32-MFMA path (pp_dense.hip) ----
bool pp_dense_supported(const PPShape& sh);
size_t pp_dense_bwd_weight_workspace_bytes(const PPShape& sh);
void pp_launch_chunk_reduce(const float* part, float* gwt, long total, int nchunk, float scale,
                            hipStream_t st);
hipError_t pp_launch_fwd_dense(const float* x, const float* wt, float* y, int32_t* seg,
                               const PPShape& sh, const BasisTab& tab, hipStream_t st);
hipError_t pp_launch_bwd_input_dense(const float* x, const float* gy, const float* wt, float* gx,
                                     const PPShape& sh, const BasisTab& tab, hipStream_t st);
hipError_t pp_launch_bwd_weight_dense(const float* x, const float* gy, float* gwt, void* ws,
                                      const PPShape& sh, const BasisTab& tab, hipStream_t st);

// ---- row-wise normalisations (pp_norm.hip) ----
hipError_t pp_launch_layer_norm_fwd(const float* x, float* y, float* rstd, int B, int D, float eps,
                                    hipStream_t st);
hipError_t pp_launch_layer_norm_bwd(const float* y, const float* gy, const float* rstd, float* gx,
                                    int B, int D, hipStream_t st);
hipError_t pp_launch_max_abs_fwd(const float* x, float* y, int B, int D, float eps, hipStream_t st);
hipError_t pp_launch_max_abs_bwd(const float* x, const float* gy, float* gx, int B, int D, float eps,
                                 hipStream_t st);
hipError_t pp_launch_char_windows(const uint8_t* text, long text_len, const long* idx, int B, int F,
                                  float* x, long* tokens, long* y, hipStream_t st);
size_t pp_ce_workspace_bytes();
hipError_t pp_launch_ce_fwd(const float* logits, const long* target, float* out, void* ws, int B,
                            int D, hipStream_t st);
hipError_t pp_launch_ce_bwd(const float* logits, const long* target, const float* gout,
                            float* glogits, int B, int D, hipStream_t st);
hipError_t pp_launch_max_abs_bwd_y(const float* y, const float* gy, const float* den,
                                   const int32_t* amax, float* gx, int B, int D, hipStream_t st);

hipError_t pp_launch_lion(int count, float* const* p, const float* const* g, float* const* m,
                          const long* numel, float lr, float b1, float b2, float wd, int sparse,
                          hipStream_t st);

// ---- Mamba selective scan (pp_scan.hip) ----
hipError_t pp_launch_scan_fwd(const float* u, const float* delta, const float* A, const float* Bm,
                              const float* Cm, const float* D, float* y, float* xhist, int b, int l,
                              int d, int n, hipStream_t st);
hipError_t pp_launch_scan_bwd(const float* u, const float* delta, const float* A, const float* Bm,
                              const float* Cm, const float* D, const float* gy, const float* xhist,
                              float* gu, float* gdelta, float* gA_t, float* gB, float* gC, float* gD,
                              int b, int l, int d, int n, hipStream_t st);

// ---- Mamba block pieces around the projections and the scan (pp_mamba.hip) ----
size_t pp_conv_silu_bwd_workspace_bytes(int b, int d);
size_t pp_rms_norm_bwd_workspace_bytes(long rows, int D);
hipError_t pp_launch_conv_silu_fwd(const float* x, int ldx, const float* w, const float* bias,
                                   float* conv, float* y, int b, int l, int d, int k,
                                   hipStream_t st);
hipError_t pp_launch_conv_silu_bwd(const float* x, int ldx, const float* w, const float* conv,
                                   const float* gy, float* gx, float* gw, float* gb, void* ws,
                                   int b, int l, int d, int k, hipStream_t st);
hipError_t pp_launch_silu_gate_fwd(const float* s, const float* r, int ldr, float* y, long rows,
                                   int d, hipStream_t st);
hipError_t pp_launch_silu_gate_bwd(const float* s, const float* r, int ldr, const float* gy,
                                   float* gs, float* gr, long rows, int d, hipStream_t st);
hipError_t pp_launch_rms_norm_fwd(const float* x, const float* w, float* y, float* rstd, long rows,
                                  int D, float eps, hipStream_t st);
hipError_t pp_launch_rms_norm_bwd(const float* x, const float* w, const float* rstd,
                                  const float* gy, float* gx, float* gw, void* ws, long rows, int D,
                                  hipStream_t st);

// ---- dense path on bf16 MFMA with a 3-way fp32 split (pp_bf3_fwd.hip, pp_bf3_bwdx.hip, pp_bf3_bwdw.hip; shared: pp_bf3_common.h) ----
bool pp_bf3_supported(const PPShape& sh);
size_t pp_bf3_fwd_workspace_bytes(const PPShape& sh);
hipError_t pp_launch_fwd_bf3(const float* x, const float* wt, float* y, int32_t* seg, void* ws,
                             const PPShape& sh, const BasisTab& tab, hipStream_t st);
hipError_t pp_launch_fwd_h2(const float* x, const float* wt, float* y, int32_t* seg, void* ws,
                            const PPShape& sh, const BasisTab& tab, bool prepacked, float* rstd,
                            float ln_eps, int32_t* amax, hipStream_t st);  // amax: max-abs epilogue
bool pp_h2_fwd_ln_supported(const PPShape& sh);
size_t pp_h2_fwd_packed_bytes(const PPShape& sh);
void pp_launch_pack_both_h2(const float* wt, void* ws, const PPShape& sh, hipStream_t st);
void pp_launch_pack_multi_h2(int count, const float* const* wt, void* const* ws,
                             const PPShape* shapes, hipStream_t st);
size_t pp_h2_bwd_x_packed_bytes(const PPShape& sh);
bool pp_bf3_bwd_x_supported(const PPShape& sh);
size_t pp_bf3_bwd_x_workspace_bytes(const PPShape& sh);
hipError_t pp_launch_bwd_x_bf3(const float* x, const float* gy, const float* wt, float* gx,
                               void* ws, const PPShape& sh, const BasisTab& tab, hipStream_t st);
// rowstats (or nullptr): [pp_bwd_x_rowstats_tiles][B] float2 = (sum_i gx, sum_i gx*x) per tile of
// 32 inputs, for the LayerNorm backward of the layer below (pp_launch_bwd_w_h2_ln)
hipError_t pp_launch_bwd_x_h2(const float* x, const float* gy, const float* wt, float* gx,
                              void* ws, const PPShape& sh, const BasisTab& tab, bool prepacked,
                              size_t q_off, float* rowstats, hipStream_t st);
int pp_bwd_x_rowstats_tiles(const PPShape& sh);
bool pp_bf3_bwd_w_supported(const PPShape& sh);
size_t pp_bf3_bwd_w_workspace_bytes(const PPShape& sh);
hipError_t pp_launch_bwd_w_bf3(const float* x, const float* gy, float* gwt, void* ws,
                               const PPShape& sh, const BasisTab& tab, hipStream_t st);
hipError_t pp_launch_bwd_w_h2(const float* x, const float* gy, float* gwt, void* ws,
                              const PPShape& sh, const BasisTab& tab, hipStream_t st);
// the same with the LayerNorm backward of the layer's output folded into the pack pass:
// g_out = rstd*(gy - m1 - yn*m2) (row means from `ntile` tiles of rowstats), gwt from g_out
hipError_t pp_launch_bwd_w_h2_ln(const float* x, const float* gy, const float* yn,
                                 const float* rstd, const float* rowstats, int ntile, float* g_out,
                                 float* gwt, void* ws, const PPShape& sh, const BasisTab& tab,
                                 hipStream_t st);
