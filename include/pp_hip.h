/*
 * pp_hip.h -- C ABI of libpphip.so: the piecewise-polynomial fully-connected
 * layer (continuous + discontinuous Lagrange layers) as hand-written gfx950
 * HIP kernels.
 *
 * What it replaces.  In the reference the whole path is Python calling the
 * third-party package high-order-layers-torch (not vendored, pyproject.toml:21);
 * there is no FFI to mirror, so each entry point cites the reference call site
 * whose work it performs.  All of them sit behind
 *     high_order_fc_layers(layer_type, n, in_features, out_features, segments, ...)
 *         /root/reference/language_interpolation/networks.py:399-426, :799-806
 *     HighOrderMLP(...).forward
 *         /root/reference/language_interpolation/networks.py:866-882,
 *         /root/reference/language_interpolation/state_space_network.py:186-196
 * and the autograd backward that Lightning's loss.backward() triggers
 * (networks.py:122-132, :164-165).
 *
 * Conventions (every entry point):
 *   - extern "C", returns 0 on success, a negative PP_ERR_* on a rejected
 *     argument, or a positive hipError_t on a launch failure; never throws.
 *   - all data pointers are DEVICE pointers, fp32, contiguous row-major,
 *     16-byte aligned; the caller owns every buffer, the library allocates
 *     nothing and keeps no state between calls (safe from any host thread --
 *     autograd calls backward from its own engine thread; the only runtime
 *     state touched is hipFuncSetAttribute on the launching device, set on
 *     every call that needs it).  One exception, stated at its declaration:
 *     pp_forward_noalloc_f32 keeps the per-device address of its scratch array.
 *   - `stream` is a hipStream_t passed as void*; every call is asynchronous and
 *     stream-ordered; no call synchronises the device.
 *   - B rows (samples), I in_features, O out_features, n nodes per segment,
 *     S segments, `discontinuous` 0/1, `length` the layer's `length`/`scale`
 *     (2.0 at every reference call site).
 *     K = n*S (discontinuous) or (n-1)*S+1 (continuous, n >= 2).
 *   - weights are taken in the KERNEL layout w_t[I][K][O] (out contiguous):
 *     the n active rows of one (sample, input) pair are one contiguous n*O
 *     slab.  The reference's parameter layout is w[O][I][K];
 *     pp_transpose_w_f32 / pp_transpose_wt_f32 convert (the Python module
 *     stores its nn.Parameter with logical shape [O,I,K] and physical layout
 *     [I,K,O], so no conversion runs in the training loop).
 *   - `nodes`: HOST pointer to n floats (the Chebyshev-Lobatto nodes exactly as
 *     the caller's framework computed them in fp32), or NULL for the library's
 *     own -cos(pi k/(n-1))*length/2.
 */
#ifndef PP_HIP_H
#define PP_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PP_MAX_N 8 /* nodes per segment supported by the kernels */

enum {
  PP_OK = 0,
  PP_ERR_NULL = -1,      /* a required pointer is NULL */
  PP_ERR_SHAPE = -2,     /* B/I/O/n/S out of range (n > PP_MAX_N, continuous n < 2, ...) */
  PP_ERR_ALIGN = -3,     /* a pointer is not 16-byte aligned */
  PP_ERR_WORKSPACE = -4, /* workspace missing or smaller than pp_workspace_bytes() */
  PP_ERR_METHOD = -5,    /* unknown / inapplicable method id */
  PP_ERR_RANGE = -6,     /* index arithmetic would overflow int32 */
  PP_ERR_NOALLOC_SIZE = -7 /* pp_forward_noalloc_f32: the shape needs more than PP_NOALLOC_SCRATCH_BYTES */
};

/* method ids */
enum {
  PP_METHOD_AUTO = 0,
  PP_FWD_GATHER = 1,  /* wavefront-cooperative gather of the active slab (any S) */
  PP_FWD_DENSE = 2,   /* expanded-basis x weight on f32 MFMA (small K only) */
  PP_FWD_BF16X3 = 3,  /* same contraction on bf16 MFMA, operands split 3-way (fp32-level accuracy);
                         needs a workspace: pp_forward_ws_f32 */
  PP_FWD_FP16X2 = 4,  /* same on fp16 MFMA, operands split 2-way, weights scaled per tensor into fp16's
                         range, fp32 fix-up of tiles whose basis does not fit (fp32-level accuracy);
                         needs a workspace */
  PP_FWD_SLAB = 5,    /* many segments, few inputs, large batch (the character-grid input layers):
                         the weight slab of one input staged in LDS by LDS-DMA, rows gathered from
                         there; PP_METHOD_AUTO picks it where it applies, else PP_ERR_METHOD */
  PP_FWD_TILE = 6,    /* many segments, large batch (cfg3 hidden, cfg4 input): a workgroup keeps the
                         accumulators of a tall tile of rows x 64 outs in registers, the slab of the
                         current input streams through LDS by LDS-DMA, one ds_read_b128 serves four
                         rows (lanes rotate over the rows by DPP); the inputs are split over
                         workgroups when the output is too small to fill the chip -- partial outputs
                         in the workspace (pp_forward_workspace_bytes), summed in a fixed order.
                         O % 4 == 0, n <= 4, K <= 320; bitwise reproducible */
  PP_BWD_W_ATOMIC = 1, /* global float atomics into gw */
  PP_BWD_W_SORT = 2,   /* counting sort by (input, segment) + segmented reduce; float atomics only
                          for buckets shared between segments (continuous layers) or spanning
                          two reduce units */
  PP_BWD_W_DENSE = 3,  /* Phi^T x gy on f32 MFMA (small K only) */
  PP_BWD_W_FP16X2 = 5, /* same on fp16 MFMA, operands split 2-way, gy scaled per (512 rows, out) */
  PP_BWD_W_BF16X3 = 4, /* Phi^T x gy on bf16 MFMA, operands split 3-way (small K only) */
  PP_BWD_W_REGACC = 6, /* many segments (weight rows per input K <= 192, n <= 3): a wave owns one input
                          x 64 outs, keeps one register per weight row and adds the batch rows in
                          their natural order through the VGPR index mode; no sort, no zero fill, no
                          atomics, bitwise reproducible; needs a workspace.  Discontinuous layers
                          with n = 2, 3 up to K <= 576: two or three waves an input, whole segments
                          each.  Large layers x batches share gy through LDS (round 3) */
  PP_BWD_W_DIRECT = 7, /* small batches (B <= 1024, n <= 4) of many-segment layers: the wave that owns a
                          weight row finds the batch rows that land on it (segment ids of the input
                          column in LDS, ballots in row order) and stores the finished row once; no
                          zero fill, no atomics, no workspace, bitwise reproducible */
  PP_BWD_X_GATHER = 1,
  PP_BWD_X_DENSE = 2,
  PP_BWD_X_FP16X2 = 4, /* same on fp16 MFMA, operands split 2-way, weights scaled per tensor and gy per
                          (row, 128-out chunk) by powers of two; needs a workspace */
  PP_BWD_X_TILE = 5,   /* many-segment layers with n >= 2 at large batches (round 3): the row-tile
                          forward's slab stream and LDS gathers, gy of a 1024-row x 64-out tile in
                          registers, dot-product pieces handed to the row's lane by DPP rotation;
                          needs a workspace (records + one partial gx per out chunk); I % 4 == 0 */
  PP_BWD_X_BF16X3 = 3 /* gy x w^T on bf16 MFMA, operands split 3-way; needs a workspace:
                         pp_backward_input_ws_f32 */
};

/* Library version (major*10000 + minor*100 + patch). No reference counterpart. */
int pp_version(void);

/* Text for a return code (static storage). */
const char* pp_error_string(int code);

/* A1 of SURVEY.md 8(a): seg[b,i] = clamp(trunc(((x+L/2)/L)*S), 0, S-1), fp32 in
 * that op order.  Replaces `Piecewise.forward` step 1 of high-order-layers-torch
 * as reached from networks.py:799-806.  seg: int32[B*I]. */
int pp_segment_ids_f32(const float* x, int32_t* seg, int B, int I, int S, float length,
                       void* stream);

/* Forward, A1-A4 fused: y[B,O] = out_scale * sum_i sum_j l_j(xi[b,i]) * w_t[i, s*stride+j, :].
 * Replaces the layer's forward as called from nn.Sequential inside HighOrderMLP
 * (networks.py:160-162 -> :866-882) and from reshape_apply (utils.py:36).
 * seg_out: optional int32[B*I] (NULL to skip) for the bit-exact index check.
 * out_scale: 1.0, or 1/I for rescale_output=True. */
int pp_forward_f32(const float* x, const float* w_t, float* y, int32_t* seg_out, int B, int I,
                   int O, int n, int S, int discontinuous, float length, const float* nodes,
                   float out_scale, int method, void* stream);

/* Same as pp_forward_f32 with caller-owned scratch (pp_forward_workspace_bytes; may be
 * NULL/0 for methods that need none).  PP_METHOD_AUTO may pick a workspace method
 * only through this entry point. */
size_t pp_forward_workspace_bytes(int B, int I, int O, int n, int S, int discontinuous,
                                  float length, int method);
int pp_forward_ws_f32(const float* x, const float* w_t, float* y, int32_t* seg_out,
                      void* workspace, size_t workspace_bytes, int B, int I, int O, int n, int S,
                      int discontinuous, float length, const float* nodes, float out_scale,
                      int method, void* stream);

/* Forward for callers that cannot allocate device memory or hand over a workspace (code under
 * HIP graph capture, tight serving loops, harnesses that forbid hipMalloc): pp_forward_ws_f32
 * with the scratch taken from a fixed array inside the library (PP_NOALLOC_SCRATCH_BYTES per
 * GPU the library runs on, held until the process ends).  It has no counterpart in the
 * reference.  Same arguments and results as pp_forward_f32 / pp_forward_ws_f32 (bit-identical
 * to the latter); PP_METHOD_AUTO may pick the workspace methods and, where the one it would pick
 * needs more scratch than that, runs what pp_forward_f32 runs for the shape (no scratch).  An
 * explicitly requested method that needs more returns PP_ERR_NOALLOC_SIZE
 * (pp_forward_workspace_bytes tells; a method without scratch, e.g. PP_FWD_GATHER, always
 * fits).  Allocates nothing, frees nothing.  State: the first call on a device looks up that
 * device's copy of the array (hipGetSymbolAddress, under a mutex) and keeps the address -- make
 * that first call outside stream capture; later calls are capture-safe.  Every call on a GPU
 * shares that GPU's array: calls that overlap on one GPU (two streams, two threads) race --
 * serialise them or use pp_forward_ws_f32. */
#define PP_NOALLOC_SCRATCH_BYTES (8u << 20)
int pp_forward_noalloc_f32(const float* x, const float* w_t, float* y, int32_t* seg_out, int B, int I,
                           int O, int n, int S, int discontinuous, float length, const float* nodes,
                           float out_scale, int method, void* stream);

/* Pre-split weights for the fp16x2 kernels: one pass per weight UPDATE instead of one inside
 * every forward and backward-input call.  pp_pack_weights_f32 writes both kernels' layouts
 * (and the tensor's scale) into caller-owned memory of pp_packed_weights_bytes (0 = this
 * layer has no fp16x2 kernels); pass that buffer as `workspace` with method
 * PP_FWD_FP16X2 | PP_WS_PREPACKED (forward) or PP_BWD_X_FP16X2 | PP_WS_PREPACKED
 * (backward-input): the call then only reads it.  It must be rewritten whenever w_t
 * changes.  No reference counterpart (layout adaptor, like pp_transpose_w_f32). */
#define PP_WS_PREPACKED 0x100
size_t pp_packed_weights_bytes(int I, int O, int n, int S, int discontinuous, float length);
int pp_pack_weights_f32(const float* w_t, void* packed, size_t packed_bytes, int I, int O, int n,
                        int S, int discontinuous, float length, void* stream);

/* The same for up to PP_PACK_MAX_LAYERS layers in two launches (one absmax, one pack, the
 * layers' blocks side by side): what a training step calls once for all its dense-regime layers
 * instead of pp_pack_weights_f32 per layer (2 x count launches of a few microseconds each).
 * Arrays of `count` entries, meaning as in pp_pack_weights_f32.  No reference counterpart. */
#define PP_PACK_MAX_LAYERS 16
int pp_pack_weights_multi_f32(int count, const float* const* w_t, void* const* packed,
                              const size_t* packed_bytes, const int* I, const int* O, const int* n,
                              const int* S, const int* discontinuous, const float* length,
                              void* stream);

/* Backward w.r.t. the input (A4' of SURVEY.md 8(a)):
 * gx[b,i] = out_scale * (L/(x_max-x_min)) * sum_o gy[b,o] * sum_j l'_j(xi) * w_t[i, s*stride+j, o].
 * Replaces autograd's backward through gather/einsum (loss.backward(), networks.py:164-165).
 * n == 1 gives gx == 0 (written, not skipped). */
int pp_backward_input_f32(const float* x, const float* gy, const float* w_t, float* gx, int B,
                          int I, int O, int n, int S, int discontinuous, float length,
                          const float* nodes, float out_scale, int method, void* stream);

/* Same as pp_backward_input_f32 with caller-owned scratch (pp_backward_input_workspace_bytes;
 * NULL/0 for methods that need none).  PP_METHOD_AUTO may pick a workspace method only
 * through this entry point. */
size_t pp_backward_input_workspace_bytes(int B, int I, int O, int n, int S, int discontinuous,
                                         float length, int method);
int pp_backward_input_ws_f32(const float* x, const float* gy, const float* w_t, float* gx,
                             void* workspace, size_t workspace_bytes, int B, int I, int O, int n,
                             int S, int discontinuous, float length, const float* nodes,
                             float out_scale, int method, void* stream);

/* Bytes of scratch pp_backward_weight_f32 needs for `method` (0 for methods
 * that need none).  Scratch is caller-owned device memory, 16-byte aligned,
 * contents undefined on entry and exit. */
size_t pp_workspace_bytes(int B, int I, int O, int n, int S, int discontinuous, float length,
                          int method);

/* Backward w.r.t. the weights (A4'): gw_t[i,k,o] = out_scale * sum_{(b,j): s[b,i]*stride+j==k}
 * gy[b,o] * l_j(xi[b,i]); DENSE result in the kernel layout [I][K][O], every
 * element written (untouched weights get 0), the callee does any zero-fill it
 * needs.  Replaces autograd's index_put_(accumulate=True) scatter into w.grad.
 * PP_BWD_W_SORT is bitwise reproducible only up to the order of rows inside a
 * bucket; PP_BWD_W_ATOMIC depends on atomic arrival order. */
int pp_backward_weight_f32(const float* x, const float* gy, float* gw_t, void* workspace,
                           size_t workspace_bytes, int B, int I, int O, int n, int S,
                           int discontinuous, float length, const float* nodes, float out_scale,
                           int method, void* stream);

/* Table form of a layer whose inputs lie on a uniform grid (round 3).  The first layer of every
 * reference network sees characters, x = (c - 63.5)/64 for a byte c (single_text_dataset.py:15-16,
 * fed to the input layer at networks.py:793-823 / :866-882): one of `points` = 128 values
 * x_c = (c + 1/2) * 2/points - 1.  On such inputs the layer is exactly the piecewise-constant layer
 * (n = 1, S = points, discontinuous; the segment id of x_c is c, exact in fp32) on
 *   table[i][c][:] = out_scale * sum_j l_j(xi(x_c)) * w_t[i, s(x_c)*stride + j, :]      ([I][points][O]),
 * so a caller who KNOWS its inputs are on the grid (the library takes float x and does not check)
 * runs pp_grid_table_f32 once per step, then pp_forward_* / pp_forward_layer_norm_f32 with
 * (table, n = 1, S = points, discontinuous = 1, out_scale = 1) -- one table row gathered per
 * (row, input) instead of n -- and in the backward pp_backward_weight_f32 with the same n = 1
 * arguments (one indexed add per (row, input); result g_table[I][points][O]) followed by
 *   pp_grid_table_grad_f32: gw_t[i][k][:] = out_scale * sum_{c,j: s(x_c)*stride+j = k} l_j(xi(x_c)) * g_table[i][c][:]
 * (dense [I][K][O], every element written, grid points added in ascending order).
 * points: a power of two <= 128; length must be 2.  Replaces nothing in the reference by itself:
 * same y and gw_t as the general entry points on grid inputs (up to fp32 summation order).
 * Measured (MI355X, 16384 rows, table build and map back included): 16 -> 250, n = 2: forward +
 * LayerNorm 35 -> 26 us, weight gradient level; 16 -> 1000, n = 2 (K = 256): forward 57 -> 52 us,
 * weight gradient 110 -> 58 us.  At a few hundred rows the table costs more than it saves. */
int pp_grid_table_supported(int I, int O, int n, int S, int discontinuous, float length,
                            int points);
int pp_grid_table_f32(const float* w_t, float* table, int I, int O, int n, int S, int discontinuous,
                      float length, const float* nodes, float out_scale, int points, void* stream);
int pp_grid_table_grad_f32(const float* g_table, float* gw_t, int I, int O, int n, int S,
                           int discontinuous, float length, const float* nodes, float out_scale,
                           int points, void* stream);

/* w[O][I][K] (reference parameter layout, e.g. a checkpoint of the reference)
 * -> w_t[I][K][O] and back.  No reference counterpart (layout adaptor). */
int pp_transpose_w_f32(const float* w, float* w_t, int O, int I, int K, void* stream);
int pp_transpose_wt_f32(const float* w_t, float* w, int O, int I, int K, void* stream);

/* Forward of a layer with the FOLLOWING non-affine LayerNorm fused into the kernel's epilogue
 * (the layer -> LazyLayerNormLastDim pairs of HighOrderMLP, networks.py:866-882 with
 * normalization=LazyLayerNormLastDim): y = LN(layer(x)), rstd[B] as pp_layer_norm_fwd_f32
 * writes it (pp_layer_norm_bwd_f32 is the matching backward).  Two kernels:
 *   packed != NULL: the fp16x2 matrix-core kernel on weights pre-split by pp_pack_weights_f32
 *                   (dense regime, O <= 256);
 *   packed == NULL: the gather kernel with whole rows per wave (many-segment layers, first
 *                   layers, lengths off the matrix-core path; n <= 4, O <= 1024, O >= 4),
 *                   bit-identical to pp_forward_f32(PP_FWD_GATHER) + pp_layer_norm_fwd_f32.
 * pp_forward_layer_norm_supported: 1 / 2 when the first / second is what a caller should run
 * for this shape (the kernel PP_METHOD_AUTO would pick for the layer alone has it), else 0. */
int pp_forward_layer_norm_supported(int B, int I, int O, int n, int S, int discontinuous,
                                    float length);
/* ... 3: the row-tile kernel (PP_FWD_TILE) with the normalisation in the kernel that sums its
 * partial outputs; the caller then passes pp_forward_layer_norm_workspace_bytes() of scratch as
 * `packed` / `packed_bytes` (contents undefined on entry and exit).  0 for the other modes. */
size_t pp_forward_layer_norm_workspace_bytes(int B, int I, int O, int n, int S, int discontinuous,
                                             float length);
int pp_forward_layer_norm_f32(const float* x, const float* w_t, float* y, float* rstd,
                              void* packed, size_t packed_bytes, int B, int I, int O, int n, int S,
                              int discontinuous, float length, const float* nodes, float out_scale,
                              float eps, void* stream);

/* Non-affine LayerNorm over the last dim (what LazyLayerNormLastDim computes:
 * /root/reference/language_interpolation/networks.py:72-91, eps 1e-5, weight/bias
 * frozen at 1/0).  y[B,D] = (x - mean)/sqrt(var + eps); rstd[B] is saved for the
 * backward, which works from the OUTPUT y: gx = rstd*(gy - mean(gy) - y*mean(gy*y)).
 * D <= 1024. */
int pp_layer_norm_fwd_f32(const float* x, float* y, float* rstd, int B, int D, float eps,
                          void* stream);
int pp_layer_norm_bwd_f32(const float* y, const float* gy, const float* rstd, float* gx, int B,
                          int D, void* stream);

/* LayerNorm backward folded into the backward of the layer BELOW the normalisation (round 3;
 * the layer -> LazyLayerNormLastDim -> layer chains of HighOrderMLP, networks.py:866-882 with
 * normalization=LazyLayerNormLastDim; replaces autograd's backward through torch.nn.LayerNorm,
 * networks.py:72-91).  Two halves, both on the fp16x2 matrix-core kernels:
 *
 *  pp_backward_input_stats_ws_f32 -- pp_backward_input_ws_f32 of the layer ABOVE the
 *    normalisation (its x is the normalisation's output), which also leaves the two row sums the
 *    LayerNorm backward needs: row_stats[t][b] = (sum_i gx[b,i], sum_i gx[b,i] * x[b,i]) over
 *    the inputs of tile t (32 inputs a tile, pp_row_stats_tiles(I) tiles, float2 per row:
 *    8 * tiles * B bytes).  method: PP_METHOD_AUTO or PP_BWD_X_FP16X2 (also | PP_WS_PREPACKED);
 *    PP_ERR_METHOD when that resolves to a kernel without the epilogue
 *    (pp_backward_input_stats_supported says which shapes have it under PP_METHOD_AUTO).
 *
 *  pp_backward_weight_layer_norm_f32 -- pp_backward_weight_f32 of the layer BELOW, given the
 *    gradient at the normalisation's OUTPUT (gy = the gx above), the normalisation's output
 *    y_norm, its rstd and those row sums (stats_tiles = pp_row_stats_tiles(O) of this layer):
 *    the pass that splits the gradient for the matrix cores computes
 *    g = rstd * (gy - mean(gy) - y_norm * mean(gy * y_norm)) on the way (the expression of
 *    pp_layer_norm_bwd_f32, the means from the row sums), writes it to g_out[B,O] for the layer's
 *    own backward-input call, and the weight gradient is taken from it.  method: PP_METHOD_AUTO
 *    or PP_BWD_W_FP16X2; PP_ERR_METHOD where PP_METHOD_AUTO would not run the fp16x2 kernel
 *    (pp_backward_weight_layer_norm_supported).  Workspace: pp_workspace_bytes(PP_BWD_W_FP16X2).
 *    Same gw_t as pp_layer_norm_bwd_f32 + pp_backward_weight_f32 up to the summation order of
 *    the two means.
 * Measured on MI355X (250 -> 250, 16384 rows): the folded pass 20.5 us against 9.3 + 8.4 us for the
 * two kernels it replaces, +1.6 us for the row sums above -- slower, so the Python host keeps it
 * switched off (functional.FOLD_LAYER_NORM_BACKWARD); profiles/r03_k_ln_fold.txt. */
int pp_row_stats_tiles(int I);
int pp_backward_input_stats_supported(int B, int I, int O, int n, int S, int discontinuous,
                                      float length);
int pp_backward_input_stats_ws_f32(const float* x, const float* gy, const float* w_t, float* gx,
                                   float* row_stats, void* workspace, size_t workspace_bytes, int B,
                                   int I, int O, int n, int S, int discontinuous, float length,
                                   const float* nodes, float out_scale, int method, void* stream);
int pp_backward_weight_layer_norm_supported(int B, int I, int O, int n, int S, int discontinuous,
                                            float length);
int pp_backward_weight_layer_norm_f32(const float* x, const float* gy, const float* y_norm,
                                      const float* rstd, const float* row_stats, int stats_tiles,
                                      float* g_out, float* gw_t, void* workspace,
                                      size_t workspace_bytes, int B, int I, int O, int n, int S,
                                      int discontinuous, float length, const float* nodes,
                                      float out_scale, int method, void* stream);

/* MaxAbsNormalizationLast of high-order-layers-torch (selected at networks.py:58-59):
 * y = x / (max|x| over the last dim + eps); backward needs x and gy. D <= 1024. */
int pp_max_abs_norm_fwd_f32(const float* x, float* y, int B, int D, float eps, void* stream);
int pp_max_abs_norm_bwd_f32(const float* x, const float* gy, float* gx, int B, int D, float eps,
                            void* stream);

/* Forward of a layer with the FOLLOWING MaxAbsNormalizationLast fused into the gather kernel's
 * epilogue (HighOrderMLP with normalization=MaxAbsNormalizationLast, networks.py:58-59 and
 * :866-882): y = layer(x) / (max|layer(x)| + eps), bit-identical to pp_forward_f32(PP_FWD_GATHER)
 * + pp_max_abs_norm_fwd_f32; the raw layer output is never stored.  den[B] receives the
 * denominators and amax[B] the first index attaining each row's maximum;
 * pp_max_abs_norm_bwd_y_f32 is the backward from those and the NORMALISED row:
 *   gx_j = (gy_j - [j == amax] * sign(y_amax) * sum_k gy_k y_k) / den.
 * n <= 4, 4 <= O <= 1024.  pp_forward_max_abs_supported: 2 when this is what a caller should run
 * for the shape (a layer the gather kernels serve), 1 for the layers of the dense regime, which
 * have the same epilogue in the fp16x2 matrix-core kernel (round 3; O <= 256, a block holds whole
 * rows): pp_forward_max_abs_ws_f32 with `packed` = the weights pre-split by pp_pack_weights_f32
 * (bit-identical to pp_forward_ws_f32(PP_FWD_FP16X2) + pp_max_abs_norm_fwd_f32; packed == NULL is
 * pp_forward_max_abs_f32).  0: neither. */
int pp_forward_max_abs_supported(int B, int I, int O, int n, int S, int discontinuous,
                                 float length);
int pp_forward_max_abs_f32(const float* x, const float* w_t, float* y, float* den, int32_t* amax,
                           int B, int I, int O, int n, int S, int discontinuous, float length,
                           const float* nodes, float out_scale, float eps, void* stream);
int pp_forward_max_abs_ws_f32(const float* x, const float* w_t, float* y, float* den, int32_t* amax,
                              void* packed, size_t packed_bytes, int B, int I, int O, int n, int S,
                              int discontinuous, float length, const float* nodes, float out_scale,
                              float eps, void* stream);
int pp_max_abs_norm_bwd_y_f32(const float* y, const float* gy, const float* den,
                              const int32_t* amax, float* gx, int B, int D, void* stream);

/* A batch of character windows from the text held on the device as one uint8 buffer (what
 * generate_dataset + the DataLoader collate of the reference produce on the host:
 * single_text_dataset.py:102-123 and :15-16, lightning_datamodule.py:272-291).  Window b starts at
 * text[idx[b]] (clamped into the text) and has features + 1 characters.  Outputs, each optional
 * (NULL):  x[B][features] = (c + 0.5 - 64) / 64 of the first `features` characters;
 * y[B] = the last character (the target of the MLP models);  tokens[B][features + 1] = all of
 * them as int64 (the Mamba model's indices: inputs = [:, :-1], targets = [:, 1:]). */
int pp_char_windows(const uint8_t* text, int64_t text_len, const int64_t* idx, int B, int features,
                    float* x, int64_t* tokens, int64_t* y, void* stream);

/* Cross entropy of logits[B][C] (C <= 1024) against int64 class indices with the arg-max accuracy,
 * as ASCIIPredictionNet.eval_step computes them (networks.py:121-132: CrossEntropyLoss, mean
 * reduction; accuracy = mean(argmax == target)): loss_acc[0] = mean loss, loss_acc[1] = accuracy.
 * Targets must lie in [0, C).  workspace: pp_cross_entropy_workspace_bytes() of device scratch.
 * The backward recomputes the softmax: glogits = (softmax - onehot) * gout[0] / B, gout a device
 * scalar (the gradient of the mean loss).  Bitwise reproducible. */
size_t pp_cross_entropy_workspace_bytes(void);
int pp_cross_entropy_fwd_f32(const float* logits, const int64_t* target, float* loss_acc,
                             void* workspace, size_t workspace_bytes, int B, int C, void* stream);
int pp_cross_entropy_bwd_f32(const float* logits, const int64_t* target, const float* gout,
                             float* glogits, int B, int C, void* stream);

/* Selective scan of the reference's Mamba block (state_space_network.py:372-420: the Python
 * loop over the sequence in MambaBlock.selective_scan):
 *   x_t = exp(delta_t * A) * x_{t-1} + delta_t * B_t * u_t ;  y_t = <C_t, x_t> + D * u_t
 * u, delta, y: [b][l][d];  A: [d][n];  Bm, Cm: [b][l][n];  D: [d]  (fp32, contiguous).
 * xhist: optional [b][l][n][d] buffer that receives every x_t (the backward needs it; NULL
 * for inference).  The backward returns gu, gdelta [b][l][d], gB, gC [b][l][n], gD [d] and the
 * gradient of A TRANSPOSED, gA_t [n][d] (coalesced atomics; see csrc/pp_scan.hip).  All are
 * overwritten; sums over channels / batch use float atomics, so the low bits of gA_t, gB, gC,
 * gD depend on arrival order. */
int pp_selective_scan_fwd_f32(const float* u, const float* delta, const float* A, const float* Bm,
                              const float* Cm, const float* D, float* y, float* xhist, int b, int l,
                              int d, int n, void* stream);
int pp_selective_scan_bwd_f32(const float* u, const float* delta, const float* A, const float* Bm,
                              const float* Cm, const float* D, const float* gy, const float* xhist,
                              float* gu, float* gdelta, float* gA_t, float* gB, float* gC, float* gD,
                              int b, int l, int d, int n, void* stream);

/* Which method PP_METHOD_AUTO resolves to for a shape (for logs and tests).
 * pass: 0 forward, 1 backward-input, 2 backward-weight.  `length` matters: the matrix-core
 * kernels need a power-of-two length (and S) so that A1/A2's divisions stay exact; any other
 * length (the reference accepts every input_scale, networks.py:386-425) takes the gather path. */
int pp_auto_method(int pass, int B, int I, int O, int n, int S, int discontinuous, float length);

/* The pieces of the reference's Mamba block around the projections and the scan
 * (/root/reference/language_interpolation/state_space_network.py), one kernel per direction each.
 * Channels (d) are contiguous; `ldx` / `ldr` are row strides in floats of inputs that are column
 * slices of a wider tensor (x and res are the two halves of in_proj's output, :311-313).
 *
 * Causal depthwise conv1d + SiLU (:315-324: conv1d(x^T)[:, :, :l]^T then F.silu), x [b,l,d],
 * w [d,k] (nn.Conv1d's [d,1,k]), bias [d] or NULL, k <= 8:
 *   conv[b,t,c] = bias[c] + sum_j w[c,j] * x[b, t-(k-1)+j, c]   (kept for the backward)
 *   y = silu(conv).
 * Backward: gx [b,l,d] contiguous, gw [d,k], gbias [d] or NULL; per-chunk partial sums in the
 * workspace, summed in a fixed order (no atomics). */
size_t pp_causal_conv_silu_bwd_workspace_bytes(int b, int d);
int pp_causal_conv_silu_fwd_f32(const float* x, int ldx, const float* w, const float* bias,
                                float* conv, float* y, int b, int l, int d, int k, void* stream);
int pp_causal_conv_silu_bwd_f32(const float* x, int ldx, const float* w, const float* conv,
                                const float* gy, float* gx, float* gw, float* gbias,
                                void* workspace, size_t workspace_bytes, int b, int l, int d, int k,
                                void* stream);
/* The gate y = s * silu(r) (:326) and its backward (gs, gr contiguous [rows,d]). */
int pp_silu_gate_fwd_f32(const float* s, const float* r, int ldr, float* y, long rows, int d,
                         void* stream);
int pp_silu_gate_bwd_f32(const float* s, const float* r, int ldr, const float* gy, float* gs,
                         float* gr, long rows, int d, void* stream);
/* RMSNorm (:423-433): y = x * rsqrt(mean(x^2) + eps) * weight, rstd[rows] kept for the backward;
 * gx, gweight [D] (per-wave partial sums in the workspace, fixed order).  D <= 1024. */
size_t pp_rms_norm_bwd_workspace_bytes_f32(long rows, int D);
int pp_rms_norm_fwd_f32(const float* x, const float* weight, float* y, float* rstd, long rows,
                        int D, float eps, void* stream);
int pp_rms_norm_bwd_f32(const float* x, const float* weight, const float* rstd, const float* gy,
                        float* gx, float* gweight, void* workspace, size_t workspace_bytes,
                        long rows, int D, void* stream);

/* Lion / SparseLion optimizer step (`optimizer=lion` / `sparse_lion` of the reference's configs:
 * the yaml files under /root/reference/config/optimizer, built at networks.py:183-190 from
 * lion_pytorch.Lion and high-order-layers-torch's SparseLion) for `count` <= PP_OPT_MAX_TENSORS
 * parameters in one launch:  u = sign(b1*m + (1-b1)*g);  p = p*(1 - lr*wd) - lr*u;
 * m = b2*m + (1-b2)*g;  sparse != 0: entries with g == 0 keep p (not decayed either) and m
 * (the upstream SparseLion is absent from /root/reference: parity unpinned).
 * p / g / m: device pointers to `numel[t]` floats each, same element order (storage order of the
 * parameter), updated in place.  Host arrays of `count` entries. */
#define PP_OPT_MAX_TENSORS 32
int pp_lion_step_f32(int count, float* const* p, const float* const* g, float* const* m,
                     const long* numel, float lr, float beta1, float beta2, float weight_decay,
                     int sparse, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* PP_HIP_H */
