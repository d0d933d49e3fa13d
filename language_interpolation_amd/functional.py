"""torch.autograd.Function over the C ABI of libpphip.so.

Host side of the hot path: it validates tensors, hands raw device pointers and
the current HIP stream to the library, and owns nothing else.  The forward
saves only (x, w): segment ids, local coordinates and basis values are
recomputed in the backward kernels instead of keeping the reference's
[B, I, O, n] gathered tensor alive.
"""
from __future__ import annotations

import ctypes
import math
from dataclasses import dataclass, field
from typing import Optional, Tuple

import torch

from . import _lib


def chebyshev_lobatto(n: int, length: float = 2.0) -> torch.Tensor:
    """fp32 nodes -cos(pi k/(n-1)) * length/2, computed in torch fp32 like the
    upstream layer does (so the middle node of an odd n is ~4.4e-8, not 0)."""
    if n == 1:
        return torch.zeros(1, dtype=torch.float32)
    k = torch.arange(0, n)
    ans = -torch.cos(k * math.pi / (n - 1))
    ans = torch.where(torch.abs(ans) < 1e-15, 0 * ans, ans)
    return ((length / 2.0) * ans).to(torch.float32)


def num_weights(n: int, segments: int, discontinuous: bool) -> int:
    return n * segments if discontinuous else (n - 1) * segments + 1


@dataclass(frozen=True)
class LayerSpec:
    """Static description of one layer, hashable so it can ride through autograd."""

    n: int
    segments: int
    discontinuous: bool
    length: float = 2.0
    out_scale: float = 1.0
    fwd_method: str = "auto"
    bwd_x_method: str = "auto"
    bwd_w_method: str = "auto"
    nodes: Tuple[float, ...] = field(default=())
    # > 0: the caller states that every input is one of `input_grid` points (c + 1/2)*2/P - 1
    # (P = 128: the character encoding (c - 63.5)/64 of single_text_dataset.py:15-16); forward and
    # weight gradient then run the layer in its table form (csrc/pp_table.hip).  Not checked.
    input_grid: int = 0

    def node_array(self):
        vals = self.nodes if self.nodes else tuple(chebyshev_lobatto(self.n, self.length).tolist())
        return (ctypes.c_float * self.n)(*vals)


def make_spec(n, segments, discontinuous, length=2.0, rescale_output=False, in_features=1,
              fwd_method="auto", bwd_x_method="auto", bwd_w_method="auto") -> LayerSpec:
    if n < 1 or n > _lib.PP_MAX_N:
        raise ValueError(f"n={n}: the HIP kernels support 1 <= n <= {_lib.PP_MAX_N}")
    if not discontinuous and n < 2:
        raise ValueError("a continuous piecewise layer needs n >= 2 nodes per segment")
    return LayerSpec(
        n=int(n), segments=int(segments), discontinuous=bool(discontinuous), length=float(length),
        out_scale=(1.0 / in_features) if rescale_output else 1.0,
        fwd_method=fwd_method, bwd_x_method=bwd_x_method, bwd_w_method=bwd_w_method,
        nodes=tuple(chebyshev_lobatto(n, length).tolist()),
    )


def _require_gpu(t: torch.Tensor, name: str) -> None:
    if not t.is_cuda:
        raise RuntimeError(
            f"{name} is on {t.device}: the piecewise-polynomial layer runs on MI355X (HIP) only; "
            "there is no CPU path in this package (oracle/ is test infrastructure, not a fallback)"
        )
    if t.dtype != torch.float32:
        raise TypeError(f"{name} must be float32, got {t.dtype}")


def _stream(device) -> ctypes.c_void_p:
    return ctypes.c_void_p(torch.cuda.current_stream(device).cuda_stream)


def _ptr(t: Optional[torch.Tensor]) -> ctypes.c_void_p:
    return ctypes.c_void_p(0 if t is None else t.data_ptr())


def kernel_layout(w: torch.Tensor) -> torch.Tensor:
    """[I,K,O]-contiguous view (or copy) of a logical [O,I,K] weight."""
    wt = w.permute(1, 2, 0)
    return wt if wt.is_contiguous() else wt.contiguous()


def _packed_bytes(w_t, spec: LayerSpec, B: int) -> int:
    """Size of the pre-split weight buffer, 0 when the layer / method choice does not use one."""
    if spec.fwd_method != "auto" or spec.bwd_x_method != "auto" or B == 0:
        return 0
    lib = _lib.load()
    I, K, O = w_t.shape
    args = (B, I, O, spec.n, spec.segments, int(spec.discontinuous), spec.length)
    # the backward-input decides: one buffer holds both layouts, and a batch just above the
    # matrix-core threshold runs the gather forward with the fp16x2 backward-input
    if lib.pp_auto_method(1, *args) != _lib.BWD_X_FP16X2:
        return 0
    return lib.pp_packed_weights_bytes(I, O, spec.n, spec.segments, int(spec.discontinuous),
                                       spec.length)


def _fwd_takes_packed(w_t, spec: LayerSpec, B: int) -> bool:
    I, K, O = w_t.shape
    return _lib.load().pp_auto_method(0, B, I, O, spec.n, spec.segments, int(spec.discontinuous),
                                      spec.length) == _lib.FWD_FP16X2


# buffers written by prepack_layers for the step in flight: (storage pointer, version) -> bytes
_prepacked = {}


def prepack_layers(layers, B: int) -> int:
    """Pre-split the weights of every layer in `layers` (objects with .w and ._spec) that runs on
    the fp16x2 kernels at batch B, all in two launches (pp_pack_weights_multi_f32); the layers'
    forwards of this step then find their buffer instead of packing one by one.  Returns the
    number of layers packed.  The buffers are keyed by the parameter's storage and version
    counter, so an optimizer step in between makes them unusable rather than stale."""
    _prepacked.clear()
    todo = []
    for layer in layers:
        w = layer.w
        if not w.is_cuda:
            continue
        w_t = kernel_layout(w.detach())
        nbytes = _packed_bytes(w_t, layer._spec, B)
        if nbytes:
            todo.append((w, w_t, layer._spec, nbytes))
    for lo in range(0, len(todo), _lib.PACK_MAX_LAYERS):
        part = todo[lo:lo + _lib.PACK_MAX_LAYERS]
        cnt = len(part)
        bufs = [torch.empty(nb, device=w_t.device, dtype=torch.uint8) for _, w_t, _, nb in part]
        rc = _lib.load().pp_pack_weights_multi_f32(
            cnt,
            (ctypes.c_void_p * cnt)(*[w_t.data_ptr() for _, w_t, _, _ in part]),
            (ctypes.c_void_p * cnt)(*[b.data_ptr() for b in bufs]),
            (ctypes.c_size_t * cnt)(*[nb for *_, nb in part]),
            (ctypes.c_int * cnt)(*[w_t.shape[0] for _, w_t, _, _ in part]),
            (ctypes.c_int * cnt)(*[w_t.shape[2] for _, w_t, _, _ in part]),
            (ctypes.c_int * cnt)(*[sp.n for _, _, sp, _ in part]),
            (ctypes.c_int * cnt)(*[sp.segments for _, _, sp, _ in part]),
            (ctypes.c_int * cnt)(*[int(sp.discontinuous) for _, _, sp, _ in part]),
            (ctypes.c_float * cnt)(*[sp.length for _, _, sp, _ in part]),
            _stream(part[0][1].device))
        _lib.check(rc, "pp_pack_weights_multi_f32")
        for (w, w_t, _, _), buf in zip(part, bufs):
            _prepacked[(w_t.data_ptr(), w._version)] = buf
    return len(todo)


def pack_weights(w_t, spec: LayerSpec, B: int, version=None):
    """Pre-split weights for the fp16x2 kernels (one pass serving this step's forward and
    backward-input), or None when the layer / method choice does not use them.  `version`: the
    parameter's version counter, to pick up a buffer prepack_layers wrote for it."""
    nbytes = _packed_bytes(w_t, spec, B)
    if not nbytes:
        return None
    if version is not None:
        buf = _prepacked.get((w_t.data_ptr(), version))
        if buf is not None and buf.numel() == nbytes and buf.device == w_t.device:
            return buf
    lib = _lib.load()
    I, K, O = w_t.shape
    packed = torch.empty(nbytes, device=w_t.device, dtype=torch.uint8)
    with torch.cuda.device(w_t.device):
        rc = lib.pp_pack_weights_f32(_ptr(w_t), _ptr(packed), nbytes, I, O, spec.n, spec.segments,
                                     int(spec.discontinuous), spec.length, _stream(w_t.device))
    _lib.check(rc, "pp_pack_weights_f32")
    return packed


def forward_raw(x, w_t, spec: LayerSpec, want_seg: bool = False, packed=None):
    """y (and optionally int32 segment ids) for x[B,I], w_t[I,K,O]; no autograd.
    packed: buffer from pack_weights (same w_t), used instead of a per-call workspace."""
    lib = _lib.load()
    _require_gpu(x, "x")
    _require_gpu(w_t, "w")
    B, I = x.shape
    I2, K, O = w_t.shape
    if I2 != I or K != num_weights(spec.n, spec.segments, spec.discontinuous):
        raise ValueError(f"weight {tuple(w_t.shape)} does not match x {tuple(x.shape)} / {spec}")
    y = torch.empty(B, O, device=x.device, dtype=torch.float32)
    seg = torch.empty(B, I, device=x.device, dtype=torch.int32) if want_seg else None
    if B == 0:
        return y, seg
    if packed is not None:
        method, ws, nbytes = _lib.FWD_FP16X2 | _lib.WS_PREPACKED, packed, packed.numel()
    else:
        method = _lib.FWD_METHODS[spec.fwd_method]
        nbytes = lib.pp_forward_workspace_bytes(B, I, O, spec.n, spec.segments,
                                                int(spec.discontinuous), spec.length, method)
        ws = torch.empty(nbytes, device=x.device, dtype=torch.uint8) if nbytes else None
    with torch.cuda.device(x.device):
        rc = lib.pp_forward_ws_f32(
            _ptr(x), _ptr(w_t), _ptr(y), _ptr(seg), _ptr(ws), nbytes, B, I, O, spec.n,
            spec.segments, int(spec.discontinuous), spec.length, spec.node_array(),
            spec.out_scale, method, _stream(x.device))
    _lib.check(rc, "pp_forward_ws_f32")
    return y, seg


def backward_input_raw(x, gy, w_t, spec: LayerSpec, packed=None, row_stats=None):
    """gx[B,I].  row_stats: a float32 [tiles, B, 2] tensor (tiles = pp_row_stats_tiles(I)) that
    receives the row sums of gx and gx*x the LayerNorm backward of the layer below needs
    (pp_backward_input_stats_ws_f32: fp16x2 kernels only)."""
    lib = _lib.load()
    B, I = x.shape
    O = w_t.shape[2]
    gx = torch.empty_like(x)
    if B == 0:
        return gx
    if packed is not None:
        method, ws, nbytes = _lib.BWD_X_FP16X2 | _lib.WS_PREPACKED, packed, packed.numel()
    else:
        method = _lib.BWD_X_METHODS[spec.bwd_x_method]
        nbytes = lib.pp_backward_input_workspace_bytes(B, I, O, spec.n, spec.segments,
                                                       int(spec.discontinuous), spec.length,
                                                       method)
        ws = torch.empty(nbytes, device=x.device, dtype=torch.uint8) if nbytes else None
    with torch.cuda.device(x.device):
        if row_stats is not None:
            rc = lib.pp_backward_input_stats_ws_f32(
                _ptr(x), _ptr(gy), _ptr(w_t), _ptr(gx), _ptr(row_stats), _ptr(ws), nbytes, B, I, O,
                spec.n, spec.segments, int(spec.discontinuous), spec.length, spec.node_array(),
                spec.out_scale, method, _stream(x.device))
            _lib.check(rc, "pp_backward_input_stats_ws_f32")
            return gx
        rc = lib.pp_backward_input_ws_f32(
            _ptr(x), _ptr(gy), _ptr(w_t), _ptr(gx), _ptr(ws), nbytes, B, I, O, spec.n,
            spec.segments, int(spec.discontinuous), spec.length, spec.node_array(),
            spec.out_scale, method, _stream(x.device))
    _lib.check(rc, "pp_backward_input_ws_f32")
    return gx


# LayerNorm backward folded into the backward of the layer below.  A layer whose
# input IS the output of a fused layer+LayerNorm node (HighOrderMLP.forward says so through
# `input_is_layer_norm`) lets its backward-input kernel leave the two row sums of gx the
# LayerNorm backward needs; the node below finds them here, keyed by the gradient tensor it is
# handed, and runs pp_backward_weight_layer_norm_f32 instead of the LayerNorm-backward kernel +
# the plain weight gradient.  One slot: the producer's backward is directly followed by the
# consumer's; anything that does not match (another gradient tensor, an accumulated one, a
# changed version counter) falls back to the separate kernels.
#
# OFF by default: measured on MI355X (cfg2 hidden layer, 16384 rows, profiles/r03_k_ln_fold.txt)
# the folded pack pass takes 20.5 us against 9.3 (LayerNorm backward) + 8.4 (pack) for the two
# kernels it replaces -- it reads two tensors and writes two in 128-byte row pieces, 6 MB of
# them live per XCD against 4 MB of L2 -- and the row-sum epilogue costs the backward-input
# kernel 1.6 us: the step is 13 us SLOWER with it.  Kept as a tested option
# (FOLD_LAYER_NORM_BACKWARD = True) because the traffic argument (64 MB instead of 80) would
# favour it on a layer whose rows are multiples of 128 bytes.
FOLD_LAYER_NORM_BACKWARD = False
_input_is_layer_norm = False   # set by HighOrderMLP.forward around the call of such a layer
_row_stats_slot = None         # (gx, gx._version, x of the producer, stats)


def _emit_row_stats(ctx_flag: bool, x, w_t, spec: LayerSpec) -> bool:
    if not FOLD_LAYER_NORM_BACKWARD or not ctx_flag or spec.bwd_x_method != "auto" or \
            x.shape[0] == 0:
        return False
    I, _, O = w_t.shape
    return bool(_lib.load().pp_backward_input_stats_supported(
        x.shape[0], I, O, spec.n, spec.segments, int(spec.discontinuous), spec.length))


def _backward_input_with_stats(x, gy, w_t, spec: LayerSpec, packed, emit: bool):
    """backward_input_raw; with `emit` the row sums go to the slot for the node below."""
    global _row_stats_slot
    if not emit or not _emit_row_stats(True, x, w_t, spec):
        return backward_input_raw(x, gy, w_t, spec, packed=packed)
    B, I = x.shape
    stats = torch.empty(_lib.load().pp_row_stats_tiles(I), B, 2, device=x.device,
                        dtype=torch.float32)
    gx = backward_input_raw(x, gy, w_t, spec, packed=packed, row_stats=stats)
    _row_stats_slot = (gx, gx._version, x, stats)
    return gx


def _take_row_stats(gy, y):
    """The row sums left for gradient `gy` at LayerNorm output `y`, or None; empties the slot."""
    global _row_stats_slot
    slot, _row_stats_slot = _row_stats_slot, None
    if slot is None:
        return None
    gx, version, x_above, stats = slot
    if gx.data_ptr() != gy.data_ptr() or gx.shape != gy.shape or gx._version != version or \
            gy._version != version or x_above.data_ptr() != y.data_ptr() or \
            x_above.shape != y.shape or stats.device != y.device:
        return None
    return stats


def backward_weight_layer_norm_raw(x, gy, y, rstd, stats, w_shape_t, spec: LayerSpec, out=None):
    """(g, gw_t): g[B,O] = LayerNorm backward of gy at output y (row means from `stats`), gw_t
    [I,K,O] the weight gradient taken from g -- one pack pass instead of two kernels."""
    lib = _lib.load()
    B, I = x.shape
    I2, K, O = w_shape_t
    gw_t = out if out is not None else torch.empty(I2, K, O, device=x.device, dtype=torch.float32)
    g = torch.empty_like(y)
    nbytes = lib.pp_workspace_bytes(B, I, O, spec.n, spec.segments, int(spec.discontinuous),
                                    spec.length, _lib.BWD_W_FP16X2)
    ws = torch.empty(max(nbytes, 16), device=x.device, dtype=torch.uint8)
    with torch.cuda.device(x.device):
        rc = lib.pp_backward_weight_layer_norm_f32(
            _ptr(x), _ptr(gy), _ptr(y), _ptr(rstd), _ptr(stats), stats.shape[0], _ptr(g),
            _ptr(gw_t), _ptr(ws), nbytes, B, I, O, spec.n, spec.segments, int(spec.discontinuous),
            spec.length, spec.node_array(), spec.out_scale, _lib.BWD_W_FP16X2, _stream(x.device))
    _lib.check(rc, "pp_backward_weight_layer_norm_f32")
    return g, gw_t


# Gradient buffers assigned ahead of the backward (parallel.FlatGradBucket): storage pointer of a
# layer's parameter -> the [I,K,O] tensor its weight gradient is written into (a slice of one
# flat buffer that is all-reduced in place).  Empty unless a bucket registered itself.
_grad_out = {}


def _assigned_grad(w: torch.Tensor, shape_t):
    """The pre-assigned gradient buffer of parameter w, if there is one and autograd will take
    the returned tensor as w.grad (w.grad is None: an existing gradient is accumulated into,
    which must not alias the tensor being added)."""
    if not _grad_out or w.grad is not None:
        return None
    out = _grad_out.get(w.data_ptr())
    if out is None or tuple(out.shape) != tuple(shape_t) or out.device != w.device:
        return None
    return out


def backward_weight_raw(x, gy, w_shape_t, spec: LayerSpec, out=None):
    """gw_t[I,K,O] (kernel layout), dense; written into `out` when given."""
    lib = _lib.load()
    B, I = x.shape
    I2, K, O = w_shape_t
    gw_t = out if out is not None else torch.empty(I2, K, O, device=x.device, dtype=torch.float32)
    if B == 0:
        return gw_t.zero_()
    method = _lib.BWD_W_METHODS[spec.bwd_w_method]
    nbytes = lib.pp_workspace_bytes(B, I, O, spec.n, spec.segments, int(spec.discontinuous),
                                    spec.length, method)
    ws = torch.empty(max(nbytes, 16), device=x.device, dtype=torch.uint8) if nbytes else None
    with torch.cuda.device(x.device):
        rc = lib.pp_backward_weight_f32(
            _ptr(x), _ptr(gy), _ptr(gw_t), _ptr(ws), nbytes, B, I, O, spec.n, spec.segments,
            int(spec.discontinuous), spec.length, spec.node_array(), spec.out_scale, method,
            _stream(x.device))
    _lib.check(rc, "pp_backward_weight_f32")
    return gw_t


def segment_ids(x: torch.Tensor, segments: int, length: float = 2.0) -> torch.Tensor:
    """int32 segment ids on the GPU (A1 of SURVEY.md 8a)."""
    lib = _lib.load()
    _require_gpu(x, "x")
    x = x.contiguous()
    seg = torch.empty(x.shape, device=x.device, dtype=torch.int32)
    if x.numel() == 0:
        return seg
    with torch.cuda.device(x.device):
        rc = lib.pp_segment_ids_f32(_ptr(x), _ptr(seg), x.numel(), 1, segments, length,
                                    _stream(x.device))
    _lib.check(rc, "pp_segment_ids_f32")
    return seg


def use_table_form(spec: LayerSpec, B: int) -> bool:
    """The table form pays once the rows outnumber the table: building T and mapping its
    gradient back are I*P*O each, the saving is (n-1) table rows per (row, input).  Measured on
    the 16 -> 1000 layer (P = 128): 110 vs 166 us at 16384 rows, 34 vs 22 us at 256."""
    return spec.input_grid > 0 and B >= 16 * spec.input_grid


def table_spec(spec: LayerSpec) -> LayerSpec:
    """The piecewise-constant layer (n = 1, one segment per grid point) a layer with
    `input_grid` runs as; out_scale is folded into the table."""
    return make_spec(1, spec.input_grid, True, spec.length)


def grid_table(w_t, spec: LayerSpec):
    """T[I, P, O] = out_scale * sum_j l_j(xi(x_c)) * w_t[i, s(x_c)*stride + j, :] for the P grid
    points x_c of spec.input_grid (pp_grid_table_f32)."""
    lib = _lib.load()
    I, K, O = w_t.shape
    T = torch.empty(I, spec.input_grid, O, device=w_t.device, dtype=torch.float32)
    with torch.cuda.device(w_t.device):
        rc = lib.pp_grid_table_f32(_ptr(w_t), _ptr(T), I, O, spec.n, spec.segments,
                                   int(spec.discontinuous), spec.length, spec.node_array(),
                                   spec.out_scale, spec.input_grid, _stream(w_t.device))
    _lib.check(rc, "pp_grid_table_f32")
    return T


def grid_table_grad(g_table, w_shape_t, spec: LayerSpec, out=None):
    """gw_t[I, K, O] from the gradient of the table (pp_grid_table_grad_f32); into `out` if given."""
    lib = _lib.load()
    I, K, O = w_shape_t
    gw_t = out if out is not None else torch.empty(I, K, O, device=g_table.device,
                                                   dtype=torch.float32)
    with torch.cuda.device(g_table.device):
        rc = lib.pp_grid_table_grad_f32(_ptr(g_table), _ptr(gw_t), I, O, spec.n, spec.segments,
                                        int(spec.discontinuous), spec.length, spec.node_array(),
                                        spec.out_scale, spec.input_grid, _stream(g_table.device))
    _lib.check(rc, "pp_grid_table_grad_f32")
    return gw_t


def _table_weight_gradient(x, g, w, w_t, spec: LayerSpec):
    """Weight gradient of a layer in table form: one indexed add per (row, input) into the
    table's gradient, then the map back to the layer's own basis."""
    I, K, O = w_t.shape
    g_table = backward_weight_raw(x, g, (I, spec.input_grid, O), table_spec(spec))
    return grid_table_grad(g_table, (I, K, O), spec, out=_assigned_grad(w, w_t.shape))


def on_input_grid(x: torch.Tensor, points: int = 128) -> bool:
    """True when every element of x is one of the `points` grid values (a debugging aid: the
    layers do not call it; it synchronises)."""
    c = (x.double() + 1.0) * (points / 2.0) - 0.5
    return bool(((c == c.round()) & (c >= 0) & (c <= points - 1)).all().item())


class PiecewisePolyFunction(torch.autograd.Function):
    """y = piecewise_poly(x[B,I], w[O,I,K]) with hand-written HIP forward/backward."""

    @staticmethod
    def forward(ctx, x: torch.Tensor, w: torch.Tensor, spec: LayerSpec):
        if x.dim() != 2:
            raise ValueError(
                f"piecewise layer forward is 2-D only, got {tuple(x.shape)} (use reshape_apply)")
        _require_gpu(x, "x")
        _require_gpu(w, "w")
        x = x.contiguous()
        w_t = kernel_layout(w.detach())
        if use_table_form(spec, x.shape[0]):
            # inputs on a grid (the caller's statement): the layer in its table form
            packed = None
            y, _ = forward_raw(x, grid_table(w_t, spec), table_spec(spec))
        else:
            packed = pack_weights(w_t, spec, x.shape[0], version=w._version)
            fwd_packed = packed if packed is not None and \
                _fwd_takes_packed(w_t, spec, x.shape[0]) else None
            y, _ = forward_raw(x, w_t, spec, packed=fwd_packed)
        ctx.spec = spec
        # the pre-split weights serve this step's backward-input too (w cannot change in
        # between: autograd checks the saved tensor's version)
        ctx.packed = packed if ctx.needs_input_grad[0] else None
        ctx.emit_stats = _input_is_layer_norm
        ctx.save_for_backward(x, w)
        return y

    @staticmethod
    def backward(ctx, gy: torch.Tensor):
        x, w = ctx.saved_tensors
        spec = ctx.spec
        gy = gy.contiguous()
        if gy.dtype != torch.float32:
            gy = gy.float()
        w_t = kernel_layout(w.detach())
        gx = gw = None
        if ctx.needs_input_grad[1]:
            if use_table_form(spec, x.shape[0]):
                gw_t = _table_weight_gradient(x, gy, w, w_t, spec)
            else:
                gw_t = backward_weight_raw(x, gy, tuple(w_t.shape), spec,
                                           out=_assigned_grad(w, w_t.shape))
            gw = gw_t.permute(2, 0, 1)  # logical [O,I,K]; same strides as the module's parameter
        if ctx.needs_input_grad[0]:  # last: its row sums wait in the slot for the node below
            gx = _backward_input_with_stats(x, gy, w_t, spec, ctx.packed, ctx.emit_stats)
        return gx, gw, None


def piecewise_poly(x: torch.Tensor, w: torch.Tensor, spec: LayerSpec) -> torch.Tensor:
    return PiecewisePolyFunction.apply(x, w, spec)


def layer_norm_fusable(x: torch.Tensor, w: torch.Tensor, spec: LayerSpec) -> bool:
    """True when `LayerNorm(piecewise_poly(x, w))` has the fused kernel and that kernel is
    what the automatic method choice would run for the layer alone."""
    if spec.fwd_method != "auto" or spec.bwd_x_method != "auto" or x.dim() != 2 or \
            not x.is_cuda or x.shape[0] == 0:
        return False
    O, I, _ = w.shape
    if use_table_form(spec, x.shape[0]):  # the table layer's kernels decide
        spec = table_spec(spec)
    return bool(_lib.load().pp_forward_layer_norm_supported(
        x.shape[0], I, O, spec.n, spec.segments, int(spec.discontinuous), spec.length))


class PiecewisePolyLayerNormFunction(torch.autograd.Function):
    """y = LayerNorm_lastdim(piecewise_poly(x, w)), the normalisation fused into the forward
    kernel's epilogue (the block -- matrix-core kernel -- or wave -- gather kernel -- that
    produces a row normalises it before storing); backward =
    the LayerNorm backward kernel, then the layer's two backward kernels.  Saves (x, w, y, rstd):
    the un-normalised layer output never exists in memory."""

    @staticmethod
    def forward(ctx, x: torch.Tensor, w: torch.Tensor, spec: LayerSpec, eps: float):
        lib = _lib.load()
        _require_gpu(x, "x")
        _require_gpu(w, "w")
        x = x.contiguous()
        w_t = kernel_layout(w.detach())
        B, I = x.shape
        O = w_t.shape[2]
        # fs / fw: the layer the forward kernel runs -- the layer itself, or its table form
        fs, fw = spec, w_t
        tabled = use_table_form(spec, B)
        if tabled:
            fs, fw = table_spec(spec), grid_table(w_t, spec)
        mode = lib.pp_forward_layer_norm_supported(B, I, O, fs.n, fs.segments,
                                                   int(fs.discontinuous), fs.length)
        packed = None if tabled else \
            pack_weights(w_t, spec, B, version=w._version)  # (the backward-input's too)
        if mode == 0 or (mode == 1 and packed is None):
            raise _lib.PPHipError("fused layer+LayerNorm called on a layer that has no such kernel")
        fwd_packed = packed if mode == 1 else None
        if mode == 3:  # row-tile kernel: its scratch goes in through the same argument
            nbytes = lib.pp_forward_layer_norm_workspace_bytes(
                B, I, O, fs.n, fs.segments, int(fs.discontinuous), fs.length)
            fwd_packed = torch.empty(nbytes, device=x.device, dtype=torch.uint8)
        y = torch.empty(B, O, device=x.device, dtype=torch.float32)
        rstd = torch.empty(B, device=x.device, dtype=torch.float32)
        with torch.cuda.device(x.device):
            # mode 2: no packed weights -> the gather kernel with whole rows per wave;
            # mode 3: the row-tile kernel, normalisation in the kernel that sums its partials
            rc = lib.pp_forward_layer_norm_f32(
                _ptr(x), _ptr(fw), _ptr(y), _ptr(rstd), _ptr(fwd_packed),
                fwd_packed.numel() if fwd_packed is not None else 0, B, I, O,
                fs.n, fs.segments, int(fs.discontinuous), fs.length, fs.node_array(),
                fs.out_scale, eps, _stream(x.device))
        _lib.check(rc, "pp_forward_layer_norm_f32")
        ctx.spec = spec
        ctx.packed = packed if ctx.needs_input_grad[0] else None
        ctx.emit_stats = _input_is_layer_norm
        ctx.save_for_backward(x, w, y, rstd)
        return y

    @staticmethod
    def backward(ctx, gy: torch.Tensor):
        lib = _lib.load()
        x, w, y, rstd = ctx.saved_tensors
        spec = ctx.spec
        gy = gy.contiguous()
        if gy.dtype != torch.float32:
            gy = gy.float()
        B, O = y.shape
        w_t = kernel_layout(w.detach())
        gx = gw = None
        # the node above left the row sums of gy: the LayerNorm backward rides in the pass that
        # splits the gradient for the weight-gradient kernel (no LayerNorm-backward kernel)
        stats = _take_row_stats(gy, y)
        if stats is not None and ctx.needs_input_grad[1] and spec.bwd_w_method == "auto" and \
                lib.pp_backward_weight_layer_norm_supported(
                    B, x.shape[1], O, spec.n, spec.segments, int(spec.discontinuous), spec.length):
            g, gw_t = backward_weight_layer_norm_raw(x, gy, y, rstd, stats, tuple(w_t.shape), spec,
                                                     out=_assigned_grad(w, w_t.shape))
            gw = gw_t.permute(2, 0, 1)
        else:
            g = torch.empty_like(y)  # gradient w.r.t. the layer's (never stored) raw output
            with torch.cuda.device(y.device):
                rc = lib.pp_layer_norm_bwd_f32(_ptr(y), _ptr(gy), _ptr(rstd), _ptr(g), B, O,
                                               _stream(y.device))
            _lib.check(rc, "pp_layer_norm_bwd_f32")
            if ctx.needs_input_grad[1]:
                if use_table_form(spec, B):
                    gw = _table_weight_gradient(x, g, w, w_t, spec).permute(2, 0, 1)
                else:
                    gw = backward_weight_raw(x, g, tuple(w_t.shape), spec,
                                             out=_assigned_grad(w, w_t.shape)).permute(2, 0, 1)
        if ctx.needs_input_grad[0]:
            gx = _backward_input_with_stats(x, g, w_t, spec, ctx.packed, ctx.emit_stats)
        return gx, gw, None, None


def piecewise_poly_layer_norm(x, w, spec: LayerSpec, eps: float = 1e-5) -> torch.Tensor:
    return PiecewisePolyLayerNormFunction.apply(x, w, spec, eps)


def max_abs_fusable(x: torch.Tensor, w: torch.Tensor, spec: LayerSpec) -> bool:
    """True when `MaxAbsNormalizationLast(piecewise_poly(x, w))` has a fused kernel and the
    automatic method choice is in force."""
    if spec.fwd_method != "auto" or spec.bwd_x_method != "auto" or x.dim() != 2 or \
            not x.is_cuda or x.shape[0] == 0:
        return False
    O, I, _ = w.shape
    return bool(_lib.load().pp_forward_max_abs_supported(
        x.shape[0], I, O, spec.n, spec.segments, int(spec.discontinuous), spec.length))


class PiecewisePolyMaxAbsFunction(torch.autograd.Function):
    """y = MaxAbsNormalizationLast(piecewise_poly(x, w)) with the normalisation in the forward
    kernel's epilogue (gather kernel, or the fp16x2 matrix-core kernel for dense-regime layers).
    Saves (x, w, y, den, amax); the backward is the max-abs backward from the
    normalised row, then the layer's two backward kernels."""

    @staticmethod
    def forward(ctx, x: torch.Tensor, w: torch.Tensor, spec: LayerSpec, eps: float):
        lib = _lib.load()
        _require_gpu(x, "x")
        _require_gpu(w, "w")
        x = x.contiguous()
        w_t = kernel_layout(w.detach())
        B, I = x.shape
        O = w_t.shape[2]
        mode = lib.pp_forward_max_abs_supported(B, I, O, spec.n, spec.segments,
                                                int(spec.discontinuous), spec.length)
        # mode 1: the fp16x2 matrix-core kernel on the pre-split weights (the backward-input's too)
        packed = pack_weights(w_t, spec, B, version=w._version) if mode == 1 else None
        if mode == 0 or (mode == 1 and packed is None):
            raise _lib.PPHipError("fused layer+max-abs called on a layer that has no such kernel")
        y = torch.empty(B, O, device=x.device, dtype=torch.float32)
        den = torch.empty(B, device=x.device, dtype=torch.float32)
        amax = torch.empty(B, device=x.device, dtype=torch.int32)
        with torch.cuda.device(x.device):
            rc = lib.pp_forward_max_abs_ws_f32(
                _ptr(x), _ptr(w_t), _ptr(y), _ptr(den), _ptr(amax), _ptr(packed),
                packed.numel() if packed is not None else 0, B, I, O, spec.n, spec.segments,
                int(spec.discontinuous), spec.length, spec.node_array(), spec.out_scale, eps,
                _stream(x.device))
        _lib.check(rc, "pp_forward_max_abs_ws_f32")
        ctx.spec = spec
        ctx.packed = packed if ctx.needs_input_grad[0] else None
        ctx.save_for_backward(x, w, y, den, amax)
        return y

    @staticmethod
    def backward(ctx, gy: torch.Tensor):
        lib = _lib.load()
        x, w, y, den, amax = ctx.saved_tensors
        spec = ctx.spec
        gy = gy.contiguous()
        if gy.dtype != torch.float32:
            gy = gy.float()
        B, O = y.shape
        g = torch.empty_like(y)  # gradient w.r.t. the layer's (never stored) raw output
        with torch.cuda.device(y.device):
            rc = lib.pp_max_abs_norm_bwd_y_f32(_ptr(y), _ptr(gy), _ptr(den), _ptr(amax), _ptr(g),
                                               B, O, _stream(y.device))
        _lib.check(rc, "pp_max_abs_norm_bwd_y_f32")
        w_t = kernel_layout(w.detach())
        gx = gw = None
        if ctx.needs_input_grad[0]:
            gx = backward_input_raw(x, g, w_t, spec, packed=ctx.packed)
        if ctx.needs_input_grad[1]:
            gw = backward_weight_raw(x, g, tuple(w_t.shape), spec,
                                     out=_assigned_grad(w, w_t.shape)).permute(2, 0, 1)
        return gx, gw, None, None


def piecewise_poly_max_abs(x, w, spec: LayerSpec, eps: float = 1e-6) -> torch.Tensor:
    return PiecewisePolyMaxAbsFunction.apply(x, w, spec, eps)


# ---------------------------------------------------------------------------
# row-wise normalisations between the layers (HIP kernels in csrc/pp_norm.hip)
# ---------------------------------------------------------------------------
NORM_MAX_D = 1024


def _rows(t: torch.Tensor):
    D = t.shape[-1]
    return t.reshape(-1, D).contiguous(), D


class LayerNormLastFunction(torch.autograd.Function):
    """Non-affine LayerNorm over the last dim; saves (y, rstd), not x."""

    @staticmethod
    def forward(ctx, x: torch.Tensor, eps: float):
        lib = _lib.load()
        _require_gpu(x, "x")
        x2, D = _rows(x)
        B = x2.shape[0]
        y = torch.empty_like(x2)
        rstd = torch.empty(max(B, 1), device=x.device, dtype=torch.float32)
        if B > 0:
            with torch.cuda.device(x.device):
                rc = lib.pp_layer_norm_fwd_f32(_ptr(x2), _ptr(y), _ptr(rstd), B, D, eps,
                                               _stream(x.device))
            _lib.check(rc, "pp_layer_norm_fwd_f32")
        ctx.save_for_backward(y, rstd)
        ctx.shape = x.shape
        return y.view(x.shape)

    @staticmethod
    def backward(ctx, gy: torch.Tensor):
        lib = _lib.load()
        y, rstd = ctx.saved_tensors
        B, D = y.shape
        gy2 = gy.reshape(-1, D).contiguous()
        if gy2.dtype != torch.float32:
            gy2 = gy2.float()
        gx = torch.empty_like(y)
        if B > 0:
            with torch.cuda.device(y.device):
                rc = lib.pp_layer_norm_bwd_f32(_ptr(y), _ptr(gy2), _ptr(rstd), _ptr(gx), B, D,
                                               _stream(y.device))
            _lib.check(rc, "pp_layer_norm_bwd_f32")
        return gx.view(ctx.shape), None


class MaxAbsLastFunction(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x: torch.Tensor, eps: float):
        lib = _lib.load()
        _require_gpu(x, "x")
        x2, D = _rows(x)
        B = x2.shape[0]
        y = torch.empty_like(x2)
        if B > 0:
            with torch.cuda.device(x.device):
                rc = lib.pp_max_abs_norm_fwd_f32(_ptr(x2), _ptr(y), B, D, eps, _stream(x.device))
            _lib.check(rc, "pp_max_abs_norm_fwd_f32")
        ctx.save_for_backward(x2)
        ctx.eps = eps
        ctx.shape = x.shape
        return y.view(x.shape)

    @staticmethod
    def backward(ctx, gy: torch.Tensor):
        lib = _lib.load()
        (x2,) = ctx.saved_tensors
        B, D = x2.shape
        gy2 = gy.reshape(-1, D).contiguous()
        if gy2.dtype != torch.float32:
            gy2 = gy2.float()
        gx = torch.empty_like(x2)
        if B > 0:
            with torch.cuda.device(x2.device):
                rc = lib.pp_max_abs_norm_bwd_f32(_ptr(x2), _ptr(gy2), _ptr(gx), B, D, ctx.eps,
                                                 _stream(x2.device))
            _lib.check(rc, "pp_max_abs_norm_bwd_f32")
        return gx.view(ctx.shape), None


def _hip_norm_ok(x: torch.Tensor) -> bool:
    """GPU fp32 tensors with a last dim the row-in-registers kernels hold go to HIP (and fail
    loudly there if the library is missing); CPU tensors and wider rows use plain torch ops."""
    return x.is_cuda and x.dtype == torch.float32 and 1 <= x.shape[-1] <= NORM_MAX_D


def layer_norm_last(x: torch.Tensor, eps: float = 1e-5) -> torch.Tensor:
    if _hip_norm_ok(x):
        return LayerNormLastFunction.apply(x, eps)
    return torch.nn.functional.layer_norm(x, (x.shape[-1],), None, None, eps)


def max_abs_norm_last(x: torch.Tensor, eps: float = 1e-6) -> torch.Tensor:
    if _hip_norm_ok(x):
        return MaxAbsLastFunction.apply(x, eps)
    return x / (x.abs().max(dim=-1, keepdim=True)[0] + eps)


class CrossEntropyAccuracyFunction(torch.autograd.Function):
    """(mean cross entropy, arg-max accuracy) of logits [B, C] against class indices [B]: one
    kernel over the logits plus a fixed-order sum (csrc/pp_norm.hip); the backward recomputes the
    softmax.  What ASCIIPredictionNet.eval_step computes (networks.py:121-132)."""

    @staticmethod
    def forward(ctx, logits: torch.Tensor, target: torch.Tensor):
        lib = _lib.load()
        _require_gpu(logits, "logits")
        logits = logits.contiguous()
        target = target.contiguous()
        B, C = logits.shape
        out = torch.empty(2, device=logits.device, dtype=torch.float32)
        nbytes = lib.pp_cross_entropy_workspace_bytes()
        ws = torch.empty(nbytes, device=logits.device, dtype=torch.uint8)
        with torch.cuda.device(logits.device):
            rc = lib.pp_cross_entropy_fwd_f32(_ptr(logits), _ptr(target), _ptr(out), _ptr(ws), nbytes,
                                              B, C, _stream(logits.device))
        _lib.check(rc, "pp_cross_entropy_fwd_f32")
        ctx.save_for_backward(logits, target)
        loss, acc = out[0], out[1]
        ctx.mark_non_differentiable(acc)
        return loss, acc

    @staticmethod
    def backward(ctx, gloss, _gacc):
        lib = _lib.load()
        logits, target = ctx.saved_tensors
        B, C = logits.shape
        g = gloss.reshape(1).to(torch.float32).contiguous()
        glogits = torch.empty_like(logits)
        with torch.cuda.device(logits.device):
            rc = lib.pp_cross_entropy_bwd_f32(_ptr(logits), _ptr(target), _ptr(g), _ptr(glogits), B,
                                              C, _stream(logits.device))
        _lib.check(rc, "pp_cross_entropy_bwd_f32")
        return glogits, None


def cross_entropy_accuracy(logits: torch.Tensor, target: torch.Tensor):
    """(loss, accuracy) as `CrossEntropyLoss()(logits, target)` and
    `(argmax(logits, 1) == target).float().mean()`; logits [B, C] fp32, target int64 [B]."""
    if logits.is_cuda and logits.dtype == torch.float32 and logits.dim() == 2 and \
            1 <= logits.shape[1] <= NORM_MAX_D and logits.shape[0] > 0 and \
            target.dtype == torch.int64:
        return CrossEntropyAccuracyFunction.apply(logits, target)
    loss = torch.nn.functional.cross_entropy(logits, target)
    return loss, (torch.argmax(logits, dim=1) == target).float().mean()


class SelectiveScanFunction(torch.autograd.Function):
    """Mamba's selective scan (state_space_network.py:372-420 of the reference) as one HIP
    kernel per direction; the forward keeps every state x_t for the backward."""

    @staticmethod
    def forward(ctx, u, delta, A, Bm, Cm, D):
        lib = _lib.load()
        for name, t in (("u", u), ("delta", delta), ("A", A), ("B", Bm), ("C", Cm), ("D", D)):
            _require_gpu(t, name)
        u, delta, A, Bm, Cm, D = (t.contiguous() for t in (u, delta, A, Bm, Cm, D))
        b, l, d = u.shape
        n = A.shape[1]
        if delta.shape != u.shape or A.shape != (d, n) or Bm.shape != (b, l, n) or \
                Cm.shape != (b, l, n) or D.shape != (d,):
            raise ValueError("selective_scan: u, delta [b,l,d]; A [d,n]; B, C [b,l,n]; D [d]")
        y = torch.empty_like(u)
        need_grad = any(ctx.needs_input_grad)
        xhist = torch.empty(b, l, n, d, device=u.device, dtype=torch.float32) if need_grad else None
        if u.numel() > 0:
            with torch.cuda.device(u.device):
                rc = lib.pp_selective_scan_fwd_f32(_ptr(u), _ptr(delta), _ptr(A), _ptr(Bm), _ptr(Cm),
                                                   _ptr(D), _ptr(y), _ptr(xhist), b, l, d, n,
                                                   _stream(u.device))
            _lib.check(rc, "pp_selective_scan_fwd_f32")
        if need_grad:
            ctx.save_for_backward(u, delta, A, Bm, Cm, D, xhist)
        return y

    @staticmethod
    def backward(ctx, gy):
        lib = _lib.load()
        u, delta, A, Bm, Cm, D, xhist = ctx.saved_tensors
        b, l, d = u.shape
        n = A.shape[1]
        gy = gy.contiguous().float()
        gu, gdelta = torch.empty_like(u), torch.empty_like(u)
        gB, gC, gD = (torch.empty_like(t) for t in (Bm, Cm, D))
        gA_t = torch.empty(n, d, device=u.device, dtype=torch.float32)
        if u.numel() > 0:
            with torch.cuda.device(u.device):
                rc = lib.pp_selective_scan_bwd_f32(
                    _ptr(u), _ptr(delta), _ptr(A), _ptr(Bm), _ptr(Cm), _ptr(D), _ptr(gy),
                    _ptr(xhist), _ptr(gu), _ptr(gdelta), _ptr(gA_t), _ptr(gB), _ptr(gC), _ptr(gD),
                    b, l, d, n, _stream(u.device))
            _lib.check(rc, "pp_selective_scan_bwd_f32")
        return gu, gdelta, gA_t.t(), gB, gC, gD


def selective_scan(u, delta, A, Bm, Cm, D):
    """y [b,l,d] from u, delta [b,l,d], A [d,n], B, C [b,l,n], D [d] (GPU fp32 only)."""
    return SelectiveScanFunction.apply(u, delta, A, Bm, Cm, D)


# ---------------------------------------------------------------------------
# The Mamba block's pieces around the projections and the scan (csrc/pp_mamba.hip)
# ---------------------------------------------------------------------------
def _rows_view(t: torch.Tensor):
    """(tensor usable in place, rows, d, row stride): a [..., d] tensor whose last dim is
    contiguous and whose leading dims collapse to rows with one stride -- e.g. a column slice of
    a contiguous [..., 2d] tensor -- is passed as is; anything else is made contiguous."""
    d = t.shape[-1]
    if t.dim() >= 2 and t.stride(-1) == 1:
        ld = t.stride(-2)
        ok = ld >= d
        for i in range(t.dim() - 2):
            ok = ok and t.stride(i) == t.stride(i + 1) * t.shape[i + 1]
        if ok:
            return t, t.numel() // d, d, ld
    t = t.contiguous()
    return t, t.numel() // d, d, d


class CausalConvSiluFunction(torch.autograd.Function):
    """silu(causal depthwise conv1d(x) + bias) on [b, l, d] (state_space_network.py:315-324)."""

    @staticmethod
    def forward(ctx, x, weight, bias):
        lib = _lib.load()
        _require_gpu(x, "x")
        b, l, d = x.shape
        k = weight.shape[-1]
        x, _, _, ldx = _rows_view(x.float())
        w = weight.reshape(d, k).contiguous().float()
        bs = bias.contiguous().float() if bias is not None else None
        conv = torch.empty(b, l, d, device=x.device, dtype=torch.float32)
        y = torch.empty_like(conv)
        if conv.numel():
            with torch.cuda.device(x.device):
                rc = lib.pp_causal_conv_silu_fwd_f32(_ptr(x), ldx, _ptr(w), _ptr(bs), _ptr(conv),
                                                     _ptr(y), b, l, d, k, _stream(x.device))
            _lib.check(rc, "pp_causal_conv_silu_fwd_f32")
        ctx.save_for_backward(x, w, conv)
        ctx.meta = (ldx, bias is not None, tuple(weight.shape))
        return y

    @staticmethod
    def backward(ctx, gy):
        lib = _lib.load()
        x, w, conv = ctx.saved_tensors
        ldx, has_bias, wshape = ctx.meta
        b, l, d = conv.shape
        k = w.shape[1]
        gy = gy.contiguous().float()
        gx = torch.empty_like(conv)
        gw = torch.empty_like(w)
        gb = torch.empty(d, device=w.device, dtype=torch.float32) if has_bias else None
        if conv.numel():
            nbytes = lib.pp_causal_conv_silu_bwd_workspace_bytes(b, d)
            ws = torch.empty(nbytes, device=w.device, dtype=torch.uint8)
            with torch.cuda.device(w.device):
                rc = lib.pp_causal_conv_silu_bwd_f32(_ptr(x), ldx, _ptr(w), _ptr(conv), _ptr(gy),
                                                     _ptr(gx), _ptr(gw), _ptr(gb), _ptr(ws), nbytes,
                                                     b, l, d, k, _stream(w.device))
            _lib.check(rc, "pp_causal_conv_silu_bwd_f32")
        return gx, gw.reshape(wshape), gb


def causal_conv_silu(x, weight, bias=None):
    return CausalConvSiluFunction.apply(x, weight, bias)


class SiluGateFunction(torch.autograd.Function):
    """y = s * silu(r) (state_space_network.py:326)."""

    @staticmethod
    def forward(ctx, s, r):
        lib = _lib.load()
        _require_gpu(s, "s")
        s = s.contiguous().float()
        r, rows, d, ldr = _rows_view(r.float())
        y = torch.empty_like(s)
        if y.numel():
            with torch.cuda.device(s.device):
                rc = lib.pp_silu_gate_fwd_f32(_ptr(s), _ptr(r), ldr, _ptr(y), rows, d,
                                              _stream(s.device))
            _lib.check(rc, "pp_silu_gate_fwd_f32")
        ctx.save_for_backward(s, r)
        ctx.meta = (rows, d, ldr)
        return y

    @staticmethod
    def backward(ctx, gy):
        lib = _lib.load()
        s, r = ctx.saved_tensors
        rows, d, ldr = ctx.meta
        gy = gy.contiguous().float()
        gs = torch.empty_like(s)
        gr = torch.empty(r.shape, device=s.device, dtype=torch.float32)
        if gs.numel():
            with torch.cuda.device(s.device):
                rc = lib.pp_silu_gate_bwd_f32(_ptr(s), _ptr(r), ldr, _ptr(gy), _ptr(gs), _ptr(gr),
                                              rows, d, _stream(s.device))
            _lib.check(rc, "pp_silu_gate_bwd_f32")
        return gs, gr


def silu_gate(s, r):
    return SiluGateFunction.apply(s, r)


class RMSNormFunction(torch.autograd.Function):
    """x * rsqrt(mean(x^2) + eps) * weight over the last dim (state_space_network.py:423-433)."""

    @staticmethod
    def forward(ctx, x, weight, eps):
        lib = _lib.load()
        _require_gpu(x, "x")
        x2, D = _rows(x.float())
        if D > NORM_MAX_D:
            raise ValueError(f"rms_norm: last dim {D} > {NORM_MAX_D}")
        w = weight.contiguous().float()
        rows = x2.shape[0]
        y = torch.empty_like(x2)
        rstd = torch.empty(rows, device=x.device, dtype=torch.float32)
        if x2.numel():
            with torch.cuda.device(x.device):
                rc = lib.pp_rms_norm_fwd_f32(_ptr(x2), _ptr(w), _ptr(y), _ptr(rstd), rows, D, eps,
                                             _stream(x.device))
            _lib.check(rc, "pp_rms_norm_fwd_f32")
        ctx.save_for_backward(x2, w, rstd)
        ctx.shape = x.shape
        return y.reshape(x.shape)

    @staticmethod
    def backward(ctx, gy):
        lib = _lib.load()
        x2, w, rstd = ctx.saved_tensors
        rows, D = x2.shape
        gy2 = gy.reshape(rows, D).contiguous().float()
        gx = torch.empty_like(x2)
        gw = torch.empty_like(w)
        if x2.numel():
            nbytes = lib.pp_rms_norm_bwd_workspace_bytes_f32(rows, D)
            ws = torch.empty(nbytes, device=w.device, dtype=torch.uint8)
            with torch.cuda.device(w.device):
                rc = lib.pp_rms_norm_bwd_f32(_ptr(x2), _ptr(w), _ptr(rstd), _ptr(gy2), _ptr(gx),
                                             _ptr(gw), _ptr(ws), nbytes, rows, D, _stream(w.device))
            _lib.check(rc, "pp_rms_norm_bwd_f32")
        else:
            gw.zero_()
        return gx.reshape(ctx.shape), gw, None


def rms_norm(x, weight, eps: float = 1e-5):
    return RMSNormFunction.apply(x, weight, eps)
