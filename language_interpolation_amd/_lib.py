"""ctypes binding of libpphip.so (C ABI: include/pp_hip.h).

The library is built in-tree by ``language_interpolation_amd/csrc/Makefile``
(``__graft_entry__.build()`` drives it).  There is no fallback: if the shared
object is missing or a call fails, this module raises.
"""
from __future__ import annotations

import ctypes
import os
from ctypes import c_float, c_int, c_long, c_size_t, c_void_p

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libpphip.so")

PP_MAX_N = 8
NOALLOC_SCRATCH_BYTES = 8 << 20
PACK_MAX_LAYERS = 16
OPT_MAX_TENSORS = 32

# method ids (include/pp_hip.h)
METHOD_AUTO = 0
WS_PREPACKED = 0x100
FWD_GATHER, FWD_DENSE, FWD_BF16X3, FWD_FP16X2, FWD_SLAB, FWD_TILE = 1, 2, 3, 4, 5, 6
BWD_X_GATHER, BWD_X_DENSE, BWD_X_BF16X3, BWD_X_FP16X2, BWD_X_TILE = 1, 2, 3, 4, 5
BWD_W_ATOMIC, BWD_W_SORT, BWD_W_DENSE, BWD_W_BF16X3, BWD_W_FP16X2, BWD_W_REGACC = 1, 2, 3, 4, 5, 6
BWD_W_DIRECT = 7

FWD_METHODS = {"auto": 0, "gather": 1, "dense": 2, "bf16x3": 3, "fp16x2": 4, "slab": 5,
               "tile": 6}
BWD_X_METHODS = {"auto": 0, "gather": 1, "dense": 2, "bf16x3": 3, "fp16x2": 4, "tile": 5}
BWD_W_METHODS = {"auto": 0, "atomic": 1, "sort": 2, "dense": 3, "bf16x3": 4, "fp16x2": 5,
                 "regacc": 6, "direct": 7}

# every symbol include/pp_hip.h declares: name -> (restype, argtypes)
_P = c_void_p
SIGNATURES = {
    "pp_version": (c_int, []),
    "pp_error_string": (ctypes.c_char_p, [c_int]),
    "pp_auto_method": (c_int, [c_int] * 7 + [c_float]),
    "pp_segment_ids_f32": (c_int, [_P, _P, c_int, c_int, c_int, c_float, _P]),
    "pp_forward_f32": (
        c_int,
        [_P, _P, _P, _P, c_int, c_int, c_int, c_int, c_int, c_int, c_float, _P, c_float, c_int, _P],
    ),
    "pp_forward_noalloc_f32": (
        c_int,
        [_P, _P, _P, _P, c_int, c_int, c_int, c_int, c_int, c_int, c_float, _P, c_float, c_int, _P],
    ),
    "pp_forward_workspace_bytes": (c_size_t, [c_int] * 6 + [c_float, c_int]),
    "pp_forward_ws_f32": (
        c_int,
        [_P, _P, _P, _P, _P, c_size_t, c_int, c_int, c_int, c_int, c_int, c_int, c_float, _P,
         c_float, c_int, _P],
    ),
    "pp_backward_input_f32": (
        c_int,
        [_P, _P, _P, _P, c_int, c_int, c_int, c_int, c_int, c_int, c_float, _P, c_float, c_int, _P],
    ),
    "pp_packed_weights_bytes": (c_size_t, [c_int] * 5 + [c_float]),
    "pp_pack_weights_f32": (c_int, [_P, _P, c_size_t, c_int, c_int, c_int, c_int, c_int, c_float, _P]),
    "pp_pack_weights_multi_f32": (c_int, [c_int] + [_P] * 10),
    "pp_causal_conv_silu_bwd_workspace_bytes": (c_size_t, [c_int, c_int]),
    "pp_causal_conv_silu_fwd_f32": (c_int, [_P, c_int, _P, _P, _P, _P, c_int, c_int, c_int, c_int, _P]),
    "pp_causal_conv_silu_bwd_f32": (c_int, [_P, c_int, _P, _P, _P, _P, _P, _P, _P, c_size_t,
                                            c_int, c_int, c_int, c_int, _P]),
    "pp_silu_gate_fwd_f32": (c_int, [_P, _P, c_int, _P, c_long, c_int, _P]),
    "pp_silu_gate_bwd_f32": (c_int, [_P, _P, c_int, _P, _P, _P, c_long, c_int, _P]),
    "pp_rms_norm_bwd_workspace_bytes_f32": (c_size_t, [c_long, c_int]),
    "pp_rms_norm_fwd_f32": (c_int, [_P, _P, _P, _P, c_long, c_int, c_float, _P]),
    "pp_rms_norm_bwd_f32": (c_int, [_P, _P, _P, _P, _P, _P, _P, c_size_t, c_long, c_int, _P]),
    "pp_lion_step_f32": (c_int, [c_int, _P, _P, _P, _P, c_float, c_float, c_float, c_float, c_int, _P]),
    "pp_backward_input_workspace_bytes": (c_size_t, [c_int] * 6 + [c_float, c_int]),
    "pp_backward_input_ws_f32": (
        c_int,
        [_P, _P, _P, _P, _P, c_size_t, c_int, c_int, c_int, c_int, c_int, c_int, c_float, _P,
         c_float, c_int, _P],
    ),
    "pp_row_stats_tiles": (c_int, [c_int]),
    "pp_backward_input_stats_supported": (c_int, [c_int] * 6 + [c_float]),
    "pp_backward_input_stats_ws_f32": (
        c_int,
        [_P, _P, _P, _P, _P, _P, c_size_t, c_int, c_int, c_int, c_int, c_int, c_int, c_float, _P,
         c_float, c_int, _P],
    ),
    "pp_backward_weight_layer_norm_supported": (c_int, [c_int] * 6 + [c_float]),
    "pp_backward_weight_layer_norm_f32": (
        c_int,
        [_P, _P, _P, _P, _P, c_int, _P, _P, _P, c_size_t, c_int, c_int, c_int, c_int, c_int, c_int,
         c_float, _P, c_float, c_int, _P],
    ),
    "pp_grid_table_supported": (c_int, [c_int] * 5 + [c_float, c_int]),
    "pp_grid_table_f32": (c_int, [_P, _P, c_int, c_int, c_int, c_int, c_int, c_float, _P, c_float,
                                  c_int, _P]),
    "pp_grid_table_grad_f32": (c_int, [_P, _P, c_int, c_int, c_int, c_int, c_int, c_float, _P,
                                       c_float, c_int, _P]),
    "pp_workspace_bytes": (c_size_t, [c_int] * 6 + [c_float, c_int]),
    "pp_backward_weight_f32": (
        c_int,
        [_P, _P, _P, _P, c_size_t, c_int, c_int, c_int, c_int, c_int, c_int, c_float, _P, c_float,
         c_int, _P],
    ),
    "pp_forward_layer_norm_supported": (c_int, [c_int] * 6 + [c_float]),
    "pp_forward_layer_norm_workspace_bytes": (c_size_t, [c_int] * 6 + [c_float]),
    "pp_forward_layer_norm_f32": (
        c_int,
        [_P, _P, _P, _P, _P, c_size_t, c_int, c_int, c_int, c_int, c_int, c_int, c_float, _P,
         c_float, c_float, _P],
    ),
    "pp_layer_norm_fwd_f32": (c_int, [_P, _P, _P, c_int, c_int, c_float, _P]),
    "pp_layer_norm_bwd_f32": (c_int, [_P, _P, _P, _P, c_int, c_int, _P]),
    "pp_max_abs_norm_fwd_f32": (c_int, [_P, _P, c_int, c_int, c_float, _P]),
    "pp_max_abs_norm_bwd_f32": (c_int, [_P, _P, _P, c_int, c_int, c_float, _P]),
    "pp_forward_max_abs_supported": (c_int, [c_int] * 6 + [c_float]),
    "pp_forward_max_abs_f32": (c_int, [_P] * 5 + [c_int] * 6 + [c_float, _P, c_float, c_float, _P]),
    "pp_forward_max_abs_ws_f32": (c_int, [_P] * 6 + [c_size_t] + [c_int] * 6 +
                                  [c_float, _P, c_float, c_float, _P]),
    "pp_max_abs_norm_bwd_y_f32": (c_int, [_P] * 5 + [c_int, c_int, _P]),
    "pp_char_windows": (c_int, [_P, c_long, _P, c_int, c_int, _P, _P, _P, _P]),
    "pp_cross_entropy_workspace_bytes": (c_size_t, []),
    "pp_cross_entropy_fwd_f32": (c_int, [_P, _P, _P, _P, c_size_t, c_int, c_int, _P]),
    "pp_cross_entropy_bwd_f32": (c_int, [_P, _P, _P, _P, c_int, c_int, _P]),
    "pp_selective_scan_fwd_f32": (c_int, [_P] * 8 + [c_int] * 4 + [_P]),
    "pp_selective_scan_bwd_f32": (c_int, [_P] * 14 + [c_int] * 4 + [_P]),
    "pp_transpose_w_f32": (c_int, [_P, _P, c_int, c_int, c_int, _P]),
    "pp_transpose_wt_f32": (c_int, [_P, _P, c_int, c_int, c_int, _P]),
}

_lib = None


class PPHipError(RuntimeError):
    pass


def load() -> ctypes.CDLL:
    """Load libpphip.so once; raise loudly when it is not there."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise PPHipError(
            f"{LIB_PATH} is missing: build it with `make -C {os.path.join(_HERE, 'csrc')}` "
            "(or __graft_entry__.build()). There is no CPU or PyTorch fallback for this layer."
        )
    lib = ctypes.CDLL(LIB_PATH)
    for name, (restype, argtypes) in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if the export is missing
        fn.restype = restype
        fn.argtypes = argtypes
    _lib = lib
    return lib


def check(code: int, what: str) -> None:
    if code != 0:
        msg = load().pp_error_string(code)
        raise PPHipError(f"{what} failed: {msg.decode() if msg else code} (code {code})")
