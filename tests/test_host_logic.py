"""CPU: host-side mirror of the reference interface + the C-ABI library's exports
(no compute calls: there is no GPU in the CPU test tier)."""
import ctypes
import os
import re

import pytest
import torch

import language_interpolation_amd as L
from language_interpolation_amd import _lib
from language_interpolation_amd import functional as PF

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_loads_and_exports_every_declared_symbol():
    lib = _lib.load()
    header = open(os.path.join(ROOT, "include", "pp_hip.h")).read()
    declared = set(re.findall(r"\b(pp_[a-z0-9_]+)\s*\(", header))
    assert declared, "no declarations found in include/pp_hip.h"
    for name in sorted(declared):
        assert hasattr(lib, name), f"{name} declared in pp_hip.h but not exported"
    assert declared == set(_lib.SIGNATURES), (declared ^ set(_lib.SIGNATURES))
    assert lib.pp_version() >= 100
    assert b"NULL" in lib.pp_error_string(-1)


def test_argument_validation_without_a_gpu():
    lib = _lib.load()
    null = ctypes.c_void_p(0)
    # NULL pointers are rejected before anything touches the device
    assert lib.pp_forward_f32(null, null, null, null, 4, 4, 4, 2, 2, 0, 2.0, null, 1.0, 0, null) == -1
    fake = ctypes.c_void_p(4096)  # aligned, never dereferenced on the host
    bad_n = lib.pp_forward_f32(fake, fake, fake, null, 4, 4, 4, 9, 2, 0, 2.0, null, 1.0, 0, null)
    assert bad_n == -2
    cont_n1 = lib.pp_forward_f32(fake, fake, fake, null, 4, 4, 4, 1, 2, 0, 2.0, null, 1.0, 0, null)
    assert cont_n1 == -2
    misaligned = ctypes.c_void_p(4100)
    assert lib.pp_forward_f32(misaligned, fake, fake, null, 4, 4, 4, 2, 2, 0, 2.0, null, 1.0, 0, null) == -3
    assert lib.pp_forward_f32(fake, fake, fake, null, 4, 4, 4, 2, 2, 0, 2.0, null, 1.0, 77, null) == -5
    # sort path needs a workspace
    need = lib.pp_workspace_bytes(256, 16, 1000, 2, 128, 1, 2.0, _lib.BWD_W_SORT)
    assert need > 0
    rc = lib.pp_backward_weight_f32(fake, fake, fake, null, 0, 256, 16, 1000, 2, 128, 1, 2.0, null,
                                    1.0, _lib.BWD_W_SORT, null)
    assert rc == -4
    assert lib.pp_workspace_bytes(256, 16, 1000, 2, 128, 1, 2.0, _lib.BWD_W_ATOMIC) == 0


def test_auto_method_selection():
    lib = _lib.load()
    # many segments -> gather / sort
    assert lib.pp_auto_method(0, 1024, 16, 1000, 2, 128, 1, 2.0) == _lib.FWD_GATHER
    assert lib.pp_auto_method(2, 1024, 16, 1000, 4, 64, 1, 2.0) == _lib.BWD_W_SORT
    # ... up to 192 weight rows per input (or whole segments to 2-3 waves, discontinuous) and
    # n <= 3: accumulators in registers
    assert lib.pp_auto_method(2, 1000, 200, 200, 1, 150, 1, 2.0) == _lib.BWD_W_REGACC
    assert lib.pp_auto_method(2, 16384, 16, 250, 2, 128, 0, 2.0) == _lib.BWD_W_REGACC
    assert lib.pp_auto_method(2, 64, 16, 250, 2, 128, 0, 2.0) == _lib.BWD_W_ATOMIC  # tiny
    assert lib.pp_auto_method(2, 4096, 200, 200, 2, 150, 1, 2.0) == _lib.BWD_W_REGACC  # K = 300: two parts
    assert lib.pp_auto_method(2, 1024, 16, 1000, 2, 128, 1, 2.0) == _lib.BWD_W_REGACC  # cfg4 input
    assert lib.pp_auto_method(2, 256, 16, 1000, 2, 128, 1, 2.0) == _lib.BWD_W_DIRECT   # ... reference batch
    assert lib.pp_auto_method(2, 4096, 64, 64, 2, 200, 0, 2.0) == _lib.BWD_W_SORT      # K = 201 continuous
    assert lib.pp_auto_method(2, 4096, 64, 64, 1, 300, 1, 2.0) == _lib.BWD_W_SORT      # K = 300, n = 1
    assert lib.pp_auto_method(2, 4096, 64, 64, 4, 40, 1, 2.0) == _lib.BWD_W_SORT     # n = 4
    assert lib.pp_workspace_bytes(1000, 200, 200, 1, 150, 1, 2.0, _lib.BWD_W_REGACC) > 0
    assert lib.pp_workspace_bytes(1000, 16, 1000, 2, 128, 0, 2.0, _lib.BWD_W_REGACC) > 0
    assert lib.pp_workspace_bytes(1000, 16, 1000, 2, 200, 0, 2.0, _lib.BWD_W_REGACC) == 0  # K = 201 continuous
    null = ctypes.c_void_p(0)
    fake = ctypes.c_void_p(0x1000)
    assert lib.pp_backward_weight_f32(fake, fake, fake, null, 0, 1000, 16, 1000, 2, 200, 0, 2.0,
                                      null, 1.0, _lib.BWD_W_REGACC, null) == -5  # not this shape
    assert lib.pp_backward_weight_f32(fake, fake, fake, null, 0, 1000, 200, 200, 1, 150, 1, 2.0,
                                      null, 1.0, _lib.BWD_W_REGACC, null) == -4  # no workspace
    # few segments, wide layer -> expanded basis on the matrix cores (cfg2 hidden, cfg1 hidden)
    for p, want in ((0, _lib.FWD_FP16X2), (1, _lib.BWD_X_FP16X2), (2, _lib.BWD_W_DENSE)):
        assert lib.pp_auto_method(p, 512, 250, 250, 3, 2, 0, 2.0) == want
        assert lib.pp_auto_method(p, 1000, 200, 200, 2, 2, 0, 2.0) == want
    # small batches of the same layers: gather forward under 384 rows, everything under 256
    assert lib.pp_auto_method(0, 256, 250, 250, 3, 2, 0, 2.0) == _lib.FWD_GATHER
    assert lib.pp_auto_method(1, 256, 250, 250, 3, 2, 0, 2.0) == _lib.BWD_X_FP16X2
    assert lib.pp_auto_method(0, 240, 256, 256, 3, 2, 0, 2.0) == _lib.FWD_GATHER
    assert lib.pp_auto_method(1, 240, 256, 256, 3, 2, 0, 2.0) == _lib.BWD_X_GATHER
    assert lib.pp_auto_method(2, 240, 256, 256, 3, 2, 0, 2.0) == _lib.BWD_W_DENSE
    # K = 9 (n=5, S=2) is beyond the dense kernels -> gather
    assert lib.pp_auto_method(0, 35, 64, 64, 5, 2, 0, 2.0) == _lib.FWD_GATHER
    # a length that is not a power of two (the reference accepts any input_scale,
    # networks.py:386-425) keeps the layer off the matrix-core kernels
    assert lib.pp_auto_method(0, 256, 250, 250, 3, 2, 0, 3.0) == _lib.FWD_GATHER
    assert lib.pp_auto_method(1, 256, 250, 250, 3, 2, 0, 3.0) == _lib.BWD_X_GATHER
    # many segments with a derivative (n >= 2), large batch: the row-tile backward-input
    assert lib.pp_auto_method(1, 16384, 200, 200, 2, 150, 1, 2.0) == _lib.BWD_X_TILE
    assert lib.pp_auto_method(1, 2048, 200, 200, 2, 150, 1, 2.0) == _lib.BWD_X_GATHER
    assert lib.pp_backward_input_workspace_bytes(16384, 200, 200, 2, 150, 1, 2.0, _lib.BWD_X_TILE) > \
        4 * 16384 * 200 * 4
    assert lib.pp_auto_method(2, 256, 250, 250, 3, 2, 0, 3.0) == _lib.BWD_W_REGACC
    assert lib.pp_packed_weights_bytes(250, 250, 3, 2, 0, 3.0) == 0
    assert lib.pp_forward_layer_norm_supported(256, 250, 250, 3, 2, 0, 3.0) == 2  # gather-fused
    assert lib.pp_forward_layer_norm_supported(1000, 200, 200, 1, 150, 1, 2.0) == 2
    assert lib.pp_forward_layer_norm_supported(256, 16, 1000, 2, 128, 1, 2.0) == 2
    assert lib.pp_forward_layer_norm_supported(256, 16, 1030, 2, 128, 1, 2.0) == 0  # O > 1024
    assert lib.pp_forward_layer_norm_supported(256, 16, 250, 5, 30, 1, 2.0) == 0    # n > 4
    assert lib.pp_forward_layer_norm_supported(512, 250, 250, 3, 2, 0, 4.0) == 1
    assert lib.pp_forward_layer_norm_supported(256, 250, 250, 3, 2, 0, 4.0) == 2
    # fp16x2 layers with K >= 7 keep the fused LayerNorm up to 256 outs
    assert lib.pp_forward_layer_norm_supported(512, 64, 256, 4, 2, 0, 2.0) == 1
    # the max-abs epilogue: gather kernels (2) and the fp16x2 matrix-core kernel (1)
    assert lib.pp_forward_max_abs_supported(1000, 200, 200, 1, 150, 1, 2.0) == 2
    assert lib.pp_forward_max_abs_supported(256, 16, 1000, 2, 128, 1, 2.0) == 2
    assert lib.pp_forward_max_abs_supported(512, 250, 250, 3, 2, 0, 2.0) == 1   # matrix-core layer
    assert lib.pp_forward_max_abs_supported(512, 300, 300, 3, 2, 0, 2.0) == 0   # more than 256 outs
    assert lib.pp_forward_max_abs_supported(256, 16, 1030, 2, 128, 1, 2.0) == 0  # O > 1024


def test_layer_constructor_and_parameter_layout():
    layer = L.high_order_fc_layers(layer_type="continuous", n=3, in_features=7, out_features=5,
                                   segments=4, device="cpu", scale=2.0)
    assert isinstance(layer, L.PiecewisePolynomial)
    assert tuple(layer.w.shape) == (5, 7, 9)  # [out, in, (n-1)*S+1]
    assert layer.w.permute(1, 2, 0).is_contiguous()  # physical [in, K, out]
    assert layer.w.abs().max() <= 1.0 / 7
    d = L.high_order_fc_layers("discontinuous", n=2, in_features=3, out_features=4, segments=8)
    assert tuple(d.w.shape) == (4, 3, 16)
    assert d.in_features == 3 and d.out_features == 4
    with pytest.raises(ValueError):
        L.high_order_fc_layers("fourier", n=3, in_features=2, out_features=2, segments=2)
    with pytest.raises(ValueError):
        L.high_order_fc_layers("continuous", n=1, in_features=2, out_features=2, segments=2)
    with pytest.raises(ValueError):
        L.high_order_fc_layers("discontinuous", n=9, in_features=2, out_features=2, segments=2)


def test_state_dict_round_trip_with_reference_layout():
    a = L.PiecewisePolynomial(n=2, in_features=4, out_features=3, segments=5)
    sd = a.state_dict()
    assert list(sd) == ["w"] and tuple(sd["w"].shape) == (3, 4, 6)
    # a checkpoint written by the reference holds a plain contiguous [O, I, K] tensor
    ref_w = torch.randn(3, 4, 6)
    b = L.PiecewisePolynomial(n=2, in_features=4, out_features=3, segments=5)
    b.load_state_dict({"w": ref_w})
    assert torch.equal(b.w.detach(), ref_w)
    assert b.w.permute(1, 2, 0).is_contiguous()  # kernel layout survives the load
    import copy
    import pickle
    c = pickle.loads(pickle.dumps(b))
    assert torch.equal(c.w.detach(), ref_w)
    assert copy.deepcopy(b).w.permute(1, 2, 0).is_contiguous()


def test_high_order_mlp_ordering_matches_reference_tests():
    # tests/test_utils.py:9-20 of the reference
    net = L.HighOrderMLP(layer_type="continuous", n=5, in_width=10, out_width=11, hidden_layers=1,
                         hidden_width=5, in_segments=2, out_segments=2, hidden_segments=2,
                         device="cpu")
    assert [tuple(m.w.shape) for m in net.model] == [(5, 10, 9), (5, 5, 9), (11, 5, 9)]
    # tests/test_dataset_from_representation.py:67-70: "model.model.5" has the hidden width
    net = L.HighOrderMLP(layer_type="discontinuous", n=2, n_in=2, n_hidden=2, in_width=16,
                         out_width=128, hidden_layers=2, hidden_width=10, in_segments=128,
                         out_segments=8, hidden_segments=8,
                         normalization=L.LazyLayerNormLastDim, layer_type_in="continuous")
    kinds = [type(m).__name__ for m in net.model]
    assert kinds == ["PiecewisePolynomial", "LazyLayerNormLastDim",
                     "PiecewiseDiscontinuousPolynomial", "LazyLayerNormLastDim",
                     "PiecewiseDiscontinuousPolynomial", "LazyLayerNormLastDim",
                     "PiecewiseDiscontinuousPolynomial"]
    assert net.model[4].out_features == 10
    # hidden_layers=0 still gives two polynomial layers (state_space_network.py:186-196)
    proj = L.HighOrderMLP(layer_type="continuous", in_width=64, out_width=256, hidden_width=256,
                          hidden_layers=0, n=3, in_segments=2, out_segments=2, hidden_segments=2)
    assert len(proj.model) == 2


def test_cpu_tensors_fail_loudly():
    layer = L.PiecewisePolynomial(n=2, in_features=4, out_features=3, segments=2)
    with pytest.raises(RuntimeError, match="no CPU path"):
        layer(torch.rand(2, 4))
    with pytest.raises(ValueError, match="2-D|batch"):
        layer(torch.rand(2, 3, 4))


def test_nodes_match_oracle_nodes():
    from oracle import pp_oracle as po
    for n in range(1, 9):
        assert torch.equal(PF.chebyshev_lobatto(n), po.chebyshev_lobatto(n))


def test_normalizations_and_reshape_apply_on_cpu():
    from oracle import pp_oracle as po
    x = torch.randn(6, 5, 9)
    assert torch.allclose(L.MaxAbsNormalizationLast()(x), po.max_abs_normalization_last(x))
    assert torch.allclose(L.MaxCenterNormalizationLast()(x), po.max_center_normalization_last(x))
    assert torch.allclose(L.LazyLayerNormLastDim()(x), po.layer_norm_last_dim(x))
    out = L.reshape_apply(x, torch.nn.Linear(9, 4))
    assert out.shape == (6, 5, 4)
