"""`HighOrderMLP` / `LowOrderMLP` with the constructor the reference calls
(/root/reference/language_interpolation/networks.py:866-882, :812-820;
state_space_network.py:186-196; tests/test_utils.py:9-20).

Sequential ordering (pinned by
tests/test_dataset_from_representation.py:67-70 of the reference):
    [PP_in, (norm, [non_linearity], PP_hidden) * hidden_layers, norm, [non_linearity], PP_out]
`hidden_layers=0` still yields two polynomial layers (in -> hidden_width -> out).
"""
from __future__ import annotations

from typing import Callable, Optional

import torch.nn as nn
from torch import Tensor

from . import functional as PF
from .layers import LazyLayerNormLastDim, MaxAbsNormalizationLast, high_order_fc_layers


class HighOrderMLP(nn.Module):
    def __init__(
        self,
        layer_type: str,
        n: int,
        in_width: int,
        out_width: int,
        hidden_layers: int,
        hidden_width: int,
        scale: float = 2.0,
        n_in: Optional[int] = None,
        n_out: Optional[int] = None,
        n_hidden: Optional[int] = None,
        rescale_output: bool = False,
        periodicity: Optional[float] = None,
        non_linearity: Optional[nn.Module] = None,
        in_segments: Optional[int] = None,
        out_segments: Optional[int] = None,
        hidden_segments: Optional[int] = None,
        normalization: Optional[Callable[[], nn.Module]] = None,
        resnet: bool = False,
        device=None,
        layer_type_in: Optional[str] = None,
    ) -> None:
        super().__init__()
        if resnet:
            raise ValueError("resnet=True is outside the hot path (no reference call site uses it)")
        n_in = n_in or n
        n_hidden = n_hidden or n
        n_out = n_out or n
        common = dict(rescale_output=rescale_output, scale=scale, periodicity=periodicity,
                      device=device)
        layer_list = [
            high_order_fc_layers(layer_type=layer_type_in or layer_type, n=n_in,
                                 in_features=in_width, out_features=hidden_width,
                                 segments=in_segments, **common)
        ]
        for _ in range(hidden_layers):
            if normalization is not None:
                layer_list.append(normalization())
            if non_linearity is not None:
                layer_list.append(non_linearity)
            layer_list.append(
                high_order_fc_layers(layer_type=layer_type, n=n_hidden, in_features=hidden_width,
                                     out_features=hidden_width, segments=hidden_segments, **common))
        if normalization is not None:
            layer_list.append(normalization())
        if non_linearity is not None:
            layer_list.append(non_linearity)
        layer_list.append(
            high_order_fc_layers(layer_type=layer_type, n=n_out, in_features=hidden_width,
                                 out_features=out_width, segments=out_segments, **common))
        self.model = nn.Sequential(*layer_list)

    def forward(self, x: Tensor) -> Tensor:
        # same result as self.model(x); a polynomial layer directly followed by
        # LazyLayerNormLastDim (or MaxAbsNormalizationLast) runs as one kernel where the layer
        # has the fused epilogue
        mods = list(self.model)
        PF._prepacked.clear()
        if x.is_cuda and x.dim() == 2:
            # the weights of all matrix-core layers are pre-split for this step in two launches
            PF.prepack_layers([m for m in mods if hasattr(m, "forward_layer_norm")], x.shape[0])
        i = 0
        # after_ln: x is the output of a fused layer+LayerNorm node and feeds the next polynomial
        # layer directly -- that layer's backward-input then leaves the row sums with which the
        # LayerNorm backward rides in the weight-gradient pass of the layer below (PF._row_stats_slot)
        after_ln = False
        try:
            while i < len(mods):
                m = mods[i]
                PF._input_is_layer_norm = after_ln and hasattr(m, "forward_layer_norm")
                after_ln = False
                if i + 1 < len(mods) and isinstance(mods[i + 1], LazyLayerNormLastDim) and \
                        hasattr(m, "forward_layer_norm"):
                    fused = m.forward_layer_norm(x, LazyLayerNormLastDim.EPS)
                    if fused is not None:
                        x = fused
                        after_ln = True
                        i += 2
                        continue
                if i + 1 < len(mods) and isinstance(mods[i + 1], MaxAbsNormalizationLast) and \
                        hasattr(m, "forward_max_abs"):
                    fused = m.forward_max_abs(x, mods[i + 1]._eps)
                    if fused is not None:
                        x = fused
                        i += 2
                        continue
                x = m(x)
                i += 1
        finally:
            PF._input_is_layer_norm = False
        # the pre-split buffers belong to this step only (each layer's autograd node holds its
        # own reference for the backward): a later call of self.model / a sub-layer must not find
        # them after an optimizer that writes the weights through raw pointers or .data
        PF._prepacked.clear()
        return x


class LowOrderMLP(nn.Module):
    """Linear(+norm)+non-linearity stack used below the polynomial input layer of
    `net=large_standard` (networks.py:812-820).  GEMM territory: plain
    `torch.nn.Linear` (rocBLAS/hipBLASLt), deliberately not re-implemented."""

    def __init__(
        self,
        in_width: int,
        out_width: int,
        hidden_layers: int,
        hidden_width: int,
        non_linearity: Optional[nn.Module] = None,
        normalization: Optional[Callable[[], nn.Module]] = None,
        device=None,
    ) -> None:
        super().__init__()
        layer_list = [nn.Linear(in_width, hidden_width, device=device)]
        for _ in range(hidden_layers):
            if normalization is not None:
                layer_list.append(normalization())
            if non_linearity is not None:
                layer_list.append(non_linearity)
            layer_list.append(nn.Linear(hidden_width, hidden_width, device=device))
        if normalization is not None:
            layer_list.append(normalization())
        if non_linearity is not None:
            layer_list.append(non_linearity)
        layer_list.append(nn.Linear(hidden_width, out_width, device=device))
        self.model = nn.Sequential(*layer_list)

    def forward(self, x: Tensor) -> Tensor:
        return self.model(x)
