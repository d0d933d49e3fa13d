"""Drop-in modules for the piecewise-polynomial layers of high-order-layers-torch.

Mirrors what the reference imports from ``high_order_layers_torch.layers``
(/root/reference/language_interpolation/networks.py:2, :14-18):
``high_order_fc_layers``, ``MaxAbsNormalizationLast``,
``MaxCenterNormalizationLast``, plus ``LazyLayerNormLastDim`` from
networks.py:72-91.  Constructor keywords, attribute names, parameter name/shape
(``w``: [out, in, K]) and the 2-D-only forward follow SURVEY.md 8(b).

The parameter is stored with LOGICAL shape [out, in, K] (what ``state_dict``,
optimizers and checkpoints of the reference see) but PHYSICAL layout [in, K, out]
-- a permuted, non-contiguous leaf -- because that is the layout the HIP kernels
read: the n active rows of a (sample, input) pair are one contiguous slab.
``load_state_dict`` copies element-wise, so reference checkpoints load unchanged.
"""
from __future__ import annotations

import math  # noqa: F401  (re-exported: networks.py:105 uses `math` via the star import)
from typing import Optional

import torch  # noqa: F401
import torch.nn as nn
from torch import Tensor

from . import functional as PF


def make_periodic(x: Tensor, periodicity: float) -> Tensor:
    """Triangle-wave fold of x into [-periodicity/2, periodicity/2] ([RECALLED];
    no reference call site passes `periodicity`, networks.py:866-882)."""
    xp = x + 0.5 * periodicity
    xp = torch.remainder(xp, 2 * periodicity)
    xp = torch.where(xp > periodicity, 2 * periodicity - xp, xp)
    return xp - 0.5 * periodicity


class _Piecewise(nn.Module):
    _discontinuous = False

    def __init__(
        self,
        n: int,
        in_features: int,
        out_features: int,
        segments: int,
        length: float = 2.0,
        weight_magnitude: float = 1.0,
        poly=None,  # accepted and ignored (upstream passes its LagrangePoly class)
        periodicity: Optional[float] = None,
        device=None,
        rescale_output: bool = False,
        scale: Optional[float] = None,  # call sites' name for `length` (networks.py:406)
        fwd_method: str = "auto",
        bwd_x_method: str = "auto",
        bwd_w_method: str = "auto",
        **kwargs,  # unknown upstream keywords are swallowed
    ):
        super().__init__()
        if scale is not None:
            length = scale
        self._n = int(n)
        self._segments = int(segments)
        self.in_features = int(in_features)
        self.out_features = int(out_features)
        self._length = float(length)
        self._half = 0.5 * float(length)
        self.periodicity = periodicity
        self.rescale_output = bool(rescale_output)
        self._spec = PF.make_spec(
            self._n, self._segments, self._discontinuous, self._length, self.rescale_output,
            self.in_features, fwd_method, bwd_x_method, bwd_w_method)
        K = PF.num_weights(self._n, self._segments, self._discontinuous)
        if device in ("gpu", "cuda") and not torch.cuda.is_available():
            raise RuntimeError("device='cuda' requested but no HIP device is visible")
        base = torch.empty(self.in_features, K, self.out_features, device=device,
                           dtype=torch.float32)
        base.uniform_(-weight_magnitude / in_features, weight_magnitude / in_features)
        # logical [O, I, K], physical [I, K, O]
        self.w = nn.Parameter(base.permute(2, 0, 1))

    @property
    def n(self) -> int:
        return self._n

    @property
    def segments(self) -> int:
        return self._segments

    def which_segment(self, x: Tensor) -> Tensor:
        return PF.segment_ids(x, self._segments, self._length).long()

    def set_methods(self, fwd: str = "auto", bwd_x: str = "auto", bwd_w: str = "auto"):
        import dataclasses
        self._spec = dataclasses.replace(PF.make_spec(
            self._n, self._segments, self._discontinuous, self._length, self.rescale_output,
            self.in_features, fwd, bwd_x, bwd_w), input_grid=self._spec.input_grid)
        return self

    def assume_input_grid(self, points: int = 128):
        """State that every input of this layer is one of `points` grid values
        (c + 1/2) * 2/points - 1 -- for points = 128 the character encoding (c - 63.5)/64 of the
        reference's datasets (single_text_dataset.py:15-16), i.e. what every first layer sees.
        Forward and weight gradient then run the layer in its table form (one table row
        gathered / one indexed add per (row, input) instead of n; csrc/pp_table.hip): same
        results up to fp32 summation order; at 16384 rows the 16 -> 1000 layer's weight gradient goes
        from 110 to 58 us, the 16 -> 250 layer's forward + LayerNorm from 35 to 26 us.  The inputs are NOT checked
        (functional.on_input_grid does, for debugging); points = 0 switches it off."""
        import dataclasses
        from . import _lib
        if points:
            if self.periodicity is not None:
                raise ValueError("assume_input_grid: a periodic layer folds its inputs off the grid")
            if not _lib.load().pp_grid_table_supported(
                    self.in_features, self.out_features, self._n, self._segments,
                    int(self._discontinuous), self._length, int(points)):
                raise ValueError(f"assume_input_grid: no table form for points={points}, "
                                 f"length={self._length} (needs a power of two <= 128, length 2)")
        self._spec = dataclasses.replace(self._spec, input_grid=int(points))
        return self

    def _prepare(self, x: Tensor) -> Tensor:
        if x.dim() != 2:
            raise ValueError(
                f"{type(self).__name__}.forward takes [batch, in_features], got {tuple(x.shape)}; "
                "wrap with reshape_apply for higher-rank inputs")
        if x.shape[1] != self.in_features:
            raise ValueError(f"expected in_features={self.in_features}, got {x.shape[1]}")
        if x.dtype != torch.float32:
            x = x.to(torch.float32)
        if self.periodicity is not None:
            x = make_periodic(x, self.periodicity)
        return x

    def forward(self, x: Tensor) -> Tensor:
        return PF.piecewise_poly(self._prepare(x), self.w, self._spec)

    def forward_layer_norm(self, x: Tensor, eps: float = 1e-5):
        """`LazyLayerNormLastDim()(self(x))` in one kernel when this layer has the fused
        epilogue (None otherwise: the caller then runs the two modules)."""
        if x.dim() != 2 or not PF.layer_norm_fusable(x, self.w, self._spec):
            return None
        return PF.piecewise_poly_layer_norm(self._prepare(x), self.w, self._spec, eps)

    def forward_max_abs(self, x: Tensor, eps: float = 1e-6):
        """`MaxAbsNormalizationLast()(self(x))` in one kernel where the layer has one (gather
        kernels, fp16x2 matrix-core kernel up to 256 outs; None otherwise)."""
        if x.dim() != 2 or not PF.max_abs_fusable(x, self.w, self._spec):
            return None
        return PF.piecewise_poly_max_abs(self._prepare(x), self.w, self._spec, eps)

    def extra_repr(self) -> str:
        kind = "discontinuous" if self._discontinuous else "continuous"
        return (f"{kind}, n={self._n}, in_features={self.in_features}, "
                f"out_features={self.out_features}, segments={self._segments}")


class PiecewisePolynomial(_Piecewise):
    """Continuous piecewise Lagrange layer: K = (n-1)*segments + 1 weights per (out, in)."""

    _discontinuous = False


class PiecewiseDiscontinuousPolynomial(_Piecewise):
    """Discontinuous piecewise Lagrange layer: K = n*segments weights per (out, in)."""

    _discontinuous = True


fc_layers = {
    "continuous": PiecewisePolynomial,
    "discontinuous": PiecewiseDiscontinuousPolynomial,
}

_OUT_OF_SCOPE = ("polynomial", "fourier", "continuous_prod", "discontinuous_prod", "polynomial_prod",
                 "switch_continuous", "switch_discontinuous")


def high_order_fc_layers(layer_type: str, **kwargs) -> nn.Module:
    """Factory with the reference's signature (networks.py:399-407, :799-806)."""
    if layer_type in fc_layers:
        return fc_layers[layer_type](**kwargs)
    if layer_type in _OUT_OF_SCOPE:
        raise ValueError(
            f"layer_type {layer_type!r} exists upstream but is outside this package's hot path "
            f"(supported: {sorted(fc_layers)})")
    raise ValueError(f"Fully connected layer type {layer_type!r} not recognized. "
                     f"Must be one of {sorted(fc_layers)}")


# ---------------------------------------------------------------------------
# normalisations that sit between the layers (networks.py:48-91)
# ---------------------------------------------------------------------------
class MaxAbsNormalizationLast(nn.Module):
    """x / (max|x| over the last dim + eps), eps = 1e-6 ([RECALLED])."""

    def __init__(self, eps: float = 1e-6):
        super().__init__()
        self._eps = eps

    def forward(self, x: Tensor) -> Tensor:
        return PF.max_abs_norm_last(x, self._eps)


class MaxCenterNormalizationLast(nn.Module):
    """(x - midrange) / (half range + eps) over the last dim ([RECALLED])."""

    def __init__(self, eps: float = 1e-6):
        super().__init__()
        self._eps = eps

    def forward(self, x: Tensor) -> Tensor:
        mx = x.max(dim=-1, keepdim=True)[0]
        mn = x.min(dim=-1, keepdim=True)[0]
        return (x - 0.5 * (mx + mn)) / (0.5 * (mx - mn) + self._eps)


class LazyLayerNormLastDim(nn.Module):
    """LayerNorm over the last dim, lazily sized, eps 1e-5 (networks.py:72-91).

    The reference creates weight=1 / bias=0 on the first forward, i.e. after
    `configure_optimizers`, and `.to(device)` turns them into plain tensors: they
    are never trained, so this is a non-affine LayerNorm.
    """

    EPS = 1e-5

    def __init__(self, bias: bool = True):
        super().__init__()
        self.weight = None
        self.bias = bias

    def forward(self, input: Tensor) -> Tensor:
        return PF.layer_norm_last(input, self.EPS)
