"""MI355X-native piecewise-polynomial layers (drop-in for the high-order layers
that jloveric/language-interpolation builds its networks from)."""
from .layers import (  # noqa: F401
    LazyLayerNormLastDim,
    MaxAbsNormalizationLast,
    MaxCenterNormalizationLast,
    PiecewiseDiscontinuousPolynomial,
    PiecewisePolynomial,
    fc_layers,
    high_order_fc_layers,
)
from .mamba import Mamba  # noqa: F401
from .networks import HighOrderMLP, LowOrderMLP  # noqa: F401
from .utils import reshape_apply  # noqa: F401

__all__ = [
    "PiecewisePolynomial",
    "PiecewiseDiscontinuousPolynomial",
    "high_order_fc_layers",
    "fc_layers",
    "HighOrderMLP",
    "LowOrderMLP",
    "Mamba",
    "MaxAbsNormalizationLast",
    "MaxCenterNormalizationLast",
    "LazyLayerNormLastDim",
    "reshape_apply",
]
