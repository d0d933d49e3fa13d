"""CPU oracle for the piecewise-polynomial fully-connected layer.

TEST INFRASTRUCTURE ONLY.  Nothing under ``language_interpolation_amd/`` may
import this file; only ``tests/``, ``__graft_entry__.smoke()`` and the
``cpu_baseline`` leg of ``bench.py`` use it, and there only as the checker /
the CPU baseline -- never as the thing shipped.

PARITY UNPINNED.  The arithmetic of this path lives in the third-party package
``high-order-layers-torch ^2.6.0`` (/root/reference/pyproject.toml:21; no lock
file, so the caret range is the only pin).  That package is not installed in
the build container, there is no network, and the reference's own tests hold
shapes only (/root/reference/tests/test_utils.py:25,
/root/reference/tests/test_dataset_from_representation.py:69-70).  This file is
therefore a restatement of the package's *published* algorithm, anchored on the
reference's call sites:

* ``high_order_fc_layers(layer_type, n, in_features, out_features, segments,
  device[, scale])``  /root/reference/language_interpolation/networks.py:399-407,
  :799-806
* ``HighOrderMLP(...)``  /root/reference/language_interpolation/networks.py:866-882,
  /root/reference/language_interpolation/state_space_network.py:186-196,
  /root/reference/tests/test_utils.py:9-20
* normalisation classes  /root/reference/language_interpolation/networks.py:48-91
* ``reshape_apply``  /root/reference/language_interpolation/utils.py:21-37

and on the spec rows A1-A8 of SURVEY.md section 8(a).  The known-answer tests
KA1-KA4 / P1-P8 of SURVEY.md section 8(c) (tests/test_oracle.py) are what pins
this file, in place of the missing reference goldens.

Formulation notes (what "the reference formulation" means here):

A1  segment id   t = ((x + L/2) / L) * S  in fp32, in that order;
                 s = clamp(trunc(t), 0, S-1)           (``.long()`` + two wheres)
A2  local coord  x_min = (s/S)*2 - 1, x_max = ((s+1)/S)*2 - 1,
                 xi = L*((x - x_min)/(x_max - x_min)) - L/2      (not clamped)
A3  basis        Chebyshev-Lobatto nodes X_k = -cos(pi k/(n-1)) * L/2 (fp32),
                 l_j(xi) = prod_{m != j} (xi - X_m)/(X_j - X_m)
A4  contraction  Wg[b,i,o,j] = w[o,i,s*stride+j] (materialised gather) and
                 y = einsum("ijk,jkli->jl", basis[n,B,I], Wg[B,I,O,n])
    stride = n-1 (continuous, K=(n-1)S+1) or n (discontinuous, K=nS).
"""
from __future__ import annotations

import math
from typing import Callable, Optional

import torch
import torch.nn as nn
import torch.nn.functional as F

__all__ = [
    "chebyshev_lobatto",
    "num_weights",
    "weight_stride",
    "segment_ids",
    "local_coordinate",
    "lagrange_basis",
    "piecewise_forward",
    "piecewise_forward_dense",
    "piecewise_forward_f64",
    "piecewise_fwd_bwd",
    "make_periodic",
    "max_abs_normalization_last",
    "max_center_normalization_last",
    "layer_norm_last_dim",
    "OraclePiecewise",
    "OracleHighOrderMLP",
    "OracleLazyLayerNormLastDim",
    "OracleMaxAbsNormalizationLast",
    "OracleMaxCenterNormalizationLast",
    "oracle_fc_layer",
    "reshape_apply",
    "ascii_to_float",
]


# --------------------------------------------------------------------------
# A3: nodes
# --------------------------------------------------------------------------
def chebyshev_lobatto(n: int, length: float = 2.0) -> torch.Tensor:
    """Nodes X_k = -cos(pi k/(n-1)) * length/2 as an fp32 tensor.

    Computed in torch fp32 the way the upstream package does
    (``-torch.cos(k * math.pi / (n - 1))`` on an ``arange``), so the middle node
    of an odd n is ~4.4e-8 rather than 0.  n == 1 is special-cased to [0].
    """
    if n == 1:
        return torch.zeros(1, dtype=torch.float32)
    k = torch.arange(0, n)
    ans = -torch.cos(k * math.pi / (n - 1))
    ans = torch.where(torch.abs(ans) < 1e-15, 0 * ans, ans)
    return ((length / 2.0) * ans).to(torch.float32)


def weight_stride(n: int, discontinuous: bool) -> int:
    return n if discontinuous else n - 1


def num_weights(n: int, segments: int, discontinuous: bool) -> int:
    """K of SURVEY 8a: n*S (discontinuous) or (n-1)*S+1 (continuous)."""
    return n * segments if discontinuous else (n - 1) * segments + 1


# --------------------------------------------------------------------------
# optional periodic wrap [RECALLED]; not used by any call site in the reference
# (select_network never passes it, networks.py:866-882).
# --------------------------------------------------------------------------
def make_periodic(x: torch.Tensor, periodicity: float) -> torch.Tensor:
    xp = x + 0.5 * periodicity
    xp = torch.remainder(xp, 2 * periodicity)
    xp = torch.where(xp > periodicity, 2 * periodicity - xp, xp)
    return xp - 0.5 * periodicity


# --------------------------------------------------------------------------
# A1 / A2
# --------------------------------------------------------------------------
def segment_ids(x: torch.Tensor, segments: int, length: float = 2.0) -> torch.Tensor:
    """int64 segment index, computed in fp32 in the reference's op order (A1)."""
    x32 = x.detach().to(torch.float32)
    half = 0.5 * length
    id_min = (((x32 + half) / length) * segments).long()
    id_min = torch.where(id_min <= segments - 1, id_min, torch.tensor(segments - 1))
    id_min = torch.where(id_min >= 0, id_min, torch.tensor(0))
    return id_min


def _eta(index: torch.Tensor, segments: int, dtype: torch.dtype) -> torch.Tensor:
    eta = index.to(dtype) / float(segments)
    return eta * 2 - 1


def local_coordinate(
    x: torch.Tensor, s: torch.Tensor, segments: int, length: float = 2.0
) -> torch.Tensor:
    """xi in the dtype of x (A2).  The `*2-1` is hard-coded upstream, so
    x_max - x_min == 2/S whatever `length` is."""
    x_min = _eta(s, segments, x.dtype)
    x_max = _eta(s + 1, segments, x.dtype)
    return length * ((x - x_min) / (x_max - x_min)) - 0.5 * length


def lagrange_basis(xi: torch.Tensor, nodes: torch.Tensor) -> torch.Tensor:
    """[n, *xi.shape] stack of l_j(xi), product taken in the upstream order."""
    n = nodes.numel()
    X = nodes.to(xi.dtype)
    if n == 1:
        return torch.ones_like(xi).unsqueeze(0)
    basis = []
    for j in range(n):
        b = [(xi - X[m]) / (X[j] - X[m]) for m in range(n) if m != j]
        basis.append(torch.prod(torch.stack(b), dim=0))
    return torch.stack(basis)


# --------------------------------------------------------------------------
# A4: the reference formulation (gather + einsum)
# --------------------------------------------------------------------------
def piecewise_forward(
    x: torch.Tensor,
    w: torch.Tensor,
    n: int,
    segments: int,
    discontinuous: bool,
    length: float = 2.0,
    rescale_output: bool = False,
    periodicity: Optional[float] = None,
) -> torch.Tensor:
    """y[B,O] for x[B,I], w[O,I,K]; same dtype as x/w.  Materialises the
    [B,I,O,n] gathered weights exactly like the upstream layer does, which is
    what makes this the "reference CPU path" that bench.py times."""
    if x.dim() != 2:
        raise ValueError("piecewise layer forward is 2-D only (utils.py:21-37)")
    out_features, in_features, K = w.shape
    assert K == num_weights(n, segments, discontinuous), (K, n, segments)
    if periodicity is not None:
        x = make_periodic(x, periodicity)
    stride = weight_stride(n, discontinuous)
    s = segment_ids(x, segments, length)  # [B,I] int64, no grad
    B = x.shape[0]
    wid_min = s * stride
    wrange = wid_min.unsqueeze(-1) + torch.arange(n).view(1, 1, n)  # [B,I,n]
    windex = torch.arange(in_features).view(1, in_features, 1).expand(B, in_features, n)
    wg = w[:, windex.reshape(-1), wrange.reshape(-1)]  # [O, B*I*n]
    wg = wg.view(out_features, B, in_features, n).permute(1, 2, 0, 3)  # [B,I,O,n]
    xi = local_coordinate(x, s, segments, length)
    basis = lagrange_basis(xi, chebyshev_lobatto(n, length))  # [n,B,I]
    y = torch.einsum("ijk,jkli->jl", basis, wg)
    if rescale_output:
        y = y * (1.0 / in_features)
    return y


def piecewise_forward_dense(
    x: torch.Tensor,
    w: torch.Tensor,
    n: int,
    segments: int,
    discontinuous: bool,
    length: float = 2.0,
    rescale_output: bool = False,
) -> torch.Tensor:
    """Same function as `piecewise_forward`, written as Phi[B, I*K] @ W^T (the
    "fairer CPU baseline" of BASELINE.md section 2)."""
    out_features, in_features, K = w.shape
    stride = weight_stride(n, discontinuous)
    s = segment_ids(x, segments, length)
    xi = local_coordinate(x, s, segments, length)
    basis = lagrange_basis(xi, chebyshev_lobatto(n, length))  # [n,B,I]
    B = x.shape[0]
    phi = x.new_zeros(B, in_features, K)
    idx = (s * stride).unsqueeze(-1) + torch.arange(n).view(1, 1, n)
    phi = phi.scatter_add(2, idx, basis.permute(1, 2, 0))
    y = phi.reshape(B, in_features * K) @ w.reshape(out_features, in_features * K).t()
    if rescale_output:
        y = y * (1.0 / in_features)
    return y


def piecewise_forward_f64(
    x32: torch.Tensor,
    w: torch.Tensor,
    n: int,
    segments: int,
    discontinuous: bool,
    length: float = 2.0,
    rescale_output: bool = False,
) -> torch.Tensor:
    """The judge of accumulation error: the *same function* as the fp32
    reference, evaluated in fp64 from the point where rounding stops being part
    of the function's definition.

    * segment ids come from the fp32 computation (A1) -- they are part of the
      function (bit-exact requirement);
    * the VALUE of xi is the fp32 reference value (A2 in fp32, cast up): for a
      segment count that is not a power of two the fp32 x_min/x_max carry a
      rounding of ~S*1e-7 in xi that every fp32 implementation of the
      reference shares; its GRADIENT is the exact L*S/2;
    * nodes are the fp32 node values cast up (A3);
    * basis products and the contraction run in fp64.

    `x32` must be an fp32 tensor (may require grad); `w` may be fp32 or fp64.
    """
    assert x32.dtype == torch.float32
    out_features, in_features, K = w.shape
    stride = weight_stride(n, discontinuous)
    s = segment_ids(x32, segments, length)
    xi32 = local_coordinate(x32.detach(), s, segments, length)
    x64 = x32.to(torch.float64)
    xi_exact = local_coordinate(x64, s, segments, length)
    xi = xi_exact + (xi32.to(torch.float64) - xi_exact).detach()
    basis = lagrange_basis(xi, chebyshev_lobatto(n, length))  # [n,B,I] fp64
    B = x32.shape[0]
    w64 = w.to(torch.float64)
    idx = (s * stride).unsqueeze(-1) + torch.arange(n).view(1, 1, n)  # [B,I,n]
    phi = torch.zeros(B, in_features, K, dtype=torch.float64)
    phi = phi.scatter_add(2, idx, basis.permute(1, 2, 0))
    y = phi.reshape(B, in_features * K) @ w64.reshape(out_features, in_features * K).t()
    if rescale_output:
        y = y * (1.0 / in_features)
    return y


def piecewise_fwd_bwd(
    x: torch.Tensor,
    w: torch.Tensor,
    gy: torch.Tensor,
    n: int,
    segments: int,
    discontinuous: bool,
    length: float = 2.0,
    rescale_output: bool = False,
    precision: str = "f64",
) -> dict:
    """Forward + autograd backward.  precision: "f32" (reference formulation in
    fp32), "f32dense", or "f64" (`piecewise_forward_f64`).  Returns
    dict(y, seg, gx, gw) with gw in the reference layout [O,I,K]."""
    x32 = x.detach().to(torch.float32).clone().requires_grad_(True)
    if precision == "f64":
        wq = w.detach().to(torch.float64).clone().requires_grad_(True)
        y = piecewise_forward_f64(x32, wq, n, segments, discontinuous, length, rescale_output)
        y.backward(gy.to(torch.float64))
    else:
        wq = w.detach().to(torch.float32).clone().requires_grad_(True)
        fn = piecewise_forward if precision == "f32" else piecewise_forward_dense
        y = fn(x32, wq, n, segments, discontinuous, length, rescale_output)
        y.backward(gy.to(torch.float32))
    gx = x32.grad if x32.grad is not None else torch.zeros_like(x32)
    return {
        "y": y.detach(),
        "seg": segment_ids(x32, segments, length),
        "gx": gx.detach(),
        "gw": wq.grad.detach(),
    }


# --------------------------------------------------------------------------
# A6 / A7 normalisations
# --------------------------------------------------------------------------
def max_abs_normalization_last(x: torch.Tensor, eps: float = 1e-6) -> torch.Tensor:
    """x / (max|x| over the last dim + eps)  ([RECALLED])."""
    return x / (x.abs().max(dim=-1, keepdim=True)[0] + eps)


def max_center_normalization_last(x: torch.Tensor, eps: float = 1e-6) -> torch.Tensor:
    """Mid-range centred variant [RECALLED]: (x - (max+min)/2) / ((max-min)/2 + eps)."""
    mx = x.max(dim=-1, keepdim=True)[0]
    mn = x.min(dim=-1, keepdim=True)[0]
    return (x - 0.5 * (mx + mn)) / (0.5 * (mx - mn) + eps)


def layer_norm_last_dim(x: torch.Tensor, eps: float = 1e-5) -> torch.Tensor:
    """Non-affine LayerNorm over the last dim: what LazyLayerNormLastDim
    (networks.py:72-91) computes, its weight/bias being created after the
    optimizer and therefore frozen at 1/0."""
    return F.layer_norm(x, (x.shape[-1],), None, None, eps)


class OracleLazyLayerNormLastDim(nn.Module):
    def __init__(self, bias: bool = True):
        super().__init__()

    def forward(self, x):
        return layer_norm_last_dim(x)


class OracleMaxAbsNormalizationLast(nn.Module):
    def __init__(self, eps: float = 1e-6):
        super().__init__()
        self.eps = eps

    def forward(self, x):
        return max_abs_normalization_last(x, self.eps)


class OracleMaxCenterNormalizationLast(nn.Module):
    def __init__(self, eps: float = 1e-6):
        super().__init__()
        self.eps = eps

    def forward(self, x):
        return max_center_normalization_last(x, self.eps)


# --------------------------------------------------------------------------
# Modules mirroring the reference-visible API (A5, 8b)
# --------------------------------------------------------------------------
class OraclePiecewise(nn.Module):
    """`w: [out, in, K]`, uniform(-m/in, m/in) init (U3), 2-D forward."""

    def __init__(
        self,
        n: int,
        in_features: int,
        out_features: int,
        segments: int,
        discontinuous: bool,
        length: float = 2.0,
        weight_magnitude: float = 1.0,
        periodicity: Optional[float] = None,
        rescale_output: bool = False,
        dtype: torch.dtype = torch.float32,
        formulation: str = "gather",
        **_ignored,
    ):
        super().__init__()
        self.n, self.segments, self.discontinuous = n, segments, discontinuous
        self.in_features, self.out_features = in_features, out_features
        self.length, self.periodicity, self.rescale_output = length, periodicity, rescale_output
        self.formulation = formulation
        K = num_weights(n, segments, discontinuous)
        w = torch.empty(out_features, in_features, K, dtype=dtype)
        w.uniform_(-weight_magnitude / in_features, weight_magnitude / in_features)
        self.w = nn.Parameter(w)

    def forward(self, x):
        if self.formulation == "dense":
            return piecewise_forward_dense(
                x, self.w, self.n, self.segments, self.discontinuous, self.length, self.rescale_output
            )
        return piecewise_forward(
            x, self.w, self.n, self.segments, self.discontinuous, self.length,
            self.rescale_output, self.periodicity,
        )


def oracle_fc_layer(layer_type: str, **kwargs) -> nn.Module:
    if "scale" in kwargs:  # `scale` is the call sites' name for `length` (networks.py:406)
        kwargs["length"] = kwargs.pop("scale")
    kwargs.pop("device", None)
    if layer_type == "continuous":
        return OraclePiecewise(discontinuous=False, **kwargs)
    if layer_type == "discontinuous":
        return OraclePiecewise(discontinuous=True, **kwargs)
    raise ValueError(f"layer_type {layer_type!r} is outside the hot path")


class OracleHighOrderMLP(nn.Module):
    """Sequential[PP_in, (norm, [nl], PP_hidden)*hidden_layers, norm, [nl], PP_out]
    (ordering pinned by tests/test_dataset_from_representation.py:67-70)."""

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
        device=None,
        layer_type_in: Optional[str] = None,
        formulation: str = "gather",
    ):
        super().__init__()
        n_in, n_hidden, n_out = n_in or n, n_hidden or n, n_out or n
        common = dict(length=scale, rescale_output=rescale_output, periodicity=periodicity,
                      formulation=formulation)
        layers = [
            oracle_fc_layer(layer_type_in or layer_type, n=n_in, in_features=in_width,
                            out_features=hidden_width, segments=in_segments, **common)
        ]
        for _ in range(hidden_layers):
            if normalization is not None:
                layers.append(normalization())
            if non_linearity is not None:
                layers.append(non_linearity)
            layers.append(
                oracle_fc_layer(layer_type, n=n_hidden, in_features=hidden_width,
                                out_features=hidden_width, segments=hidden_segments, **common)
            )
        if normalization is not None:
            layers.append(normalization())
        if non_linearity is not None:
            layers.append(non_linearity)
        layers.append(
            oracle_fc_layer(layer_type, n=n_out, in_features=hidden_width,
                            out_features=out_width, segments=out_segments, **common)
        )
        self.model = nn.Sequential(*layers)

    def forward(self, x):
        return self.model(x)


def reshape_apply(x: torch.Tensor, layer) -> torch.Tensor:
    """Flatten leading dims, call the 2-D-only layer, unflatten (utils.py:21-37)."""
    shape = x.shape
    first = int(math.prod(shape[:-1]))
    return layer(x.reshape(first, shape[-1])).view(*shape[:-1], -1)


def ascii_to_float(c: torch.Tensor) -> torch.Tensor:
    """(c - 64 + 0.5)/64  (single_text_dataset.py:15-16)."""
    return (c.to(torch.float32) - 64 + 0.5) / 64.0
