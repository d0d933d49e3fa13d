"""CPU oracle for the reference's Mamba variant.

TEST INFRASTRUCTURE ONLY: imported by ``tests/`` (and nothing under
``language_interpolation_amd/``).

PARITY UNPINNED.  /root/reference/language_interpolation/state_space_network.py cannot
be imported here (it needs ``high_order_layers_torch`` and, through
``language_interpolation.utils``, ``pytorch_lightning``: both absent, no network), and the
reference's tests hold no numbers for it.  This file restates what that file computes, in
plain torch, function by function:

* ``selective_scan``   state_space_network.py:372-420
* ``rms_norm``         state_space_network.py:423-434
* ``OracleMambaBlock`` state_space_network.py:201-370  (projections: ``layer_generator``
  :182-198 -> ``HighOrderMLP(hidden_layers=0, no normalization)`` = two polynomial layers)
* ``OracleResidualBlock`` :124-179, ``OracleMamba`` :39-121 (tied embedding / lm_head,
  vocabulary padded to a multiple, ``input_ids`` of shape [b, l, 1])

It runs in whatever dtype its parameters have (the tests use float64).
"""
from __future__ import annotations

import math

import torch
import torch.nn as nn
import torch.nn.functional as F

from .pp_oracle import OracleHighOrderMLP, reshape_apply

__all__ = ["selective_scan", "rms_norm", "OracleMambaBlock", "OracleResidualBlock",
           "OracleMamba", "projection"]


def selective_scan(u, delta, A, B, C, D):
    """x_t = exp(delta_t A) * x_{t-1} + delta_t B_t u_t ;  y_t = <C_t, x_t> + D u_t.

    u, delta: [b, l, d];  A: [d, n];  B, C: [b, l, n];  D: [d]  ->  y: [b, l, d]
    (zero-order hold on A, Euler on B: state_space_network.py:402-418)."""
    b, l, d = u.shape
    dA = torch.exp(delta.unsqueeze(-1) * A)                       # [b, l, d, n]
    dBu = delta.unsqueeze(-1) * B.unsqueeze(2) * u.unsqueeze(-1)  # [b, l, d, n]
    x = torch.zeros(b, d, A.shape[1], dtype=u.dtype, device=u.device)
    ys = []
    for t in range(l):
        x = dA[:, t] * x + dBu[:, t]
        ys.append((x * C[:, t].unsqueeze(1)).sum(-1))
    return torch.stack(ys, dim=1) + u * D


def rms_norm(x, weight, eps: float = 1e-5):
    return x * torch.rsqrt(x.pow(2).mean(-1, keepdim=True) + eps) * weight


def projection(layer_type, in_width, out_width, segments, n, bias, hidden_layers):
    """layer_generator (state_space_network.py:182-198)."""
    if layer_type == "linear":
        return nn.Linear(in_width, out_width, bias=bias)
    return OracleHighOrderMLP(layer_type=layer_type, in_width=in_width, out_width=out_width,
                              hidden_width=out_width, hidden_layers=hidden_layers, n=n,
                              in_segments=segments, out_segments=segments,
                              hidden_segments=segments)


class OracleMambaBlock(nn.Module):
    def __init__(self, d_inner, d_model, d_conv, d_state, dt_rank, conv_bias, bias,
                 layer_type="continuous", n=2, segments=2, hidden_layers=0):
        super().__init__()
        self.d_inner, self.d_state, self.dt_rank = d_inner, d_state, dt_rank
        kw = dict(layer_type=layer_type, n=n, segments=segments, hidden_layers=hidden_layers)
        self.in_proj = projection(in_width=d_model, out_width=2 * d_inner, bias=bias, **kw)
        self.conv1d = nn.Conv1d(d_inner, d_inner, kernel_size=d_conv, groups=d_inner,
                                padding=d_conv - 1, bias=conv_bias)
        self.x_proj = projection(in_width=d_inner, out_width=dt_rank + 2 * d_state, bias=False, **kw)
        self.dt_proj = projection(in_width=dt_rank, out_width=d_inner, bias=True, **kw)
        A = torch.arange(1, d_state + 1, dtype=torch.float32).repeat(d_inner, 1)
        self.A_log = nn.Parameter(torch.log(A))
        self.D = nn.Parameter(torch.ones(d_inner))
        self.out_proj = projection(in_width=d_inner, out_width=d_model, bias=bias, **kw)

    def ssm(self, x):
        A = -torch.exp(self.A_log)
        x_dbl = reshape_apply(x, self.x_proj)
        delta, B, C = x_dbl.split([self.dt_rank, self.d_state, self.d_state], dim=-1)
        delta = F.softplus(reshape_apply(delta, self.dt_proj))
        return selective_scan(x, delta, A, B, C, self.D)

    def forward(self, x):
        l = x.shape[1]
        x, res = reshape_apply(x, self.in_proj).split([self.d_inner, self.d_inner], dim=-1)
        x = self.conv1d(x.transpose(1, 2))[:, :, :l].transpose(1, 2)  # causal depthwise conv
        y = self.ssm(F.silu(x)) * F.silu(res)
        return reshape_apply(y, self.out_proj)


class OracleResidualBlock(nn.Module):
    def __init__(self, d_model, **kw):
        super().__init__()
        self.mixer = OracleMambaBlock(d_model=d_model, **kw)
        self.norm_weight = nn.Parameter(torch.ones(d_model))

    def forward(self, x):
        return self.mixer(rms_norm(x, self.norm_weight)) + x


class OracleMamba(nn.Module):
    def __init__(self, d_model, vocab_size, n_layer, d_state=16, d_conv=4, conv_bias=True,
                 bias=False, expand=2, dt_rank="auto", pad_vocab_size_multiple=8,
                 layer_type="linear", n=2, segments=2, hidden_layers=0):
        super().__init__()
        d_inner = int(expand * d_model)
        if dt_rank == "auto":
            dt_rank = math.ceil(d_model / 16)
        if vocab_size % pad_vocab_size_multiple:
            vocab_size += pad_vocab_size_multiple - vocab_size % pad_vocab_size_multiple
        self.embedding = nn.Embedding(vocab_size, d_model)
        self.layers = nn.ModuleList([
            OracleResidualBlock(d_model=d_model, d_inner=d_inner, d_conv=d_conv, d_state=d_state,
                                dt_rank=dt_rank, conv_bias=conv_bias, bias=bias,
                                layer_type=layer_type, n=n, segments=segments,
                                hidden_layers=hidden_layers)
            for _ in range(n_layer)])
        self.norm_f_weight = nn.Parameter(torch.ones(d_model))

    def forward(self, input_ids):
        x = self.embedding(input_ids.reshape(input_ids.shape[0], -1))
        for layer in self.layers:
            x = layer(x)
        return rms_norm(x, self.norm_f_weight) @ self.embedding.weight.t()  # tied lm_head
