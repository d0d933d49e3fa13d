"""The reference's Mamba variant on the HIP layers.

Mirror of /root/reference/language_interpolation/state_space_network.py: same constructor
arguments, module / parameter names (``embedding``, ``layers.<i>.mixer.{in_proj,conv1d,x_proj,
dt_proj,A_log,D,out_proj}``, ``layers.<i>.norm.weight``, ``norm_f.weight``, tied ``lm_head``)
and tensor shapes.  The four projections of a block are ``HighOrderMLP`` (two piecewise-
polynomial layers each: the hot path) unless ``layer_type == "linear"``; the time loop of
``selective_scan`` (:372-420) is one HIP kernel per direction (csrc/pp_scan.hip); the causal
depthwise conv + SiLU, the gate ``y * silu(res)`` and RMSNorm are one HIP kernel per direction
each (csrc/pp_mamba.hip).  Embedding, softplus and lm_head stay torch ops; the weight gradients of
the (tied) embedding and head are computed as split-K batched products (_tn_matmul).
"""
from __future__ import annotations

import math
from typing import Union

import torch
import torch.nn as nn
import torch.nn.functional as F

from . import functional as PF
from .networks import HighOrderMLP
from .utils import reshape_apply

__all__ = ["Mamba", "ResidualBlock", "MambaBlock", "RMSNorm", "layer_generator", "reshape_apply"]


def layer_generator(layer_type, in_width, out_width, segments, n, bias, hidden_layers):
    """state_space_network.py:182-198."""
    if layer_type == "linear":
        return nn.Linear(in_width, out_width, bias=bias)
    return HighOrderMLP(layer_type=layer_type, in_width=in_width, out_width=out_width,
                        hidden_width=out_width, hidden_layers=hidden_layers, n=n,
                        in_segments=segments, out_segments=segments, hidden_segments=segments)


def causal_depthwise_conv(x: torch.Tensor, weight: torch.Tensor, bias) -> torch.Tensor:
    """``conv1d(x^T)[:, :, :l]^T`` of the reference (:315-322: depthwise, padding k-1, truncated
    to l, i.e. causal) on [b, l, d] directly, as k shifted multiply-adds.  MIOpen runs this
    depthwise shape as one im2col + GEMM per batch row (16384 launches per step at b=1024)."""
    k = weight.shape[-1]
    l = x.shape[1]
    xp = F.pad(x, (0, 0, k - 1, 0))
    w = weight[:, 0, :]  # [d, k]
    y = xp[:, 0:l] * w[:, 0]
    for j in range(1, k):
        y = y + xp[:, j:j + l] * w[:, j]
    return y if bias is None else y + bias


def _tn_matmul(a: torch.Tensor, b: torch.Tensor, chunk: int = 256) -> torch.Tensor:
    """a^T @ b for a [M, p], b [M, q] with M >> p, q: the rows are cut into chunks, one batched
    matmul gives a [p, q] product per chunk and the chunks are added.  (rocBLAS runs the plain
    [p, M] x [M, q] product of the tied head's weight gradient -- 128 x 15360 x 64 at b = 1024 -- as
    a handful of tiles walking all M rows: 109 us; this way 10.)"""
    M = a.shape[0]
    if M <= 2 * chunk:
        return a.t() @ b
    pad = (-M) % chunk
    if pad:
        a = F.pad(a, (0, 0, 0, pad))
        b = F.pad(b, (0, 0, 0, pad))
    c = a.shape[0] // chunk
    return torch.bmm(a.view(c, chunk, -1).transpose(1, 2), b.view(c, chunk, -1)).sum(0)


class _HeadFunction(torch.autograd.Function):
    """logits = x @ W^T (the tied lm_head, state_space_network.py:90-93, :118) with the weight gradient
    as a split-K product."""

    @staticmethod
    def forward(ctx, x, w):
        ctx.save_for_backward(x, w)
        return x @ w.t()

    @staticmethod
    def backward(ctx, g):
        x, w = ctx.saved_tensors
        x2, g2 = x.reshape(-1, x.shape[-1]), g.reshape(-1, g.shape[-1])
        gx = (g2 @ w).view(x.shape) if ctx.needs_input_grad[0] else None
        gw = _tn_matmul(g2, x2) if ctx.needs_input_grad[1] else None
        return gx, gw


class _EmbeddingFunction(torch.autograd.Function):
    """embedding(ids) with the weight gradient as one_hot(ids)^T @ g through the same split-K
    product (torch's embedding backward sorts the indices and scatters: 80 us at b = 1024 for a
    128-row table; deterministic either way)."""

    @staticmethod
    def forward(ctx, ids, w):
        ctx.save_for_backward(ids)
        ctx.rows = w.shape[0]
        return F.embedding(ids, w)

    @staticmethod
    def backward(ctx, g):
        (ids,) = ctx.saved_tensors
        onehot = F.one_hot(ids.reshape(-1), ctx.rows).to(g.dtype)
        return None, _tn_matmul(onehot, g.reshape(-1, g.shape[-1]))


class RMSNorm(nn.Module):
    def __init__(self, d_model: int, eps: float = 1e-5):
        super().__init__()
        self.eps = eps
        self.weight = nn.Parameter(torch.ones(d_model))

    def forward(self, x):
        return PF.rms_norm(x, self.weight, self.eps)


class MambaBlock(nn.Module):
    def __init__(self, d_inner: int, d_model: int, d_conv: int, d_state: int, dt_rank: int,
                 conv_bias: bool, bias: bool, layer_type: str = "continuous", n: int = 2,
                 segments: int = 2, hidden_layers: int = 0):
        super().__init__()
        self.d_inner, self.d_model, self.d_conv = d_inner, d_model, d_conv
        self.d_state, self.dt_rank, self.conv_bias, self.bias = d_state, dt_rank, conv_bias, bias
        kw = dict(layer_type=layer_type, n=n, segments=segments, hidden_layers=hidden_layers)
        self.in_proj = layer_generator(in_width=d_model, out_width=2 * d_inner, bias=bias, **kw)
        self.conv1d = nn.Conv1d(in_channels=d_inner, out_channels=d_inner, bias=conv_bias,
                                kernel_size=d_conv, groups=d_inner, padding=d_conv - 1)
        self.x_proj = layer_generator(in_width=d_inner, out_width=dt_rank + 2 * d_state,
                                      bias=False, **kw)
        self.dt_proj = layer_generator(in_width=dt_rank, out_width=d_inner, bias=True, **kw)
        A = torch.arange(1, d_state + 1, dtype=torch.float32).repeat(d_inner, 1)
        self.A_log = nn.Parameter(torch.log(A))
        self.D = nn.Parameter(torch.ones(d_inner))
        self.out_proj = layer_generator(in_width=d_inner, out_width=d_model, bias=bias, **kw)

    def forward(self, x):
        x, res = reshape_apply(x, self.in_proj).split([self.d_inner, self.d_inner], dim=-1)
        x = PF.causal_conv_silu(x, self.conv1d.weight, self.conv1d.bias)
        y = PF.silu_gate(self.ssm(x), res)
        return reshape_apply(y, self.out_proj)

    def ssm(self, x):
        A = -torch.exp(self.A_log.float())
        x_dbl = reshape_apply(x, self.x_proj)
        delta, B, C = x_dbl.split([self.dt_rank, self.d_state, self.d_state], dim=-1)
        delta = F.softplus(reshape_apply(delta, self.dt_proj))
        return self.selective_scan(x, delta, A, B, C, self.D.float())

    def selective_scan(self, u, delta, A, B, C, D):
        return PF.selective_scan(u, delta, A, B, C, D)


class ResidualBlock(nn.Module):
    def __init__(self, d_inner: int, d_model: int, d_conv: int, d_state: int, dt_rank: int,
                 conv_bias: bool, bias: bool, layer_type: str = "continuous", n: int = 2,
                 segments: int = 2, hidden_layers: int = 0):
        super().__init__()
        self.mixer = MambaBlock(d_inner=d_inner, d_model=d_model, d_conv=d_conv, d_state=d_state,
                                dt_rank=dt_rank, conv_bias=conv_bias, bias=bias,
                                layer_type=layer_type, segments=segments, n=n,
                                hidden_layers=hidden_layers)
        self.norm = RMSNorm(d_model)

    def forward(self, x):
        return self.mixer(self.norm(x)) + x


class Mamba(nn.Module):
    def __init__(self, d_model: int, vocab_size: int, n_layer: int, d_state: int = 16,
                 d_conv: int = 4, conv_bias: bool = True, bias: bool = False, expand: int = 2,
                 dt_rank: Union[str, int] = "auto", pad_vocab_size_multiple: int = 8,
                 layer_type: str = "linear", n: int = 2, segments: int = 2,
                 hidden_layers: int = 0):
        super().__init__()
        d_inner = int(expand * d_model)
        if dt_rank == "auto":
            dt_rank = math.ceil(d_model / 16)
        if vocab_size % pad_vocab_size_multiple != 0:
            vocab_size += pad_vocab_size_multiple - vocab_size % pad_vocab_size_multiple
        self.embedding = nn.Embedding(vocab_size, d_model)
        self.layers = nn.ModuleList([
            ResidualBlock(d_inner=d_inner, d_model=d_model, d_conv=d_conv, d_state=d_state,
                          dt_rank=dt_rank, conv_bias=conv_bias, bias=bias, layer_type=layer_type,
                          segments=segments, n=n, hidden_layers=hidden_layers)
            for _ in range(n_layer)])
        self.norm_f = RMSNorm(d_model)
        self.lm_head = nn.Linear(d_model, vocab_size, bias=False)
        self.lm_head.weight = self.embedding.weight  # tied

    def forward(self, input_ids):
        """input_ids: long [b, l, 1] (or [b, l]) -> logits [b, l, vocab]."""
        ids = input_ids.reshape(input_ids.shape[0], -1)
        # (same results as self.embedding(ids) / self.lm_head(...): the weight is tied)
        x = _EmbeddingFunction.apply(ids, self.embedding.weight)
        for layer in self.layers:
            x = layer(x)
        return _HeadFunction.apply(self.norm_f(x), self.lm_head.weight)
