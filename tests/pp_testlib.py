"""Shared helpers for the parity tests (test infrastructure; may import oracle/)."""
from __future__ import annotations

import json
import os

import numpy as np
import torch

from oracle import pp_oracle as po

TOL = 1e-5  # north_star: fp32 outputs and gradients within 1e-5 relative of the reference


RAVEN_HIST = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "raven_byte_hist.json")


def raven_char_probs():
    """128 byte probabilities of the reference's training text books/the_raven.txt (SURVEY 8d;
    the counts are a committed fixture, tests/golden/make_raven_hist.py)."""
    with open(RAVEN_HIST) as f:
        counts = torch.tensor(json.load(f)["counts"], dtype=torch.float64)
    return counts / counts.sum()


def char_grid(B, I, gen):
    """x = (c - 63.5)/64 with c drawn from the byte histogram of the reference's text
    (single_text_dataset.py:15-16, :102-123): 90 distinct bytes, 24 % spaces."""
    c = torch.multinomial(raven_char_probs(), B * I, replacement=True, generator=gen).view(B, I)
    return po.ascii_to_float(c)


def make_inputs(B, I, O, n, S, disc, dist="normal", seed=0):
    gen = torch.Generator().manual_seed(seed)
    K = po.num_weights(n, S, disc)
    if dist == "char":
        x = char_grid(B, I, gen)
    elif dist == "uniform":
        x = torch.rand(B, I, generator=gen) * 2 - 1
    elif dist == "edges":
        # exact segment boundaries, +-1 ulp, and far out-of-range values
        m = torch.randint(0, S + 1, (B, I), generator=gen).float()
        x = (-1.0 + 2.0 * m / S).float()
        bump = torch.randint(-2, 3, (B, I), generator=gen)
        xn = x.numpy().copy()
        for _ in range(2):
            xn = np.where(bump.numpy() > 0, np.nextafter(xn, np.float32(4)), xn)
            xn = np.where(bump.numpy() < 0, np.nextafter(xn, np.float32(-4)), xn)
            bump = bump - torch.sign(bump)
        x = torch.from_numpy(xn.astype(np.float32))
        far = torch.rand(B, I, generator=gen) < 0.05
        x = torch.where(far, torch.randn(B, I, generator=gen) * 3, x)
    else:
        x = torch.randn(B, I, generator=gen)
    w = (torch.rand(O, I, K, generator=gen) * 2 - 1) / I
    gy = torch.randn(B, O, generator=gen) / O
    return x.float().contiguous(), w.float().contiguous(), gy.float().contiguous()


def rel_err(got: torch.Tensor, ref: torch.Tensor) -> float:
    """max |got - ref| / max |ref|  (per tensor; individual outputs can cancel to ~0)."""
    ref = ref.double().cpu()
    got = got.double().cpu()
    denom = max(ref.abs().max().item(), 1e-30)
    return (got - ref).abs().max().item() / denom


def oracle_all(x, w, gy, n, S, disc, **kw):
    return po.piecewise_fwd_bwd(x, w, gy, n, S, disc, precision="f64", **kw)
