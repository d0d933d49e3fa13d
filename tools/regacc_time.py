"""Developer timing of the register-accumulator weight gradient (library chosen by PP_LIB)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from language_interpolation_amd import _lib  # noqa: E402

if os.environ.get("PP_LIB"):
    _lib.LIB_PATH = os.environ["PP_LIB"]
from language_interpolation_amd import functional as PF  # noqa: E402

SHAPES = [(True, 1, 150, 200, 200, 16384), (False, 2, 128, 16, 250, 16384), (True, 2, 64, 16, 1000, 16384),
          (True, 2, 128, 16, 1000, 16384), (True, 2, 150, 200, 200, 16384)]
if os.environ.get("REGACC_B"):
    SHAPES = [s[:5] + (int(os.environ["REGACC_B"]),) for s in SHAPES]
dev = torch.device("cuda:0")
gen = torch.Generator().manual_seed(1)
for disc, n, S, I, O, B in SHAPES:
    K = PF.num_weights(n, S, disc)
    x = (torch.rand(B, I, generator=gen) * 2 - 1).to(dev)
    gy = torch.randn(B, O, generator=gen).to(dev)
    spec = PF.make_spec(n, S, disc, bwd_w_method="regacc")
    for _ in range(5):
        PF.backward_weight_raw(x, gy, (I, K, O), spec)
    torch.cuda.synchronize()
