"""Developer check of the register-accumulator weight gradient against the sorted path."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from language_interpolation_amd import _lib  # noqa: E402

if os.environ.get("PP_LIB"):
    _lib.LIB_PATH = os.environ["PP_LIB"]
from language_interpolation_amd import functional as PF  # noqa: E402

CASES = [
    # disc, n, S, I, O, B
    (True, 1, 150, 200, 200, 16384),
    (True, 1, 150, 200, 200, 1000),
    (False, 2, 128, 16, 250, 16384),
    (False, 2, 128, 16, 250, 1000),
    (False, 2, 128, 16, 200, 256),
    (True, 2, 64, 16, 1000, 4096),
    (False, 3, 50, 37, 70, 777),
    (True, 3, 64, 5, 64, 129),
    (True, 1, 192, 3, 1, 17),
    (True, 1, 150, 200, 200, 2049),
    (True, 2, 80, 6, 72, 3000),
    (False, 3, 50, 37, 68, 2111),
    (True, 1, 160, 9, 4, 100),
    (True, 2, 128, 16, 1000, 16384),
    (True, 2, 128, 16, 1000, 1000),
    (True, 2, 150, 200, 200, 4099),
    (True, 3, 190, 6, 70, 333),
]


def main():
    dev = torch.device("cuda:0")
    gen = torch.Generator().manual_seed(1)
    bad = 0
    for disc, n, S, I, O, B in sorted(CASES, key=lambda c: c[4] % 4 != 0):
        K = PF.num_weights(n, S, disc)
        x = (torch.rand(B, I, generator=gen) * 2.2 - 1.1).to(dev)
        gy = torch.randn(B, O, generator=gen).to(dev)
        ref = PF.backward_weight_raw(x, gy, (I, K, O), PF.make_spec(n, S, disc, bwd_w_method="sort"))
        got = PF.backward_weight_raw(x, gy, (I, K, O), PF.make_spec(n, S, disc, bwd_w_method="regacc"))
        got2 = PF.backward_weight_raw(x, gy, (I, K, O), PF.make_spec(n, S, disc, bwd_w_method="regacc"))
        torch.cuda.synchronize()
        err = (got - ref).abs().max().item() / ref.abs().max().item()
        same = torch.equal(got, got2)
        ok = err < 2e-6 and same
        bad += not ok
        print(f"disc={int(disc)} n={n} S={S} I={I} O={O} B={B} K={K}: rel err {err:.2e} repeatable={same} {'ok' if ok else 'FAIL'}",
              flush=True)
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
