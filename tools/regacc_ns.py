"""Developer sweep of the slice count of the register-accumulator weight gradient on one shape
(whole entry point by HIP events; needs a -DPP_TUNING library, PP_LIB=...)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from language_interpolation_amd import _lib  # noqa: E402

_lib.LIB_PATH = os.environ["PP_LIB"]
from language_interpolation_amd import functional as PF  # noqa: E402

dev = torch.device("cuda:0")
gen = torch.Generator().manual_seed(1)
for disc, n, S, I, O, B in [(False, 2, 128, 16, 250, 16384), (False, 2, 128, 16, 200, 16384),
                            (True, 2, 128, 16, 1000, 16384), (True, 1, 150, 200, 200, 4096)]:
    K = PF.num_weights(n, S, disc)
    x = (torch.rand(B, I, generator=gen) * 2 - 1).to(dev)
    gy = torch.randn(B, O, generator=gen).to(dev)
    spec = PF.make_spec(n, S, disc, bwd_w_method="regacc")
    for lds in ("0", "1"):
        os.environ["PP_REGACC_LDS"] = lds
        row = []
        for ns in (1, 2, 4, 8, 12, 16, 24, 32, 48, 64):
            os.environ["PP_REGACC_NSLICE"] = str(ns)
            f = lambda: PF.backward_weight_raw(x, gy, (I, K, O), spec)
            for _ in range(3):
                f()
            torch.cuda.synchronize()
            t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            t0.record()
            for _ in range(20):
                f()
            t1.record()
            torch.cuda.synchronize()
            row.append(f"{ns}:{t0.elapsed_time(t1) / 20 * 1e3:.1f}")
        print(f"n={n} S={S} {I}->{O} B={B} lds={lds}  " + "  ".join(row), flush=True)
