"""Developer sweep: register-accumulator weight gradient, gy per wave (v1) against gy through LDS,
whole entry point (prepass + kernel + slice sum) by HIP events.  Needs a -DPP_TUNING library
(PP_LIB=...), which reads PP_REGACC_LDS / PP_REGACC_NSLICE at every launch."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from language_interpolation_amd import _lib  # noqa: E402

if os.environ.get("PP_LIB"):
    _lib.LIB_PATH = os.environ["PP_LIB"]
from language_interpolation_amd import functional as PF  # noqa: E402

SHAPES = [(True, 1, 150, 200, 200), (True, 1, 150, 200, 128), (True, 1, 150, 16, 200),
          (False, 2, 128, 16, 250), (True, 2, 64, 16, 1000), (False, 2, 128, 16, 200),
          (True, 3, 50, 64, 256), (True, 1, 192, 512, 512)]
dev = torch.device("cuda:0")
gen = torch.Generator().manual_seed(1)


def timed(fn, reps=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0.record()
    for _ in range(reps):
        fn()
    t1.record()
    torch.cuda.synchronize()
    return t0.elapsed_time(t1) / reps * 1e3


for disc, n, S, I, O in SHAPES:
    K = PF.num_weights(n, S, disc)
    for B in (512, 1000, 2048, 4096, 16384):
        x = (torch.rand(B, I, generator=gen) * 2 - 1).to(dev)
        gy = torch.randn(B, O, generator=gen).to(dev)
        spec = PF.make_spec(n, S, disc, bwd_w_method="regacc")
        out = []
        for lds in ("0", "1"):
            os.environ["PP_REGACC_LDS"] = lds
            out.append(timed(lambda: PF.backward_weight_raw(x, gy, (I, K, O), spec)))
        print(f"disc={int(disc)} n={n} S={S} {I}->{O} B={B}: v1 {out[0]:7.1f} us   lds {out[1]:7.1f} us", flush=True)
