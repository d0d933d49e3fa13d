"""Developer stress of the packed-FMA register-accumulator body (discontinuous n = 2) against the
two-FMA body: same bits expected on every launch.  Needs a -DPP_TUNING library (PP_LIB=...)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from language_interpolation_amd import _lib  # noqa: E402

_lib.LIB_PATH = os.environ["PP_LIB"]
from language_interpolation_amd import functional as PF  # noqa: E402

dev = torch.device("cuda:0")
gen = torch.Generator().manual_seed(3)
bad = 0
for disc, n, S, I, O, B, reps in [(True, 2, 128, 16, 1000, 16384, 300), (True, 2, 64, 16, 1000, 16384, 100),
                                  (True, 2, 150, 200, 200, 8192, 60), (True, 2, 96, 64, 256, 8200, 100)]:
    K = PF.num_weights(n, S, disc)
    x = (torch.rand(B, I, generator=gen) * 2.2 - 1.1).to(dev)
    gy = torch.randn(B, O, generator=gen).to(dev)
    spec = PF.make_spec(n, S, disc, bwd_w_method="regacc")
    os.environ["PP_REGACC_PK"] = "0"
    ref = PF.backward_weight_raw(x, gy, (I, K, O), spec).clone()
    os.environ["PP_REGACC_PK"] = "1"
    diff = 0
    for _ in range(reps):
        got = PF.backward_weight_raw(x, gy, (I, K, O), spec)
        if not torch.equal(got, ref):
            diff += 1
            worst = (got - ref).abs().max().item()
    print(f"n={n} S={S} {I}->{O} B={B}: {reps} launches, {diff} differ" + (f" (max abs {worst:.3e})" if diff else ""), flush=True)
    bad += diff
    t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    for pk in ("0", "1"):
        os.environ["PP_REGACC_PK"] = pk
        torch.cuda.synchronize()
        t0.record()
        for _ in range(20):
            PF.backward_weight_raw(x, gy, (I, K, O), spec)
        t1.record()
        torch.cuda.synchronize()
        print(f"   PK={pk}: {t0.elapsed_time(t1) / 20 * 1e3:.1f} us", flush=True)
sys.exit(1 if bad else 0)
