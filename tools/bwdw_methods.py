"""Time the weight-gradient methods of the sparse regime: python tools/bwdw_methods.py"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from language_interpolation_amd import functional as PF

SHAPES = [  # disc, n, S, I, O, B
    (False, 2, 128, 16, 250, 16384), (False, 2, 128, 16, 250, 65536), (False, 2, 128, 16, 250, 1024),
    (True, 2, 128, 16, 1000, 16384), (True, 1, 150, 200, 256, 16384), (True, 2, 150, 200, 128, 4096),
]
ONLY = int(os.environ.get("PP_ONLY", "-1"))
for disc, n, S, I, O, B in (SHAPES if ONLY < 0 else SHAPES[ONLY:ONLY + 1]):
    K = PF.num_weights(n, S, disc)
    c = torch.randint(0, 128, (B, I), device="cuda")
    x = (c.float() - 63.5) / 64
    gy = torch.randn(B, O, device="cuda")
    out = {}
    for m in ("sort", "tilesort", "atomic"):
        spec = PF.make_spec(n, S, disc, bwd_w_method=m)
        try:
            g = PF.backward_weight_raw(x, gy, (I, K, O), spec)
        except Exception as e:
            out[m] = "n/a"
            continue
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            g = PF.backward_weight_raw(x, gy, (I, K, O), spec)
        e1.record()
        torch.cuda.synchronize()
        out[m] = "%.1f us" % (e0.elapsed_time(e1) * 50)
        if m == "sort": ref = g
        elif m == "tilesort": out["err"] = "%.1e" % ((g - ref).abs().max() / ref.abs().max()).item()
    print((disc, n, S, I, O, B), out, flush=True)
