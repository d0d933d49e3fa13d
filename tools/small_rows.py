"""Developer timing: f32-MFMA (dense) vs fp16x2 kernels of the Mamba projections' layers at
small row counts (pre-split weights, as a step runs them)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from language_interpolation_amd import functional as PF  # noqa: E402
from tools.layer_bench import time_fn  # noqa: E402

SHAPES = [(64, 256), (256, 256), (128, 68), (68, 68), (128, 128), (128, 64), (64, 64), (250, 250)]
n, S = 3, 2
K = PF.num_weights(n, S, False)
g = torch.Generator().manual_seed(0)
for B in [int(v) for v in sys.argv[1:]] or [64, 128, 240]:
    for I, O in SHAPES:
        x = torch.randn(B, I, generator=g).cuda()
        w_t = ((torch.rand(I, K, O, generator=g) * 2 - 1) / I).cuda()
        gy = (torch.randn(B, O, generator=g) / O).cuda()
        out = []
        for m in ("dense", "fp16x2", "gather"):
            spec = PF.make_spec(n, S, False, fwd_method=m, bwd_x_method=m)
            packed = PF.pack_weights(w_t, spec, B) if m == "fp16x2" else None
            try:
                t1 = time_fn(lambda: PF.forward_raw(x, w_t, spec, packed=packed), 30)
                t2 = time_fn(lambda: PF.backward_input_raw(x, gy, w_t, spec, packed=packed), 30)
            except Exception as e:  # noqa: BLE001
                out.append(f"{m}: {type(e).__name__}")
                continue
            out.append(f"{m}: fwd {t1:5.1f} bwd_x {t2:5.1f}")
        print(f"B={B:4d} {I:3d}->{O:3d}  " + " | ".join(out), flush=True)
