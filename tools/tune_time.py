"""Developer timing of one layer's three entry points with the tuning build (PP_LIB) and its
environment overrides."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from language_interpolation_amd import _lib  # noqa: E402

if os.environ.get("PP_LIB"):
    _lib.LIB_PATH = os.environ["PP_LIB"]
from language_interpolation_amd import functional as PF  # noqa: E402
from tools.layer_bench import LAYERS, time_fn  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "cfg2_hidden"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 16384
disc, n, I, O, S, dist, first = LAYERS[name]
K = PF.num_weights(n, S, disc)
g = torch.Generator().manual_seed(0)
x = torch.randn(B, I, generator=g).cuda()
w_t = ((torch.rand(I, K, O, generator=g) * 2 - 1) / I).cuda()
gy = (torch.randn(B, O, generator=g) / O).cuda()
spec = PF.make_spec(n, S, disc)
packed = PF.pack_weights(w_t, spec, B)
t1 = time_fn(lambda: PF.forward_raw(x, w_t, spec, packed=packed), 30)
t2 = time_fn(lambda: PF.backward_input_raw(x, gy, w_t, spec, packed=packed), 30)
t3 = time_fn(lambda: PF.backward_weight_raw(x, gy, tuple(w_t.shape), spec), 30)
print(f"{os.environ.get('TAG', '')}: fwd {t1:.1f} bwd_x {t2:.1f} bwd_w {t3:.1f} us (prepacked weights)")
