"""Developer probe: the gather forward moving the same 160 MB of table rows with the table inside
every XCD's L2 (25 inputs x 8000 rows, 3 MB table) and outside it (200 inputs x 1000 rows, 24 MB)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from language_interpolation_amd import functional as PF  # noqa: E402

dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(0)
for I, B in ((200, 1000), (25, 8000), (25, 1000), (50, 4000), (100, 2000)):
    n, S, disc, O = 1, 150, True, 200
    K = PF.num_weights(n, S, disc)
    x = torch.randn(B, I, generator=g).to(dev)
    w_t = ((torch.rand(I, K, O, generator=g) * 2 - 1) / I).to(dev)
    spec = PF.make_spec(n, S, disc, fwd_method="gather")
    for _ in range(5):
        PF.forward_raw(x, w_t, spec)
    torch.cuda.synchronize()
    t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0.record()
    for _ in range(50):
        PF.forward_raw(x, w_t, spec)
    t1.record()
    torch.cuda.synchronize()
    us = t0.elapsed_time(t1) / 50 * 1e3
    print(f"I={I:4d} B={B:5d}: table {I * K * O * 4 / 1e6:5.1f} MB, gathered {B * I * O * 4 / 1e6:6.1f} MB: {us:6.1f} us "
          f"= {B * I * O * 4 / us / 1e6:5.1f} TB/s", flush=True)
