"""Developer probe: backward-input and backward-weight of one dense-regime hidden layer, one after
the other on one stream against side by side on two streams (HIP events over 20 pairs)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from language_interpolation_amd import functional as PF  # noqa: E402

dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(0)
B, I, O, n, S, disc = 16384, 250, 250, 3, 2, False
K = PF.num_weights(n, S, disc)
x = torch.randn(B, I, generator=g).to(dev)
w_t = ((torch.rand(I, K, O, generator=g) * 2 - 1) / I).to(dev)
gy = (torch.randn(B, O, generator=g) / O).to(dev)
spec = PF.make_spec(n, S, disc)
side = torch.cuda.Stream()


def seq():
    PF.backward_input_raw(x, gy, w_t, spec)
    PF.backward_weight_raw(x, gy, tuple(w_t.shape), spec)


def par():
    ev = torch.cuda.Event()
    ev.record()
    with torch.cuda.stream(side):
        side.wait_event(ev)
        PF.backward_weight_raw(x, gy, tuple(w_t.shape), spec)
        done = torch.cuda.Event()
        done.record()
    PF.backward_input_raw(x, gy, w_t, spec)
    torch.cuda.current_stream().wait_event(done)


for name, fn in (("one stream", seq), ("two streams", par), ("one stream", seq), ("two streams", par)):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0.record()
    for _ in range(20):
        fn()
    t1.record()
    torch.cuda.synchronize()
    print(f"{name}: {t0.elapsed_time(t1) / 20 * 1e3:.1f} us per (bwd_x + bwd_w) pair", flush=True)
