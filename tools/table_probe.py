"""Probe: the first layers on character inputs as they run now (n = 2, S = 128) against the
equivalent table layer (discontinuous n = 1, S = 128 on T[i][c][:]) -- existing kernels only."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from language_interpolation_amd import functional as PF, _lib
from bench import char_batch

lib = _lib.load()
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(0)
B = int(sys.argv[1]) if len(sys.argv) > 1 else 16384

def ev(fn, it=30):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(it): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / it * 1e3

for name, disc, n, S, I, O in (("cfg2 L0 as is", False, 2, 128, 16, 250), ("cfg2 L0 as table", True, 1, 128, 16, 250),
                               ("cfg4 L0 as is", True, 2, 128, 16, 1000), ("cfg4 L0 as table", True, 1, 128, 16, 1000)):
    K = PF.num_weights(n, S, disc)
    x = char_batch(B, I, g).to(dev)
    w = ((torch.rand(O, I, K, generator=g) * 2 - 1) / I).to(dev)
    w = w.permute(1, 2, 0).contiguous().permute(2, 0, 1)
    w_t = w.permute(1, 2, 0)
    gy = (torch.randn(B, O, generator=g) / O).to(dev)
    spec = PF.make_spec(n, S, disc)
    t_f = ev(lambda: PF.forward_raw(x, w_t, spec))
    wp = w.detach().clone().requires_grad_(False)
    t_fl = ev(lambda: PF.piecewise_poly_layer_norm(x, wp, spec, 1e-5)) if PF.layer_norm_fusable(x, wp, spec) else float("nan")
    t_w = ev(lambda: PF.backward_weight_raw(x, gy, tuple(w_t.shape), spec))
    am = [lib.pp_auto_method(p, B, I, O, n, S, int(disc), 2.0) for p in range(3)]
    print(f"{name:18s} B={B}: fwd {t_f:6.1f} us  fwd+LN {t_fl:6.1f} us  bwd_w {t_w:6.1f} us  auto={am}", flush=True)
