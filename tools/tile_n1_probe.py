"""Developer probe: gather against row-tile forward for the n = 1 table layers (16 inputs, 128 rows a slab)
at three batches -- the figures behind pp_tile_fwd_preferred's rule for unsplit few-input layers."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from language_interpolation_amd import functional as PF
from bench import char_batch
dev = torch.device("cuda:0"); g = torch.Generator().manual_seed(0)
def ev(fn, it=30):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(it): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / it * 1e3
for I, O in ((16, 1000), (16, 250), (16, 200)):
    for B in (16384, 8192, 4096):
        x = char_batch(B, I, g).to(dev)
        w_t = ((torch.rand(I, 128, O, generator=g) * 2 - 1) / I).to(dev)
        out = []
        for m in ("gather", "tile"):
            spec = PF.make_spec(1, 128, True, fwd_method=m)
            try:
                out.append(f"{m} {ev(lambda: PF.forward_raw(x, w_t, spec)):6.1f} us")
            except Exception as e:
                out.append(f"{m} refused ({str(e)[:40]})")
        print(f"n=1 S=128 {I}->{O} B={B}: " + "   ".join(out), flush=True)
