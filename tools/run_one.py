"""Run one kernel of one layer a few times (for rocprofv3 --pmc runs). Developer tool.
python tools/run_one.py <pass: fwd|bwd_x|bwd_w> [B] [layer name from layer_bench.LAYERS]"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from language_interpolation_amd import _lib
if os.environ.get("PP_LIB"):
    _lib.LIB_PATH = os.environ["PP_LIB"]
from language_interpolation_amd import functional as PF
from tools.layer_bench import LAYERS

which = sys.argv[1]
B = int(sys.argv[2]) if len(sys.argv) > 2 else 16384
name = sys.argv[3] if len(sys.argv) > 3 else "cfg2_hidden"
disc, n, I, O, S, dist, first = LAYERS[name]
K = PF.num_weights(n, S, disc)
g = torch.Generator().manual_seed(0)
if dist == "char":  # first layers: character windows drawn from the reference text's histogram
    from bench import char_batch
    x = char_batch(B, I, g).cuda()
else:
    x = torch.randn(B, I, generator=g).cuda()
w_t = ((torch.rand(I, K, O, generator=g) * 2 - 1) / I).cuda()
gy = (torch.randn(B, O, generator=g) / O).cuda()
spec = PF.make_spec(n, S, disc)
for _ in range(5):
    if which == "fwd":
        PF.forward_raw(x, w_t, spec)
    elif which == "bwd_x":
        PF.backward_input_raw(x, gy, w_t, spec)
    else:
        PF.backward_weight_raw(x, gy, tuple(w_t.shape), spec)
torch.cuda.synchronize()
