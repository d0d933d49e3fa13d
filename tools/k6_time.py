import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from language_interpolation_amd import functional as PF
for disc, n, S, I, O, B in ((True, 3, 2, 250, 250, 16384), (False, 4, 2, 250, 250, 16384), (False, 3, 2, 250, 250, 16384), (True, 2, 4, 250, 250, 16384), (False, 2, 4, 250, 250, 16384)):
    K = PF.num_weights(n, S, disc)
    x = torch.randn(B, I, device="cuda"); gy = torch.randn(B, O, device="cuda")
    spec = PF.make_spec(n, S, disc, bwd_w_method="fp16x2")
    for _ in range(3): PF.backward_weight_raw(x, gy, (I, K, O), spec)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(30): PF.backward_weight_raw(x, gy, (I, K, O), spec)
    e1.record(); torch.cuda.synchronize()
    t_w = e0.elapsed_time(e1) / 30 * 1e3
    w_t = torch.randn(I, K, O, device="cuda") / I
    spec2 = PF.make_spec(n, S, disc, bwd_x_method="fp16x2", fwd_method="fp16x2")
    res = []
    for f in (lambda: PF.backward_input_raw(x, gy, w_t, spec2), lambda: PF.forward_raw(x, w_t, spec2)):
        for _ in range(3): f()
        torch.cuda.synchronize()
        e0.record()
        for _ in range(30): f()
        e1.record(); torch.cuda.synchronize()
        res.append(e0.elapsed_time(e1) / 30 * 1e3)
    print((disc, n, S, "K=%d" % K), "bwd_w %.1f us, bwd_x %.1f us, fwd %.1f us" % (t_w, res[0], res[1]), flush=True)
