"""Selective scan and Mamba (cfg5) timings: python tools/mamba_bench.py [b]
(PP_MB_OPT=adam for Adam instead of the reference's default optimizer, sparse_lion)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import language_interpolation_amd as L
from language_interpolation_amd import functional as PF
from language_interpolation_amd import train as T


def torch_loop_scan(u, delta, A, B, C, D):
    """The reference's formulation (state_space_network.py:399-420) on the same GPU."""
    b, l, d = u.shape
    dA = torch.exp(delta.unsqueeze(-1) * A)
    dBu = delta.unsqueeze(-1) * B.unsqueeze(2) * u.unsqueeze(-1)
    x = torch.zeros(b, d, A.shape[1], device=u.device)
    ys = []
    for i in range(l):
        x = dA[:, i] * x + dBu[:, i]
        ys.append((x * C[:, i].unsqueeze(1)).sum(-1))
    return torch.stack(ys, dim=1) + u * D


def timeit(fn, iters=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


b = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
l, d, n = 15, 128, 32
g = torch.Generator(device="cuda").manual_seed(0)
u = torch.randn(b, l, d, device="cuda", generator=g, requires_grad=True)
delta = torch.rand(b, l, d, device="cuda", generator=g).requires_grad_()
A = (-torch.arange(1, n + 1, device="cuda").float().repeat(d, 1)).requires_grad_()
Bm = torch.randn(b, l, n, device="cuda", generator=g, requires_grad=True)
Cm = torch.randn(b, l, n, device="cuda", generator=g, requires_grad=True)
D = torch.ones(d, device="cuda", requires_grad=True)
gy = torch.randn(b, l, d, device="cuda", generator=g)
ins = [u, delta, A, Bm, Cm, D]
for name, f in (() if os.environ.get("PP_MB_ONLY") else (("hip", PF.selective_scan), ("torch loop", torch_loop_scan))):
    def fwd_bwd():
        torch.autograd.grad(f(*ins), ins, gy)
    with torch.no_grad():
        t_f = timeit(lambda: f(*ins))
    t_fb = timeit(fwd_bwd)
    print(f"selective scan b={b} l={l} d={d} n={n} [{name}]: fwd {t_f:.1f} us, fwd+bwd {t_fb:.1f} us", flush=True)

root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
cfg = T.load_config(os.path.join(root, "configs", "high_order_interpolation.yaml"),
                    ["net=mamba", "net.n_layer=2", f"optimizer={os.environ.get('PP_MB_OPT', 'sparse_lion')}",
                     f"batch_size={b}", "data.max_features=16"])
data = T.make_data(cfg, T.synthetic_text(1 << 20))
VARIANTS = ("hip scan, graph replay", "hip scan", "torch-loop scan")
if os.environ.get("PP_MB_ONLY"):
    VARIANTS = tuple(v for v in VARIANTS if os.environ["PP_MB_ONLY"] in v)
for name in VARIANTS:
    torch.manual_seed(0)
    module = T.prediction_net(cfg, device="cuda")
    if name == "torch-loop scan":
        for blk in module.model.layers:
            blk.mixer.selective_scan = torch_loop_scan
    graph = "graph" in name
    opt = module.configure_optimizers(capturable=graph)
    gen = torch.Generator(device="cuda").manual_seed(0)
    graphed = T.GraphedTrainStep(module, opt, data.sample(b, gen)) if graph else None
    def step():
        if graphed is not None:
            graphed(data.sample(b, gen))
            return
        opt.zero_grad(set_to_none=True)
        loss, _ = module.eval_step(data.sample(b, gen))
        loss.backward()
        opt.step()
    us = timeit(step, 10)
    print(f"mamba cfg5 (2 layers, b={b}, l=15) train step [{name}]: {us / 1e3:.2f} ms = {b * 15 / us:.2f} M chars/s", flush=True)
