"""bench.py -- the hot path's headline measurement (contract: see the task brief).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--batch B]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

Workload (BASELINE.json configs[1]): net=small, width 250, n=3, hidden.layers=3,
segments=2, continuous, synthetic char batches -- i.e. the polynomial layers
    L0  continuous n=2  16 -> 250  S=128      (character-grid input layer)
    3 x continuous n=3 250 -> 250  S=2        (hidden)
    out continuous n=3 250 -> 128  S=2
with the (non-affine) LayerNorm the reference places between them.  One step =
one forward + backward (loss = <y, gy>) of that stack over one
batch of B rows per GPU, inputs resident in HBM; with N > 1 ranks every rank
runs its own B rows (weak scaling) and the weight gradients are all-reduced
over RCCL, overlapped with backward.  value = N*B*K / time(K steps) samples/s.

Extra objects in the JSON line: "roofline" for the dominant kernel (timed live
with HIP events on the launch stream) and "cpu_baseline" (the CPU oracle in the
reference formulation, on this box's host cores, bounded sample, rank 0, N=1).
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_F32_TFLOPS = 157.3  # MI355X dense fp32 (vector = f32-input MFMA), MI355X_MICROARCH.md
PEAK_HBM_GBS = 8000.0

NET = dict(layer_type="continuous", n=3, n_in=2, n_hidden=3, n_out=3, in_width=16, out_width=128,
           hidden_layers=3, hidden_width=250, in_segments=128, out_segments=2, hidden_segments=2)


def char_batch(B, I, gen):
    """x = (c - 63.5)/64 with c distributed like ASCII text (single_text_dataset.py:15-16)."""
    c = torch.randint(32, 127, (B, I), generator=gen)
    c[torch.rand(B, I, generator=gen) < 0.18] = 32
    c[torch.rand(B, I, generator=gen) < 0.02] = 10
    return (c.float() - 64 + 0.5) / 64


def time_kernel(fn, iters=20, warm=3):
    """Average launch duration of `fn`'s kernels: HIP events on torch's current stream,
    which is the stream the C ABI launches on (functional._stream)."""
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e-3 / iters  # seconds


def roofline_for_hidden_layer(B, dev):
    """Times the three kernels of one hidden layer (continuous n=3 250->250 S=2) and
    reports the roofline of the one that takes longest."""
    from language_interpolation_amd import _lib
    from language_interpolation_amd import functional as PF

    n, S, I, O = 3, 2, 250, 250
    K = PF.num_weights(n, S, False)
    gen = torch.Generator().manual_seed(1)
    x = torch.randn(B, I, generator=gen).to(dev)
    w_t = ((torch.rand(I, K, O, generator=gen) * 2 - 1) / I).to(dev)
    gy = (torch.randn(B, O, generator=gen) / O).to(dev)
    spec = PF.make_spec(n, S, False)
    # preallocate outputs/workspace so the timed region is exactly the library's launches
    lib = _lib.load()
    y = torch.empty(B, O, device=dev)
    gx = torch.empty(B, I, device=dev)
    gw = torch.empty(I, K, O, device=dev)
    nbytes = lib.pp_workspace_bytes(B, I, O, n, S, 0, 2.0, 0)
    ws = torch.empty(max(nbytes, 16), dtype=torch.uint8, device=dev)
    fbytes = lib.pp_forward_workspace_bytes(B, I, O, n, S, 0, 2.0, 0)
    fws = torch.empty(max(fbytes, 16), dtype=torch.uint8, device=dev)
    xbytes = lib.pp_backward_input_workspace_bytes(B, I, O, n, S, 0, 2.0, 0)
    xws = torch.empty(max(xbytes, 16), dtype=torch.uint8, device=dev)
    nodes = spec.node_array()
    st = PF._stream(dev)
    P = PF._ptr

    # the same entry points, methods (auto) and workspaces the module's autograd function uses
    def fwd():
        lib.pp_forward_ws_f32(P(x), P(w_t), P(y), None, P(fws), fbytes, B, I, O, n, S, 0, 2.0,
                              nodes, 1.0, 0, st)

    def bwd_x():
        lib.pp_backward_input_ws_f32(P(x), P(gy), P(w_t), P(gx), P(xws), xbytes, B, I, O, n, S, 0,
                                     2.0, nodes, 1.0, 0, st)

    def bwd_w():
        lib.pp_backward_weight_f32(P(x), P(gy), P(gw), P(ws), nbytes, B, I, O, n, S, 0, 2.0, nodes,
                                   1.0, 0, st)

    methods = [lib.pp_auto_method(p, B, I, O, n, S, 0, 2.0) for p in range(3)]
    fname = {1: "pp_fwd_gather_kernel", 2: "pp_fwd_dense_kernel",
             3: "pp_fwd_bf3_kernel<bf16x3>(+pack)",
             4: "pp_fwd_bf3_kernel<fp16x2>(+absmax,+pack)"}[methods[0]]
    xname = {1: "pp_bwd_input_gather_kernel", 2: "pp_bwd_input_dense_kernel",
             3: "pp_bwd_x_bf3_kernel<bf16x3>(+pack)",
             4: "pp_bwd_x_bf3_kernel<fp16x2>(+absmax,+pack)"}[methods[1]]
    wname = {1: "pp_bwd_weight_atomic_kernel", 2: "pp_bwd_weight_segreduce_kernel(+sort)",
             3: "pp_bwd_weight_dense_kernel(+reduce)",
             4: "pp_bwd_w_bf3_kernel<bf16x3>(+pack,+reduce)",
             5: "pp_bwd_w_bf3_kernel<fp16x2>(+pack,+reduce)"}[methods[2]]
    times = {fname: time_kernel(fwd), xname: time_kernel(bwd_x), wname: time_kernel(bwd_w)}
    name = max(times, key=times.get)
    flops = 2.0 * B * I * O * n  # algorithmic flops of one pass (6*B*I*O*n for three)
    achieved = flops / times[name] * 1e-12
    # HBM-side bytes per launch of the dominant entry point: measured offline with rocprofv3
    # --pmc (separate passes; profiles/r01_i_traffic.json says how), valid for the shape and
    # method it was taken on (weight gradient and backward-input, fp16x2, 16384 rows)
    traffic = None
    tpath = os.path.join(ROOT, "profiles", "r01_i_traffic.json")
    if B == 16384 and os.path.exists(tpath):
        with open(tpath) as f:
            measured = json.load(f)
        if name == wname and methods[2] == 5:
            traffic = measured["traffic_bytes"]
        elif name == xname and methods[1] == 4:
            traffic = measured["backward_input"]["traffic_bytes"]
    return {
        "bound": "mfma", "achieved": round(achieved, 3), "peak": PEAK_F32_TFLOPS,
        "unit": "TFLOP/s", "frac": round(achieved / PEAK_F32_TFLOPS, 4), "traffic": traffic,
        "kernel": name, "layer": "continuous n=3 250->250 S=2 (cfg2 hidden)", "rows": B,
        "algorithmic_flops_per_launch": flops,
        "launch_us": {k: round(v * 1e6, 2) for k, v in times.items()},
        "auto_methods": methods,
    }


def step_bytes(B):
    """Algorithmic HBM bytes of one fwd+bwd step of the network (every operand
    touched once per pass): 4*(3*B*I + 2*B*O + 3*W) per interior layer, 4*(2*B*I + 2*B*O + 2*W)
    for the first layer; LayerNorm passes not counted."""
    layers = [(16, 250, 129, True)] + [(250, 250, 5, False)] * 3 + [(250, 128, 5, False)]
    total = 0
    for I, O, K, first in layers:
        W = I * K * O
        total += 4 * ((2 * B * I + 2 * B * O + 2 * W) if first else (3 * B * I + 2 * B * O + 3 * W))
    return total


class _TrainModule(torch.nn.Module):
    """eval_step of ASCIIPredictionNet (cross entropy, accuracy) around an existing network."""

    def __init__(self, net):
        super().__init__()
        self.model = net
        self.loss = torch.nn.CrossEntropyLoss()

    def eval_step(self, batch):
        y_hat = self.model(batch[0])
        y = batch[1].flatten()
        return self.loss(y_hat, y), (torch.argmax(y_hat, dim=1) == y).float().mean()


def train_chars_per_s(net, dev, B, steps, world, reducer, graph):
    """The other half of BASELINE's metric: optimizer steps of the same network on character
    windows gathered on the device (cross entropy, Adam).  One process: the step is replayed
    from a HIP graph (train.GraphedTrainStep); several: eager launches with the RCCL
    all-reduce of the gradients overlapping the backward."""
    from language_interpolation_amd.train import CharWindows, GraphedTrainStep, synthetic_text

    data = CharWindows(synthetic_text(1 << 20, seed=7), 16, device=dev)
    gen = torch.Generator(device=dev).manual_seed(11)
    module = _TrainModule(net)
    graph = graph and reducer is None
    opt = torch.optim.Adam(net.parameters(), lr=1e-4, capturable=graph)
    graphed = GraphedTrainStep(module, opt, data.sample(B, gen)) if graph else None

    def one():
        batch = data.sample(B, gen)
        if graphed is not None:
            graphed(batch)
            return
        opt.zero_grad(set_to_none=True)
        module.eval_step(batch)[0].backward()
        if reducer is not None:
            reducer.finish()
        opt.step()

    for _ in range(3):
        one()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    t0 = time.perf_counter()
    for _ in range(steps):
        one()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    return world * B * steps / dt


def cpu_baseline(rows, steps=2):
    """The CPU oracle in the reference formulation (gather [B,I,O,n] + einsum +
    autograd), same network, on this box's host cores."""
    from oracle import pp_oracle as po

    cores = max(1, min(16, os.cpu_count() or 1))
    torch.set_num_threads(cores)
    torch.manual_seed(0)
    net = po.OracleHighOrderMLP(normalization=po.OracleLazyLayerNormLastDim, **NET)
    gen = torch.Generator().manual_seed(0)
    x = char_batch(rows, 16, gen)
    gy = torch.randn(rows, 128, generator=gen) / 128

    def step():
        for p in net.parameters():
            p.grad = None
        y = net(x)
        y.backward(gy)

    step()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    dt = (time.perf_counter() - t0) / steps
    return {"value": round(rows / dt, 1), "unit": "samples/s", "cores": cores, "kind": "port",
            "sample": f"{steps} fwd+bwd steps of the same 5-layer net at {rows} rows "
                      f"(torch CPU oracle, reference gather+einsum formulation), {dt:.3f} s/step"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--batch", type=int, default=16384, help="rows per GPU per step")
    ap.add_argument("--no-graph", action="store_true", help="eager launches instead of a HIP graph")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-rows", type=int, default=4096)
    ap.add_argument("--train-steps", type=int, default=20,
                    help="optimizer steps for the extra train chars/s figure (0 = skip)")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    if args.gpus != world and rank == 0 and world > 1:
        print(f"warning: --gpus {args.gpus} but WORLD_SIZE {world}", file=sys.stderr)
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)

    import language_interpolation_amd as L
    from language_interpolation_amd.parallel import GradAllReducer, broadcast_parameters

    torch.manual_seed(0)
    net = L.HighOrderMLP(normalization=L.LazyLayerNormLastDim, device=dev, **NET)
    broadcast_parameters(net)
    B = args.batch
    gen = torch.Generator().manual_seed(100 + rank)
    x = char_batch(B, 16, gen).to(dev)
    gy = (torch.randn(B, 128, generator=gen) / 128).to(dev)
    reducer = GradAllReducer(net.parameters()) if world > 1 else None
    params = list(net.parameters())

    def step():
        for p in params:
            p.grad = None
        y = net(x)
        y.backward(gy)
        if reducer is not None:
            reducer.finish()

    use_graph = (world == 1) and not args.no_graph
    graph = None
    if use_graph:
        try:
            s = torch.cuda.Stream()
            s.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(s):
                for _ in range(3):
                    step()
            torch.cuda.current_stream().wait_stream(s)
            torch.cuda.synchronize()
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph):
                step()
            run = graph.replay
        except Exception as e:  # capture refused: measure eager, and say so
            print(f"HIP graph capture failed ({type(e).__name__}: {e}); running eager",
                  file=sys.stderr)
            graph = None
            torch.cuda.synchronize()
            run = step
    else:
        run = step

    for _ in range(args.warmup):
        run()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        run()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())

    train_cps = None
    if args.train_steps > 0:
        graph = graph  # (the captured graph keeps its own buffers; training runs eager)
        train_cps = train_chars_per_s(net, dev, B, args.train_steps, world, reducer, not args.no_graph)

    if rank == 0:
        ms = dt / args.steps * 1e3
        out = {
            "metric": "piecewise-poly layer fwd+bwd samples/s",
            "value": round(world * B * args.steps / dt, 1),
            "unit": "samples/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(ms, 4),
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {
                "workload": "BASELINE configs[1]: net=small width=250 n=3 hidden.layers=3 segments=2 "
                            "continuous; fwd+bwd of the 5 polynomial layers (+LayerNorm between), "
                            "loss=<y,gy>, synthetic char batches",
                "rows_per_gpu": B, "global_rows": world * B,
                "parallelism": f"dp{world}" if world > 1 else "single",
                "launch": "hipgraph" if graph is not None else "eager",
                "grad_allreduce": "rccl, per-parameter, overlapped with backward" if world > 1 else None,
            },
        }
        if train_cps is not None:
            out["train_chars_per_s"] = round(train_cps, 1)  # cross entropy + Adam (graph replay at N=1)
        hb = step_bytes(B)
        out["hbm_roofline"] = {"algorithmic_bytes_per_step": hb,
                               "achieved_GBs": round(hb / (dt / args.steps) * 1e-9, 1),
                               "frac_of_8TBs": round(hb / (dt / args.steps) / (PEAK_HBM_GBS * 1e9), 4),
                               "note": "segments=2 layers are MFMA-bound (SURVEY 8d): structurally low"}
        if world == 1:
            out["roofline"] = roofline_for_hidden_layer(B, dev)
            if not args.no_cpu_baseline:
                out["cpu_baseline"] = cpu_baseline(args.cpu_rows)
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
