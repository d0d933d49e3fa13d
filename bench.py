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


RAVEN_HIST = os.path.join(ROOT, "tests", "golden", "raven_byte_hist.json")


def raven_char_probs():
    """Byte probabilities of the reference's own training text (books/the_raven.txt: 2,822,602
    ASCII characters, 90 distinct, 24 % spaces) -- SURVEY.md 8d's distribution for the
    character-grid input layers.  A committed histogram (tests/golden/make_raven_hist.py), not
    the book: /root/reference does not exist on the GPU box."""
    with open(RAVEN_HIST) as f:
        counts = torch.tensor(json.load(f)["counts"], dtype=torch.float64)
    return counts / counts.sum()


def char_batch(B, I, gen):
    """x = (c - 63.5)/64 (single_text_dataset.py:15-16) with c drawn i.i.d. from the byte
    histogram of the reference's text."""
    c = torch.multinomial(raven_char_probs(), B * I, replacement=True, generator=gen).view(B, I)
    return (c.float() - 64 + 0.5) / 64


def raven_like_text(nchar, seed):
    """Bytes drawn i.i.d. from the same histogram (the train chars/s figure gathers its
    character windows from this buffer on the device)."""
    g = torch.Generator().manual_seed(seed)
    c = torch.multinomial(raven_char_probs(), nchar, replacement=True, generator=g)
    return bytes(c.to(torch.uint8).tolist())


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


def time_layer(B, disc, n, S, I, O, dev, dist_kind="normal", first=False, iters=20,
               prepacked=False, table=0):
    """Seconds per launch of the three entry points of one layer (auto methods, preallocated
    outputs and workspaces: the timed region is exactly the library's launches).
    table = P > 0: a first layer on character inputs in its table form (include/pp_hip.h,
    pp_grid_table_f32): forward = table build + the n = 1 forward on it, weight gradient = the
    n = 1 weight gradient + the map back to the layer's basis."""
    from language_interpolation_amd import _lib
    from language_interpolation_amd import functional as PF

    K = PF.num_weights(n, S, disc)
    gen = torch.Generator().manual_seed(1)
    x = (char_batch(B, I, gen) if dist_kind == "char" else torch.randn(B, I, generator=gen)).to(dev)
    w_t = ((torch.rand(I, K, O, generator=gen) * 2 - 1) / I).to(dev)
    gy = (torch.randn(B, O, generator=gen) / O).to(dev)
    spec = PF.make_spec(n, S, disc)
    lib = _lib.load()
    d = int(disc)
    y = torch.empty(B, O, device=dev)
    gx = torch.empty(B, I, device=dev)
    gw = torch.empty(I, K, O, device=dev)
    nbytes = lib.pp_workspace_bytes(B, I, O, n, S, d, 2.0, 0)
    ws = torch.empty(max(nbytes, 16), dtype=torch.uint8, device=dev)
    fbytes = lib.pp_forward_workspace_bytes(B, I, O, n, S, d, 2.0, 0)
    fws = torch.empty(max(fbytes, 16), dtype=torch.uint8, device=dev)
    xbytes = lib.pp_backward_input_workspace_bytes(B, I, O, n, S, d, 2.0, 0)
    xws = torch.empty(max(xbytes, 16), dtype=torch.uint8, device=dev)
    nodes = spec.node_array()
    st = PF._stream(dev)
    P = PF._ptr

    # the same entry points, methods (auto) and workspaces the module's autograd function uses
    def fwd():
        lib.pp_forward_ws_f32(P(x), P(w_t), P(y), None, P(fws), fbytes, B, I, O, n, S, d, 2.0,
                              nodes, 1.0, 0, st)

    def bwd_x():
        lib.pp_backward_input_ws_f32(P(x), P(gy), P(w_t), P(gx), P(xws), xbytes, B, I, O, n, S, d,
                                     2.0, nodes, 1.0, 0, st)

    def bwd_w():
        lib.pp_backward_weight_f32(P(x), P(gy), P(gw), P(ws), nbytes, B, I, O, n, S, d, 2.0, nodes,
                                   1.0, 0, st)

    methods = [lib.pp_auto_method(p, B, I, O, n, S, d, 2.0) for p in range(3)]
    if prepacked and methods[1] == _lib.BWD_X_FP16X2:
        # the way a training step runs them: weights pre-split once per step
        # (pp_pack_weights_multi_f32), forward and backward-input are one kernel each (just
        # above the matrix-core threshold the forward is the gather kernel and only the
        # backward-input reads the pre-split buffer)
        packed = PF.pack_weights(w_t, spec, B)
        pk = packed.numel()
        PRE = 0x100  # PP_WS_PREPACKED

        if methods[0] == _lib.FWD_FP16X2:
            def fwd():  # noqa: F811
                lib.pp_forward_ws_f32(P(x), P(w_t), P(y), None, P(packed), pk, B, I, O, n, S, d,
                                      2.0, nodes, 1.0, _lib.FWD_FP16X2 | PRE, st)

        def bwd_x():  # noqa: F811
            lib.pp_backward_input_ws_f32(P(x), P(gy), P(w_t), P(gx), P(packed), pk, B, I, O, n, S,
                                         d, 2.0, nodes, 1.0, _lib.BWD_X_FP16X2 | PRE, st)
    if table:
        T = torch.empty(I, table, O, device=dev)
        GT = torch.empty(I, table, O, device=dev)
        tfb = lib.pp_forward_workspace_bytes(B, I, O, 1, table, 1, 2.0, 0)
        tfws = torch.empty(max(tfb, 16), dtype=torch.uint8, device=dev)
        twb = lib.pp_workspace_bytes(B, I, O, 1, table, 1, 2.0, 0)
        tws = torch.empty(max(twb, 16), dtype=torch.uint8, device=dev)
        n1 = PF.make_spec(1, table, True).node_array()
        methods = [lib.pp_auto_method(p, B, I, O, 1, table, 1, 2.0) for p in range(3)]

        def fwd():  # noqa: F811
            lib.pp_grid_table_f32(P(w_t), P(T), I, O, n, S, d, 2.0, nodes, 1.0, table, st)
            lib.pp_forward_ws_f32(P(x), P(T), P(y), None, P(tfws), tfb, B, I, O, 1, table, 1, 2.0,
                                  n1, 1.0, 0, st)

        def bwd_w():  # noqa: F811
            lib.pp_backward_weight_f32(P(x), P(gy), P(GT), P(tws), twb, B, I, O, 1, table, 1, 2.0,
                                       n1, 1.0, 0, st)
            lib.pp_grid_table_grad_f32(P(GT), P(gw), I, O, n, S, d, 2.0, nodes, 1.0, table, st)
    times = {"fwd": time_kernel(fwd, iters), "bwd_w": time_kernel(bwd_w, iters)}
    if not first:
        times["bwd_x"] = time_kernel(bwd_x, iters)
    return times, methods, K


def roofline_for_hidden_layer(B, dev):
    """Times the three kernels of one hidden layer (continuous n=3 250->250 S=2) and
    reports the roofline of the one that takes longest."""
    n, S, I, O = 3, 2, 250, 250
    t, methods, K = time_layer(B, False, n, S, I, O, dev, prepacked=True)
    fname = {1: "pp_fwd_gather_kernel", 2: "pp_fwd_dense_kernel",
             3: "pp_fwd_bf3_kernel<bf16x3>(+pack)",
             4: "pp_fwd_bf3_kernel<fp16x2>", 5: "pp_fwd_slab_kernel"}[methods[0]]
    xname = {1: "pp_bwd_input_gather_kernel", 2: "pp_bwd_input_dense_kernel",
             3: "pp_bwd_x_bf3_kernel<bf16x3>(+pack)",
             4: "pp_bwd_x_bf3_kernel<fp16x2>"}[methods[1]]
    wname = {1: "pp_bwd_weight_atomic_kernel", 2: "pp_bwd_weight_segreduce_kernel(+sort)",
             3: "pp_bwd_weight_dense_kernel(+reduce)",
             4: "pp_bwd_w_bf3_kernel<bf16x3>(+pack,+reduce)",
             5: "pp_bwd_w_bf3_kernel<fp16x2>(+pack,+reduce)"}[methods[2]]
    times = {fname: t["fwd"], xname: t["bwd_x"], wname: t["bwd_w"]}
    name = max(times, key=times.get)
    flops = 2.0 * B * I * O * n  # algorithmic flops of one pass (6*B*I*O*n for three)
    achieved = flops / times[name] * 1e-12
    # HBM-side bytes per launch of the dominant entry point: measured offline with rocprofv3
    # --pmc (separate passes; profiles/*_traffic.json says how), valid for the shape and
    # method it was taken on (weight gradient and backward-input, fp16x2, 16384 rows)
    traffic = None
    tpath = os.path.join(ROOT, "profiles", "r03_k_traffic.json")
    if not os.path.exists(tpath):
        tpath = os.path.join(ROOT, "profiles", "r02_traffic.json")
    if B == 16384 and os.path.exists(tpath):
        with open(tpath) as f:
            measured = json.load(f)
        if name == wname and methods[2] == 5:
            traffic = measured["traffic_bytes"]
        elif name == xname and methods[1] == 4:
            traffic = measured["backward_input"]["traffic_bytes"]
        elif name == fname and methods[0] == 4 and "forward" in measured:
            traffic = measured["forward"]["traffic_bytes"]
    return {
        "bound": "mfma", "achieved": round(achieved, 3), "peak": PEAK_F32_TFLOPS,
        "unit": "TFLOP/s", "frac": round(achieved / PEAK_F32_TFLOPS, 4), "traffic": traffic,
        "kernel": name, "layer": "continuous n=3 250->250 S=2 (cfg2 hidden)", "rows": B,
        "algorithmic_flops_per_launch": flops,
        "launch_us": {k: round(v * 1e6, 2) for k, v in times.items()},
        "auto_methods": methods,
        "arith": "fp16 MFMA on 2-way split fp32 operands (3 cross terms, ~22 significant bits), "
                 "fp32 accumulate; `peak` is the fp32 matrix peak the fp32 contract is priced "
                 "against -- the kernel issues 5x the algorithmic flops as fp16 MFMAs "
                 "(3 cross terms x K/n = 5/3 dense padding); forward and backward-input timed on "
                 "weights pre-split once per step (pp_pack_weights_multi_f32), as the step runs them",
    }


def roofline_hbm(B, dev):
    """The HBM-bound layers of BASELINE (many segments: SURVEY.md 8d rows 'cfg3 hidden' and
    'cfg4 input', plus the cfg3 layer with the reference's n_hidden quirk), timed live per entry
    point at the bench batch and at the reference batch:
    algorithmic bytes (every operand touched once per pass, dense gw) over the sum of the
    entry-point times, as a fraction of 8 TB/s."""
    out = []
    for name, disc, n, S, I, O, kind, first, ref_rows, table in (
            ("cfg3 hidden (discontinuous n=1 200->200 S=150)", True, 1, 150, 200, 200, "normal", False, 1000, 0),
            ("cfg4 input (discontinuous n=2 16->1000 S=128)", True, 2, 128, 16, 1000, "char", True, 256, 0),
            # the same layer told that its inputs are characters (layer.assume_input_grid(128)):
            # table build + n = 1 kernels + map back, all inside the timed launches
            ("cfg4 input in table form (inputs declared on the 128-point character grid)", True, 2, 128,
             16, 1000, "char", True, 256, 128),
            # the same cfg3 layer as the reference's select_network really builds it: hidden layers get
            # n_hidden = cfg.net.n_in = 2 (networks.py:870; SURVEY.md 8a "reference quirk"), K = 300
            ("cfg3 hidden as select_network builds it (discontinuous n=2 200->200 S=150)", True, 2, 150,
             200, 200, "normal", False, 1000, 0)):
        for rows in (B, ref_rows):
            t, methods, K = time_layer(rows, disc, n, S, I, O, dev, kind, first, iters=10,
                                       table=table)
            W = I * K * O
            by = 4 * ((2 * rows * I + 2 * rows * O + 2 * W) if first
                      else (3 * rows * I + 2 * rows * O + 3 * W))
            tot = sum(t.values())
            out.append({"layer": name, "rows": rows, "algorithmic_bytes": by,
                        "launch_us": {k: round(v * 1e6, 2) for k, v in t.items()},
                        "achieved": round(by / tot * 1e-9, 1), "peak": PEAK_HBM_GBS, "unit": "GB/s",
                        "frac": round(by / tot / (PEAK_HBM_GBS * 1e9), 4), "auto_methods": methods})
    return out


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
        from language_interpolation_amd import functional as PF
        return PF.cross_entropy_accuracy(self.model(batch[0]), batch[1].flatten())


def train_chars_per_s(net, dev, B, steps, world, reducer, graph, optimizer="sparse_lion"):
    """The other half of BASELINE's metric: optimizer steps of the same network on character
    windows gathered on the device: cross entropy and the reference's default optimizer
    (config/high_order_interpolation.yaml:70 `optimizer: sparse_lion`, lr 1e-5; one fused
    kernel per step, train.Lion) or Adam.  One process: the step is replayed from a HIP graph
    (train.GraphedTrainStep); several: eager launches with the RCCL all-reduce of the
    gradients overlapping the backward."""
    from language_interpolation_amd.train import (CharWindows, GraphedTrainStep, Lion,
                                                  declare_character_inputs)

    data = CharWindows(raven_like_text(1 << 20, seed=7), 16, device=dev)
    if data.ascii_only:
        # as train.main does: the data set vouches for the character grid, the input layer
        # runs in table form (this figure only; the headline step is timed before, on float x)
        declare_character_inputs(net)
    gen = torch.Generator(device=dev).manual_seed(11)
    module = _TrainModule(net)
    graph = graph and reducer is None
    if optimizer == "adam":
        opt = torch.optim.Adam(net.parameters(), lr=1e-4, capturable=graph)
    else:
        opt = Lion(net.parameters(), lr=1e-5, weight_decay=0.0, sparse=optimizer == "sparse_lion")
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
    """The CPU oracle on this box's host cores, same network: the reference formulation
    (gather [B,I,O,n] + einsum + autograd) is `value`; the dense-expansion formulation
    (Phi @ W^T matmul, SURVEY.md 8d "fairer CPU baseline") is reported beside it."""
    from oracle import pp_oracle as po

    # the GPU box gives a one-GPU job 16 CPUs (os.cpu_count() there shows the whole machine)
    cores = max(1, min(16, os.cpu_count() or 1))
    torch.set_num_threads(cores)
    gen = torch.Generator().manual_seed(0)
    x = char_batch(rows, 16, gen)
    gy = torch.randn(rows, 128, generator=gen) / 128

    def timed(net):
        def step():
            for p in net.parameters():
                p.grad = None
            y = net(x)
            y.backward(gy)

        step()
        t0 = time.perf_counter()
        for _ in range(steps):
            step()
        return (time.perf_counter() - t0) / steps

    torch.manual_seed(0)
    dt = timed(po.OracleHighOrderMLP(normalization=po.OracleLazyLayerNormLastDim, **NET))
    out = {"value": round(rows / dt, 1), "unit": "samples/s", "cores": cores, "kind": "port",
           "sample": f"{steps} fwd+bwd steps of the same 5-layer net at {rows} rows "
                     f"(torch CPU oracle, reference gather+einsum formulation), {dt:.3f} s/step"}
    try:
        torch.manual_seed(0)
        dt2 = timed(po.OracleHighOrderMLP(normalization=po.OracleLazyLayerNormLastDim,
                                          formulation="dense", **NET))
        out["dense_formulation"] = {"value": round(rows / dt2, 1), "unit": "samples/s",
                                    "sample": f"same net and rows, expanded basis Phi @ W^T "
                                              f"(oracle piecewise_forward_dense), {dt2:.3f} s/step"}
    except TypeError:
        pass
    return out


def spawn_ranks(n, argv, script=None, need_gpus=True):
    """`python bench.py --gpus N` without a launcher: start N copies of this script as child
    processes (one rank per GPU, env as torch.distributed.run would set it) BEFORE this process
    touches a GPU, pass rank 0's JSON line through, and fail loudly when a rank does."""
    import socket
    import subprocess

    ndev = torch.cuda.device_count()  # (counting devices does not initialise the GPU)
    if need_gpus and ndev < n:
        print(f"bench.py: --gpus {n} asked but only {ndev} GPU(s) are visible", file=sys.stderr)
        return 2
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, script or os.path.abspath(__file__)] + argv, env=env,
                                      stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL))
    out0 = procs[0].communicate()[0].decode()
    codes = [procs[0].returncode] + [p.wait() for p in procs[1:]]
    if any(codes):
        print(f"bench.py: rank exit codes {codes}: fewer ranks finished than asked", file=sys.stderr)
        return 1
    sys.stdout.write(out0)
    return 0


def trace(msg):
    if os.environ.get("PP_BENCH_TRACE"):
        print(f"[rank {os.environ.get('RANK', '0')}] {msg}", file=sys.stderr, flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--batch", type=int, default=16384, help="rows per GPU per step")
    ap.add_argument("--no-graph", action="store_true", help="eager launches instead of a HIP graph")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-overlap-probe", action="store_true",
                    help="N > 1: keep graph replay + flat all-reduce without timing the hook path")
    ap.add_argument("--no-hbm-layers", action="store_true", help="skip the roofline_hbm object")
    ap.add_argument("--no-char-table", action="store_true",
                    help="skip the extra timing of the step with the input layer in table form")
    ap.add_argument("--cpu-rows", type=int, default=4096)
    ap.add_argument("--train-optimizer", choices=("sparse_lion", "lion", "adam"),
                    default="sparse_lion",
                    help="optimizer of the train chars/s figure (the reference's default: sparse_lion)")
    ap.add_argument("--train-steps", type=int, default=20,
                    help="optimizer steps for the extra train chars/s figure (0 = skip)")
    args = ap.parse_args()

    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        sys.exit(spawn_ranks(args.gpus, sys.argv[1:]))

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        if rank == 0:
            print(f"bench.py: --gpus {args.gpus} but the launcher started WORLD_SIZE={world} ranks",
                  file=sys.stderr)
        sys.exit(2)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))  # = RCCL
        if dist.get_world_size() != world:
            sys.exit(3)
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    trace("process group up")

    import language_interpolation_amd as L
    from language_interpolation_amd.parallel import (GradAllReducer, GraphedGradAllReduce,
                                                     broadcast_parameters)

    torch.manual_seed(0)
    net = L.HighOrderMLP(normalization=L.LazyLayerNormLastDim, device=dev, **NET)
    broadcast_parameters(net)
    trace("parameters broadcast")
    B = args.batch
    gen = torch.Generator().manual_seed(100 + rank)
    x = char_batch(B, 16, gen).to(dev)
    gy = (torch.randn(B, 128, generator=gen) / 128).to(dev)
    params = list(net.parameters())

    # N > 1: every gradient is a slice of one flat buffer (the layers' backward kernels write
    # there directly), so the exchange is one in-place all-reduce
    flat_reduce = GraphedGradAllReduce(params) if world > 1 else None

    def compute():
        if flat_reduce is not None:
            flat_reduce.prepare()
        else:
            for p in params:
                p.grad = None
        y = net(x)
        y.backward(gy)
        if flat_reduce is not None:
            flat_reduce.collect()

    def capture():
        s = torch.cuda.Stream()
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):
            for _ in range(3):
                compute()
        torch.cuda.current_stream().wait_stream(s)
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        # (thread_local: RCCL's watchdog thread may query events while this thread captures)
        with torch.cuda.graph(g, capture_error_mode="thread_local" if world > 1 else "global"):
            compute()
        return g

    def timed(fn, steps):
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            fn()
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        if world > 1:
            t = torch.tensor([dt], device=dev, dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt = float(t.item())
        return dt

    timed_raw = timed

    # N = 1 and N > 1 run the SAME captured compute graph per rank; with N > 1 the gradients
    # (static buffers of the graph) are all-reduced after each replay, so weak-scaling
    # efficiency compares graph with graph.  --no-graph: eager launches, and for N > 1 the
    # per-parameter all-reduce started from autograd hooks so that it overlaps the backward.
    graph = None
    reducer = None
    launch = "eager"
    if not args.no_graph:
        try:
            graph = capture()
            launch = "hipgraph"
            trace("graph captured")
        except Exception as e:  # capture refused: measure eager, and say so
            print(f"HIP graph capture failed ({type(e).__name__}: {e}); running eager",
                  file=sys.stderr)
            graph = None
            torch.cuda.synchronize()
    if world > 1:
        # two ways to run the step, both timed outside the measured region, the faster kept
        # (the ranks agree through a MAX all-reduce of the probe times):
        #  (a) HIP graph replay, then ONE in-place all-reduce of the flat gradient bucket;
        #  (b) eager launches with the all-reduce of each parameter started from its autograd
        #      hook, i.e. overlapped with the rest of the backward (what DDP does in the
        #      reference: examples/high_order_interpolation.py:154-160).
        def run_flat():
            (graph.replay if graph is not None else compute)()
            flat_reduce()

        hook_reducer = [None]

        def compute_hooks():
            for p in params:
                p.grad = None
            y = net(x)
            y.backward(gy)

        def run_hooks():
            compute_hooks()
            hook_reducer[0].finish()

        run, launch = run_flat, ("hipgraph" if graph is not None else "eager") + \
            " + one in-place all-reduce of the flat gradient bucket after the step"
        if not args.no_overlap_probe:
            for _ in range(3):
                run_flat()
            t_flat = timed_raw(run_flat, 10)
            flat_reduce.bucket.release()       # the hook path owns p.grad itself
            hook_reducer[0] = GradAllReducer(params)
            for _ in range(3):
                run_hooks()
            t_hooks = timed_raw(run_hooks, 10)
            if t_hooks < t_flat:
                run = run_hooks
                launch = "eager, per-parameter all-reduce overlapped with backward (autograd hooks)"
                reducer = hook_reducer[0]
            else:
                hook_reducer[0].remove()
                flat_reduce.bucket.register()   # the layers write into the bucket's slices again
    else:
        run = graph.replay if graph is not None else compute

    for _ in range(args.warmup):
        run()
    trace("warm-up done")
    # One process: the step runs from the HIP graph or from eager launches, whichever this box
    # replays faster (at 16384 rows the replay's per-node dependency gaps cost more than the
    # launches it saves; at a few hundred rows the graph wins by a wide margin).  Both are
    # probed outside the timed region, the faster one is timed for K steps, the other one's
    # figure is reported beside it.
    other_dt = None
    if graph is not None and world == 1:
        for _ in range(3):
            compute()
        probe = max(5, min(20, args.steps))
        t_graph, t_eager = timed(graph.replay, probe), timed(compute, probe)
        if t_eager < t_graph:
            run, launch = compute, "eager (faster than the HIP graph replay on this box at this size)"
        other = graph.replay if run is compute else compute
    dt = timed(run, args.steps)
    trace("timed region done")
    eager_dt = graph_dt = None
    if graph is not None and world == 1:
        other_dt = timed(other, args.steps)
        eager_dt, graph_dt = (dt, other_dt) if run is compute else (other_dt, dt)

    # The same step with the input layer told that its inputs are characters
    # (layer.assume_input_grid(128): table form, include/pp_hip.h pp_grid_table_f32).  Reported
    # beside the headline, which stays on the general kernels -- the statement about the inputs
    # is the caller's, the reference's layer has no such argument.
    table_dt = table_launch = None
    if world == 1 and not args.no_char_table:
        first_layer = net.model[0]
        first_layer.assume_input_grid(128)
        try:
            for _ in range(3):
                compute()
            tab_run, table_launch = compute, "eager"
            if graph is not None:
                g2 = capture()
                probe = max(5, min(20, args.steps))
                if timed(g2.replay, probe) < timed(compute, probe):
                    tab_run, table_launch = g2.replay, "hipgraph"
            for _ in range(args.warmup):
                tab_run()
            table_dt = timed(tab_run, args.steps)
        finally:
            first_layer.assume_input_grid(0)

    train_cps = None
    if args.train_steps > 0:
        if reducer is None and world > 1:
            reducer = GradAllReducer(params)
        train_cps = train_chars_per_s(net, dev, B, args.train_steps, world, reducer, not args.no_graph,
                                      args.train_optimizer)

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
            "arith": "fp32 in/out/accumulate; the segments=2 layers run fp16 MFMA on 2-way split "
                     "operands (~22 significant bits, 1e-5 parity bar), the rest plain fp32",
            "data": "synthetic: characters drawn i.i.d. from the byte histogram of the reference's "
                    "books/the_raven.txt (SURVEY 8d), random-init weights, gy ~ N(0,1)/O",
            "config": {
                "workload": "BASELINE configs[1]: net=small width=250 n=3 hidden.layers=3 segments=2 "
                            "continuous; fwd+bwd of the 5 polynomial layers (+LayerNorm between), "
                            "loss=<y,gy>, synthetic char batches",
                "rows_per_gpu": B, "global_rows": world * B,
                "parallelism": f"dp{world}" if world > 1 else "single",
                "launch": launch,
                "grad_allreduce": ("rccl, " + launch) if world > 1 else None,
            },
        }
        if eager_dt is not None:
            out["eager_ms_per_step"] = round(eager_dt / args.steps * 1e3, 4)
            out["hipgraph_ms_per_step"] = round(graph_dt / args.steps * 1e3, 4)
        if table_dt is not None:
            out["char_table"] = {
                "ms_per_step": round(table_dt / args.steps * 1e3, 4),
                "value": round(B * args.steps / table_dt, 1), "launch": table_launch,
                "note": "same step, input layer in table form (net.model[0].assume_input_grid(128): "
                        "inputs declared on the 128-point character grid; table build and the map "
                        "back of the gradient inside the step); not the headline"}
        if train_cps is not None:
            # cross entropy + optimizer step (graph replay at N=1)
            out["train_chars_per_s"] = round(train_cps, 1)
            out["train_optimizer"] = args.train_optimizer
            out["train_input_layer"] = "table form (the data set vouches for the character grid)"
        hb = step_bytes(B)
        out["hbm_roofline"] = {"algorithmic_bytes_per_step": hb,
                               "achieved_GBs": round(hb / (dt / args.steps) * 1e-9, 1),
                               "frac_of_8TBs": round(hb / (dt / args.steps) / (PEAK_HBM_GBS * 1e9), 4),
                               "note": "segments=2 layers are MFMA-bound (SURVEY 8d): structurally low; "
                                       "the HBM-bound layers are in roofline_hbm"}
        if world == 1:
            out["roofline"] = roofline_for_hidden_layer(B, dev)
            if not args.no_hbm_layers:
                out["roofline_hbm"] = roofline_hbm(B, dev)
            if not args.no_cpu_baseline:
                out["cpu_baseline"] = cpu_baseline(args.cpu_rows)
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
