"""CPU, world_size 2, gloo: the gradient all-reduce that bench.py / the DDP harness
use for N > 1 (the layers themselves are HIP-only, so the replicas here are small
torch modules with the same parameter layout trick: a permuted, non-contiguous leaf)."""
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp
import torch.nn as nn


class PermutedLinear(nn.Module):
    """y = x @ w.sum(-1).T with w stored like the polynomial layers' parameter:
    logical [O, I, K], physical [I, K, O]."""

    def __init__(self, I, O, K):
        super().__init__()
        self.w = nn.Parameter(torch.randn(I, K, O).permute(2, 0, 1))

    def forward(self, x):
        return x @ self.w.sum(-1).t()


def build():
    torch.manual_seed(0)
    return nn.Sequential(PermutedLinear(6, 300, 40), nn.Tanh(), nn.Linear(300, 5))


def data(rank):
    g = torch.Generator().manual_seed(10 + rank)
    return torch.randn(8, 6, generator=g), torch.randn(8, 5, generator=g)


def worker(rank, world, port, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from language_interpolation_amd.parallel import GradAllReducer, broadcast_parameters

    torch.manual_seed(123 + rank)  # replicas start different; broadcast must fix that
    net = nn.Sequential(PermutedLinear(6, 300, 40), nn.Tanh(), nn.Linear(300, 5))
    broadcast_parameters(net)
    red = GradAllReducer(net.parameters(), small_numel=4096)
    for _ in range(2):  # two steps: state must reset between them
        for p in net.parameters():
            p.grad = None
        x, t = data(rank)
        ((net(x) - t) ** 2).sum().backward()
        red.finish()
    if rank == 0:
        torch.save({"grads": [p.grad.clone() for p in net.parameters()],
                    "params": [p.detach().clone() for p in net.parameters()],
                    "strides": [p.grad.stride() for p in net.parameters()]}, out)
    dist.destroy_process_group()


def free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def test_grad_allreduce_matches_mean_of_per_rank_grads(tmp_path):
    out = str(tmp_path / "r0.pt")
    mp.spawn(worker, args=(2, free_port(), out), nprocs=2, join=True)
    got = torch.load(out, weights_only=True)
    # single-process reference: same start weights as rank 0 (seed 123), mean of both ranks' grads
    torch.manual_seed(123)
    net = nn.Sequential(PermutedLinear(6, 300, 40), nn.Tanh(), nn.Linear(300, 5))
    for a, b in zip(net.parameters(), got["params"]):
        assert torch.equal(a.detach(), b)
    want = [torch.zeros_like(p) for p in net.parameters()]
    for rank in range(2):
        for p in net.parameters():
            p.grad = None
        x, t = data(rank)
        ((net(x) - t) ** 2).sum().backward()
        for acc, p in zip(want, net.parameters()):
            acc += p.grad / 2
    for g, w_, st, p in zip(got["grads"], want, got["strides"], net.parameters()):
        assert torch.allclose(g, w_, rtol=1e-5, atol=1e-6)
        assert tuple(st) == tuple(p.stride())  # layout of the permuted leaf is kept


def graphed_worker(rank, world, port, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from language_interpolation_amd.parallel import GraphedGradAllReduce, broadcast_parameters

    torch.manual_seed(123 + rank)
    net = nn.Sequential(PermutedLinear(6, 300, 40), nn.Tanh(), nn.Linear(300, 5))
    broadcast_parameters(net)
    red = GraphedGradAllReduce(list(net.parameters()), small_numel=4096)
    flat = red.bucket.flat
    inplace = True
    for _ in range(2):   # (what a graph replay leaves behind: fresh gradients in the same buffers)
        red.prepare()
        x, t = data(rank)
        ((net(x) - t) ** 2).sum().backward()
        red.collect()
        red()
        for p, v in zip(red.bucket.params, red.bucket.views):
            inplace = inplace and p.grad.data_ptr() == v.data_ptr() and \
                flat.data_ptr() <= p.grad.data_ptr() < flat.data_ptr() + flat.numel() * 4
    if rank == 0:
        torch.save({"grads": [p.grad.clone() for p in net.parameters()],
                    "strides": [p.grad.stride() for p in net.parameters()],
                    "inplace": inplace,
                    "flat": flat.clone()}, out)
    dist.destroy_process_group()


def test_graphed_grad_allreduce_matches_mean_of_per_rank_grads(tmp_path):
    """The exchange bench.py uses for N > 1 when the step is a HIP graph: the gradients are
    views of one flat buffer, all-reduced in place after the replay."""
    out = str(tmp_path / "g0.pt")
    mp.spawn(graphed_worker, args=(2, free_port(), out), nprocs=2, join=True)
    got = torch.load(out, weights_only=True)
    torch.manual_seed(123)
    net = nn.Sequential(PermutedLinear(6, 300, 40), nn.Tanh(), nn.Linear(300, 5))
    want = [torch.zeros_like(p) for p in net.parameters()]
    for rank in range(2):
        for p in net.parameters():
            p.grad = None
        x, t = data(rank)
        ((net(x) - t) ** 2).sum().backward()
        for acc, p in zip(want, net.parameters()):
            acc += p.grad / 2
    for g, w_, st, p in zip(got["grads"], want, got["strides"], net.parameters()):
        assert torch.allclose(g, w_, rtol=1e-5, atol=1e-6)
        assert tuple(st) == tuple(p.stride())  # strides of the permuted leaf kept
    # every gradient IS a slice of the one flat buffer (no concatenation, no copy back), laid
    # out in storage order: the flat buffer is the gradients' physical views end to end
    assert got["inplace"]
    phys = [g.permute(1, 2, 0) if g.dim() == 3 else g for g in got["grads"]]
    assert all(v.is_contiguous() for v in phys)
    assert torch.equal(got["flat"], torch.cat([v.reshape(-1) for v in phys]))


def test_bench_spawns_its_own_ranks_and_reports_failures(capsys):
    """`python bench.py --gpus N` without torchrun: bench.spawn_ranks starts N ranks, passes
    rank 0's JSON line through and returns non-zero when a rank does not finish."""
    import json
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    import bench
    probe = os.path.join(os.path.dirname(os.path.abspath(__file__)), "_rank_probe.py")
    assert bench.spawn_ranks(2, [], script=probe, need_gpus=False) == 0
    line = json.loads(capsys.readouterr().out.strip().splitlines()[-1])
    assert line == {"n_ranks": 2, "sum": 3.0, "world_env": 2}
    assert bench.spawn_ranks(2, ["1"], script=probe, need_gpus=False) == 1   # rank 1 exits 7
    # asking for more GPUs than are visible fails before anything is started
    if torch.cuda.device_count() < 64:
        assert bench.spawn_ranks(64, [], need_gpus=True) == 2
