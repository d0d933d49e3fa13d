"""The gradient exchange of bench.py's N > 1 path on the device, in a one-rank RCCL group: the
collective is the identity there, so the flat-bucket gather / scale / copy-back on the real
(permuted) gradient buffers is what is checked.  (World size 2 on CPU: test_parallel_gloo.py.)"""
import socket

import pytest
import torch
import torch.distributed as dist

pytestmark = pytest.mark.gpu


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def test_graphed_grad_allreduce_on_device_buffers():
    import language_interpolation_amd as lia
    from language_interpolation_amd.parallel import GraphedGradAllReduce

    torch.cuda.set_device(0)
    dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{_free_port()}", rank=0,
                            world_size=1)
    try:
        torch.manual_seed(3)
        net = lia.HighOrderMLP(layer_type="continuous", n=3, in_width=16, out_width=24,
                               hidden_layers=1, hidden_width=64, in_segments=8, out_segments=2,
                               hidden_segments=2, normalization=lia.LazyLayerNormLastDim,
                               device="cuda")
        x = torch.rand(300, 16, device="cuda") * 2 - 1
        net(x).square().sum().backward()
        params = list(net.parameters())
        before = [p.grad.clone() for p in params]
        ptrs = [p.grad.data_ptr() for p in params]
        red = GraphedGradAllReduce(params)
        red.world, red.scale = 2, 0.5   # take the N > 1 branch; a one-rank sum is the identity
        red()
        torch.cuda.synchronize()
        for p, b, ptr in zip(params, before, ptrs):
            assert p.grad.data_ptr() == ptr              # in place: a graph's static buffers stay
            assert p.grad.stride() == b.stride()
            assert torch.equal(p.grad, b * 0.5)
    finally:
        dist.destroy_process_group()
