"""The gradient exchange of bench.py's N > 1 path on the device, in a one-rank RCCL group: the
collective is the identity there, so what is checked is that the real (permuted) gradient
buffers ARE the flat bucket's slices and that the scale acts on them in place.  (World size 2 on CPU: test_parallel_gloo.py.)"""
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
    """The layers' backward kernels write their weight gradients straight into the slices of the
    flat bucket (no copy in, none back), over several steps and under a HIP-graph capture; the
    one all-reduce + scale acts on those same buffers."""
    import language_interpolation_amd as lia
    from language_interpolation_amd import functional as PF
    from language_interpolation_amd.parallel import GraphedGradAllReduce

    torch.cuda.set_device(0)
    dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{_free_port()}", rank=0,
                            world_size=1)
    red = None
    try:
        torch.manual_seed(3)
        net = lia.HighOrderMLP(layer_type="continuous", n=3, in_width=16, out_width=24,
                               hidden_layers=1, hidden_width=64, in_segments=8, out_segments=2,
                               hidden_segments=2, normalization=lia.LazyLayerNormLastDim,
                               device="cuda")
        x = torch.rand(300, 16, device="cuda") * 2 - 1
        params = list(net.parameters())
        # reference gradients without any bucket
        net(x).square().sum().backward()
        want = [p.grad.clone() for p in params]
        for p in params:
            p.grad = None
        red = GraphedGradAllReduce(params)
        bucket = red.bucket
        assert all(bucket.direct) and len(PF._grad_out) == len(params)
        lo, hi = bucket.flat.data_ptr(), bucket.flat.data_ptr() + bucket.flat.numel() * 4

        def compute():
            red.prepare()
            net(x).square().sum().backward()
            red.collect()

        for _ in range(2):
            compute()
            for p, w_, v in zip(params, want, bucket.views):
                assert p.grad.data_ptr() == v.data_ptr() and lo <= p.grad.data_ptr() < hi
                assert p.grad.stride() == p.stride()
                # (same values; not the same bits where a small layer's gradient uses atomics)
                assert torch.allclose(p.grad, w_, rtol=1e-5, atol=1e-6 * float(w_.abs().max()))
        # captured: the replay writes the same slices
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            compute()
        torch.cuda.current_stream().wait_stream(side)
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, capture_error_mode="thread_local"):
            compute()
        bucket.flat.fill_(7.0)
        g.replay()
        bucket.world, bucket.scale = 2, 0.5   # take the N > 1 branch; a one-rank sum is the identity
        red()
        torch.cuda.synchronize()
        for p, w_, v in zip(params, want, bucket.views):
            assert p.grad.data_ptr() == v.data_ptr()
            assert torch.allclose(p.grad, w_ * 0.5, rtol=1e-5, atol=1e-6 * float(w_.abs().max()))
        # a gradient that already exists is accumulated into, never aliased
        for p in params:
            p.grad = None
        net(x).square().sum().backward()
        net(x).square().sum().backward()
        for p, w_ in zip(params, want):
            assert torch.allclose(p.grad, 2 * w_, rtol=1e-5, atol=2e-6 * float(w_.abs().max()))
    finally:
        if red is not None:
            red.bucket.release()
        dist.destroy_process_group()
    assert not PF._grad_out
