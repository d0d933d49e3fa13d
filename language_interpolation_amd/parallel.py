"""Data-parallel gradient exchange for the piecewise-polynomial networks.

The path shards by rows (samples): y and gx are row-local, only the weight
gradient gw is a sum over rows.  One process per GPU; each rank
holds a full replica and its own rows; the single exchange step is an all-reduce
(sum) of every parameter gradient -- RCCL over xGMI when the process group's
backend is "nccl" (that IS RCCL on ROCm), gloo on CPU for the tests.

`GradAllReducer` launches the all-reduce of a parameter's gradient from its
post-accumulate-grad hook, i.e. as soon as autograd has produced it, so the
collective of layer L overlaps the backward kernels of layers < L (RCCL runs on
its own stream).  Large gradients (the polynomial layers' 0.3-50 MB tensors) go
out individually, in place, on their physical storage; small ones are packed
into one flat bucket sent at `finish()`.  The mean (1/world) is folded into a
single post-scale.

In the reference this step is implicit (Lightning's default DDP strategy,
/root/reference/examples/high_order_interpolation.py:154-160); there is no
explicit collective call site to mirror.
"""
from __future__ import annotations

from typing import Iterable, List, Optional

import torch
import torch.distributed as dist


def _physical_view(t: torch.Tensor) -> torch.Tensor:
    """A contiguous view over t's storage (the polynomial layers keep w and w.grad
    as permuted views of an [I,K,O] block), or t itself when already contiguous."""
    if t.is_contiguous():
        return t
    if t.dim() == 3:
        v = t.permute(1, 2, 0)
        if v.is_contiguous():
            return v
    return t  # caller falls back to a copy


class GradAllReducer:
    def __init__(self, params: Iterable[torch.nn.Parameter], process_group=None,
                 small_numel: int = 65536, average: bool = True):
        self.params: List[torch.nn.Parameter] = [p for p in params if p.requires_grad]
        self.group = process_group
        self.world = dist.get_world_size(process_group) if dist.is_initialized() else 1
        self.small_numel = small_numel
        self.average = average
        self._pending = []  # (work, param, tensor_sent, needs_copy_back)
        self._small: List[torch.nn.Parameter] = []
        self._handles = []
        if self.world > 1:
            for p in self.params:
                self._handles.append(p.register_post_accumulate_grad_hook(self._hook))

    def _hook(self, p: torch.nn.Parameter) -> None:
        g = p.grad
        if g is None:
            return
        if g.numel() < self.small_numel:
            self._small.append(p)
            return
        v = _physical_view(g)
        copy_back = not v.is_contiguous()
        if copy_back:
            v = g.contiguous()
        work = dist.all_reduce(v, op=dist.ReduceOp.SUM, group=self.group, async_op=True)
        self._pending.append((work, p, v, copy_back))

    def finish(self) -> None:
        """Send the small bucket, wait for everything, apply the 1/world scale."""
        if self.world <= 1:
            return
        flat: Optional[torch.Tensor] = None
        if self._small:
            flat = torch.cat([p.grad.reshape(-1) for p in self._small])
            dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=self.group)
        scale = 1.0 / self.world if self.average else 1.0
        for work, p, v, copy_back in self._pending:
            work.wait()
            if copy_back:
                p.grad.copy_(v)
            if scale != 1.0:
                _physical_view(p.grad).mul_(scale)
        if flat is not None:
            if scale != 1.0:
                flat.mul_(scale)
            off = 0
            for p in self._small:
                n = p.grad.numel()
                p.grad.copy_(flat[off:off + n].view_as(p.grad))
                off += n
        self._pending.clear()
        self._small.clear()

    def remove(self) -> None:
        for h in self._handles:
            h.remove()
        self._handles.clear()


class GraphedGradAllReduce:
    """Gradient exchange for a step whose forward+backward is replayed from a HIP graph: the
    gradients then live in the graph's static buffers, so after each replay they are summed
    over the ranks through one flat bucket and scaled by 1/world, in place.  No hooks (nothing of
    autograd runs at replay)."""

    def __init__(self, params: Iterable[torch.nn.Parameter], process_group=None,
                 small_numel: int = 65536, average: bool = True):
        self.params = [p for p in params if p.requires_grad]
        self.group = process_group
        self.world = dist.get_world_size(process_group) if dist.is_initialized() else 1
        self.small_numel = small_numel
        self.scale = (1.0 / self.world) if average else 1.0

    def __call__(self) -> None:
        if self.world <= 1:
            return
        # ONE bucket: every gradient's storage-order view copied into a flat buffer, one
        # all-reduce, one scale, one multi-tensor copy back -- a collective per parameter
        # costs its launch latency five times a step for these 0.6-2 MB tensors.  (small_numel
        # is kept for the constructor's signature; everything shares the bucket.)
        views = []
        for p in self.params:
            g = p.grad
            if g is None:
                continue
            v = _physical_view(g)
            if not v.is_contiguous():  # no storage-order view: reduce a copy, write it back
                c = g.contiguous()
                dist.all_reduce(c, op=dist.ReduceOp.SUM, group=self.group)
                g.copy_(c.mul_(self.scale) if self.scale != 1.0 else c)
                continue
            views.append(v.reshape(-1))
        if not views:
            return
        flat = torch.cat(views)
        # plain stream-ordered collective: RCCL enqueues on its own stream and the current
        # stream waits for it
        dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=self.group)
        if self.scale != 1.0:
            flat.mul_(self.scale)
        torch._foreach_copy_(views, list(flat.split([v.numel() for v in views])))


def broadcast_parameters(module: torch.nn.Module, src: int = 0, process_group=None) -> None:
    """Make every replica start from rank `src`'s weights (what DDP does at wrap time)."""
    if not dist.is_initialized() or dist.get_world_size(process_group) <= 1:
        return
    for p in module.parameters():
        v = _physical_view(p.data)
        if v.is_contiguous():
            dist.broadcast(v, src=src, group=process_group)
        else:
            tmp = p.data.contiguous()
            dist.broadcast(tmp, src=src, group=process_group)
            p.data.copy_(tmp)
    # writes through .data do not bump the parameters' version counters, which key the
    # pre-split weight buffers of the step in flight: drop them
    from . import functional as PF
    PF._prepacked.clear()
