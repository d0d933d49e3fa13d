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


class FlatGradBucket:
    """All gradients of a replica as views of ONE flat buffer, so that the exchange is a single
    in-place all-reduce of that buffer: no concatenation before it, no copy back after it.

    * polynomial layers (permuted leaves on the GPU): the layer's backward writes its weight
      gradient straight into the layer's slice (functional._grad_out) and autograd takes that
      tensor as `p.grad` -- the bucket costs them nothing;
    * every other parameter: `p.grad` is pre-set to its (zeroed) slice and autograd accumulates
      into it in place.

    Per step: `prepare()` before the forward/backward (also inside a HIP-graph capture: the
    replay then reproduces the same in-place writes), `all_reduce()` after it.  `p.grad` keeps
    the parameter's strides (the optimizer kernels walk p, g, m in storage order)."""

    def __init__(self, params: Iterable[torch.nn.Parameter], process_group=None,
                 average: bool = True):
        from . import functional as PF
        self.params = [p for p in params if p.requires_grad]
        self.group = process_group
        self.world = dist.get_world_size(process_group) if dist.is_initialized() else 1
        self.scale = (1.0 / self.world) if average else 1.0
        if not self.params:
            raise ValueError("FlatGradBucket: no parameter requires a gradient")
        dev = self.params[0].device
        total = sum(p.numel() for p in self.params)
        self.flat = torch.zeros(total, device=dev, dtype=torch.float32)
        self.views = []    # p.grad-shaped views (logical shape, the parameter's strides)
        self.direct = []   # True: the layer's backward writes the slice itself
        self._keys = []
        self._slices = []
        off = 0
        for p in self.params:
            n = p.numel()
            phys = _physical_view(p.data)
            if not phys.is_contiguous():
                raise ValueError("FlatGradBucket: parameter without a storage-order view")
            sl = self.flat[off:off + n].view(phys.shape)
            permuted = phys.shape != p.shape or phys.stride() != p.stride()
            self.views.append(sl.permute(2, 0, 1) if permuted else sl)
            self.direct.append(bool(permuted and p.is_cuda and p.dim() == 3))
            self._slices.append(sl)
            off += n
        self._generic = [v for v, d in zip(self.views, self.direct) if not d]
        self.register()

    def register(self) -> None:
        """Tell the layers' backward where their weight gradients go (undone by release())."""
        from . import functional as PF
        for p, sl, d in zip(self.params, self._slices, self.direct):
            if d and p.data_ptr() not in self._keys:
                PF._grad_out[p.data_ptr()] = sl
                self._keys.append(p.data_ptr())

    def prepare(self) -> None:
        """Before the step's backward: slices that autograd accumulates into start at zero."""
        if self._generic:
            torch._foreach_zero_(self._generic)
        for p, v, d in zip(self.params, self.views, self.direct):
            p.grad = None if d else v

    def collect(self) -> None:
        """After the backward: every p.grad must BE its slice.  A gradient autograd produced
        elsewhere (cloned instead of taken, a layer run through another path) is copied in."""
        for p, v in zip(self.params, self.views):
            g = p.grad
            if g is None:
                v.zero_()
                p.grad = v
            elif g.data_ptr() != v.data_ptr() or g.stride() != v.stride():
                v.copy_(g)
                p.grad = v

    def all_reduce(self, async_op: bool = False):
        """One collective over the whole bucket (sum, then 1/world), in place."""
        if self.world <= 1:
            return None
        work = dist.all_reduce(self.flat, op=dist.ReduceOp.SUM, group=self.group, async_op=async_op)
        if async_op:
            return work
        if self.scale != 1.0:
            self.flat.mul_(self.scale)
        return None

    def finish(self, work) -> None:
        """Wait for an async all_reduce() and apply the 1/world scale."""
        if work is not None:
            work.wait()
            if self.scale != 1.0:
                self.flat.mul_(self.scale)

    def release(self) -> None:
        from . import functional as PF
        for k in self._keys:
            PF._grad_out.pop(k, None)
        self._keys.clear()


class GraphedGradAllReduce:
    """Gradient exchange for a step whose forward+backward is replayed from a HIP graph: the
    gradients live in the slices of a FlatGradBucket (the graph's static gradient buffers ARE the
    bucket), so after each replay the exchange is one in-place all-reduce and one scale -- no
    concatenation, no copy back.  No hooks (nothing of autograd runs at replay).

        red = GraphedGradAllReduce(params)
        def compute(): red.prepare(); loss(...).backward(); red.collect()    # captured once
        ... graph.replay(); red()                                             # every step
    """

    def __init__(self, params: Iterable[torch.nn.Parameter], process_group=None,
                 small_numel: int = 65536, average: bool = True):
        self.bucket = FlatGradBucket(params, process_group, average)
        self.world = self.bucket.world

    def prepare(self) -> None:
        self.bucket.prepare()

    def collect(self) -> None:
        self.bucket.collect()

    def __call__(self) -> None:
        self.bucket.all_reduce()


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
