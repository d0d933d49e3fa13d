"""Minimal training harness around the HIP layer.

What the reference does with Lightning + Hydra
(/root/reference/language_interpolation/networks.py:121-132 ClassificationMixin,
:159-223 PredictionNetMixin, :750-954 select_network, :957-967 ASCIIPredictionNet;
examples/high_order_interpolation.py) reduced to what feeds the hot path:

* ``load_config``     reads the reference's yaml keys (``net.*``, ``optimizer.*``,
                      ``batch_size``, ``data.max_features`` ...), ``defaults:`` groups and
                      ``${a.b}`` interpolation, with PyYAML only;
* ``select_network``  ``net.model_type`` ``high_order`` and ``high_order_input``;
* ``ASCIIPredictionNet``  cross-entropy + accuracy = mean(argmax == y), optimizers
                      ``adam`` / ``lion`` / ``sparse_lion``;
* ``CharWindows``     text -> uint8 buffer ON THE DEVICE -> windows gathered on the GPU,
                      ``x = (c - 64 + 0.5)/64`` (single_text_dataset.py:15-16, :102-123
                      builds the same windows as Python lists);
* ``fit``             the loop, one process per GPU (gradients summed by
                      ``parallel.GradAllReducer`` when torch.distributed is initialised).

Everything outside that (checkpoints, text generation, Gutenberg download, schedulers'
plateau logic beyond lr decay) is the reference's control plane and stays out of scope.
"""
from __future__ import annotations

import argparse
import json
import os
import re
import time
from types import SimpleNamespace
from typing import Any, Dict, Iterable, Optional

import torch
import torch.nn as nn

from . import functional as PF
from .layers import (LazyLayerNormLastDim, MaxAbsNormalizationLast, MaxCenterNormalizationLast,
                     high_order_fc_layers)
from .networks import HighOrderMLP, LowOrderMLP


# ---------------------------------------------------------------------------
# configuration (the reference's yaml files, no Hydra)
# ---------------------------------------------------------------------------
class Config(SimpleNamespace):
    """Attribute access + ``get`` like the DictConfig the reference passes around."""

    def get(self, key, default=None):
        return getattr(self, key, default)

    def to_dict(self):
        return {k: v.to_dict() if isinstance(v, Config) else v for k, v in vars(self).items()}


def _wrap(obj):
    if isinstance(obj, dict):
        return Config(**{k: _wrap(v) for k, v in obj.items()})
    return obj


_INTERP = re.compile(r"\$\{([A-Za-z0-9_.]+)\}")
# PyYAML (YAML 1.1) reads `1e-5` as a string; the reference's optimizer files use that form
_FLOAT = re.compile(r"[-+]?(\d+\.?\d*|\.\d+)[eE][-+]?\d+")


def _resolve(tree: Dict[str, Any]) -> Dict[str, Any]:
    """``${net.features}``-style references, resolved against the root (a few passes)."""

    def lookup(path):
        node = tree
        for part in path.split("."):
            node = node[part]
        return node

    def visit(node):
        if isinstance(node, dict):
            return {k: visit(v) for k, v in node.items()}
        if isinstance(node, list):
            return [visit(v) for v in node]
        if isinstance(node, str):
            if _FLOAT.fullmatch(node):
                return float(node)
            m = _INTERP.fullmatch(node)
            if m:
                return lookup(m.group(1))
            return _INTERP.sub(lambda mm: str(lookup(mm.group(1))), node)
        return node

    for _ in range(4):
        tree = visit(tree)
    return tree


def _set_path(tree, dotted, value):
    node = tree
    parts = dotted.split(".")
    for part in parts[:-1]:
        node = node.setdefault(part, {})
    node[parts[-1]] = value


def load_config(path_or_dict, overrides: Iterable[str] = ()) -> Config:
    """A reference-style config: a root yaml whose ``defaults: [{group: name}, ...]`` pull
    ``<dir>/<group>/<name>.yaml`` under key ``group``; ``overrides`` are ``a.b=value``
    strings (``net=small`` switches a group) as on the reference's command line."""
    import yaml

    overrides = list(overrides)
    if isinstance(path_or_dict, dict):
        tree = json.loads(json.dumps(path_or_dict))
    else:
        root_dir = os.path.dirname(os.path.abspath(path_or_dict))
        with open(path_or_dict) as f:
            tree = yaml.safe_load(f) or {}
        groups = {}
        for entry in tree.pop("defaults", []) or []:
            if isinstance(entry, dict):
                groups.update({k: v for k, v in entry.items() if not k.startswith("override")})
        for ov in list(overrides):
            k, _, v = ov.partition("=")
            if k in groups and "." not in k:
                groups[k] = v
                overrides.remove(ov)
        for group, name in groups.items():
            with open(os.path.join(root_dir, group, f"{name}.yaml")) as f:
                sub = yaml.safe_load(f) or {}
            merged = dict(sub)
            merged.update(tree.get(group, {}) or {})
            tree[group] = merged
    for ov in overrides:
        k, _, v = ov.partition("=")
        _set_path(tree, k, yaml.safe_load(v))
    return _wrap(_resolve(tree))


# ---------------------------------------------------------------------------
# network selection (networks.py:46-69, :750-954 of the reference)
# ---------------------------------------------------------------------------
class _NoOpNormalizer(nn.Module):
    def forward(self, x):
        return x


def select_normalization(normalizer: str):
    table = {"maxabs": MaxAbsNormalizationLast, "maxcenter": MaxCenterNormalizationLast,
             "layer": LazyLayerNormLastDim, "none": _NoOpNormalizer}
    if normalizer not in table:
        raise ValueError(f"Normalization {normalizer} not recognized")
    return table[normalizer]


def select_network(cfg: Config, device=None) -> nn.Module:
    net = cfg.net
    device = device or cfg.get("accelerator", "cuda")
    normalization = select_normalization(net.normalize)
    if net.model_type == "high_order":
        # (n_hidden is passed n_in, as the reference does: networks.py:871)
        return HighOrderMLP(
            layer_type=net.layer_type, n=net.n, n_in=net.n_in, n_hidden=net.n_in,
            n_out=net.get("n_out"), in_width=net.input.width, in_segments=net.input.segments,
            out_width=net.output.width, out_segments=net.output.segments,
            hidden_width=net.hidden.width, hidden_layers=net.hidden.layers,
            hidden_segments=net.hidden.segments, normalization=normalization, device=device,
            layer_type_in=net.input.get("layer_type"))
    if net.model_type == "high_order_input":
        layers = [high_order_fc_layers(layer_type=net.layer_type, n=net.n,
                                       in_features=net.input.width,
                                       out_features=net.hidden.width,
                                       segments=net.input.segments, device=device),
                  normalization(),
                  LowOrderMLP(in_width=net.hidden.width, out_width=net.output.width,
                              hidden_width=net.hidden.width, hidden_layers=net.hidden.layers - 1,
                              non_linearity=nn.ReLU(), normalization=normalization,
                              device=device)]
        return nn.Sequential(*layers)
    if net.model_type == "mamba":  # networks.py:918-934
        from .mamba import Mamba
        return Mamba(d_model=net.d_model, n_layer=net.n_layer, vocab_size=net.vocab_size,
                     d_state=net.d_state, expand=net.expand, dt_rank=net.dt_rank,
                     d_conv=net.d_conv, pad_vocab_size_multiple=net.pad_vocab_size_multiple,
                     conv_bias=net.conv_bias, bias=net.bias, layer_type=net.layer_type, n=net.n,
                     segments=net.segments, hidden_layers=net.hidden_layers).to(device)
    raise ValueError(f"model_type {net.model_type!r} is outside the hot path "
                     "(supported: high_order, high_order_input, mamba)")


# ---------------------------------------------------------------------------
# optimizers the reference's configs name (lion_pytorch.Lion, high-order-layers-torch's
# SparseLion: Lion that leaves entries with a zero gradient untouched)
# ---------------------------------------------------------------------------
def _dense_non_overlapping(t: torch.Tensor) -> bool:
    """The tensor's elements fill numel() consecutive storage slots (any permutation of a
    contiguous tensor does)."""
    dims = sorted((st, sz) for st, sz in zip(t.stride(), t.shape) if sz > 1)
    expect = 1
    for st, sz in dims:
        if st != expect:
            return False
        expect *= sz
    return True


class Lion(torch.optim.Optimizer):
    """update = sign(b1*m + (1-b1)*g); p -= lr*(update + wd*p); m = b2*m + (1-b2)*g.
    On the GPU the whole step of a parameter group is one kernel (pp_lion_step_f32); the torch
    expressions below are the same rule for tensors it does not take (CPU, other dtypes)."""

    def __init__(self, params, lr=1e-4, betas=(0.9, 0.99), weight_decay=0.0, sparse=False):
        super().__init__(params, dict(lr=lr, betas=betas, weight_decay=weight_decay))
        self.sparse = sparse

    @staticmethod
    def _fusable(p, g, m) -> bool:
        """One flat pass in storage order: same dense layout for all three (the layers'
        parameters are permuted leaves; their gradients and zeros_like momenta share the strides)."""
        return (p.is_cuda and p.dtype == torch.float32 and g.dtype == torch.float32
                and g.layout == torch.strided and p.stride() == g.stride() == m.stride()
                and _dense_non_overlapping(p))

    def _fused_step(self, todo, lr, b1, b2, wd):
        """pp_lion_step_f32: all parameters of a group in one launch (per PP_OPT_MAX_TENSORS)."""
        import ctypes
        from . import _lib
        from .functional import _stream
        lib = _lib.load()
        for lo in range(0, len(todo), _lib.OPT_MAX_TENSORS):
            part = todo[lo:lo + _lib.OPT_MAX_TENSORS]
            cnt = len(part)
            with torch.cuda.device(part[0][0].device):
                rc = lib.pp_lion_step_f32(
                    cnt,
                    (ctypes.c_void_p * cnt)(*[p.data_ptr() for p, _, _ in part]),
                    (ctypes.c_void_p * cnt)(*[g.data_ptr() for _, g, _ in part]),
                    (ctypes.c_void_p * cnt)(*[m.data_ptr() for _, _, m in part]),
                    (ctypes.c_long * cnt)(*[p.numel() for p, _, _ in part]),
                    lr, b1, b2, wd, int(self.sparse), _stream(part[0][0].device))
            _lib.check(rc, "pp_lion_step_f32")
            # the kernel wrote through raw pointers: tell autograd (saved-tensor checks) and the
            # pre-split weight cache (functional._prepacked, keyed by version) that p changed
            for p, _, _ in part:
                torch._C._increment_version(p)

    @torch.no_grad()
    def step(self, closure=None):
        loss = closure() if closure is not None else None
        for group in self.param_groups:
            lr, (b1, b2), wd = group["lr"], group["betas"], group["weight_decay"]
            fused = []
            for p in group["params"]:
                if p.grad is None:
                    continue
                g = p.grad
                st = self.state[p]
                if "exp_avg" not in st:
                    st["exp_avg"] = torch.zeros_like(p)
                m = st["exp_avg"]
                if self._fusable(p, g, m):
                    fused.append((p, g, m))
                    continue
                update = (m * b1).add_(g, alpha=1 - b1).sign_()
                new_m = (m * b2).add_(g, alpha=1 - b2)
                if self.sparse:  # weights no sample touched keep value and momentum
                    touched = g != 0
                    update = update * touched
                    new_m = torch.where(touched, new_m, m)
                if wd:
                    if self.sparse:  # ... and are not decayed either
                        p.copy_(torch.where(touched, p * (1 - lr * wd), p))
                    else:
                        p.mul_(1 - lr * wd)
                p.add_(update, alpha=-lr)
                m.copy_(new_m)
            if fused:
                self._fused_step(fused, lr, b1, b2, wd)
        return loss


class ASCIIPredictionNet(nn.Module):
    """`ASCIIPredictionNet` without Lightning: ``self.model`` + cross entropy."""

    def __init__(self, cfg: Config, device=None):
        super().__init__()
        self.cfg = cfg
        self.model = select_network(cfg, device)
        self.loss = nn.CrossEntropyLoss()

    def forward(self, x):
        return self.model(x)

    def eval_step(self, batch):
        x, y = batch[0], batch[1]
        y_hat = self(x)
        # (CrossEntropyLoss + argmax accuracy; one HIP kernel per direction on the GPU)
        return PF.cross_entropy_accuracy(y_hat, y.flatten())

    def configure_optimizers(self, capturable: bool = False):
        """`capturable`: keep Adam's step count on the device (needed under HIP graph capture)."""
        opt = self.cfg.optimizer
        if opt.name == "adam":
            return torch.optim.Adam(self.parameters(), lr=opt.lr, capturable=capturable)
        if opt.name == "lion":
            return Lion(self.parameters(), lr=opt.lr, weight_decay=0.0)
        if opt.name == "sparse_lion":
            return Lion(self.parameters(), lr=opt.lr, weight_decay=0.0, sparse=True)
        raise ValueError(f"Optimizer {opt.name} not recognized")


class MambaASCIIPredictionNet(ASCIIPredictionNet):
    """`MambaASCIIPredictionNet` (networks.py:135-147, :969-980): a target per position."""

    def eval_step(self, batch):
        x, y = batch[0], batch[1]
        y_hat = self(x)
        return PF.cross_entropy_accuracy(y_hat.reshape(y.shape[0] * y.shape[1], -1), y.flatten())


def prediction_net(cfg: Config, device=None) -> ASCIIPredictionNet:
    """The module examples/high_order_interpolation.py:162-165 builds for cfg.net.model_type."""
    cls = MambaASCIIPredictionNet if cfg.net.model_type == "mamba" else ASCIIPredictionNet
    return cls(cfg, device)


# ---------------------------------------------------------------------------
# data: character windows gathered on the device
# ---------------------------------------------------------------------------
def ascii_to_float(c: torch.Tensor) -> torch.Tensor:
    return (c - 64 + 0.5) / 64


def _char_windows(buf, idx, features, x, tokens, y):
    """pp_char_windows on the current stream of buf's device (idx: int64 window starts)."""
    from . import _lib
    lib = _lib.load()
    idx = idx.contiguous()
    if idx.dtype != torch.int64:
        idx = idx.long()
    if idx.numel() == 0:
        return
    with torch.cuda.device(buf.device):
        rc = lib.pp_char_windows(PF._ptr(buf), buf.numel(), PF._ptr(idx), idx.numel(), features,
                                 PF._ptr(x), PF._ptr(tokens), PF._ptr(y), PF._stream(buf.device))
    _lib.check(rc, "pp_char_windows")


class CharWindows:
    """All windows of ``features`` characters + the next one as target, held as ONE uint8
    buffer on the device; a batch is an index gather (what generate_dataset builds as
    Python lists of ord() values, single_text_dataset.py:102-123)."""

    def __init__(self, text, features: int, device="cuda"):
        if isinstance(text, str):
            text = text.encode("ascii", "ignore")
        self.features = features
        self.buf = torch.frombuffer(bytearray(text), dtype=torch.uint8).to(device)
        self.n = self.buf.numel() - (features + 1)
        if self.n <= 0:
            raise ValueError("text shorter than one window")
        # every byte below 128: every x this dataset hands out is one of the 128 grid points
        # (c - 63.5)/64, and the input layer may run in table form (declare_character_inputs)
        self.ascii_only = bool((self.buf < 128).all().item())
        self._ar = torch.arange(features, device=self.buf.device)

    def __len__(self):
        return self.n

    def batch(self, idx: torch.Tensor):
        idx = idx.to(self.buf.device)
        if self.buf.is_cuda:  # one kernel: gather, normalise, targets (csrc/pp_data.hip)
            B = idx.numel()
            x = torch.empty(B, self.features, device=self.buf.device, dtype=torch.float32)
            y = torch.empty(B, 1, device=self.buf.device, dtype=torch.int64)
            _char_windows(self.buf, idx, self.features, x, None, y)
            return x, y, idx
        win = self.buf[idx.unsqueeze(1) + self._ar]
        x = ascii_to_float(win.float())
        y = self.buf[idx + self.features].long().unsqueeze(1)
        return x, y, idx

    def sample(self, batch_size: int, generator=None):
        """Random windows.  With a generator on the buffer's device the indices never exist on
        the host (a host-side draw costs a blocking copy per step, which keeps the next step's
        launches from being queued behind the running one)."""
        dev = generator.device if generator is not None else self.buf.device
        idx = torch.randint(0, self.n, (batch_size,), generator=generator, device=dev)
        return self.batch(idx)


class CharSequences(CharWindows):
    """Windows of ``max_features`` characters as indices; features = all but the last, targets
    = all but the first (MambaMixin.collate_fn, lightning_datamodule.py:272-291, as_index)."""

    def __init__(self, text, max_features: int, device="cuda"):
        super().__init__(text, max_features - 1, device)

    def batch(self, idx: torch.Tensor):
        idx = idx.to(self.buf.device)
        if self.buf.is_cuda:
            win = torch.empty(idx.numel(), self.features + 1, device=self.buf.device,
                              dtype=torch.int64)
            _char_windows(self.buf, idx, self.features, None, win, None)
        else:
            win = self.buf[idx.unsqueeze(1) + torch.arange(self.features + 1)].long()
        win = win.unsqueeze(2)
        return win[:, :-1], win[:, 1:], idx


def declare_character_inputs(net: nn.Module, points: int = 128):
    """Tell the first polynomial layer of `net` (HighOrderMLP, or the Sequential of
    model_type high_order_input) that its inputs are character codes on the `points`-point
    grid, so that its forward and weight gradient run in table form (layers.assume_input_grid;
    csrc/pp_table.hip).  Only for a dataset that guarantees it: CharWindows.ascii_only.
    Returns the layer, or None when there is none / it has no table form."""
    seq = net.model if isinstance(getattr(net, "model", None), nn.Sequential) else net
    first = seq[0] if isinstance(seq, nn.Sequential) and len(seq) else None
    if first is None or not hasattr(first, "assume_input_grid"):
        return None
    try:
        return first.assume_input_grid(points)
    except ValueError:
        return None


def make_data(cfg: Config, text, device="cuda") -> CharWindows:
    if cfg.net.model_type == "mamba":
        return CharSequences(text, cfg.data.max_features, device)
    return CharWindows(text, cfg.net.features, device)


def synthetic_text(nchar: int, seed: int = 0) -> bytes:
    """Printable-ASCII text with word structure (no dataset ships with the build)."""
    g = torch.Generator().manual_seed(seed)
    words = [bytes(torch.randint(97, 123, (int(n),), generator=g).tolist())
             for n in torch.randint(2, 9, (200,), generator=g)]
    out = bytearray()
    while len(out) < nchar:
        out += words[int(torch.randint(0, len(words), (1,), generator=g))] + b" "
    return bytes(out[:nchar])


# ---------------------------------------------------------------------------
# the loop
# ---------------------------------------------------------------------------
class GraphedTrainStep:
    """One optimizer step (forward, loss, backward, clip, optimizer) captured once in a HIP
    graph and replayed on batches copied into static tensors: the step of these small networks
    is a few hundred short launches, and the host cannot issue them as fast as the GPU retires
    them.  The library allocates nothing and never synchronises, so the whole step is
    capturable; torch's allocations inside the capture come from the graph's private pool.

    The warm-up iterations PyTorch asks for before a capture run real steps, so parameters are
    restored and the optimizer state is zeroed in place afterwards (state tensors must exist
    before the capture, or their initialisation would be replayed with every step)."""

    def __init__(self, module: "ASCIIPredictionNet", optimizer, example_batch,
                 gradient_clip: Optional[float] = None, warmup: int = 3):
        self.module, self.optimizer, self.clip = module, optimizer, gradient_clip
        self.x = example_batch[0].clone()
        self.y = example_batch[1].clone()
        saved = [p.detach().clone() for p in module.parameters()]
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(warmup):
                self._body()
        torch.cuda.current_stream().wait_stream(side)
        with torch.no_grad():
            for p, q in zip(module.parameters(), saved):
                p.copy_(q)
            for st in optimizer.state.values():
                for v in st.values():
                    if torch.is_tensor(v):
                        v.zero_()
        optimizer.zero_grad(set_to_none=True)
        self.graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph):
            self.loss, self.acc = self._body()

    def _body(self):
        self.optimizer.zero_grad(set_to_none=True)
        loss, acc = self.module.eval_step((self.x, self.y))
        loss.backward()
        if self.clip:
            torch.nn.utils.clip_grad_norm_(self.module.parameters(), self.clip)
        self.optimizer.step()
        return loss.detach(), acc.detach()

    def __call__(self, batch):
        self.x.copy_(batch[0])
        self.y.copy_(batch[1])
        self.graph.replay()
        return self.loss, self.acc


def fit(module: ASCIIPredictionNet, data: CharWindows, steps: int, batch_size: int,
        seed: int = 0, log_every: int = 0, gradient_clip: Optional[float] = None,
        graph: bool = False):
    """`steps` optimizer steps; returns the list of (loss, accuracy) per step.
    `graph`: replay the step from a HIP graph (single process only; see GraphedTrainStep).
    `gradient_clip`: max gradient norm (the Trainer's gradient_clip_val,
    examples/high_order_interpolation.py:158; `gradient_clip: null` in the root config)."""
    import torch.distributed as dist

    from .parallel import GradAllReducer, broadcast_parameters

    distributed = dist.is_available() and dist.is_initialized()
    rank = dist.get_rank() if distributed else 0
    gen = torch.Generator(device=data.buf.device).manual_seed(seed + rank)
    graph = graph and not distributed
    optimizer = module.configure_optimizers(capturable=graph)
    reducer = None
    if distributed:
        broadcast_parameters(module)
        reducer = GradAllReducer(module.parameters())
    history = []
    graphed = None
    t0 = time.perf_counter()
    for step in range(steps):
        batch = data.sample(batch_size, gen)
        if graph:
            if graphed is None:
                graphed = GraphedTrainStep(module, optimizer, batch, gradient_clip)
            loss, acc = graphed(batch)
            history.append((loss.clone(), acc.clone()))
        else:
            optimizer.zero_grad(set_to_none=True)
            loss, acc = module.eval_step(batch)
            loss.backward()
            if reducer is not None:
                reducer.finish()
            if gradient_clip:
                torch.nn.utils.clip_grad_norm_(module.parameters(), gradient_clip)
            optimizer.step()
            history.append((loss.detach(), acc.detach()))
        if log_every and rank == 0 and (step + 1) % log_every == 0:
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
            print(f"step {step + 1}: loss {loss.item():.4f} acc {acc.item():.3f} "
                  f"{(step + 1) * batch_size / dt:.0f} chars/s per GPU", flush=True)
    return [(float(l), float(a)) for l, a in history]


def main(argv=None):
    ap = argparse.ArgumentParser(description=__doc__.splitlines()[0])
    ap.add_argument("--config", required=True, help="root yaml in the reference's layout")
    ap.add_argument("--text", default=None, help="ASCII text file (default: synthetic text)")
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--graph", action="store_true", help="replay the step from a HIP graph")
    ap.add_argument("overrides", nargs="*", help="a.b=value, or group=name")
    args = ap.parse_args(argv)
    cfg = load_config(args.config, args.overrides)
    text = open(args.text, "rb").read() if args.text else synthetic_text(1 << 20)
    module = prediction_net(cfg, device="cuda")
    data = make_data(cfg, text, device="cuda")
    if cfg.net.model_type != "mamba" and data.ascii_only:
        declare_character_inputs(module.model)  # the data set vouches for the grid
    hist = fit(module, data, args.steps, cfg.batch_size, log_every=max(1, args.steps // 10),
               gradient_clip=cfg.get("gradient_clip"), graph=args.graph)
    print(json.dumps({"final_loss": hist[-1][0], "final_acc": hist[-1][1], "steps": len(hist)}))


if __name__ == "__main__":
    main()
