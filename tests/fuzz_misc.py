"""Randomised run of the row-wise kernels around the layer (csrc/pp_norm.hip, csrc/pp_mamba.hip)
against their torch expressions in fp64 (test infrastructure, not collected by pytest):

    python tests/fuzz_misc.py --seconds 120 --seed 1 [--out gpurun_out/fuzz/misc.txt]

Each case draws (rows, width) and one of: LayerNorm over the last dimension, max-abs
normalisation (networks.py:48-91 of the reference; oracle/pp_oracle.py), cross entropy + accuracy
(networks.py:121-132), causal conv + SiLU, SiLU gate and RMSNorm (state_space_network.py:315-326,
:423-433); values and every gradient.
"""
from __future__ import annotations

import argparse
import os
import random
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

ROWS = [1, 2, 3, 63, 64, 65, 255, 256, 257, 1000, 1024, 4097, 16384, 20000]
WIDTH = [1, 2, 3, 5, 31, 32, 33, 63, 64, 65, 100, 127, 128, 129, 200, 250, 256, 257, 300, 512, 1000, 1030]
KINDS = ["layer_norm", "max_abs", "cross_entropy", "conv_silu", "silu_gate", "rms_norm"]


def draw_case(rng):
    while True:
        kind, B, C = rng.choice(KINDS), rng.choice(ROWS), rng.choice(WIDTH)
        # (widths under 8 leave the normalisations' gradients to cancellation: a LayerNorm row
        # of two values is +-1 whatever the input; rms_norm serves rows of up to 1024)
        if B * C <= 6e6 and (kind == "cross_entropy" or 8 <= C <= 1024):
            return kind, B, C, rng.randrange(1 << 30)


def _err(a, r):
    return ((a.double().cpu() - r.double().cpu()).abs().max() / r.abs().max().clamp_min(1e-30)).item()


def run_case(case):
    import torch
    import torch.nn.functional as F
    from language_interpolation_amd import functional as PF
    from language_interpolation_amd.mamba import causal_depthwise_conv
    from oracle import pp_oracle as po

    kind, B, C, seed = case
    g = torch.Generator().manual_seed(seed)
    bad = []

    def compare(outs, refs, names, bar):
        for a, r, nm in zip(outs, refs, names):
            if not torch.isfinite(a).all():
                bad.append(f"{nm} not finite")
            elif not _err(a, r) <= bar:
                bad.append(f"{nm} {_err(a, r):.3e}")

    if kind in ("layer_norm", "max_abs"):
        x = torch.randn(B, C, generator=g) * (10.0 ** torch.randint(-3, 4, (B, 1), generator=g).float())
        gy = torch.randn(B, C, generator=g)
        xd = x.cuda().requires_grad_(True)
        xr = x.double().requires_grad_(True)
        if kind == "layer_norm":
            y, yr = PF.layer_norm_last(xd, 1e-5), po.layer_norm_last_dim(xr, 1e-5)
        else:
            y, yr = PF.max_abs_norm_last(xd, 1e-6), po.max_abs_normalization_last(xr, 1e-6)
        y.backward(gy.cuda())
        yr.backward(gy.double())
        # gradients row by row: rows are scaled 1e-3 .. 1e3, so a whole-tensor maximum would
        # only ever look at the smallest rows
        compare([y], [yr.detach()], ["y"], 1e-5)
        ge = (xd.grad.double().cpu() - xr.grad).abs().amax(dim=1)
        gs = xr.grad.abs().amax(dim=1).clamp_min(1e-30)
        # (the LayerNorm backward cancels gy against its projections: bar 1e-4, as the fused test)
        if not (ge / gs).max().item() <= 1e-4:
            bad.append(f"gx row-relative {(ge / gs).max().item():.3e}")
    elif kind == "cross_entropy":
        if C < 1:
            return bad
        logits = (4 * torch.randn(B, C, generator=g)).cuda().requires_grad_(True)
        target = torch.randint(0, C, (B,), generator=g).cuda()
        loss, acc = PF.cross_entropy_accuracy(logits, target)
        ref_in = logits.detach().double().requires_grad_(True)
        ref = F.cross_entropy(ref_in, target)
        ref_acc = (torch.argmax(ref_in, dim=1) == target).double().mean()
        if not abs(loss.item() - ref.item()) <= 2e-6 * max(1.0, abs(ref.item())):
            bad.append(f"loss {loss.item()} vs {ref.item()}")
        if not abs(acc.item() - ref_acc.item()) <= 1e-6:
            bad.append(f"accuracy {acc.item()} vs {ref_acc.item()}")
        loss.backward()
        ref.backward()
        # glogits = (softmax - onehot) / B: a row whose target has p ~ 1 leaves 1 - p to
        # cancellation, so the bar has an absolute part of a few fp32 ulps of 1 / B
        err = (logits.grad.double() - ref_in.grad).abs().max().item()
        if not err <= 2e-6 * ref_in.grad.abs().max().item() + 3e-7 / B:
            bad.append(f"glogits abs err {err:.3e} (largest entry {ref_in.grad.abs().max().item():.3e})")
    else:
        l = 1 + seed % 17
        b = max(1, B // l)
        d = C
        k = 2 + seed % 3
        both = torch.randn(b, l, 2 * d, generator=g).cuda().requires_grad_(True)
        w = (torch.randn(d, 1, k, generator=g) * 0.5).cuda().requires_grad_(True)
        bs = torch.randn(d, generator=g).cuda().requires_grad_(True)
        nw = (torch.rand(d, generator=g) + 0.5).cuda().requires_grad_(True)
        gy = torch.randn(b, l, d, generator=g).cuda()

        def run(fused, leaves):
            bt, wt, bst, nwt = leaves
            x, res = bt.split([d, d], dim=-1)
            if kind == "conv_silu":
                return PF.causal_conv_silu(x, wt, bst) if fused else F.silu(causal_depthwise_conv(x, wt, bst))
            if kind == "silu_gate":
                return PF.silu_gate(x, res) if fused else x * F.silu(res)
            y = x * 1.0
            return PF.rms_norm(y, nwt, 1e-5) if fused else \
                y * torch.rsqrt(y.pow(2).mean(-1, keepdim=True) + 1e-5) * nwt

        used = {"conv_silu": (0, 1, 2), "silu_gate": (0,), "rms_norm": (0, 3)}[kind]
        leaves = (both, w, bs, nw)
        out = run(True, leaves)
        grads = torch.autograd.grad(out, [leaves[i] for i in used], gy)
        ref_leaves = [t.detach().double().requires_grad_(True) for t in leaves]
        ref = run(False, ref_leaves)
        ref_grads = torch.autograd.grad(ref, [ref_leaves[i] for i in used], gy.double())
        names = ["gx", "gw", "gbias", "gnorm"]
        compare([out.detach()] + list(grads), [ref.detach()] + list(ref_grads),
                ["y"] + [names[i] for i in used], 2e-5)
    return bad


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seconds", type=float, default=120.0)
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--out", default="")
    args = ap.parse_args()
    rng = random.Random(args.seed)
    t0, done, fails = time.time(), 0, 0
    out = open(args.out, "w") if args.out else None

    def emit(line):
        print(line, flush=True)
        if out:
            out.write(line + "\n")
            out.flush()

    while time.time() - t0 < args.seconds:
        case = draw_case(rng)
        if args.out:
            with open(args.out + ".current", "w") as cur:
                cur.write(f"{done} {case}\n")
        bad = run_case(case)
        done += 1
        if bad:
            fails += 1
            emit(f"FAIL {case}: {'; '.join(bad)}")
        if done % 50 == 0:
            print(f"[{time.time() - t0:6.0f} s] {done} cases, {fails} failing", flush=True)
    emit(f"done: {done} row-wise kernel cases, {fails} failing, seed {args.seed}")
    if out:
        out.close()
    return 1 if fails else 0


if __name__ == "__main__":
    sys.exit(main())
