"""Randomised parity run of the selective-scan kernels (csrc/pp_scan.hip) against
oracle/mamba_oracle.py in fp64 (test infrastructure, not collected by pytest):

    python tests/fuzz_scan.py --seconds 180 --seed 1 [--out gpurun_out/fuzz/scan.txt]

Every case draws (batch, length, channels, states) around the kernels' variant boundaries
(state-split kernels up to 64 states, tiles of 8 time steps, 8 states per wave) and checks y and
the six gradients to 1e-5 of each tensor's largest magnitude, plus the inference path's bits.
"""
from __future__ import annotations

import argparse
import os
import random
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

BATCH = [1, 2, 3, 15, 16, 17, 64, 100, 256, 1024]
LENGTH = [1, 2, 7, 8, 9, 15, 16, 17, 24, 33, 40]
CHANNELS = [1, 3, 63, 64, 65, 70, 128, 130, 256]
STATES = [1, 2, 7, 8, 9, 16, 19, 32, 33, 40, 63, 64, 65, 70, 96]


def draw_case(rng):
    while True:
        b, l, d, n = rng.choice(BATCH), rng.choice(LENGTH), rng.choice(CHANNELS), rng.choice(STATES)
        if b * l * d * n <= 2e7:
            return b, l, d, n, rng.randrange(1 << 30)


def run_case(case):
    import torch
    from language_interpolation_amd import functional as PF
    from oracle import mamba_oracle as mo
    from test_mamba_gpu import TOL, rel, scan_inputs

    b, l, d, n, seed = case
    ins, gy = scan_inputs(b, l, d, n, seed=seed)
    ref_in = [t.double().requires_grad_() for t in ins]
    ref_y = mo.selective_scan(*ref_in)
    ref_g = torch.autograd.grad(ref_y, ref_in, gy.double(), retain_graph=True)
    dev_in = [t.cuda().requires_grad_() for t in ins]
    y = PF.selective_scan(*dev_in)
    bad, soft = [], []
    if not rel(y, ref_y.detach()) <= TOL:
        bad.append(f"y {rel(y, ref_y.detach()):.3e}")
    grads = torch.autograd.grad(y, dev_in, gy.cuda())
    for name, g, r in zip(("gu", "gdelta", "gA", "gB", "gC", "gD"), grads, ref_g):
        if rel(g, r) <= TOL:
            continue
        if name == "gD":
            # gD[c] = sum over (batch, time) of gy * u: with one or two channels the whole tensor
            # can cancel; the error is then held to the bar against the largest sum of |terms|
            terms = (gy.double() * ins[0].double()).abs().sum(dim=(0, 1)).max()
            err = (g.double().cpu() - r).abs().max()
            if err <= TOL * terms:
                soft.append(f"gD {rel(g, r):.3e} (max|ref| / max sum|terms| = {(r.abs().max() / terms).item():.2e})")
                continue
        if name == "gA" and r.numel() <= 4:
            # a handful of elements, each a sum over (batch, time): the same, with the sum of
            # the per-batch-element gradients' magnitudes as a (lower) estimate of sum |terms|
            per = [torch.autograd.grad(ref_y[i], ref_in[2], gy[i].double(), retain_graph=True)[0].abs()
                   for i in range(b)]
            terms = torch.stack(per).sum(dim=0).max()
            err = (g.double().cpu() - r).abs().max()
            if err <= TOL * terms:
                soft.append(f"gA {rel(g, r):.3e} (max|ref| / sum|per-row terms| = {(r.abs().max() / terms).item():.2e})")
                continue
        bad.append(f"{name} {rel(g, r):.3e}")
    with torch.no_grad():
        y2 = PF.selective_scan(*[t.detach() for t in dev_in])
    if not torch.equal(y2, y.detach()):
        bad.append("inference path differs from the training path")
    return bad, soft


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seconds", type=float, default=120.0)
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--out", default="")
    args = ap.parse_args()
    rng = random.Random(args.seed)
    t0, done, fails, softs = time.time(), 0, 0, 0
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
        bad, soft = run_case(case)
        done += 1
        if bad:
            fails += 1
            emit(f"FAIL {case}: {'; '.join(bad)}")
        if soft:
            softs += 1
            emit(f"ILL-CONDITIONED {case}: {'; '.join(soft)}")
        if done % 20 == 0:
            print(f"[{time.time() - t0:6.0f} s] {done} cases, {fails} failing", flush=True)
    emit(f"done: {done} selective-scan cases, {fails} failing, {softs} ill-conditioned (a one-channel gD / gA that cancelled), seed {args.seed}")
    if out:
        out.close()
    return 1 if fails else 0


if __name__ == "__main__":
    sys.exit(main())
