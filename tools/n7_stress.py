"""Repeat one forward instantiation many times on fixed inputs; histogram of where a run differs
bitwise from the first (the kernels are deterministic, so any difference is a fault).
  python tools/n7_stress.py [reps]
"""
import os
import sys
from collections import Counter

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from pp_testlib import make_inputs  # noqa: E402
from language_interpolation_amd import functional as PF  # noqa: E402


def pattern(y, y0):
    d = (y != y0).nonzero()
    rows = sorted(set(d[:, 0].tolist()))
    cols = sorted(set(d[:, 1].tolist()))
    blocks = sorted(set(r // 32 for r in rows))
    inblk = sorted(set(r % 32 for r in rows))
    return (f"blocks={blocks[:6]} rows_in_block={inblk[0]}..{inblk[-1]}({len(inblk)}) "
            f"cols={cols[0]}..{cols[-1]}({len(cols)}) n={len(d)} "
            f"maxabs={(y - y0).abs().max().item():.3g}")


def run(tag, n, S, disc, I, O, B, method, variant, reps, mode):
    dev = torch.device("cuda:0")
    if variant:
        os.environ["PP_BF3_VARIANT"] = variant
    else:
        os.environ.pop("PP_BF3_VARIANT", None)
    x, w, gy = make_inputs(B, I, O, n, S, disc, "normal", seed=20 + n + S)
    xd = x.to(dev)
    w_t = w.to(dev).permute(1, 2, 0).contiguous()
    spec = PF.make_spec(n, S, disc, fwd_method=method, in_features=I)
    auto = PF.make_spec(n, S, disc, in_features=I)
    packed = PF.pack_weights(w_t, auto, B) if mode != "percall" else None
    y0 = None
    pats = Counter()
    nbad = 0
    for it in range(reps):
        if mode == "percall":
            y, _ = PF.forward_raw(xd, w_t, spec)
        elif mode == "prepack_once":
            y, _ = PF.forward_raw(xd, w_t, auto, packed=packed)
        else:
            packed = PF.pack_weights(w_t, auto, B)
            y, _ = PF.forward_raw(xd, w_t, auto, packed=packed)
        if y0 is None:
            y0 = y.clone()
            continue
        if not torch.equal(y, y0):
            nbad += 1
            pats[pattern(y.cpu(), y0.cpu())] += 1
            if nbad == 1 and os.environ.get("PP_DUMP"):
                torch.save({"y": y[:64].cpu(), "y0": y0[:64].cpu()},
                           os.path.join(ROOT, "gpurun_out", f"n7_wrong_{tag}_{mode}.pt"))
    print(f"{tag} n={n} S={S} disc={disc} I={I} O={O} B={B} {method} var={variant or '-'} "
          f"mode={mode}: {nbad}/{reps - 1} runs differ", flush=True)
    for p, c in pats.most_common(12):
        print(f"    {c}x {p}", flush=True)


def main():
    reps = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
    if os.environ.get("PP_DUMP"):
        run("A", 7, 1, True, 70, 160, 8200, "fp16x2", "12", reps, "prepack_once")
        return
    for mode in ("percall", "prepack_once", "prepack_each"):
        run("A", 7, 1, True, 70, 160, 8200, "fp16x2", "12", reps, mode)
    run("B", 7, 1, True, 70, 160, 8200, "fp16x2", "11", reps, "percall")
    run("B", 7, 1, True, 70, 160, 8200, "fp16x2", "11", reps, "prepack_each")
    run("C", 4, 2, False, 70, 160, 8200, "fp16x2", "12", reps, "percall")
    run("C", 4, 2, False, 70, 160, 8200, "fp16x2", "12", reps, "prepack_each")
    run("D", 8, 1, True, 70, 160, 8200, "fp16x2", "12", reps, "percall")
    run("D", 8, 1, True, 70, 160, 8200, "fp16x2", "12", reps, "prepack_each")
    run("E", 3, 2, False, 70, 160, 8200, "fp16x2", "12", reps, "prepack_each")
    run("E", 3, 2, False, 250, 250, 16384, "fp16x2", "", reps, "prepack_each")
    run("F", 3, 2, False, 250, 250, 8200, "bf16x3", "12", reps, "percall")
    run("F", 7, 1, True, 70, 160, 8200, "bf16x3", "12", reps, "percall")


if __name__ == "__main__":
    main()
