"""Per-kernel timing of the BASELINE layer shapes through the C ABI (developer tool).

python tools/layer_bench.py [--batch 16384] [--only cfg3] [--iters 20]
Prints, per layer and pass, the HIP-event time, algorithmic GB/s and GFLOP/s
(SURVEY.md 8d formulas) for every method that applies.
"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from language_interpolation_amd import _lib  # noqa: E402

if os.environ.get("PP_LIB"):  # a developer build of the library
    _lib.LIB_PATH = os.environ["PP_LIB"]
from language_interpolation_amd import functional as PF  # noqa: E402

# name: (disc, n, I, O, S, xdist, first_layer)
LAYERS = {
    "cfg1_L0": (False, 2, 16, 200, 128, "char", True),
    "cfg1_L1": (False, 2, 200, 200, 2, "normal", False),
    "cfg2_L0": (False, 2, 16, 250, 128, "char", True),
    "cfg2_hidden": (False, 3, 250, 250, 2, "normal", False),
    "cfg2_out": (False, 3, 250, 128, 2, "normal", False),
    "cfg3_hidden": (True, 1, 200, 200, 150, "normal", False),
    "cfg3_quirk": (True, 2, 200, 200, 150, "normal", False),
    "cfg3_out": (True, 1, 200, 128, 150, "normal", False),
    "cfg4_input": (True, 2, 16, 1000, 128, "char", True),
    "cfg5_inproj2": (False, 3, 256, 256, 2, "normal", False),
}


def time_fn(fn, iters):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ev0.record()
    for _ in range(iters):
        fn()
    ev1.record()
    torch.cuda.synchronize()
    return ev0.elapsed_time(ev1) * 1e3 / iters  # us


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, nargs="+", default=[1024, 16384])
    ap.add_argument("--only", type=str, default="")
    ap.add_argument("--iters", type=int, default=20)
    ap.add_argument("--fwd-only", action="store_true")
    args = ap.parse_args()
    lib = _lib.load()
    dev = torch.device("cuda:0")
    gen = torch.Generator().manual_seed(0)
    for name, (disc, n, I, O, S, dist, first) in LAYERS.items():
        if args.only and args.only not in name:
            continue
        K = PF.num_weights(n, S, disc)
        W = O * I * K
        for B in args.batch:
            if dist == "char":
                from bench import char_batch
                x = char_batch(B, I, gen).to(dev)
            else:
                x = torch.randn(B, I, generator=gen).to(dev)
            w_t = ((torch.rand(I, K, O, generator=gen) * 2 - 1) / I).to(dev)
            gy = (torch.randn(B, O, generator=gen) / O).to(dev)
            rows = []
            for m in ("gather", "slab", "tile", "dense", "bf16x3", "fp16x2"):
                spec = PF.make_spec(n, S, disc, fwd_method=m)
                try:
                    t = time_fn(lambda: PF.forward_raw(x, w_t, spec), args.iters)
                except Exception:
                    continue
                by = 4 * (B * I + B * O + W)
                rows.append((f"fwd/{m}", t, by, 2 * B * I * O * n))
            # layer + LayerNorm in one kernel, where the automatic choice has it (1: matrix-core,
            # 2: gather)
            mode = lib.pp_forward_layer_norm_supported(B, I, O, n, S, int(disc), 2.0)
            if mode:
                w_log = w_t.permute(2, 0, 1)
                spec = PF.make_spec(n, S, disc)
                with torch.no_grad():
                    t = time_fn(lambda: PF.piecewise_poly_layer_norm(x, w_log, spec, 1e-5),
                                args.iters)
                rows.append((f"fwd+LN/{mode}", t, 4 * (B * I + B * O + W), 2 * B * I * O * n))
            if not first and not args.fwd_only:
                for m in ("gather", "dense", "bf16x3", "fp16x2", "tile"):
                    spec = PF.make_spec(n, S, disc, bwd_x_method=m)
                    try:
                        t = time_fn(lambda: PF.backward_input_raw(x, gy, w_t, spec), args.iters)
                    except Exception:
                        continue
                    by = 4 * (2 * B * I + B * O + W)
                    rows.append((f"bwd_x/{m}", t, by, 2 * B * I * O * n))
            for m in (() if args.fwd_only else ("atomic", "sort", "regacc", "direct", "dense", "bf16x3", "fp16x2")):
                spec = PF.make_spec(n, S, disc, bwd_w_method=m)
                try:
                    t = time_fn(lambda: PF.backward_weight_raw(x, gy, tuple(w_t.shape), spec),
                                args.iters)
                except Exception:
                    continue
                by = 4 * (B * I + B * O + W)
                rows.append((f"bwd_w/{m}", t, by, 2 * B * I * O * n))
            auto = [lib.pp_auto_method(p, B, I, O, n, S, int(disc), 2.0) for p in range(3)]
            print(f"== {name} B={B} (disc={int(disc)} n={n} {I}->{O} S={S} K={K}) auto={auto}")
            for what, t, by, fl in rows:
                print(f"   {what:14s} {t:10.1f} us  {by / t * 1e-3:9.1f} GB/s  "
                      f"{fl / t * 1e-6:9.2f} TFLOP/s")
            sys.stdout.flush()


if __name__ == "__main__":
    main()
