"""Diagnosis of the intermittent wrong rows (round 1: fp16x2, n = K = 7, I = 70, O = 160, B = 8200).

Runs every fp16x2 kernel of that layer many times on the SAME inputs in one process, compares
each run bitwise with the first and against the fp64 oracle, and prints where they differ
(rows / columns / 32-row block, whether the block holds the out-of-range basis value).
  python tools/n7_diag.py [reps]
"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from pp_testlib import make_inputs, oracle_all  # noqa: E402
from language_interpolation_amd import functional as PF  # noqa: E402


def describe(name, got, ref, first, tol_abs):
    d = (got.double() - ref.double()).abs()
    bad = (d > tol_abs).nonzero()
    same = torch.equal(got, first)
    if len(bad) == 0 and same:
        return False
    rows = sorted(set(bad[:, 0].tolist()))
    cols = sorted(set(bad[:, 1].tolist()))
    print(f"  {name}: bitwise_same_as_run0={same} wrong={len(bad)} maxerr={d.max().item():.3e} "
          f"rows={rows[:12]}{'...' if len(rows) > 12 else ''} (blocks {sorted(set(r // 32 for r in rows))[:8]}) "
          f"cols={cols[:8]}{'...' if len(cols) > 8 else ''} ncols={len(cols)}", flush=True)
    if not same:
        dd = (got != first).nonzero()
        r2 = sorted(set(dd[:, 0].tolist()))
        print(f"     differs from run0 in {len(dd)} entries, rows {r2[:16]} cols "
              f"{sorted(set(dd[:, 1].tolist()))[:16]}", flush=True)
    return True


def main():
    reps = int(sys.argv[1]) if len(sys.argv) > 1 else 40
    dev = torch.device("cuda:0")
    n, S, disc, B, I = 7, 1, True, 8200, 70
    for O in (160, 100):
        x, w, gy = make_inputs(B, I, O, n, S, disc, "normal", seed=20 + n + S)
        ref = oracle_all(x, w, gy, n, S, disc)
        xd, gyd = x.to(dev), gy.to(dev)
        w_t = w.to(dev).permute(1, 2, 0).contiguous()
        for variant in ("12", "11", ""):
            if variant:
                os.environ["PP_BF3_VARIANT"] = variant
            else:
                os.environ.pop("PP_BF3_VARIANT", None)
            spec = PF.make_spec(n, S, disc, fwd_method="fp16x2", bwd_x_method="fp16x2",
                                bwd_w_method="fp16x2", in_features=I)
            auto = PF.make_spec(n, S, disc, in_features=I)
            firsts = {}
            nbad = {}
            for it in range(reps):
                # dirty the allocator's free blocks so stale workspace contents vary
                junk = torch.full((1 << 22,), float(it) * 1e30, device=dev)
                del junk
                outs = {}
                outs["y"], _ = PF.forward_raw(xd, w_t, spec)
                outs["gx"] = PF.backward_input_raw(xd, gyd, w_t, spec)
                outs["gw"] = PF.backward_weight_raw(xd, gyd, tuple(w_t.shape), spec).permute(2, 0, 1).reshape(O, -1)
                packed = PF.pack_weights(w_t, auto, B)
                outs["y_pre"], _ = PF.forward_raw(xd, w_t, auto, packed=packed)
                outs["gx_pre"] = PF.backward_input_raw(xd, gyd, w_t, auto, packed=packed)
                torch.cuda.synchronize()
                for k, v in outs.items():
                    v = v.cpu()
                    rk = k.split("_")[0]
                    r = ref[rk].reshape(v.shape)
                    tol = 1e-5 * r.abs().max().item()
                    if k not in firsts:
                        firsts[k] = v
                    if describe(f"O={O} var={variant or 'default'} it={it} {k}", v, r, firsts[k], tol):
                        nbad[k] = nbad.get(k, 0) + 1
            print(f"O={O} variant={variant or 'default'}: {reps} reps, anomalies {nbad}", flush=True)


if __name__ == "__main__":
    main()
