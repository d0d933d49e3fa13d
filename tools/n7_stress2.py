"""Second-stage diagnosis: for each wrong run of the suspect forward instantiation, identify WHICH
(input, weight index) contribution differs (the error is rank-1 over the columns), for a few
variations of the layer width / row offset.   python tools/n7_stress2.py [reps]
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
from oracle import pp_oracle as po  # noqa: E402


def basis(x, n):
    X = po.chebyshev_lobatto(n).double()
    xi = x.double()
    L = torch.ones(*x.shape, n, dtype=torch.double)
    for j in range(n):
        for m in range(n):
            if m != j:
                L[..., j] *= (xi - X[m]) / (X[j] - X[m])
    return L


def explain(y, y0, x, w, n):
    d = (y.double() - y0.double())
    rows = sorted(set((y != y0).nonzero()[:, 0].tolist()))
    dr = d[rows]
    U, S, Vh = torch.linalg.svd(dr, full_matrices=False)
    W = w.double()
    cs = torch.einsum("o,oik->ik", Vh[0], W) / (W.norm(dim=0) * Vh[0].norm() + 1e-30)
    j = int(cs.abs().flatten().argmax())
    i_, k_ = j // W.shape[2], j % W.shape[2]
    coef = dr @ W[:, i_, k_] / (W[:, i_, k_] @ W[:, i_, k_])
    true = basis(x[rows][:, i_:i_ + 1], n)[:, 0, k_]
    ratio = ((coef + true).abs().max() / true.abs().max()).item()
    blocks = sorted(set(r // 32 for r in rows))
    inblk = sorted(set(r % 32 for r in rows))
    return (f"blocks={blocks[:4]} rows={inblk[0]}..{inblk[-1]}({len(inblk)}) rank1={S[1].item() / S[0].item():.1e} "
            f"input={i_} k={k_} cos={cs.flatten()[j].item():.4f} |wrong_phi|/|phi|={ratio:.1e}")


def run(tag, n, I, O, B, variant, reps, row_off=0):
    dev = torch.device("cuda:0")
    os.environ["PP_BF3_VARIANT"] = variant
    x, w, gy = make_inputs(B + row_off, I, O, n, 1, True, "normal", seed=20 + n + 1)
    x = x[row_off:].contiguous()
    xd = x.to(dev)
    w_t = w.to(dev).permute(1, 2, 0).contiguous()
    auto = PF.make_spec(n, 1, True, in_features=I)
    packed = PF.pack_weights(w_t, auto, B)
    y0 = None
    pats = Counter()
    nbad = 0
    for it in range(reps):
        y, _ = PF.forward_raw(xd, w_t, auto, packed=packed)
        if y0 is None:
            y0 = y.clone()
            continue
        if not torch.equal(y, y0):
            nbad += 1
            pats[explain(y.cpu(), y0.cpu(), x, w, n)] += 1
    print(f"{tag} n={n} I={I} O={O} B={B} row_off={row_off} var={variant}: {nbad}/{reps - 1} differ", flush=True)
    for p, c in pats.most_common(8):
        print(f"    {c}x {p}", flush=True)


def main():
    reps = int(sys.argv[1]) if len(sys.argv) > 1 else 1500
    if os.environ.get("PP_QUICK"):
        run("base", 7, 70, 160, 8200, "12", reps)
        run("I=102", 7, 102, 160, 8200, "12", reps)
        run("I=102 B=16384", 7, 102, 160, 16384, "12", reps)
        return
    run("base", 7, 70, 160, 8200, "12", reps)
    run("rows+32", 7, 70, 160, 8200, "12", reps, row_off=32)
    run("rows+5", 7, 70, 160, 8200, "12", reps, row_off=5)
    run("I=96", 7, 96, 160, 8200, "12", reps)
    run("I=68", 7, 68, 160, 8200, "12", reps)
    run("I=71", 7, 71, 160, 8200, "12", reps)
    run("I=102", 7, 102, 160, 8200, "12", reps)
    run("I=38", 7, 38, 160, 8200, "12", reps)
    run("O=250", 7, 70, 250, 8200, "12", reps)
    run("B=512", 7, 70, 160, 512, "12", reps)
    run("B=64", 7, 70, 160, 64, "12", reps)


if __name__ == "__main__":
    main()
