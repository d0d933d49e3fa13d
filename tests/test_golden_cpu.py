"""CPU: the oracle (all three formulations) against the committed golden vectors."""
import os

import numpy as np
import pytest
import torch

from oracle import pp_oracle as po
from pp_testlib import TOL, rel_err

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "pp_golden.npz")


def load_cases():
    z = np.load(GOLDEN, allow_pickle=False)
    names = sorted({k.split("/")[0] for k in z.files})
    return z, names


Z, NAMES = load_cases()


@pytest.mark.parametrize("name", NAMES)
@pytest.mark.parametrize("prec", ["f32", "f32dense", "f64"])
def test_oracle_matches_golden(name, prec):
    disc, n, S, I, O, B = (int(v) for v in Z[name + "/meta"])
    x, w, gy = (torch.from_numpy(Z[f"{name}/{k}"]) for k in ("x", "w", "gy"))
    r = po.piecewise_fwd_bwd(x, w, gy, n, S, bool(disc), precision=prec)
    assert np.array_equal(r["seg"].numpy().astype(np.int32), Z[name + "/seg"])
    for key in ("y", "gx", "gw"):
        ref = torch.from_numpy(Z[f"{name}/{key}"])
        if ref.abs().max() == 0:
            assert r[key].abs().max() == 0
        else:
            assert rel_err(r[key], ref) <= TOL, (name, prec, key)
