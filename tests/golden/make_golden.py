"""Generates tests/golden/pp_golden.npz from oracle/pp_oracle.py (fp64 variant).

PARITY UNPINNED: the reference's layer source (high-order-layers-torch) is not
available offline and the reference's tests hold no numeric vectors for this
path, so these vectors come from the spec-derived oracle, not from the
reference itself.  They freeze the oracle's behaviour (so a later edit of the
oracle or of the kernels that changes results is caught) for every
(type, n, segments) class in BASELINE.json's configs plus edge cases.

Run from the repo root:  python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

from pp_testlib import make_inputs, oracle_all  # noqa: E402

# name: (disc, n, S, I, O, B, dist)
CASES = {
    "cfg1_input_c2_s128": (False, 2, 128, 16, 8, 48, "char"),
    "cfg1_hidden_c2_s2": (False, 2, 2, 32, 24, 40, "normal"),
    "cfg2_hidden_c3_s2": (False, 3, 2, 30, 26, 33, "normal"),
    "cfg3_hidden_d1_s150": (True, 1, 150, 8, 8, 64, "normal"),
    "cfg3_quirk_d2_s150": (True, 2, 150, 6, 8, 37, "normal"),
    "cfg4_input_d2_s128": (True, 2, 128, 16, 8, 64, "char"),
    "cfg5_proj_c3_s2": (False, 3, 2, 4, 32, 30, "normal"),
    "small_hidden_d2_s8": (True, 2, 8, 10, 10, 29, "uniform"),
    "n5_c_s2_test_utils": (False, 5, 2, 10, 5, 35, "uniform"),
    "n8_d_s3": (True, 8, 3, 7, 3, 9, "uniform"),
    "edges_c4_s5": (False, 4, 5, 9, 6, 64, "edges"),
    "edges_d3_s150": (True, 3, 150, 4, 4, 64, "edges"),
}


def main():
    out = {}
    for name, (disc, n, S, I, O, B, dist) in CASES.items():
        x, w, gy = make_inputs(B, I, O, n, S, disc, dist, seed=len(name) * 7 + n)
        r = oracle_all(x, w, gy, n, S, disc)
        out[name + "/meta"] = np.array([int(disc), n, S, I, O, B], dtype=np.int64)
        out[name + "/x"] = x.numpy()
        out[name + "/w"] = w.numpy()
        out[name + "/gy"] = gy.numpy()
        out[name + "/y"] = r["y"].numpy().astype(np.float32)
        out[name + "/seg"] = r["seg"].numpy().astype(np.int32)
        out[name + "/gx"] = r["gx"].numpy().astype(np.float32)
        out[name + "/gw"] = r["gw"].numpy().astype(np.float32)
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "pp_golden.npz")
    np.savez_compressed(path, **out)
    print(path, os.path.getsize(path), "bytes", len(CASES), "cases")


if __name__ == "__main__":
    main()
