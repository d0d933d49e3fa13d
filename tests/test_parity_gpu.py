"""GPU parity: HIP kernels (through the C ABI of libpphip.so) vs the CPU oracle.

Bar (north_star): segment ids bit-exact; fp32 outputs and gradients within 1e-5
relative (max-norm per tensor) of the fp64 evaluation of the reference function.
"""
import os

import numpy as np
import pytest
import torch

from pp_testlib import TOL, make_inputs, oracle_all, rel_err

pytestmark = pytest.mark.gpu

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "pp_golden.npz")


def _pf():
    from language_interpolation_amd import functional as PF
    return PF


def hip_all(x, w, gy, n, S, disc, fwd="auto", bwd_x="auto", bwd_w="auto", **kw):
    PF = _pf()
    dev = torch.device("cuda:0")
    spec = PF.make_spec(n, S, disc, fwd_method=fwd, bwd_x_method=bwd_x, bwd_w_method=bwd_w,
                        in_features=x.shape[1], **kw)
    xd, gyd = x.to(dev), gy.to(dev)
    w_t = w.to(dev).permute(1, 2, 0).contiguous()
    y, seg = PF.forward_raw(xd, w_t, spec, want_seg=True)
    gx = PF.backward_input_raw(xd, gyd, w_t, spec)
    gw_t = PF.backward_weight_raw(xd, gyd, tuple(w_t.shape), spec)
    torch.cuda.synchronize()
    return {"y": y.cpu(), "seg": seg.cpu(), "gx": gx.cpu(), "gw": gw_t.permute(2, 0, 1).cpu()}


def check(got, ref, what):
    assert torch.equal(got["seg"].long(), ref["seg"].long()), f"{what}: segment ids differ"
    for key in ("y", "gx", "gw"):
        r = ref[key]
        if r.abs().max() == 0:
            assert got[key].abs().max() == 0, (what, key)
        else:
            e = rel_err(got[key], r)
            assert e <= TOL, f"{what}: {key} rel err {e:.3e}"


# (disc, n, S, I, O, B, dist)
CASES = [
    (False, 2, 128, 16, 250, 257, "char"),      # cfg1/cfg2 input layer
    (False, 2, 128, 16, 200, 1000, "char"),     # cfg1 L0 at B=1000
    (True, 2, 128, 16, 1000, 256, "char"),      # cfg4 input layer
    (True, 1, 150, 200, 200, 300, "normal"),    # cfg3 hidden
    (True, 2, 150, 200, 200, 128, "normal"),    # cfg3 hidden with the n_hidden quirk
    (True, 1, 150, 200, 128, 300, "normal"),    # cfg3 output
    (False, 3, 2, 250, 250, 256, "normal"),     # cfg2 hidden
    (False, 3, 2, 250, 128, 256, "normal"),     # cfg2 output
    (False, 2, 2, 200, 200, 1000, "normal"),    # cfg1 hidden
    (False, 2, 2, 200, 128, 1000, "normal"),    # cfg1 output
    (False, 3, 2, 64, 256, 240, "normal"),      # cfg5 in_proj L1
    (False, 3, 2, 256, 256, 240, "normal"),     # cfg5 in_proj L2
    (False, 3, 2, 128, 68, 240, "normal"),      # cfg5 x_proj
    (False, 3, 2, 4, 128, 240, "normal"),       # cfg5 dt_proj
    (False, 3, 2, 128, 64, 240, "normal"),      # cfg5 out_proj
    (True, 2, 8, 10, 10, 82, "uniform"),        # net=small defaults
    (False, 5, 2, 10, 5, 35, "uniform"),        # tests/test_utils.py shapes
    (False, 5, 2, 5, 11, 35, "uniform"),
    (True, 8, 3, 7, 3, 5, "uniform"),           # max n
    (False, 4, 5, 65, 1, 3, "edges"),           # O=1, I just over one wave
    (True, 3, 7, 130, 257, 9, "edges"),         # ragged everything
    (False, 2, 2, 1, 1, 1, "normal"),           # minimum sizes
    (True, 1, 1, 3, 5, 17, "normal"),           # one segment, piecewise constant
]


@pytest.mark.parametrize("case", CASES, ids=lambda c: "-".join(str(v) for v in c))
def test_auto_methods_match_oracle(case):
    disc, n, S, I, O, B, dist = case
    x, w, gy = make_inputs(B, I, O, n, S, disc, dist, seed=1)
    ref = oracle_all(x, w, gy, n, S, disc)
    check(hip_all(x, w, gy, n, S, disc), ref, "auto")


@pytest.mark.parametrize("case", CASES, ids=lambda c: "-".join(str(v) for v in c))
def test_gather_and_both_scatter_variants(case):
    disc, n, S, I, O, B, dist = case
    x, w, gy = make_inputs(B, I, O, n, S, disc, dist, seed=2)
    ref = oracle_all(x, w, gy, n, S, disc)
    check(hip_all(x, w, gy, n, S, disc, "gather", "gather", "atomic"), ref, "gather/atomic")
    check(hip_all(x, w, gy, n, S, disc, "gather", "gather", "sort"), ref, "gather/sort")


DENSE_CASES = [c for c in CASES if (c[1] * c[2] if c[0] else (c[1] - 1) * c[2] + 1) <= 8
               and c[2] & (c[2] - 1) == 0] + [
    (True, 2, 4, 40, 70, 100, "edges"),     # K=8, non power-of-two free
    (False, 2, 4, 33, 31, 77, "normal"),    # K=5 from n=2, S=4; odd widths -> scalar loads
    (False, 3, 1, 34, 66, 50, "normal"),    # single segment: a plain polynomial layer
    (True, 1, 8, 64, 64, 300, "normal"),
    (False, 4, 2, 100, 96, 130, "normal"),  # K=7
    (False, 3, 2, 250, 250, 4096, "normal"),  # several row chunks in the weight gradient
]


@pytest.mark.parametrize("case", DENSE_CASES, ids=lambda c: "-".join(str(v) for v in c))
def test_dense_mfma_methods(case):
    disc, n, S, I, O, B, dist = case
    x, w, gy = make_inputs(B, I, O, n, S, disc, dist, seed=5)
    ref = oracle_all(x, w, gy, n, S, disc)
    check(hip_all(x, w, gy, n, S, disc, "dense", "dense", "dense"), ref, "dense")
    if dist == "normal" and not disc:
        ref3 = oracle_all(x, w, gy, n, S, disc, length=4.0, rescale_output=True)
        got3 = hip_all(x, w, gy, n, S, disc, "dense", "dense", "dense", length=4.0,
                       rescale_output=True)
        check(got3, ref3, "dense/length4")


@pytest.mark.parametrize("case", DENSE_CASES, ids=lambda c: "-".join(str(v) for v in c))
def test_bf16x3_forward(case):
    """bf16 matrix cores with 3-way split operands: same 1e-5 bar as the fp32 kernels."""
    disc, n, S, I, O, B, dist = case
    x, w, gy = make_inputs(B, I, O, n, S, disc, dist, seed=7)
    ref = oracle_all(x, w, gy, n, S, disc)
    got = hip_all(x, w, gy, n, S, disc, "bf16x3", "dense", "dense")
    check(got, ref, "bf16x3")
    print("bf16x3 y rel err", rel_err(got["y"], ref["y"]))


@pytest.mark.parametrize("case", DENSE_CASES, ids=lambda c: "-".join(str(v) for v in c))
def test_bf16x3_backward_input(case):
    """gy x w^T on the bf16 matrix cores (both operands split 3-way), 1e-5 bar."""
    disc, n, S, I, O, B, dist = case
    x, w, gy = make_inputs(B, I, O, n, S, disc, dist, seed=8)
    ref = oracle_all(x, w, gy, n, S, disc)
    got = hip_all(x, w, gy, n, S, disc, "dense", "bf16x3", "dense")
    check(got, ref, "bf16x3 bwd_x")
    if dist == "normal" and not disc:
        ref3 = oracle_all(x, w, gy, n, S, disc, length=4.0, rescale_output=True)
        got3 = hip_all(x, w, gy, n, S, disc, "bf16x3", "bf16x3", "dense", length=4.0,
                       rescale_output=True)
        check(got3, ref3, "bf16x3/length4")


@pytest.mark.parametrize("case", DENSE_CASES, ids=lambda c: "-".join(str(v) for v in c))
def test_bf16x3_backward_weight(case):
    """Phi^T x gy on the bf16 matrix cores (both operands split 3-way), 1e-5 bar."""
    disc, n, S, I, O, B, dist = case
    x, w, gy = make_inputs(B, I, O, n, S, disc, dist, seed=10)
    ref = oracle_all(x, w, gy, n, S, disc)
    got = hip_all(x, w, gy, n, S, disc, "dense", "dense", "bf16x3")
    check(got, ref, "bf16x3 bwd_w")
    if dist == "normal" and not disc:
        ref3 = oracle_all(x, w, gy, n, S, disc, length=4.0, rescale_output=True)
        got3 = hip_all(x, w, gy, n, S, disc, "bf16x3", "bf16x3", "bf16x3", length=4.0,
                       rescale_output=True)
        check(got3, ref3, "bf16x3 all/length4")


def test_bf16x3_backward_input_large_rows():
    """64-row tiles (B >= 4096) and a ragged last tile."""
    x, w, gy = make_inputs(4096 + 37, 250, 250, 3, 2, False, "normal", seed=9)
    ref = oracle_all(x, w, gy, 3, 2, False)
    check(hip_all(x, w, gy, 3, 2, False, "bf16x3", "bf16x3", "bf16x3"), ref, "bf16x3 large")


@pytest.mark.parametrize("case", DENSE_CASES, ids=lambda c: "-".join(str(v) for v in c))
def test_fp16x2_forward(case):
    """fp16 matrix cores, operands split 2-way, weights scaled per tensor: 1e-5 bar."""
    disc, n, S, I, O, B, dist = case
    x, w, gy = make_inputs(B, I, O, n, S, disc, dist, seed=12)
    ref = oracle_all(x, w, gy, n, S, disc)
    got = hip_all(x, w, gy, n, S, disc, "fp16x2", "dense", "dense")
    check(got, ref, "fp16x2")
    print("fp16x2 y rel err", rel_err(got["y"], ref["y"]))
    # tiny and huge weights: the per-tensor power-of-two scale keeps the pieces in range
    for wscale in (1e-6, 3e4):
        ref2 = oracle_all(x, w * wscale, gy, n, S, disc)
        got2 = hip_all(x, w * wscale, gy, n, S, disc, "fp16x2", "dense", "dense")
        assert rel_err(got2["y"], ref2["y"]) <= TOL, wscale


@pytest.mark.parametrize("case", DENSE_CASES, ids=lambda c: "-".join(str(v) for v in c))
def test_fp16x2_backward_input(case):
    disc, n, S, I, O, B, dist = case
    x, w, gy = make_inputs(B, I, O, n, S, disc, dist, seed=14)
    ref = oracle_all(x, w, gy, n, S, disc)
    got = hip_all(x, w, gy, n, S, disc, "dense", "fp16x2", "dense")
    check(got, ref, "fp16x2 bwd_x")
    # rows of very different magnitude (per-row scales), tiny overall scale
    gy2 = gy * torch.logspace(-12, 3, B).unsqueeze(1)
    ref2 = oracle_all(x, w * 1e-5, gy2, n, S, disc)
    got2 = hip_all(x, w * 1e-5, gy2, n, S, disc, "dense", "fp16x2", "dense")
    rr, gg = ref2["gx"], got2["gx"]
    # row-wise: each row against its own largest entry
    denom = rr.abs().amax(dim=1, keepdim=True).clamp_min(1e-30)
    assert ((gg - rr).abs() / denom).max().item() <= TOL


def test_fp16x2_backward_input_wide_out():
    """More than two 128-out chunks: accumulators are rescaled between chunks."""
    x, w, gy = make_inputs(200, 64, 600, 2, 2, False, "normal", seed=15)
    gy[:, 300:] *= 1e4   # the row maximum moves from chunk to chunk
    gy[:, :128] *= 1e-3
    ref = oracle_all(x, w, gy, 2, 2, False)
    check(hip_all(x, w, gy, 2, 2, False, "dense", "fp16x2", "dense"), ref, "fp16x2 wide")


@pytest.mark.parametrize("case", DENSE_CASES, ids=lambda c: "-".join(str(v) for v in c))
def test_fp16x2_backward_weight(case):
    disc, n, S, I, O, B, dist = case
    x, w, gy = make_inputs(B, I, O, n, S, disc, dist, seed=17)
    ref = oracle_all(x, w, gy, n, S, disc)
    check(hip_all(x, w, gy, n, S, disc, "dense", "dense", "fp16x2"), ref, "fp16x2 bwd_w")
    # tiny gradients (the usual case: gy ~ 1/(B*O)) and columns of very different size
    gy2 = gy * 1e-9 * torch.logspace(0, 6, O).unsqueeze(0)
    ref2 = oracle_all(x, w, gy2, n, S, disc)
    got2 = hip_all(x, w, gy2, n, S, disc, "dense", "dense", "fp16x2")
    rr, gg = ref2["gw"], got2["gw"]          # [O, I, K]: column-wise bar
    denom = rr.abs().amax(dim=(1, 2), keepdim=True).clamp_min(1e-38)
    assert ((gg - rr).abs() / denom).max().item() <= TOL


def test_fp16x2_backward_weight_row_blocks_and_range():
    """Several 512-row scale blocks with different magnitudes (accumulators rescaled in
    between) and a basis value beyond fp16 (that block recomputes its tile in fp32)."""
    x, w, gy = make_inputs(9000, 96, 250, 3, 2, False, "normal", seed=18)
    gy[:512] *= 1e-6
    gy[1024:2048] *= 1e5
    gy[4000:] *= 3e-3
    ref = oracle_all(x, w, gy, 3, 2, False)
    check(hip_all(x, w, gy, 3, 2, False, "fp16x2", "fp16x2", "fp16x2"), ref, "fp16x2 blocks")
    x[77, 40] = 900.0   # basis ~ 1e6 in input tile 1
    ref = oracle_all(x, w, gy, 3, 2, False)
    got = hip_all(x, w, gy, 3, 2, False, "dense", "dense", "fp16x2")
    rr, gg = ref["gw"], got["gw"]
    for sl in (slice(0, 32), slice(32, 64), slice(64, 96)):   # per 32-input tile
        assert rel_err(gg[:, sl], rr[:, sl]) <= TOL, sl


# every (n, K) instantiation of the matrix-core kernels, in the large-batch variants
# (64-row tiles need B >= 8192; O <= 128 selects the 32-out wave tiles)
_COMBOS = [(1, 1, True), (1, 2, True), (1, 4, True), (1, 8, True), (2, 1, True), (2, 1, False),
           (2, 2, True), (2, 2, False), (2, 4, True), (2, 4, False), (3, 1, True), (3, 2, False),
           (3, 2, True), (4, 1, True), (4, 2, False), (4, 2, True), (5, 1, True), (6, 1, True),
           (7, 1, True), (8, 1, True)]


@pytest.mark.parametrize("combo", _COMBOS, ids=lambda c: f"n{c[0]}-S{c[1]}-{'d' if c[2] else 'c'}")
@pytest.mark.parametrize("O", [100, 160])
def test_fp16x2_all_instantiations_large_batch(combo, O):
    n, S, disc = combo
    if not disc and n < 2:
        pytest.skip("continuous layers need n >= 2")
    B, I = 8200, 70
    x, w, gy = make_inputs(B, I, O, n, S, disc, "normal", seed=20 + n + S)
    ref = oracle_all(x, w, gy, n, S, disc)
    check(hip_all(x, w, gy, n, S, disc, "fp16x2", "fp16x2", "fp16x2"), ref, "fp16x2 large")
    check(hip_all(x, w, gy, n, S, disc), ref, "auto large")   # pre-split weights + auto


@pytest.mark.parametrize("shape", [(300, 250, 250, 3, 2, False), (9000, 250, 250, 3, 2, False),
                                   (8300, 250, 128, 3, 2, False), (700, 40, 70, 2, 4, True),
                                   (513, 64, 128, 4, 2, False)],
                         ids=lambda s: "-".join(str(v) for v in s))
def test_fused_layer_norm_epilogue(shape):
    """layer -> LazyLayerNormLastDim as one kernel: forward and all gradients against the
    two-module path on the GPU and against the fp64 oracle."""
    from oracle import pp_oracle as po
    PF = _pf()
    B, I, O, n, S, disc = shape
    x, w, gy = make_inputs(B, I, O, n, S, disc, "normal", seed=30)
    spec = PF.make_spec(n, S, disc)
    xd = x.cuda().requires_grad_(True)
    wd = w.cuda().permute(1, 2, 0).contiguous().permute(2, 0, 1).requires_grad_(True)
    assert PF.layer_norm_fusable(xd, wd, spec)
    y = PF.piecewise_poly_layer_norm(xd, wd, spec, 1e-5)
    y.backward(gy.cuda())
    x2 = x.cuda().requires_grad_(True)
    w2 = wd.detach().clone().requires_grad_(True)
    y2 = PF.layer_norm_last(PF.piecewise_poly(x2, w2, spec), 1e-5)
    y2.backward(gy.cuda())
    assert rel_err(y.detach().cpu(), y2.detach().cpu()) <= 2e-6
    assert rel_err(xd.grad.cpu(), x2.grad.cpu()) <= 2e-5
    assert rel_err(wd.grad.cpu(), w2.grad.cpu()) <= 2e-5
    xo = x.float().clone().requires_grad_(True)   # fp32 leaf: the oracle takes ids from fp32
    wo = w.double().requires_grad_(True)
    yo = po.layer_norm_last_dim(po.piecewise_forward_f64(xo, wo, n, S, disc), 1e-5)
    yo.backward(gy.double())
    assert rel_err(y.detach().cpu(), yo.detach()) <= TOL
    # Gradients through the LayerNorm backward (a difference of row sums): the yardstick is the
    # reference formulation of the pair evaluated in fp32 -- within the bar, or no more than
    # twice as far from the fp64 result as that evaluation is.
    x32 = x.float().clone().requires_grad_(True)
    w32 = w.float().clone().requires_grad_(True)
    po.layer_norm_last_dim(po.piecewise_forward(x32, w32, n, S, disc), 1e-5).backward(gy.float())
    for got, ref, ref32 in ((xd.grad.cpu(), xo.grad, x32.grad), (wd.grad.cpu(), wo.grad, w32.grad)):
        e, e32 = rel_err(got, ref), rel_err(ref32, ref)
        assert e <= max(TOL, 2 * e32), (e, e32)


def test_fused_layer_norm_out_of_range_basis_and_network_equivalence():
    import language_interpolation_amd as L
    PF = _pf()
    # far extrapolation in one row: the block recomputes its tile in fp32 and normalises it
    x, w, gy = make_inputs(300, 250, 250, 3, 2, False, "normal", seed=31)
    x[17, 5] = 900.0
    spec = PF.make_spec(3, 2, False)
    wd = w.cuda().permute(1, 2, 0).contiguous().permute(2, 0, 1)
    y = PF.piecewise_poly_layer_norm(x.cuda(), wd, spec, 1e-5)
    y2 = PF.layer_norm_last(PF.piecewise_poly(x.cuda(), wd, PF.make_spec(3, 2, False, fwd_method="dense")), 1e-5)
    assert rel_err(y.cpu(), y2.cpu()) <= TOL
    # HighOrderMLP.forward (fused pairs) == its Sequential run module by module
    torch.manual_seed(0)
    net = L.HighOrderMLP(layer_type="continuous", n=3, in_width=16, out_width=128, hidden_layers=2,
                         hidden_width=250, in_segments=128, out_segments=2, hidden_segments=2,
                         n_in=2, normalization=L.LazyLayerNormLastDim, device="cuda")
    xb = (torch.rand(600, 16) * 2 - 1).cuda()
    a = net(xb)
    b = net.model(xb)
    assert rel_err(a.detach().cpu(), b.detach().cpu()) <= 1e-5


def test_prepacked_weights_match_per_call_packing():
    """pp_pack_weights_f32 + PP_WS_PREPACKED give bit-identical results to the per-call path."""
    PF = _pf()
    x, w, gy = make_inputs(700, 250, 128, 3, 2, False, "normal", seed=16)
    xd, gyd = x.cuda(), gy.cuda()
    w_t = w.permute(1, 2, 0).contiguous().cuda()
    spec = PF.make_spec(3, 2, False)
    packed = PF.pack_weights(w_t, spec, x.shape[0])
    assert packed is not None
    y1, _ = PF.forward_raw(xd, w_t, spec, packed=packed)
    y2, _ = PF.forward_raw(xd, w_t, PF.make_spec(3, 2, False, fwd_method="fp16x2"))
    assert torch.equal(y1, y2)
    g1 = PF.backward_input_raw(xd, gyd, w_t, spec, packed=packed)
    g2 = PF.backward_input_raw(xd, gyd, w_t, PF.make_spec(3, 2, False, bwd_x_method="fp16x2"))
    assert torch.equal(g1, g2)


def test_fp16x2_forward_out_of_range_basis():
    """Far extrapolation: basis values beyond fp16 -> the tile is recomputed in fp32."""
    x, w, gy = make_inputs(300, 250, 250, 3, 2, False, "normal", seed=13)
    x[5, 17] = 700.0      # l ~ 1e6
    x[211, 0] = -2.0e4    # l ~ 1e9
    ref = oracle_all(x, w, gy, 3, 2, False)
    got = hip_all(x, w, gy, 3, 2, False, "fp16x2", "dense", "dense")
    # rows are independent: compare row-wise so the huge rows do not hide the others
    for rows in (slice(0, 5), slice(5, 6), slice(6, 211), slice(211, 212), slice(212, 300)):
        assert rel_err(got["y"][rows], ref["y"][rows]) <= TOL, rows


def test_dense_rejects_inexact_layers():
    """S or length not a power of two -> the dense kernels refuse (gather path serves them)."""
    from language_interpolation_amd._lib import PPHipError
    x, w, gy = make_inputs(16, 33, 31, 2, 3, False, "normal", seed=6)
    with pytest.raises(PPHipError):
        hip_all(x, w, gy, 2, 3, False, "dense", "gather", "atomic")
    ref = oracle_all(x, w, gy, 2, 3, False)
    check(hip_all(x, w, gy, 2, 3, False), ref, "auto S=3")


def test_golden_vectors():
    z = np.load(GOLDEN, allow_pickle=False)
    for name in sorted({k.split("/")[0] for k in z.files}):
        disc, n, S, I, O, B = (int(v) for v in z[name + "/meta"])
        x, w, gy = (torch.from_numpy(z[f"{name}/{k}"]) for k in ("x", "w", "gy"))
        ref = {k: torch.from_numpy(z[f"{name}/{k}"]) for k in ("y", "seg", "gx", "gw")}
        for bw in ("atomic", "sort", "auto"):
            check(hip_all(x, w, gy, n, S, bool(disc), bwd_w=bw), ref, f"{name}/{bw}")
        K = n * S if disc else (n - 1) * S + 1
        if K <= 8 and S & (S - 1) == 0:
            check(hip_all(x, w, gy, n, S, bool(disc), "dense", "dense", "dense"), ref,
                  f"{name}/dense")


@pytest.mark.parametrize("S", [2, 8, 128, 150, 1000])
def test_segment_ids_bit_exact_sweep(S):
    """P8 of SURVEY 8c: every fp32 value within a few ulp of each boundary, plus
    out-of-range and non-finite inputs."""
    from oracle import pp_oracle as po
    PF = _pf()
    xs = []
    for m in range(S + 1):
        v = np.float32(-1.0 + 2.0 * m / S)
        for _ in range(6):
            v = np.nextafter(v, np.float32(-3.0))
        for _ in range(13):
            xs.append(v)
            v = np.nextafter(v, np.float32(3.0))
    xs += [-3.0, 3.0, -1e9, 1e9, 0.0, -0.0, 1e30, -1e30, np.inf, -np.inf, np.nan, 1e-45]
    x = torch.tensor(np.asarray(xs, dtype=np.float32)).reshape(-1, 1)
    want = po.segment_ids(x, S).flatten()
    got = PF.segment_ids(x.cuda(), S).cpu().flatten().long()
    assert torch.equal(got, want)


def test_character_grid_is_exact():
    """P5: x=(c-63.5)/64 with S=128 gives s=c and xi=0, so y is a pure table lookup."""
    from oracle import pp_oracle as po
    PF = _pf()
    c = torch.arange(128).reshape(-1, 1)
    x = po.ascii_to_float(c)
    w = torch.randn(5, 1, 129)
    spec = PF.make_spec(2, 128, False)
    y, seg = PF.forward_raw(x.cuda(), w.cuda().permute(1, 2, 0).contiguous(), spec, want_seg=True)
    assert torch.equal(seg.cpu().flatten().long(), c.flatten())
    want = 0.5 * (w[:, 0, :128] + w[:, 0, 1:]).t()
    assert torch.allclose(y.cpu(), want, atol=1e-6)


def test_length_rescale_and_properties():
    PF = _pf()
    # non-default length and rescale_output against the oracle
    disc, n, S, I, O, B = False, 3, 4, 12, 9, 40
    x, w, gy = make_inputs(B, I, O, n, S, disc, "normal", seed=3)
    for length in (2.0, 3.0):
        ref = oracle_all(x, w, gy, n, S, disc, length=length, rescale_output=True)
        for bw in ("atomic", "sort"):
            got = hip_all(x, w, gy, n, S, disc, bwd_w=bw, length=length, rescale_output=True)
            check(got, ref, f"length={length}/{bw}")
    # P1 partition of unity at full size: w == 1 -> y == I wherever x is
    B, I, O = 4096, 200, 200
    x = (torch.randn(B, I) * 2).cuda()
    for disc, n, S in ((True, 1, 150), (False, 3, 2), (True, 2, 128)):
        K = PF.num_weights(n, S, disc)
        spec = PF.make_spec(n, S, disc)
        y, _ = PF.forward_raw(x, torch.ones(I, K, O, device="cuda"), spec)
        assert (y - I).abs().max().item() <= 1e-5 * I * 40


def test_linearity_and_scatter_consistency_at_full_size():
    """Size-independent properties at a BASELINE-sized shape: forward is linear in
    w; <gy, y(w)> == <gw, w> (the weight gradient is the adjoint of the forward)."""
    PF = _pf()
    for disc, n, S, I, O, B, dist in ((True, 1, 150, 200, 200, 16384, "normal"),
                                      (True, 2, 128, 16, 1000, 1024, "char"),
                                      (False, 3, 2, 250, 250, 8192, "normal")):
        x, w, gy = make_inputs(B, I, O, n, S, disc, dist, seed=4)
        x, gy = x.cuda(), gy.cuda()
        w_t = w.cuda().permute(1, 2, 0).contiguous()
        spec = PF.make_spec(n, S, disc)
        y1, _ = PF.forward_raw(x, w_t, spec)
        y2, _ = PF.forward_raw(x, 2.5 * w_t, spec)
        assert rel_err(y2, 2.5 * y1) <= 2e-6
        lhs = (gy.double() * y1.double()).sum().item()
        for bw in ("atomic", "sort"):
            spec_b = PF.make_spec(n, S, disc, bwd_w_method=bw)
            gw_t = PF.backward_weight_raw(x, gy, tuple(w_t.shape), spec_b)
            rhs = (gw_t.double() * w_t.double()).sum().item()
            scale = (gy.double().abs() * y1.double().abs()).sum().item()
            assert abs(lhs - rhs) <= 1e-6 * scale, (bw, lhs, rhs)


def test_module_autograd_path_and_empty_batch():
    import language_interpolation_amd as L
    from oracle import pp_oracle as po
    torch.manual_seed(0)
    net = L.HighOrderMLP(layer_type="continuous", n=3, in_width=16, out_width=12, hidden_layers=1,
                         hidden_width=20, in_segments=128, out_segments=2, hidden_segments=2,
                         n_in=2, normalization=L.LazyLayerNormLastDim, device="cuda")
    ora = po.OracleHighOrderMLP(layer_type="continuous", n=3, in_width=16, out_width=12,
                                hidden_layers=1, hidden_width=20, in_segments=128, out_segments=2,
                                hidden_segments=2, n_in=2,
                                normalization=po.OracleLazyLayerNormLastDim)
    ora = ora.double()
    with torch.no_grad():
        for a, b in zip(ora.parameters(), net.parameters()):
            a.copy_(b.detach().cpu().double())
    x = (torch.rand(33, 16) * 2 - 1)
    gy = torch.randn(33, 12)
    xg = x.cuda().requires_grad_(True)
    y = net(xg)
    y.backward(gy.cuda())
    xo = x.double().requires_grad_(True)
    # fp64 oracle network (its segment ids come from the fp32 cast inside segment_ids)
    yo = ora(xo)
    yo.backward(gy.double())
    # the same oracle network evaluated in fp32: the yardstick for a two-LayerNorm chain (within
    # the bar, or no more than twice as far from the fp64 result as the fp32 evaluation is)
    ora32 = po.OracleHighOrderMLP(layer_type="continuous", n=3, in_width=16, out_width=12,
                                  hidden_layers=1, hidden_width=20, in_segments=128, out_segments=2,
                                  hidden_segments=2, n_in=2,
                                  normalization=po.OracleLazyLayerNormLastDim)
    with torch.no_grad():
        for a, b in zip(ora32.parameters(), net.parameters()):
            a.copy_(b.detach().cpu())
    x32 = x.clone().requires_grad_(True)
    y32 = ora32(x32)
    y32.backward(gy)

    def near(got, ref, ref32, bar):
        e, e32 = rel_err(got, ref), rel_err(ref32, ref)
        assert e <= max(bar, 2 * e32), (e, e32)

    near(y.detach(), yo.detach(), y32.detach(), 5 * TOL)
    near(xg.grad, xo.grad, x32.grad, 5 * TOL)
    for a, a32, b in zip(ora.parameters(), ora32.parameters(), net.parameters()):
        assert b.grad.shape == b.shape and b.grad.stride() == b.stride()
        near(b.grad, a.grad, a32.grad, 5 * TOL)
    # ragged / empty input
    empty = net(torch.empty(0, 16, device="cuda"))
    assert empty.shape == (0, 12)
    # reshape_apply on [b, l, d]
    out = L.reshape_apply(torch.rand(7, 5, 16, device="cuda") * 2 - 1, net)
    assert out.shape == (7, 5, 12)


@pytest.mark.parametrize("D", [2, 7, 64, 128, 200, 250, 251, 256, 1000, 1024])
def test_norm_kernels_match_oracle(D):
    """LayerNorm (networks.py:72-91) and MaxAbs normalisation: forward and backward vs the
    fp64 torch restatement in oracle/pp_oracle.py."""
    from oracle import pp_oracle as po
    PF = _pf()
    gen = torch.Generator().manual_seed(D)
    B = 37
    x = torch.randn(B, D, generator=gen) * 2 + 0.3
    gy = torch.randn(B, D, generator=gen)
    for hip_fn, ora_fn, tol in ((PF.layer_norm_last, po.layer_norm_last_dim, 2e-5),
                                (PF.max_abs_norm_last, po.max_abs_normalization_last, 1e-5)):
        if D == 1 and hip_fn is PF.layer_norm_last:
            continue  # variance of one element: y == 0, gradient ill-defined
        xg = x.cuda().requires_grad_(True)
        y = hip_fn(xg)
        y.backward(gy.cuda())
        xo = x.double().requires_grad_(True)
        yo = ora_fn(xo)
        yo.backward(gy.double())
        assert rel_err(y.detach(), yo.detach()) <= tol, hip_fn.__name__
        assert rel_err(xg.grad, xo.grad) <= 5 * tol, hip_fn.__name__
    # leading dims are flattened like the reference's modules allow
    x3 = torch.randn(3, 5, D, generator=gen).cuda()
    assert PF.layer_norm_last(x3).shape == x3.shape
