"""The row-tile forward (PP_FWD_TILE, csrc/pp_tile.hip) through the C ABI against the oracle:
segment ids bit-exact, outputs within 1e-5 of the fp64 evaluation; every template variant
(rows per workgroup, nodes per segment, aligned / unaligned input rows), ragged tails in rows,
outs and inputs, the input split with its fixed-order sum, the LayerNorm fused into that sum,
and the full BASELINE shapes through size-independent properties.  Second half: the row-tile
backward-input (PP_BWD_X_TILE) against the oracle, the gather kernel, and exact linearity in gy
at the bench batch."""
import pytest
import torch

from pp_testlib import TOL, make_inputs, oracle_all, rel_err

pytestmark = pytest.mark.gpu


def _pf():
    from language_interpolation_amd import functional as PF
    return PF


def _lib():
    from language_interpolation_amd import _lib
    return _lib


# (disc, n, S, I, O, B, dist)
TILE_CASES = [
    (True, 1, 150, 200, 200, 1100, "normal"),    # cfg3 hidden: two row tiles, the second 76 rows
    (True, 2, 128, 16, 1000, 1030, "char"),      # cfg4 input: no input split at full size
    (False, 2, 128, 16, 200, 700, "char"),       # cfg1 input layer (continuous: shared nodes)
    (True, 2, 150, 200, 200, 300, "normal"),     # cfg3 with the n_hidden quirk: K = 300, 150 KiB of LDS
    (False, 3, 20, 37, 72, 200, "edges"),        # inputs not a multiple of 4, 8-out tail chunk
    (True, 4, 9, 5, 4, 65, "edges"),             # one 16-byte piece per row
    (False, 4, 30, 64, 128, 513, "uniform"),     # n = 4, one row past a 512-row tile
    (True, 3, 50, 130, 260, 257, "normal"),      # 4-out tail chunk, ragged input split
    (True, 1, 300, 24, 64, 64, "uniform"),       # K = 300 at n = 1
]


@pytest.mark.parametrize("case", TILE_CASES, ids=lambda c: "-".join(str(v) for v in c))
def test_tile_forward_matches_oracle(case):
    disc, n, S, I, O, B, dist = case
    PF = _pf()
    x, w, gy = make_inputs(B, I, O, n, S, disc, dist, seed=21)
    ref = oracle_all(x, w, gy, n, S, disc)
    spec = PF.make_spec(n, S, disc, fwd_method="tile", in_features=I)
    xd = x.cuda()
    w_t = w.cuda().permute(1, 2, 0).contiguous()
    y, seg = PF.forward_raw(xd, w_t, spec, want_seg=True)
    y2, _ = PF.forward_raw(xd, w_t, spec)
    torch.cuda.synchronize()
    assert torch.equal(seg.cpu().long(), ref["seg"].long())
    assert rel_err(y.cpu(), ref["y"]) <= TOL, rel_err(y.cpu(), ref["y"])
    assert torch.equal(y, y2)  # fixed summation order: same bits every launch


@pytest.mark.parametrize("length,rescale", [(4.0, True), (3.0, False), (0.5, False)])
def test_tile_forward_other_lengths(length, rescale):
    """length a power of two (division as an exact multiplication) and not (IEEE division), with
    rescale_output, ids bit-exact in both."""
    PF = _pf()
    disc, n, S, I, O, B = False, 3, 37, 44, 96, 600
    x, w, gy = make_inputs(B, I, O, n, S, disc, "edges", seed=22)
    x = x * (length / 2.0)
    ref = oracle_all(x, w, gy, n, S, disc, length=length, rescale_output=rescale)
    spec = PF.make_spec(n, S, disc, fwd_method="tile", in_features=I, length=length,
                        rescale_output=rescale)
    y, seg = PF.forward_raw(x.cuda(), w.cuda().permute(1, 2, 0).contiguous(), spec, want_seg=True)
    assert torch.equal(seg.cpu().long(), ref["seg"].long())
    assert rel_err(y.cpu(), ref["y"]) <= TOL


def test_tile_forward_non_finite_and_huge_inputs():
    """inf / nan / 1e30 inputs take segment 0 or S-1 exactly as the oracle's `.long()` + clamps."""
    PF = _pf()
    from oracle import pp_oracle as po
    disc, n, S, I, O, B = True, 1, 150, 8, 64, 128
    x, w, _ = make_inputs(B, I, O, n, S, disc, "normal", seed=23)
    x[0, 0], x[1, 1], x[2, 2], x[3, 3], x[4, 4] = float("inf"), float("-inf"), float("nan"), 1e30, -1e30
    x[5, 5], x[6, 6] = 9.3e18, 2.0 ** 62
    spec = PF.make_spec(n, S, disc, fwd_method="tile", in_features=I)
    y, seg = PF.forward_raw(x.cuda(), w.cuda().permute(1, 2, 0).contiguous(), spec, want_seg=True)
    assert torch.equal(seg.cpu().long(), po.segment_ids(x, S))
    # n = 1: the output is a sum of table rows whatever x is
    ref = po.piecewise_forward_f64(x, w.double(), n, S, disc)
    assert rel_err(y.cpu(), ref) <= TOL


def test_tile_refuses_what_it_cannot_run():
    PF, L = _pf(), _lib()
    for (disc, n, S, I, O) in ((True, 1, 150, 16, 66), (True, 5, 8, 16, 64), (True, 2, 200, 16, 64)):
        x, w, _ = make_inputs(64, I, O, n, S, disc, "normal", seed=1)
        spec = PF.make_spec(n, S, disc, fwd_method="tile", in_features=I)
        with pytest.raises(L.PPHipError, match="inapplicable"):
            PF.forward_raw(x.cuda(), w.cuda().permute(1, 2, 0).contiguous(), spec)


def test_auto_picks_the_tile_kernel_for_the_many_segment_layers():
    L = _lib()
    lib = L.load()
    assert lib.pp_auto_method(0, 16384, 200, 200, 1, 150, 1, 2.0) == L.FWD_TILE   # cfg3 hidden
    assert lib.pp_auto_method(0, 16384, 16, 1000, 2, 128, 1, 2.0) == L.FWD_TILE   # cfg4 input
    assert lib.pp_auto_method(0, 16384, 250, 250, 3, 2, 0, 2.0) == L.FWD_FP16X2   # dense regime
    assert lib.pp_auto_method(0, 256, 200, 200, 1, 150, 1, 2.0) == L.FWD_GATHER   # small batch
    # scratch: the per-row records, plus the partial outputs where the inputs are split
    assert lib.pp_forward_workspace_bytes(16384, 16, 1000, 2, 128, 1, 2.0, L.FWD_TILE) == 16 * 3 * 16384 * 4
    assert lib.pp_forward_workspace_bytes(16384, 200, 200, 1, 150, 1, 2.0, L.FWD_TILE) > 4 * 16384 * 200 * 4
    # without a workspace the automatic choice stays on the kernels that need none
    x = torch.zeros(4096, 200, device="cuda")
    w_t = torch.zeros(200, 150, 200, device="cuda")
    y = torch.empty(4096, 200, device="cuda")
    PF = _pf()
    spec = PF.make_spec(1, 150, True, in_features=200)
    rc = lib.pp_forward_f32(PF._ptr(x), PF._ptr(w_t), PF._ptr(y), PF._ptr(None), 4096, 200, 200, 1,
                            150, 1, 2.0, spec.node_array(), 1.0, L.METHOD_AUTO, PF._stream(x.device))
    assert rc == 0
    rc = lib.pp_forward_f32(PF._ptr(x), PF._ptr(w_t), PF._ptr(y), PF._ptr(None), 4096, 200, 200, 1,
                            150, 1, 2.0, spec.node_array(), 1.0, L.FWD_TILE, PF._stream(x.device))
    assert rc == -4  # asked for by name without its scratch: PP_ERR_WORKSPACE
    torch.cuda.synchronize()


@pytest.mark.parametrize("case", [(True, 1, 150, 200, 200, 5200, "normal"),
                                  (True, 2, 128, 16, 1000, 16384, "char"),
                                  (True, 2, 150, 200, 200, 4100, "normal")],
                         ids=lambda c: "-".join(str(v) for v in c))
def test_tile_forward_with_fused_layer_norm(case):
    """mode 3 of pp_forward_layer_norm_f32: the kernel that sums the tile kernel's partial
    outputs normalises the rows.  Output, rstd and both gradients against the two modules run
    one after the other (tile forward, LayerNorm kernel), output against the fp64 oracle."""
    disc, n, S, I, O, B, dist = case
    PF, L = _pf(), _lib()
    from oracle import pp_oracle as po
    lib = L.load()
    assert lib.pp_forward_layer_norm_supported(B, I, O, n, S, int(disc), 2.0) == 3
    assert lib.pp_forward_layer_norm_workspace_bytes(B, I, O, n, S, int(disc), 2.0) >= B * O * 4
    x, w, gy = make_inputs(B, I, O, n, S, disc, dist, seed=24)
    spec = PF.make_spec(n, S, disc, in_features=I)
    wd0 = w.cuda().permute(1, 2, 0).contiguous().permute(2, 0, 1)
    res = []
    for fn in (lambda a, b: PF.piecewise_poly_layer_norm(a, b, spec, 1e-5),
               lambda a, b: PF.layer_norm_last(PF.piecewise_poly(a, b, spec), 1e-5)):
        xd = x.cuda().requires_grad_(True)
        wd = wd0.detach().clone().requires_grad_(True)
        y = fn(xd, wd)
        y.backward(gy.cuda())
        res.append((y.detach().cpu(), xd.grad.cpu(), wd.grad.cpu()))
    assert torch.equal(res[0][0], res[1][0])  # same partial sums, same row arithmetic: same bits
    assert torch.equal(res[0][1], res[1][1])
    # (the weight gradient of K > 192 layers goes through the sorted path, whose low bits depend
    # on the order of rows inside a bucket: equal inputs, not necessarily equal bits)
    assert rel_err(res[0][2], res[1][2]) <= 1e-6
    # (the fp64 oracle on a slice of the rows: its expanded basis [B, I, K] is what limits it)
    sl = slice(0, 700)
    yo = po.layer_norm_last_dim(po.piecewise_forward_f64(x[sl], w.double(), n, S, disc), 1e-5)
    assert rel_err(res[0][0][sl], yo) <= 5 * TOL


@pytest.mark.parametrize("shape", [(True, 1, 150, 200, 200), (True, 2, 128, 16, 1000),
                                   (False, 2, 128, 16, 200), (True, 1, 150, 200, 128)],
                         ids=lambda c: "-".join(str(v) for v in c))
def test_tile_forward_at_the_bench_batch(shape):
    """16384 rows (what bench.py runs), too large for the oracle: partition of unity (w = 1 gives
    y = I for every x, out of range included), ids and outputs against the gather kernel (the
    oracle-checked fp32 kernel with a different summation order), same bits on a second launch."""
    disc, n, S, I, O = shape
    B = 16384
    PF = _pf()
    dist = "char" if I == 16 else "normal"
    x, w, _ = make_inputs(B, I, O, n, S, disc, dist, seed=25)
    xd = x.cuda()
    w_t = w.cuda().permute(1, 2, 0).contiguous()
    tile = PF.make_spec(n, S, disc, fwd_method="tile", in_features=I)
    gather = PF.make_spec(n, S, disc, fwd_method="gather", in_features=I)
    y, seg = PF.forward_raw(xd, w_t, tile, want_seg=True)
    yg, segg = PF.forward_raw(xd, w_t, gather, want_seg=True)
    y2, _ = PF.forward_raw(xd, w_t, tile)
    assert torch.equal(seg, segg) and torch.equal(y, y2)
    assert rel_err(y, yg) <= 2e-6
    ones = torch.ones_like(w_t)
    y1, _ = PF.forward_raw(xd * 1.7, ones, tile)
    assert float((y1 - I).abs().max()) <= 1e-5 * I


# (disc, n, S, I, O, B, dist): the row-tile backward-input (PP_BWD_X_TILE)
TILE_BWDX_CASES = [
    (True, 2, 150, 200, 200, 1100, "normal"),    # cfg3 hidden as select_network builds it: K = 300
    (True, 2, 128, 16, 1000, 1030, "char"),      # 16 out chunks summed, no input split
    (False, 2, 128, 16, 200, 700, "char"),       # continuous, 8-out tail chunk
    (False, 3, 20, 36, 72, 200, "edges"),        # n = 3, boundary values
    (False, 4, 30, 64, 128, 513, "uniform"),     # n = 4, one row past a 512-row tile
    (True, 3, 50, 132, 260, 257, "normal"),      # ragged input split, 4-out tail chunk
]


@pytest.mark.parametrize("case", TILE_BWDX_CASES, ids=lambda c: "-".join(str(v) for v in c))
def test_tile_backward_input_matches_oracle(case):
    disc, n, S, I, O, B, dist = case
    PF = _pf()
    x, w, gy = make_inputs(B, I, O, n, S, disc, dist, seed=27)
    ref = oracle_all(x, w, gy, n, S, disc)
    xd, gyd = x.cuda(), gy.cuda()
    w_t = w.cuda().permute(1, 2, 0).contiguous()
    tile = PF.make_spec(n, S, disc, bwd_x_method="tile", in_features=I)
    gx = PF.backward_input_raw(xd, gyd, w_t, tile)
    assert torch.equal(PF.backward_input_raw(xd, gyd, w_t, tile), gx)  # fixed order of every sum
    assert rel_err(gx.cpu(), ref["gx"]) <= TOL, rel_err(gx.cpu(), ref["gx"])
    gather = PF.make_spec(n, S, disc, bwd_x_method="gather", in_features=I)
    assert rel_err(gx, PF.backward_input_raw(xd, gyd, w_t, gather)) <= 2e-6


def test_tile_backward_input_with_scale_length_and_refusals():
    PF, L = _pf(), _lib()
    disc, n, S, I, O, B = False, 3, 37, 44, 96, 600
    x, w, gy = make_inputs(B, I, O, n, S, disc, "edges", seed=28)
    x = x * 1.5
    ref = oracle_all(x, w, gy, n, S, disc, length=3.0, rescale_output=True)
    spec = PF.make_spec(n, S, disc, bwd_x_method="tile", in_features=I, length=3.0, rescale_output=True)
    gx = PF.backward_input_raw(x.cuda(), gy.cuda(), w.cuda().permute(1, 2, 0).contiguous(), spec)
    assert rel_err(gx.cpu(), ref["gx"]) <= TOL
    for (disc, n, S, I, O) in ((True, 1, 150, 16, 64), (True, 2, 40, 18, 64), (True, 2, 40, 16, 66)):
        x, w, gy = make_inputs(64, I, O, n, S, disc, "normal", seed=1)
        spec = PF.make_spec(n, S, disc, bwd_x_method="tile", in_features=I)
        with pytest.raises(L.PPHipError, match="inapplicable"):
            PF.backward_input_raw(x.cuda(), gy.cuda(), w.cuda().permute(1, 2, 0).contiguous(), spec)


@pytest.mark.parametrize("shape", [(True, 2, 150, 200, 200), (False, 2, 128, 64, 256)],
                         ids=lambda c: "-".join(str(v) for v in c))
def test_tile_backward_input_at_the_bench_batch(shape):
    """16384 rows: against the gather kernel (oracle-checked, different summation order), same
    bits on a second launch, and linearity in gy (gx(a*gy) = a*gx(gy) for a power of two, exactly)."""
    disc, n, S, I, O = shape
    B = 16384
    PF = _pf()
    x, w, gy = make_inputs(B, I, O, n, S, disc, "normal", seed=29)
    xd, gyd = x.cuda(), gy.cuda()
    w_t = w.cuda().permute(1, 2, 0).contiguous()
    tile = PF.make_spec(n, S, disc, bwd_x_method="tile", in_features=I)
    gather = PF.make_spec(n, S, disc, bwd_x_method="gather", in_features=I)
    gx = PF.backward_input_raw(xd, gyd, w_t, tile)
    assert torch.equal(PF.backward_input_raw(xd, gyd, w_t, tile), gx)
    assert rel_err(gx, PF.backward_input_raw(xd, gyd, w_t, gather)) <= 2e-6
    assert torch.equal(PF.backward_input_raw(xd, gyd * 4.0, w_t, tile), gx * 4.0)
