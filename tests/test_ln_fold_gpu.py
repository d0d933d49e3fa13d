"""LayerNorm backward folded into the backward of the layer below (reference:
the layer -> LazyLayerNormLastDim -> layer chains of HighOrderMLP, networks.py:866-882, autograd
through torch.nn.LayerNorm, networks.py:72-91).

Two halves, both through the C ABI: the backward-input kernel of the layer above leaves the row
sums (pp_backward_input_stats_ws_f32), the pack pass of the layer below applies the LayerNorm
backward on its way (pp_backward_weight_layer_norm_f32).  Checked against the separate kernels
(pp_layer_norm_bwd_f32 + pp_backward_weight_f32), the fp64 oracle of the chain, and through
HighOrderMLP against the same network with the fold switched off."""
import pytest
import torch

from pp_testlib import TOL, make_inputs, rel_err

pytestmark = pytest.mark.gpu

PP_ERR_SHAPE, PP_ERR_WORKSPACE, PP_ERR_METHOD = -2, -4, -5   # include/pp_hip.h


def _pf():
    import language_interpolation_amd.functional as PF
    return PF


# (B, I, O, n, S, disc): layers whose backward-input runs on the fp16x2 kernel
STATS_CASES = [
    (16384, 250, 128, 3, 2, False),   # cfg2 output layer at the bench batch (8 tiles, last one 26 wide)
    (4096, 250, 250, 3, 2, False),    # cfg2 hidden layer
    (1000, 200, 200, 2, 2, False),    # cfg1 hidden at the reference's batch: ragged last row tile
    (513, 64, 128, 4, 2, False),      # two tiles, one row past a tile
    (700, 40, 70, 2, 4, True),        # discontinuous, 40 inputs: second tile 8 wide
]


@pytest.mark.parametrize("shape", STATS_CASES, ids=lambda s: "-".join(str(v) for v in s))
def test_backward_input_row_stats(shape):
    """gx is bit-identical to the plain call; the tile sums add up to sum_i gx and sum_i gx*x."""
    from language_interpolation_amd import _lib
    PF = _pf()
    lib = _lib.load()
    B, I, O, n, S, disc = shape
    assert lib.pp_backward_input_stats_supported(B, I, O, n, S, int(disc), 2.0) == 1
    x, w, gy = make_inputs(B, I, O, n, S, disc, "normal", seed=41)
    spec = PF.make_spec(n, S, disc)
    xd, gyd = x.cuda(), gy.cuda()
    w_t = w.cuda().permute(1, 2, 0).contiguous()
    tiles = lib.pp_row_stats_tiles(I)
    assert tiles == (I + 31) // 32
    for packed in (None, PF.pack_weights(w_t, spec, B)):
        stats = torch.full((tiles, B, 2), float("nan"), device="cuda")
        gx = PF.backward_input_raw(xd, gyd, w_t, spec, packed=packed, row_stats=stats)
        gx0 = PF.backward_input_raw(xd, gyd, w_t, spec, packed=packed)
        torch.cuda.synchronize()
        assert torch.equal(gx, gx0)
        assert torch.isfinite(stats).all()
        s = stats.double().sum(dim=0).cpu()
        ref1 = gx.double().sum(dim=1).cpu()
        ref2 = (gx.double() * xd.double()).sum(dim=1).cpu()
        scale = gx.double().abs().sum(dim=1).cpu().clamp_min(1e-30)
        assert ((s[:, 0] - ref1).abs() / scale).max() <= 1e-6
        assert ((s[:, 1] - ref2).abs() / scale).max() <= 4e-6   # |x| up to ~4 for N(0,1) inputs
        # each tile on its own
        gxt = torch.nn.functional.pad(gx.double(), (0, tiles * 32 - I)).view(B, tiles, 32).sum(dim=2)
        assert ((stats[:, :, 0].double().t() - gxt).abs().cpu() / scale[:, None]).max() <= 1e-6
    # a method without the epilogue is refused, not silently run
    spec_g = PF.make_spec(n, S, disc, bwd_x_method="gather")
    with pytest.raises(_lib.PPHipError, match="inapplicable"):
        PF.backward_input_raw(xd, gyd, w_t, spec_g, row_stats=stats)


# (B, I, O, n, S, disc): layers whose weight gradient PP_METHOD_AUTO runs on the fp16x2 kernel
FOLD_CASES = [
    (16384, 250, 250, 3, 2, False),   # cfg2 hidden layer at the bench batch
    (8300, 250, 250, 3, 2, False),    # ragged last scale block (8300 = 16*512 + 108)
    (4096, 200, 200, 2, 2, False),    # cfg1 hidden
    (8192, 64, 96, 4, 2, False),      # O <= 128: from 8192 rows
]


@pytest.mark.parametrize("shape", FOLD_CASES, ids=lambda s: "-".join(str(v) for v in s))
def test_weight_gradient_with_folded_layer_norm_backward(shape):
    """g and gw_t of pp_backward_weight_layer_norm_f32 against pp_layer_norm_bwd_f32 +
    pp_backward_weight_f32 (same expression, the two row means summed in another order) and
    against the fp64 LayerNorm backward."""
    from language_interpolation_amd import _lib
    PF = _pf()
    lib = _lib.load()
    B, I, O, n, S, disc = shape
    assert lib.pp_backward_weight_layer_norm_supported(B, I, O, n, S, int(disc), 2.0) == 1
    x, w, gy = make_inputs(B, I, O, n, S, disc, "normal", seed=43)
    spec = PF.make_spec(n, S, disc)
    xd, gyd = x.cuda(), (gy * 3.0).cuda()
    w_t = w.cuda().permute(1, 2, 0).contiguous()
    raw, _ = PF.forward_raw(xd, w_t, spec)
    yn = torch.empty_like(raw)
    rstd = torch.empty(B, device="cuda")
    assert lib.pp_layer_norm_fwd_f32(PF._ptr(raw), PF._ptr(yn), PF._ptr(rstd), B, O, 1e-5,
                                     PF._stream(raw.device)) == 0
    # the row sums as the kernel above would leave them: tiles of 32 columns of gy / gy*yn
    tiles = lib.pp_row_stats_tiles(O)
    pad = tiles * 32 - O
    s1 = torch.nn.functional.pad(gyd, (0, pad)).view(B, tiles, 32).sum(dim=2)
    s2 = torch.nn.functional.pad(gyd * yn, (0, pad)).view(B, tiles, 32).sum(dim=2)
    stats = torch.stack((s1.t(), s2.t()), dim=2).contiguous()
    g, gw = PF.backward_weight_layer_norm_raw(xd, gyd, yn, rstd, stats, tuple(w_t.shape), spec)
    g2, gw2 = PF.backward_weight_layer_norm_raw(xd, gyd, yn, rstd, stats, tuple(w_t.shape), spec)
    # separate kernels
    g0 = torch.empty_like(raw)
    assert lib.pp_layer_norm_bwd_f32(PF._ptr(yn), PF._ptr(gyd), PF._ptr(rstd), PF._ptr(g0), B, O,
                                     PF._stream(raw.device)) == 0
    gw0 = PF.backward_weight_raw(xd, g0, tuple(w_t.shape), spec)
    torch.cuda.synchronize()
    assert torch.equal(g, g2) and torch.equal(gw, gw2)          # reproducible
    assert rel_err(g.cpu(), g0.cpu()) <= 2e-6
    assert rel_err(gw.cpu(), gw0.cpu()) <= TOL
    # fp64 LayerNorm backward from the same saved output
    yd, gd = yn.double(), gyd.double()
    gref = rstd.double()[:, None] * (gd - gd.mean(dim=1, keepdim=True)
                                      - yd * (gd * yd).mean(dim=1, keepdim=True))
    assert rel_err(g.cpu(), gref.cpu()) <= 2e-6
    # wrong tile count / a method that has no fold: refused
    ws = torch.empty(16, device="cuda", dtype=torch.uint8)
    rc = lib.pp_backward_weight_layer_norm_f32(
        PF._ptr(xd), PF._ptr(gyd), PF._ptr(yn), PF._ptr(rstd), PF._ptr(stats), tiles + 1,
        PF._ptr(g), PF._ptr(gw), PF._ptr(ws), 16, B, I, O, n, S, int(disc), 2.0,
        spec.node_array(), 1.0, 0, PF._stream(raw.device))
    assert rc == PP_ERR_SHAPE
    rc = lib.pp_backward_weight_layer_norm_f32(
        PF._ptr(xd), PF._ptr(gyd), PF._ptr(yn), PF._ptr(rstd), PF._ptr(stats), tiles,
        PF._ptr(g), PF._ptr(gw), PF._ptr(ws), 16, B, I, O, n, S, int(disc), 2.0,
        spec.node_array(), 1.0, _lib.BWD_W_METHODS["dense"], PF._stream(raw.device))
    assert rc == PP_ERR_METHOD
    rc = lib.pp_backward_weight_layer_norm_f32(
        PF._ptr(xd), PF._ptr(gyd), PF._ptr(yn), PF._ptr(rstd), PF._ptr(stats), tiles,
        PF._ptr(g), PF._ptr(gw), PF._ptr(ws), 16, B, I, O, n, S, int(disc), 2.0,
        spec.node_array(), 1.0, 0, PF._stream(raw.device))
    assert rc == PP_ERR_WORKSPACE


@pytest.mark.parametrize("rows", [16384, 4100])
def test_high_order_mlp_folds_the_layer_norm_backward(rows, monkeypatch):
    """The bench network (cfg2): the three hidden layers take their LayerNorm backward in the
    weight-gradient pass (counted), gradients agree with the same network run with the fold
    off, and a second run gives the same bits."""
    import language_interpolation_amd as lia
    PF = _pf()
    torch.manual_seed(8)
    net = lia.HighOrderMLP(layer_type="continuous", n=3, in_width=16, out_width=128,
                           hidden_layers=3, hidden_width=250, in_segments=128, out_segments=2,
                           hidden_segments=2, n_in=2, normalization=lia.LazyLayerNormLastDim,
                           device="cuda")
    from pp_testlib import char_grid
    x = char_grid(rows, 16, torch.Generator().manual_seed(5)).cuda()
    gy = (torch.randn(rows, 128) / 128).cuda()
    calls = []
    real = PF.backward_weight_layer_norm_raw

    def counted(*a, **k):
        calls.append(a[0].shape)
        return real(*a, **k)

    monkeypatch.setattr(PF, "backward_weight_layer_norm_raw", counted)
    assert PF.FOLD_LAYER_NORM_BACKWARD is False      # measured slower (functional.py): opt-in
    monkeypatch.setattr(PF, "FOLD_LAYER_NORM_BACKWARD", True)

    def run():
        for p in net.parameters():
            p.grad = None
        y = net(x)
        y.backward(gy)
        return y.detach().clone(), [p.grad.clone() for p in net.parameters()]

    y1, g1 = run()
    assert len(calls) == 3 and PF._row_stats_slot is None
    y1b, g1b = run()
    assert torch.equal(y1, y1b) and all(torch.equal(a, b) for a, b in zip(g1, g1b))
    monkeypatch.setattr(PF, "FOLD_LAYER_NORM_BACKWARD", False)
    del calls[:]
    y2, g2 = run()
    assert not calls
    assert torch.equal(y1, y2)
    for a, b in zip(g1, g2):
        assert rel_err(a.cpu(), b.cpu()) <= TOL


def test_fold_is_skipped_when_the_gradient_is_not_the_one_the_sums_belong_to(monkeypatch):
    """A gradient that reaches the LayerNorm output from two consumers is a new tensor: the
    slot does not match and the separate kernels run (same result as without the slot)."""
    import language_interpolation_amd as lia
    PF = _pf()
    torch.manual_seed(9)
    B = 4096
    l1 = lia.high_order_fc_layers(layer_type="continuous", n=3, in_features=250, out_features=250,
                                  segments=2, device="cuda")
    l2 = lia.high_order_fc_layers(layer_type="continuous", n=3, in_features=250, out_features=128,
                                  segments=2, device="cuda")
    x = torch.randn(B, 250, device="cuda")
    gy = torch.randn(B, 128, device="cuda") / 128
    monkeypatch.setattr(PF, "FOLD_LAYER_NORM_BACKWARD", True)

    def run(flag):
        for p in list(l1.parameters()) + list(l2.parameters()):
            p.grad = None
        h = l1.forward_layer_norm(x, 1e-5)
        PF._input_is_layer_norm = flag
        try:
            y = l2(h)
        finally:
            PF._input_is_layer_norm = False
        loss_side = (h * 0.25).sum()          # second consumer of h: autograd adds the two gradients
        torch.autograd.backward([y, loss_side], [gy, torch.ones_like(loss_side)])
        return [p.grad.clone() for p in l1.parameters()]

    assert l1.forward_layer_norm(x, 1e-5) is not None
    a = run(True)
    assert PF._row_stats_slot is None
    b = run(False)
    for u, v in zip(a, b):
        assert torch.equal(u, v)


def test_fold_entry_points_on_random_shapes():
    """Seeded sweep over ragged shapes (rows off the 32 / 512 boundaries, inputs and outs off the
    tile widths): every shape the library says it supports gives bit-identical gx, tile sums that
    add up, and g / gw_t within the bar of the separate kernels."""
    import random
    from language_interpolation_amd import _lib
    PF = _pf()
    lib = _lib.load()
    rng = random.Random(77)
    done_x = done_w = 0
    for _ in range(200):
        n = rng.choice([2, 3, 4, 5])
        disc = rng.random() < 0.3
        S = rng.choice([1, 2, 2, 4]) if not disc else rng.choice([1, 2])
        K = n * S if disc else (n - 1) * S + 1
        if K > 8:
            continue
        B = rng.choice([rng.randint(256, 1100), rng.randint(4096, 9000), 8192, 4097])
        I = rng.choice([32, 33, 64, 100, 250, rng.randint(32, 300)])
        O = rng.choice([32, 70, 128, 129, 250, rng.randint(32, 300)])
        x, w, gy = make_inputs(B, I, O, n, S, disc, rng.choice(["normal", "uniform"]),
                               seed=rng.randint(0, 1 << 30))
        spec = PF.make_spec(n, S, disc)
        xd, gyd = x.cuda(), gy.cuda()
        w_t = w.cuda().permute(1, 2, 0).contiguous()
        if lib.pp_backward_input_stats_supported(B, I, O, n, S, int(disc), 2.0) and done_x < 25:
            tiles = lib.pp_row_stats_tiles(I)
            stats = torch.full((tiles, B, 2), float("nan"), device="cuda")
            gx = PF.backward_input_raw(xd, gyd, w_t, spec, row_stats=stats)
            gx0 = PF.backward_input_raw(xd, gyd, w_t, spec)
            assert torch.equal(gx, gx0), (B, I, O, n, S, disc)
            assert torch.isfinite(stats).all(), (B, I, O, n, S, disc)
            s = stats.double().sum(dim=0)
            scale = gx.double().abs().sum(dim=1).clamp_min(1e-30)
            xmax = xd.abs().max().item() + 1.0
            assert ((s[:, 0] - gx.double().sum(dim=1)).abs() / scale).max().item() <= 1e-6
            assert ((s[:, 1] - (gx.double() * xd.double()).sum(dim=1)).abs() / scale).max().item() \
                <= 1e-6 * xmax
            done_x += 1
        if lib.pp_backward_weight_layer_norm_supported(B, I, O, n, S, int(disc), 2.0) and done_w < 25:
            raw, _ = PF.forward_raw(xd, w_t, spec)
            yn = torch.empty_like(raw)
            rstd = torch.empty(B, device="cuda")
            assert lib.pp_layer_norm_fwd_f32(PF._ptr(raw), PF._ptr(yn), PF._ptr(rstd), B, O, 1e-5,
                                             PF._stream(raw.device)) == 0
            tiles = lib.pp_row_stats_tiles(O)
            pad = tiles * 32 - O
            s1 = torch.nn.functional.pad(gyd, (0, pad)).view(B, tiles, 32).sum(dim=2)
            s2 = torch.nn.functional.pad(gyd * yn, (0, pad)).view(B, tiles, 32).sum(dim=2)
            stats = torch.stack((s1.t(), s2.t()), dim=2).contiguous()
            g, gw = PF.backward_weight_layer_norm_raw(xd, gyd, yn, rstd, stats, tuple(w_t.shape), spec)
            g0 = torch.empty_like(raw)
            assert lib.pp_layer_norm_bwd_f32(PF._ptr(yn), PF._ptr(gyd), PF._ptr(rstd), PF._ptr(g0),
                                             B, O, PF._stream(raw.device)) == 0
            gw0 = PF.backward_weight_raw(xd, g0, tuple(w_t.shape), spec)
            assert rel_err(g.cpu(), g0.cpu()) <= 4e-6, (B, I, O, n, S, disc)
            assert rel_err(gw.cpu(), gw0.cpu()) <= TOL, (B, I, O, n, S, disc)
            done_w += 1
    assert done_x >= 15 and done_w >= 8, (done_x, done_w)
