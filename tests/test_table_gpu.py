"""Table form of a layer whose inputs lie on the character grid (csrc/pp_table.hip; reference:
x = (c - 63.5)/64 from single_text_dataset.py:15-16 feeding the input layer, networks.py:793-823 /
:866-882).  On grid inputs the table path must give the general path's y and gw_t (fp32 summation
order aside) and the fp64 oracle's to the usual bar."""
import dataclasses

import pytest
import torch

from pp_testlib import TOL, char_grid, oracle_all, rel_err

pytestmark = pytest.mark.gpu


def _pf():
    import language_interpolation_amd.functional as PF
    return PF


def _inputs(B, I, O, n, S, disc, seed):
    from oracle import pp_oracle as po
    gen = torch.Generator().manual_seed(seed)
    K = po.num_weights(n, S, disc)
    x = char_grid(B, I, gen)
    w = (torch.rand(O, I, K, generator=gen) * 2 - 1) / I
    gy = torch.randn(B, O, generator=gen) / O
    return x, w, gy


# (disc, n, S, I, O, B)
TABLE_CASES = [
    (False, 2, 128, 16, 250, 16384),   # cfg2 input layer at the bench batch
    (True, 2, 128, 16, 1000, 4096),    # cfg4 input layer (K = 256: two parts without the table)
    (False, 2, 128, 16, 200, 2048),    # cfg1 input layer at the smallest batch that takes the table
    (False, 3, 2, 10, 64, 2777),       # few segments: 64 grid points a segment
    (True, 1, 150, 16, 72, 2049),      # n = 1, S not a divisor of the grid
    (False, 5, 7, 3, 4, 2100),         # high order, tiny
]


@pytest.mark.parametrize("case", TABLE_CASES, ids=lambda c: "-".join(str(v) for v in c))
def test_table_form_matches_general_path_and_oracle(case):
    PF = _pf()
    disc, n, S, I, O, B = case
    x, w, gy = _inputs(B, I, O, n, S, disc, seed=51)
    assert PF.on_input_grid(x) and not PF.on_input_grid(x + 1e-3)
    spec = PF.make_spec(n, S, disc)
    tspec = dataclasses.replace(spec, input_grid=128)
    assert PF.use_table_form(tspec, B) and not PF.use_table_form(tspec, 2047)
    wd0 = w.cuda().permute(1, 2, 0).contiguous().permute(2, 0, 1)
    def run(fn):
        xd = x.cuda().requires_grad_(True)
        wd = wd0.detach().clone().requires_grad_(True)
        y = fn(xd, wd)
        y.backward(gy.cuda())
        return y.detach().cpu(), xd.grad.cpu(), wd.grad.cpu()

    ty, tgx, tgw = run(lambda a, b: PF.piecewise_poly(a, b, tspec))
    gy_, ggx, ggw = run(lambda a, b: PF.piecewise_poly(a, b, spec))
    tln = gln = None
    if PF.layer_norm_fusable(x.cuda(), wd0, tspec) and PF.layer_norm_fusable(x.cuda(), wd0, spec):
        tln = run(lambda a, b: PF.piecewise_poly_layer_norm(a, b, tspec, 1e-5))
        gln = run(lambda a, b: PF.piecewise_poly_layer_norm(a, b, spec, 1e-5))
    ref = oracle_all(x, w, gy, n, S, disc)
    assert rel_err(ty, ref["y"]) <= TOL and rel_err(gy_, ref["y"]) <= TOL
    assert rel_err(tgw, ref["gw"]) <= TOL
    assert rel_err(tgx, ref["gx"]) <= TOL          # backward-input stays on the general kernels
    assert rel_err(ty, gy_) <= 2e-6 and rel_err(tgw, ggw) <= 2e-6
    if tln is not None:
        assert rel_err(tln[0], gln[0]) <= 2e-5
        assert rel_err(tln[2], gln[2]) <= 5 * TOL
    else:
        assert O < 16 or n > 4
    # same bits on a second run
    xd = x.cuda()
    wd = wd0.detach().clone().requires_grad_(True)
    y2 = PF.piecewise_poly(xd, wd, tspec)
    y2.backward(gy.cuda())
    assert torch.equal(y2.detach().cpu(), ty)
    from language_interpolation_amd import _lib
    if _lib.load().pp_auto_method(2, B, I, O, 1, 128, 1, 2.0) != _lib.BWD_W_ATOMIC:
        assert torch.equal(wd.grad.cpu(), tgw)   # (tiny layers: atomics, arrival order)


def test_grid_table_entry_points_against_torch():
    """pp_grid_table_f32 against y = sum_i T[i][c(b,i)] (torch gather) vs the oracle forward;
    pp_grid_table_grad_f32 as the exact transpose of the table map (inner products agree)."""
    PF = _pf()
    disc, n, S, I, O, B = False, 3, 16, 6, 40, 300
    x, w, gy = _inputs(B, I, O, n, S, disc, seed=52)
    spec = dataclasses.replace(PF.make_spec(n, S, disc, rescale_output=True, in_features=I),
                               input_grid=128)
    w_t = w.cuda().permute(1, 2, 0).contiguous()
    T = PF.grid_table(w_t, spec)
    c = ((x.double() + 1.0) * 64.0 - 0.5).round().long().cuda()           # [B, I]
    y = T[torch.arange(I, device="cuda")[None, :], c].sum(dim=1)            # [B, O]
    from oracle import pp_oracle as po
    ref = po.piecewise_forward_f64(x, w.double(), n, S, disc) / I
    assert rel_err(y.cpu(), ref) <= TOL
    # <T(w), G> == <w, T^t(G)> for a random G: the gradient map is the transpose of the table map
    G = torch.randn(I, 128, O, device="cuda")
    gw_t = PF.grid_table_grad(G, tuple(w_t.shape), spec)
    lhs = (T.double() * G.double()).sum().item()
    rhs = (w_t.double() * gw_t.double()).sum().item()
    assert abs(lhs - rhs) <= 1e-6 * max(abs(lhs), 1.0)
    # weight rows no grid point touches come back as zeros, not garbage
    spec2 = dataclasses.replace(PF.make_spec(2, 300, True), input_grid=128)
    gw2 = PF.grid_table_grad(torch.randn(2, 128, 8, device="cuda"), (2, 600, 8), spec2)
    assert torch.isfinite(gw2).all() and (gw2.abs().sum(dim=2) == 0).any()


def test_high_order_mlp_with_the_input_layer_in_table_form():
    """The bench network with its first layer told about the character grid: same outputs and
    gradients as without (tolerance), and the statement survives set_methods (bench.py replays
    this step from a HIP graph: `char_table.launch`)."""
    import language_interpolation_amd as lia
    PF = _pf()
    torch.manual_seed(10)
    net = lia.HighOrderMLP(layer_type="continuous", n=3, in_width=16, out_width=128,
                           hidden_layers=2, hidden_width=250, in_segments=128, out_segments=2,
                           hidden_segments=2, n_in=2, normalization=lia.LazyLayerNormLastDim,
                           device="cuda")
    B = 8192
    x = char_grid(B, 16, torch.Generator().manual_seed(6)).cuda()
    gy = (torch.randn(B, 128) / 128).cuda()

    def run():
        for p in net.parameters():
            p.grad = None
        y = net(x)
        y.backward(gy)
        return y.detach().clone(), [p.grad.clone() for p in net.parameters()]

    y0, g0 = run()
    first = net.model[0]
    assert first.assume_input_grid(128) is first and first._spec.input_grid == 128
    first.set_methods()
    assert first._spec.input_grid == 128
    y1, g1 = run()
    assert rel_err(y1.cpu(), y0.cpu()) <= 2e-5
    for a, b in zip(g1, g0):
        assert rel_err(a.cpu(), b.cpu()) <= 5 * TOL
    # under 16 rows per grid point the statement changes nothing (the table would cost more
    # than it saves): same bits as the general path
    xs, gs = x[:1000].contiguous(), gy[:1000].contiguous()
    ya = net(xs)
    first.assume_input_grid(0)
    assert torch.equal(net(xs), ya)
    y2, _ = run()
    assert torch.equal(y2, y0)


def test_assume_input_grid_refusals():
    import language_interpolation_amd as lia
    lay = lia.high_order_fc_layers(layer_type="continuous", n=2, in_features=4, out_features=8,
                                   segments=16, device="cuda")
    with pytest.raises(ValueError):
        lay.assume_input_grid(100)           # not a power of two
    with pytest.raises(ValueError):
        lay.assume_input_grid(256)           # more than one byte of table rows
    lay3 = lia.high_order_fc_layers(layer_type="continuous", n=2, in_features=4, out_features=8,
                                    segments=16, scale=3.0, device="cuda")
    with pytest.raises(ValueError):
        lay3.assume_input_grid(128)          # the local coordinate is tied to [-1, 1]
    per = lia.high_order_fc_layers(layer_type="continuous", n=2, in_features=4, out_features=8,
                                   segments=16, periodicity=2.0, device="cuda")
    with pytest.raises(ValueError):
        per.assume_input_grid(128)
