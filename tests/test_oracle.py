"""Known-answer and property tests that pin oracle/pp_oracle.py.

The reference's own tests pin shapes only (tests/test_utils.py:25,
tests/test_dataset_from_representation.py:69-70 of /root/reference), and the
layer's source (high-order-layers-torch) is absent, so these spec-derived
checks (KA1-KA4, P1-P8) are what the oracle stands on.
"""
import math

import numpy as np
import pytest
import torch

from oracle import pp_oracle as po


def fwd(x, w, n, S, disc, prec="f32", **kw):
    x = torch.as_tensor(x, dtype=torch.float32).reshape(-1, 1) if np.ndim(x) <= 1 else torch.as_tensor(x, dtype=torch.float32)
    w = torch.as_tensor(w, dtype=torch.float32)
    if w.dim() == 1:
        w = w.reshape(1, 1, -1)
    if prec == "f32":
        return po.piecewise_forward(x, w, n, S, disc, **kw)
    if prec == "dense":
        return po.piecewise_forward_dense(x, w, n, S, disc, **kw)
    return po.piecewise_forward_f64(x, w, n, S, disc, **kw)


PRECS = ["f32", "dense", "f64"]


@pytest.mark.parametrize("prec", PRECS)
def test_ka1_continuous_linear_two_segments(prec):
    a, b, c = 0.3, -1.1, 2.5
    xs = [-1.0, -0.5, 0.0, 0.5, 1.0, 1.5, -1.5]
    want = [a, (a + b) / 2, b, (b + c) / 2, c, -0.5 * b + 1.5 * c, 1.5 * a - 0.5 * b]
    got = fwd(xs, [a, b, c], 2, 2, False, prec).flatten().double()
    assert torch.allclose(got, torch.tensor(want, dtype=torch.float64), atol=1e-6)


@pytest.mark.parametrize("prec", PRECS)
def test_ka2_discontinuous_jump(prec):
    a, b, c, d = 1.0, 2.0, -3.0, 4.0
    below0 = -1e-6  # (x+1) must round below 1.0 in fp32 for the id to drop to segment 0
    got = fwd([0.0, below0, -1.0, 1.0], [a, b, c, d], 2, 2, True, prec).flatten().double()
    assert torch.allclose(got, torch.tensor([c, b, a, d], dtype=torch.float64), atol=1e-5)


@pytest.mark.parametrize("prec", PRECS)
def test_ka3_quadratic(prec):
    w = [0.7, -0.2, 1.3, 0.4, -2.0]
    got = fwd([0.25], w, 3, 2, False, prec).item()
    want = 0.375 * w[2] + 0.75 * w[3] - 0.125 * w[4]
    assert abs(got - want) < 1e-6


@pytest.mark.parametrize("prec", PRECS)
def test_ka4_piecewise_constant(prec):
    torch.manual_seed(0)
    B, I, O, S = 9, 4, 3, 7
    x = torch.rand(B, I) * 3 - 1.5
    w = torch.randn(O, I, S)
    s = po.segment_ids(x, S)
    want = torch.stack([sum(w[:, i, s[b, i]] for i in range(I)) for b in range(B)])
    got = fwd(x, w, 1, S, True, prec)
    assert torch.allclose(got.double(), want.double(), atol=1e-5)
    r = po.piecewise_fwd_bwd(x, w, torch.randn(B, O), 1, S, True, precision="f64")
    assert torch.count_nonzero(r["gx"]) == 0


@pytest.mark.parametrize("disc", [False, True])
@pytest.mark.parametrize("n,S", [(1, 5), (2, 2), (2, 128), (3, 2), (4, 3), (5, 8), (8, 2)])
def test_p1_partition_of_unity(n, S, disc):
    if n == 1 and not disc:
        pytest.skip("continuous n=1 has K=1")
    torch.manual_seed(1)
    I, O = 6, 4
    K = po.num_weights(n, S, disc)
    x = torch.randn(33, I) * 1.5
    y = po.piecewise_forward_f64(x, torch.ones(O, I, K), n, S, disc)
    assert torch.allclose(y, torch.full_like(y, float(I)), atol=1e-9 * 10 ** n)


@pytest.mark.parametrize("n,S", [(2, 4), (3, 2), (5, 3)])
def test_p2_interpolation_at_global_nodes(n, S):
    torch.manual_seed(2)
    I, O = 3, 2
    K = po.num_weights(n, S, False)
    w = torch.randn(O, I, K)
    nodes = po.chebyshev_lobatto(n).double()
    # global position of weight k = s*(n-1)+j
    for k in range(K):
        s, j = divmod(k, n - 1)
        if s == S:
            s, j = S - 1, n - 1
        xg = -1.0 + (2.0 / S) * (s + (nodes[j].item() + 1.0) / 2.0)
        x = torch.full((1, I), xg, dtype=torch.float32)
        y = po.piecewise_forward_f64(x, w, n, S, False)
        assert torch.allclose(y[0], w[:, :, k].sum(1).double(), atol=2e-5), (k, xg)


@pytest.mark.parametrize("disc", [False, True])
@pytest.mark.parametrize("n,S", [(2, 2), (3, 2), (4, 5), (5, 2)])
def test_p3_linear_reproduction(n, S, disc):
    torch.manual_seed(3)
    I, O = 5, 3
    alpha, beta = 0.7, -0.3
    nodes = po.chebyshev_lobatto(n).double()
    K = po.num_weights(n, S, disc)
    stride = po.weight_stride(n, disc)
    xk = torch.zeros(K, dtype=torch.float64)
    for s in range(S):
        for j in range(n):
            xk[s * stride + j] = -1.0 + (2.0 / S) * (s + (nodes[j] + 1.0) / 2.0)
    w = (alpha * xk + beta).float().expand(O, I, K).contiguous()
    # includes mild extrapolation beyond [-1,1]; far extrapolation is ill-conditioned in
    # the fp32 local coordinate the reference itself computes (xi error ~S*1e-7 * |l_j'|)
    x = (torch.rand(41, I) * 2 - 1) * 1.3
    y = po.piecewise_forward_f64(x, w, n, S, disc)
    want = (alpha * x.double() + beta).sum(1, keepdim=True).expand(-1, O)
    assert torch.allclose(y, want, atol=2e-5)


@pytest.mark.parametrize("n,S", [(2, 2), (3, 4), (4, 150)])
def test_p4_continuity_at_joins(n, S):
    torch.manual_seed(4)
    K = po.num_weights(n, S, False)
    w = torch.randn(1, 1, K)
    for m in range(1, S):
        xb = np.float32(-1.0 + 2.0 * m / S)
        lo = np.nextafter(xb, np.float32(-2.0))
        hi = np.nextafter(xb, np.float32(2.0))
        x = torch.tensor([[lo], [xb], [hi]], dtype=torch.float32)
        y = po.piecewise_forward_f64(x, w, n, S, False).flatten()
        scale = w.abs().max().item() * S * n * n
        assert (y.max() - y.min()).item() < 4e-7 * scale


def test_p5_character_grid():
    c = torch.arange(128)
    x = po.ascii_to_float(c).reshape(-1, 1)
    assert torch.equal(po.segment_ids(x, 128).flatten(), c)
    xi = po.local_coordinate(x, po.segment_ids(x, 128), 128)
    assert torch.count_nonzero(xi) == 0


@pytest.mark.parametrize("disc", [False, True])
@pytest.mark.parametrize("n,S", [(2, 2), (3, 2), (3, 5), (5, 3)])
def test_p6_gradcheck_and_autograd_vs_closed_form(n, S, disc):
    torch.manual_seed(5)
    B, I, O = 7, 4, 3
    K = po.num_weights(n, S, disc)
    # stay away from segment boundaries so finite differences are valid
    centres = -1.0 + (2.0 / S) * (torch.randint(0, S, (B, I)).float() + 0.5)
    x = (centres + (torch.rand(B, I) - 0.5) * (1.2 / S)).float()
    w = torch.randn(O, I, K, dtype=torch.float64)
    gy = torch.randn(B, O, dtype=torch.float64)
    r = po.piecewise_fwd_bwd(x, w, gy, n, S, disc, precision="f64")
    # central finite differences on the fp64 function of x (w.r.t. x cast up)
    eps = 1e-3 / S  # representable step in fp32 is coarse; use a polynomial-exact-ish step
    gx_fd = torch.zeros(B, I, dtype=torch.float64)
    for b in range(B):
        for i in range(I):
            xp, xm = x.clone(), x.clone()
            xp[b, i] += eps
            xm[b, i] -= eps
            yp = po.piecewise_forward_f64(xp, w, n, S, disc)
            ym = po.piecewise_forward_f64(xm, w, n, S, disc)
            h = (xp[b, i].double() - xm[b, i].double())
            gx_fd[b, i] = ((yp - ym) * gy).sum() / h
    tol = 2e-3 * (r["gx"].abs().max().item() + 1.0) * (1.0 if n <= 3 else 20.0)
    assert (r["gx"].double() - gx_fd).abs().max().item() < tol
    # gw closed form: gw[o,i,k] = sum_b gy[b,o] * Phi[b,i,k]
    s = po.segment_ids(x, S)
    xi = po.local_coordinate(x.double(), s, S)
    basis = po.lagrange_basis(xi, po.chebyshev_lobatto(n))
    gw = torch.zeros(O, I, K, dtype=torch.float64)
    stride = po.weight_stride(n, disc)
    for b in range(B):
        for i in range(I):
            for j in range(n):
                gw[:, i, s[b, i] * stride + j] += gy[b] * basis[j, b, i]
    assert torch.allclose(r["gw"], gw, atol=1e-5)


def test_p7_shared_node_receives_both_neighbours():
    n, S = 3, 2
    w = torch.zeros(1, 1, 5)
    x = torch.tensor([[-0.1], [0.1]])
    gy = torch.ones(2, 1)
    r = po.piecewise_fwd_bwd(x, w, gy, n, S, False, precision="f64")
    nodes = po.chebyshev_lobatto(3)
    l_right_of_seg0 = po.lagrange_basis(torch.tensor([0.8], dtype=torch.float64), nodes)[2]
    l_left_of_seg1 = po.lagrange_basis(torch.tensor([-0.8], dtype=torch.float64), nodes)[0]
    assert abs(r["gw"][0, 0, 2].item() - (l_right_of_seg0 + l_left_of_seg1).item()) < 1e-6


@pytest.mark.parametrize("S", [2, 8, 128, 150])
def test_p8_segment_ids_against_numpy_fp32(S):
    """Independent restatement of A1 with explicit numpy fp32 ops."""
    xs = []
    for m in range(S + 1):
        xb = np.float32(-1.0 + 2.0 * m / S)
        v = xb
        for _ in range(4):
            v = np.nextafter(v, np.float32(-3.0))
        for _ in range(9):
            xs.append(v)
            v = np.nextafter(v, np.float32(3.0))
    xs += [-3.0, 3.0, -1e9, 1e9, 0.0]
    xs = np.asarray(xs, dtype=np.float32)
    t = ((xs + np.float32(1.0)) / np.float32(2.0)) * np.float32(S)
    want = np.clip(np.trunc(t).astype(np.int64), 0, S - 1)
    got = po.segment_ids(torch.from_numpy(xs).reshape(-1, 1), S).flatten().numpy()
    assert np.array_equal(got, want)


@pytest.mark.parametrize("disc", [False, True])
@pytest.mark.parametrize("n,S", [(2, 128), (3, 2), (1, 150), (2, 150), (5, 2)])
def test_three_formulations_agree(n, S, disc):
    if n == 1 and not disc:
        pytest.skip("continuous n=1 has K=1")
    torch.manual_seed(6)
    B, I, O = 19, 11, 7
    K = po.num_weights(n, S, disc)
    x = torch.randn(B, I)
    w = (torch.rand(O, I, K) * 2 - 1) / I
    gy = torch.randn(B, O) / O
    r64 = po.piecewise_fwd_bwd(x, w, gy, n, S, disc, precision="f64")
    for prec in ("f32", "f32dense"):
        r = po.piecewise_fwd_bwd(x, w, gy, n, S, disc, precision=prec)
        assert torch.equal(r["seg"], r64["seg"])
        for key in ("y", "gx", "gw"):
            ref = r64[key]
            err = (r[key].double() - ref).abs().max().item()
            assert err <= 1e-5 * max(ref.abs().max().item(), 1e-30), (prec, key, err)


def test_rescale_and_shapes_like_reference_tests():
    # /root/reference/tests/test_utils.py:8-25
    net = po.OracleHighOrderMLP(
        layer_type="continuous", n=5, in_width=10, out_width=11, hidden_layers=1,
        hidden_width=5, in_segments=2, out_segments=2, hidden_segments=2, device="cpu",
    )
    ans = po.reshape_apply(torch.rand(7, 5, 10), net)
    assert ans.shape == torch.Size([7, 5, 11])
    # /root/reference/tests/test_dataset_from_representation.py:67-70: hidden.layers=2
    # puts a width-10 polynomial layer output at Sequential index 5 ... index 4 is the layer,
    # 5 the norm after it; both have the hidden width.
    net = po.OracleHighOrderMLP(
        layer_type="discontinuous", n=2, n_in=2, n_hidden=2, in_width=16, out_width=128,
        hidden_layers=2, hidden_width=10, in_segments=128, out_segments=8, hidden_segments=8,
        normalization=po.OracleLazyLayerNormLastDim, layer_type_in="continuous",
    )
    kinds = [type(m).__name__ for m in net.model]
    assert kinds == ["OraclePiecewise", "OracleLazyLayerNormLastDim"] * 3 + ["OraclePiecewise"]
    h = net.model[:6](torch.rand(82, 16) * 2 - 1)
    assert h.shape == (82, 10)
    # rescale_output divides by in_features
    x = torch.rand(4, 6) * 2 - 1
    w = torch.randn(3, 6, 3)
    a = po.piecewise_forward(x, w, 2, 2, False)
    b = po.piecewise_forward(x, w, 2, 2, False, rescale_output=True)
    assert torch.allclose(a / 6, b, atol=1e-6)


def test_normalizations():
    torch.manual_seed(7)
    x = torch.randn(5, 9)
    y = po.max_abs_normalization_last(x)
    assert torch.allclose(y.abs().max(dim=-1)[0], torch.ones(5), atol=1e-5)
    z = po.max_center_normalization_last(x)
    assert torch.allclose(z.max(dim=-1)[0], torch.ones(5), atol=1e-5)
    assert torch.allclose(z.min(dim=-1)[0], -torch.ones(5), atol=1e-5)
    ln = po.layer_norm_last_dim(x)
    want = (x - x.mean(-1, keepdim=True)) / torch.sqrt(x.var(-1, unbiased=False, keepdim=True) + 1e-5)
    assert torch.allclose(ln, want, atol=1e-6)
