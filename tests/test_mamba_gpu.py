"""Selective scan kernels and the Mamba mirror against the CPU oracle (oracle/mamba_oracle.py,
fp64).  Parity unpinned: the reference holds no numbers for state_space_network.py and the
file cannot be imported here (see the oracle's header)."""
import pytest
import torch

pytestmark = pytest.mark.gpu

TOL = 1e-5


def rel(a, b):
    return ((a.double().cpu() - b.double()).abs().max() / b.double().abs().max().clamp_min(1e-30)).item()


def scan_inputs(b, l, d, n, seed):
    g = torch.Generator().manual_seed(seed)
    u = torch.randn(b, l, d, generator=g)
    delta = torch.nn.functional.softplus(torch.randn(b, l, d, generator=g))
    A = -torch.exp(torch.log(torch.arange(1, n + 1).float()).repeat(d, 1)
                   + 0.1 * torch.randn(d, n, generator=g))
    Bm = torch.randn(b, l, n, generator=g)
    Cm = torch.randn(b, l, n, generator=g)
    D = torch.randn(d, generator=g)
    gy = torch.randn(b, l, d, generator=g)
    return [u, delta, A, Bm, Cm, D], gy


# (b, l, d, n): cfg5 (d_inner 128, d_state 32, l 15), ragged channel / state counts, one step
# state-split kernels up to n = 64 (tiles of 8 time steps: l = 16, 17, 33 cross tile edges),
# one-wave kernels above
SCAN_CASES = [(16, 15, 128, 32), (3, 7, 70, 16), (2, 5, 130, 19), (1, 1, 1, 1), (5, 33, 64, 40),
              (4, 16, 64, 32), (2, 17, 64, 8), (3, 12, 96, 33), (2, 9, 64, 70)]


@pytest.mark.parametrize("case", SCAN_CASES, ids=lambda c: "-".join(map(str, c)))
def test_selective_scan_forward_and_backward(case):
    from language_interpolation_amd import functional as PF
    from oracle import mamba_oracle as mo
    b, l, d, n = case
    ins, gy = scan_inputs(b, l, d, n, seed=b * 100 + n)
    ref_in = [t.double().requires_grad_() for t in ins]
    ref_y = mo.selective_scan(*ref_in)
    ref_g = torch.autograd.grad(ref_y, ref_in, gy.double())
    dev_in = [t.cuda().requires_grad_() for t in ins]
    y = PF.selective_scan(*dev_in)
    assert rel(y, ref_y.detach()) <= TOL
    grads = torch.autograd.grad(y, dev_in, gy.cuda())
    for name, g, r in zip(("gu", "gdelta", "gA", "gB", "gC", "gD"), grads, ref_g):
        assert rel(g, r) <= TOL, name
    # inference path (no history buffer) gives the same y
    with torch.no_grad():
        y2 = PF.selective_scan(*[t.detach() for t in dev_in])
    assert torch.equal(y2, y.detach())


def test_selective_scan_rejects_cpu_and_bad_shapes():
    from language_interpolation_amd import functional as PF
    ins, _ = scan_inputs(2, 3, 4, 5, seed=0)
    with pytest.raises(RuntimeError):
        PF.selective_scan(*ins)
    dev = [t.cuda() for t in ins]
    dev[3] = dev[3][:, :, :4].contiguous()
    with pytest.raises(ValueError):
        PF.selective_scan(*dev)


def _copy_params(ora, net):
    """Same values in both models: names differ only for the norms (oracle keeps bare weights)."""
    src = dict(ora.named_parameters())
    with torch.no_grad():
        for name, p in net.named_parameters():
            key = (name.replace(".norm.weight", ".norm_weight")
                       .replace("norm_f.weight", "norm_f_weight"))
            p.copy_(src[key].to(p.dtype))


@pytest.mark.parametrize("layer_type", ["continuous", "linear"])
def test_mamba_matches_oracle_forward_and_gradients(layer_type):
    """cfg5 block sizes (d_model 64 -> d_inner 128, dt_rank 4, d_state 32, l 15), 2 layers."""
    import language_interpolation_amd as L
    from oracle import mamba_oracle as mo
    torch.manual_seed(0)
    kw = dict(d_model=64, vocab_size=128, n_layer=2, d_state=32, d_conv=4, conv_bias=True,
              bias=False, expand=2, dt_rank="auto", pad_vocab_size_multiple=128,
              layer_type=layer_type, n=3, segments=2, hidden_layers=0)
    ora = mo.OracleMamba(**kw).double()
    net = L.Mamba(**kw).cuda()
    _copy_params(ora, net)
    assert {n for n, _ in net.named_parameters()} >= {
        "embedding.weight", "layers.0.mixer.A_log", "layers.0.mixer.D", "layers.0.norm.weight",
        "layers.1.mixer.conv1d.weight", "norm_f.weight"}
    ids = torch.randint(0, 128, (6, 15, 1))
    tgt = torch.randint(0, 128, (6, 15))
    ref = ora(ids)
    out = net(ids.cuda())
    assert out.shape == (6, 15, 128)
    assert rel(out, ref.detach()) <= 2e-5
    lossf = torch.nn.CrossEntropyLoss()
    lossf(ref.reshape(90, -1), tgt.flatten()).backward()
    lossf(out.reshape(90, -1), tgt.cuda().flatten()).backward()
    src = dict(ora.named_parameters())
    for name, p in net.named_parameters():
        key = (name.replace(".norm.weight", ".norm_weight").replace("norm_f.weight", "norm_f_weight"))
        r = src[key].grad
        assert p.grad is not None, name
        assert (p.grad.double().cpu() - r).abs().max().item() <= 2e-4 * r.abs().max().item() + 1e-9, name


@pytest.mark.parametrize("b,l,d,k,bias", [(64, 15, 128, 4, True), (3, 7, 100, 2, False),
                                          (1030, 5, 64, 4, True), (2, 1, 70, 3, True)])
def test_block_pieces_match_the_torch_formulation(b, l, d, k, bias):
    """Conv + SiLU, the gate and RMSNorm kernels (csrc/pp_mamba.hip) against the torch
    expressions of state_space_network.py:315-326 / :423-433 in fp64, values and every gradient;
    x and res are the two halves of one tensor, as in the block."""
    import torch.nn.functional as F
    from language_interpolation_amd import functional as PF
    from language_interpolation_amd.mamba import causal_depthwise_conv
    g = torch.Generator().manual_seed(b + l + d)
    both = torch.randn(b, l, 2 * d, generator=g).cuda().requires_grad_(True)
    w = (torch.randn(d, 1, k, generator=g) * 0.5).cuda().requires_grad_(True)
    bs = torch.randn(d, generator=g).cuda().requires_grad_(True) if bias else None
    nw = (torch.rand(d, generator=g) + 0.5).cuda().requires_grad_(True)
    gy = torch.randn(b, l, d, generator=g).cuda()

    def run(fused, leaves):
        bt, wt, bst, nwt = leaves
        x, res = bt.split([d, d], dim=-1)
        if fused:
            c = PF.causal_conv_silu(x, wt, bst)
            y = PF.silu_gate(c, res)
            return PF.rms_norm(y, nwt, 1e-5)
        c = F.silu(causal_depthwise_conv(x, wt, bst))
        y = c * F.silu(res)
        return y * torch.rsqrt(y.pow(2).mean(-1, keepdim=True) + 1e-5) * nwt

    out = run(True, (both, w, bs, nw))
    out.backward(gy)
    ref_leaves = [t.detach().double().requires_grad_(True) if t is not None else None
                  for t in (both, w, bs, nw)]
    ref = run(False, ref_leaves)
    ref.backward(gy.double())

    def close(a, r, what):
        err = (a.double() - r).abs().max().item() / max(r.abs().max().item(), 1e-30)
        assert err <= 2e-5, (what, err)

    close(out, ref, "y")
    close(both.grad, ref_leaves[0].grad, "gx/gres")
    close(w.grad, ref_leaves[1].grad, "gw")
    if bias:
        close(bs.grad, ref_leaves[2].grad, "gbias")
    close(nw.grad, ref_leaves[3].grad, "gnorm")
    # fixed summation order: the same bits on a second run
    both.grad = w.grad = nw.grad = None
    out2 = run(True, (both, w, bs, nw))
    out2.backward(gy)
    assert torch.equal(out2, out)


def test_seeded_slice_of_the_randomised_scan_run():
    """30 cases of tests/fuzz_scan.py's distribution (batch, length, channels, states around the
    scan kernels' variant boundaries) against the fp64 oracle."""
    import random

    import fuzz_scan
    rng = random.Random(20261111)
    failures = []
    for _ in range(30):
        case = fuzz_scan.draw_case(rng)
        bad, _ill_conditioned = fuzz_scan.run_case(case)
        if bad:
            failures.append((case, bad))
    assert not failures, failures
