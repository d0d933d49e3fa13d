"""Training harness on the GPU: a few optimizer steps of the HIP network against the same
steps of the CPU oracle network (same init, same batches), and learning on a small text."""
import os

import pytest
import torch

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CFG = os.path.join(ROOT, "configs", "high_order_interpolation.yaml")


def _oracle_twin(module, cfg):
    from oracle import pp_oracle as po
    net = cfg.net
    ora = po.OracleHighOrderMLP(
        layer_type=net.layer_type, n=net.n, n_in=net.n_in, n_hidden=net.n_in,
        n_out=net.get("n_out"), in_width=net.input.width, in_segments=net.input.segments,
        out_width=net.output.width, out_segments=net.output.segments,
        hidden_width=net.hidden.width, hidden_layers=net.hidden.layers,
        hidden_segments=net.hidden.segments, normalization=po.OracleLazyLayerNormLastDim,
        layer_type_in=net.input.get("layer_type")).double()
    with torch.no_grad():
        for a, b in zip(ora.parameters(), module.model.parameters()):
            a.copy_(b.detach().cpu().double())
    return ora


@pytest.mark.parametrize("overrides", [
    ["net=small", "optimizer=adam", "optimizer.lr=1e-3"],
    ["net=baseline_small_250", "net.hidden.width=40", "net.hidden.layers=1", "optimizer=adam",
     "optimizer.lr=1e-3"],
])
def test_training_steps_match_oracle(overrides):
    from language_interpolation_amd import train as T
    torch.manual_seed(0)
    cfg = T.load_config(CFG, overrides + ["batch_size=96"])
    module = T.ASCIIPredictionNet(cfg, device="cuda")
    ora = _oracle_twin(module, cfg)
    text = T.synthetic_text(4000, seed=3)
    data = T.CharWindows(text, cfg.net.features, device="cuda")
    opt = module.configure_optimizers()
    opt_o = torch.optim.Adam(ora.parameters(), lr=cfg.optimizer.lr)
    ce = torch.nn.CrossEntropyLoss()
    gen = torch.Generator().manual_seed(5)
    for step in range(4):
        x, y, _ = data.sample(cfg.batch_size, gen)
        opt.zero_grad()
        loss, acc = module.eval_step((x, y))
        loss.backward()
        opt.step()
        opt_o.zero_grad()
        lo = ce(ora(x.cpu().double()), y.cpu().flatten())
        lo.backward()
        opt_o.step()
        assert abs(loss.item() - lo.item()) <= 2e-4 * max(1.0, abs(lo.item())), (step, loss.item(), lo.item())
        assert 0.0 <= acc.item() <= 1.0
    # after four Adam steps the weights still agree (Adam's sign-like first steps amplify
    # gradient noise, hence the looser bar)
    for a, b in zip(ora.parameters(), module.model.parameters()):
        assert (a.detach().float() - b.detach().cpu()).abs().max().item() <= 2e-4


def test_fit_learns_a_repeating_text():
    from language_interpolation_amd import train as T
    torch.manual_seed(1)
    cfg = T.load_config(CFG, ["net=baseline_small_250", "net.hidden.width=64",
                              "net.hidden.layers=1", "optimizer=adam", "optimizer.lr=2e-3",
                              "batch_size=512"])
    module = T.ASCIIPredictionNet(cfg, device="cuda")
    data = T.CharWindows(b"the cat sat on the mat and the dog sat on the log. " * 40,
                         cfg.net.features, device="cuda")
    hist = T.fit(module, data, steps=150, batch_size=cfg.batch_size, seed=2)
    first = sum(l for l, _ in hist[:5]) / 5
    last = sum(l for l, _ in hist[-5:]) / 5
    assert last < 0.5 * first, (first, last)
    assert hist[-1][1] > 0.6


def test_training_with_the_input_layer_in_table_form_follows_the_general_path():
    """CharWindows vouches for the character grid (ascii_only); with the input layer declared
    (train.declare_character_inputs) ten Adam steps at 4096 rows stay on the general path's
    losses, and a text with a byte >= 128 is not vouched for."""
    from language_interpolation_amd import train as T
    cfg = T.load_config(CFG, ["net=baseline_small_250", "net.hidden.width=64",
                              "net.hidden.layers=1", "optimizer=adam", "optimizer.lr=1e-3",
                              "batch_size=4096"])
    text = T.synthetic_text(20000, seed=3)
    data = T.CharWindows(text, cfg.net.features, device="cuda")
    assert data.ascii_only
    assert not T.CharWindows(text + bytes([200]) * 40, cfg.net.features, device="cuda").ascii_only
    hists = []
    for declare in (False, True):
        torch.manual_seed(4)
        module = T.ASCIIPredictionNet(cfg, device="cuda")
        if declare:
            layer = T.declare_character_inputs(module.model)
            assert layer is module.model.model[0] and layer._spec.input_grid == 128
        hists.append(T.fit(module, data, steps=10, batch_size=cfg.batch_size, seed=5))
    for (la, _), (lb, _) in zip(*hists):
        assert abs(la - lb) <= 2e-4 * max(1.0, abs(la)), (la, lb)


def test_high_order_input_network_and_sparse_lion_step():
    from language_interpolation_amd import train as T
    torch.manual_seed(2)
    cfg = T.load_config(CFG, ["net=large_standard", "net.hidden.width=64", "batch_size=300"])
    module = T.ASCIIPredictionNet(cfg, device="cuda")
    data = T.CharWindows(T.synthetic_text(5000), cfg.net.features, device="cuda")
    hist = T.fit(module, data, steps=3, batch_size=cfg.batch_size)
    assert len(hist) == 3 and all(l == l for l, _ in hist)
    w = module.model[0].w
    assert w.grad is not None and (w.grad == 0).any()   # untouched segments: sparse gradient


def test_mamba_model_type_trains():
    """cfg5 (config/net/mamba.yaml with 2 layers): targets per position, loss goes down.
    (Thirty steps at 5e-4: the projections are un-normalised quadratic layers feeding each other, and
    with Adam at 1e-3 their outputs overflow fp32 after 35-90 steps whichever kernels run --
    plain-fp32 gather kernels included; `layer_type=linear` trains on.)"""
    from language_interpolation_amd import train as T
    torch.manual_seed(3)
    cfg = T.load_config(CFG, ["net=mamba", "net.n_layer=2", "optimizer=adam", "optimizer.lr=5e-4",
                              "batch_size=64", "data.max_features=16"])
    module = T.prediction_net(cfg, device="cuda")
    assert isinstance(module, T.MambaASCIIPredictionNet)
    data = T.make_data(cfg, b"the cat sat on the mat and the dog sat on the log. " * 40)
    x, y, _ = data.sample(4)
    assert module(x).shape == (4, 15, 128) and y.shape == (4, 15, 1)
    hist = T.fit(module, data, steps=30, batch_size=cfg.batch_size, seed=4, gradient_clip=5.0)
    first = sum(l for l, _ in hist[:5]) / 5
    last = sum(l for l, _ in hist[-5:]) / 5
    assert last < 0.7 * first, (first, last, hist[::10])


@pytest.mark.parametrize("overrides", [
    ["net=baseline_small_250", "net.hidden.width=64", "net.hidden.layers=1", "optimizer=adam",
     "optimizer.lr=1e-3", "batch_size=512"],
    ["net=mamba", "net.n_layer=1", "optimizer=lion", "optimizer.lr=1e-4", "batch_size=32",
     "data.max_features=16"],
])
def test_graph_replayed_steps_match_eager_steps(overrides):
    """fit(graph=True) replays the captured step; same init and batches as the eager loop give
    the same losses (up to the arrival order of the float atomics)."""
    from language_interpolation_amd import train as T
    cfg = T.load_config(CFG, overrides)
    text = b"the cat sat on the mat and the dog sat on the log. " * 40
    hists = []
    for graph in (False, True):
        torch.manual_seed(5)
        module = T.prediction_net(cfg, device="cuda")
        data = T.make_data(cfg, text)
        hists.append(T.fit(module, data, steps=12, batch_size=cfg.batch_size, seed=6, graph=graph))
    for (le, ae), (lg, ag) in zip(*hists):
        assert abs(le - lg) <= 2e-3 * max(1.0, abs(le)), (hists[0], hists[1])
    assert hists[1][-1][0] < hists[1][0][0]


def test_char_windows_and_sequences_are_exact_on_the_device():
    """The device-side dataset path (text as one uint8 buffer in HBM, windows gathered on the
    GPU): every window equals the reference construction (single_text_dataset.py:15-16, :102-123;
    MambaMixin.collate_fn, lightning_datamodule.py:272-288) -- bit-exact, integer and float."""
    from language_interpolation_amd import train as T
    text = T.synthetic_text(5000, seed=3)
    F = 16
    data = T.CharWindows(text, F, device="cuda")
    gen = torch.Generator(device="cuda").manual_seed(5)
    x, y, ids = data.sample(512, gen)
    assert x.is_cuda and y.is_cuda and x.shape == (512, F) and y.shape == (512, 1)
    raw = text if isinstance(text, (bytes, bytearray)) else text.encode("ascii", "ignore")
    codes = torch.tensor(list(raw), dtype=torch.float32)
    idc = ids.cpu()
    win = torch.stack([codes[i:i + F] for i in idc.tolist()])
    assert torch.equal(x.cpu(), (win - 64 + 0.5) / 64)
    assert torch.equal(y.cpu().flatten().float(), codes[idc + F])
    # first and last window
    xb, yb, _ = data.batch(torch.tensor([0, len(data) - 1], device="cuda"))
    assert torch.equal(xb[1].cpu(), (codes[len(data) - 1:len(data) - 1 + F] - 63.5) / 64)
    seq = T.CharSequences(text, F, device="cuda")
    xs, ys, ids = seq.sample(64, gen)
    assert xs.dtype == torch.long and xs.shape == (64, F - 1, 1)
    ic = ids.cpu()
    want_x = torch.stack([codes[i:i + F - 1] for i in ic.tolist()]).long()
    want_y = torch.stack([codes[i + 1:i + F] for i in ic.tolist()]).long()
    assert torch.equal(xs.cpu()[:, :, 0], want_x) and torch.equal(ys.cpu()[:, :, 0], want_y)


@pytest.mark.parametrize("sparse", [False, True])
def test_fused_lion_step_matches_the_torch_rule(sparse):
    """pp_lion_step_f32 (one launch for all parameters, storage order) against the same rule in
    torch ops on CPU copies: permuted layer parameters, an odd-sized tensor, weight decay, zero
    gradients (SparseLion leaves those entries and their momentum alone), two steps."""
    from language_interpolation_amd import train as T
    torch.manual_seed(5)
    shapes = [(250, 16, 129), (7, 3, 5), (33,), (128, 250, 5)]
    cpu_p, gpu_p = [], []
    for sh in shapes:
        base = torch.randn(*sh)
        if len(sh) == 3:   # physical [I,K,O], logical [O,I,K]: what the layers hold
            phys = base.permute(1, 2, 0).contiguous()
            cp = phys.clone().permute(2, 0, 1)
            gp = phys.clone().cuda().permute(2, 0, 1)
        else:
            cp, gp = base.clone(), base.clone().cuda()
        cpu_p.append(torch.nn.Parameter(cp))
        gpu_p.append(torch.nn.Parameter(gp))
    opt_c = T.Lion(cpu_p, lr=3e-3, betas=(0.9, 0.99), weight_decay=0.1, sparse=sparse)
    opt_g = T.Lion(gpu_p, lr=3e-3, betas=(0.9, 0.99), weight_decay=0.1, sparse=sparse)
    assert all(T.Lion._fusable(p, torch.zeros_like(p), torch.zeros_like(p)) for p in gpu_p)
    for step in range(2):
        for pc, pg in zip(cpu_p, gpu_p):
            g = torch.randn(pc.shape)
            g[torch.rand(pc.shape) < 0.3] = 0.0
            pc.grad = g.clone()
            pg.grad = torch.empty_like(pg).copy_(g)     # same strides as the parameter
        before = [pg.detach().cpu().clone() for pg in gpu_p]
        opt_c.step()
        opt_g.step()
        for pc, pg, b4 in zip(cpu_p, gpu_p, before):
            assert torch.allclose(pg.detach().cpu(), pc.detach(), rtol=0, atol=1e-6), step
            if sparse:  # untouched entries keep their bits: no update, no weight decay
                untouched = pc.grad == 0
                assert torch.equal(pg.detach().cpu()[untouched], b4[untouched]), step
            mc, mg = opt_c.state[pc]["exp_avg"], opt_g.state[pg]["exp_avg"]
            assert torch.allclose(mg.cpu(), mc, rtol=1e-6, atol=1e-7), step


@pytest.mark.parametrize("B,C", [(16384, 128), (1000, 128), (7, 5), (130, 1000), (3, 257), (1, 1)])
def test_cross_entropy_accuracy_kernel_matches_torch(B, C):
    """Loss, accuracy and the gradient of the fused kernel against torch's CrossEntropyLoss /
    argmax in fp64 (networks.py:121-132 of the reference)."""
    from language_interpolation_amd import functional as PF
    g = torch.Generator().manual_seed(B + C)
    logits = (4 * torch.randn(B, C, generator=g)).cuda().requires_grad_(True)
    target = torch.randint(0, C, (B,), generator=g).cuda()
    loss, acc = PF.cross_entropy_accuracy(logits, target)
    ref_in = logits.detach().double().requires_grad_(True)
    ref = torch.nn.functional.cross_entropy(ref_in, target)
    ref_acc = (torch.argmax(ref_in, dim=1) == target).double().mean()
    assert abs(loss.item() - ref.item()) <= 2e-6 * max(1.0, abs(ref.item()))
    assert abs(acc.item() - ref_acc.item()) <= 1e-6
    (3.0 * loss).backward()
    (3.0 * ref).backward()
    err = (logits.grad.double() - ref_in.grad).abs().max().item()
    assert err <= 1e-6 * max(ref_in.grad.abs().max().item(), 1e-30) + 1e-12
    # the same bits every time
    loss2, acc2 = PF.cross_entropy_accuracy(logits.detach(), target)
    assert torch.equal(loss2, loss.detach()) and torch.equal(acc2, acc)
