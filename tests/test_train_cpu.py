"""Host logic of the training harness (language_interpolation_amd/train.py): config
loading in the reference's yaml layout, the Lion / SparseLion update rules, the character
window gather.  No GPU."""
import os

import torch

from language_interpolation_amd import train as T

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CFG = os.path.join(ROOT, "configs", "high_order_interpolation.yaml")


def test_config_defaults_interpolation_and_overrides():
    c = T.load_config(CFG)
    assert c.net.model_type == "high_order" and c.net.input.width == 16   # ${net.features}
    assert c.optimizer.name == "sparse_lion" and c.optimizer.lr == 1e-5   # `1e-5` is a float
    assert c.net.get("n_out") is None and c.net.input.get("layer_type") == "continuous"
    c = T.load_config(CFG, ["net=baseline_small_250", "optimizer=adam", "batch_size=64",
                            "net.hidden.width=32", "net.features=8"])
    assert (c.net.hidden.width, c.net.n, c.net.hidden.layers, c.batch_size) == (32, 3, 3, 64)
    assert c.net.input.width == 8 and c.optimizer.name == "adam"
    c = T.load_config({"net": {"features": 4, "input": {"width": "${net.features}"}}})
    assert c.net.input.width == 4


def test_select_normalization_names():
    from language_interpolation_amd import layers as L
    assert T.select_normalization("layer") is L.LazyLayerNormLastDim
    assert T.select_normalization("maxabs") is L.MaxAbsNormalizationLast
    assert T.select_normalization("maxcenter") is L.MaxCenterNormalizationLast
    x = torch.randn(3, 5)
    assert torch.equal(T.select_normalization("none")()(x), x)
    try:
        T.select_normalization("batch")
    except ValueError:
        pass
    else:
        raise AssertionError("unknown normalization accepted")


def test_lion_and_sparse_lion_updates():
    torch.manual_seed(0)
    p0 = torch.randn(6)
    g1 = torch.tensor([0.5, -2.0, 0.0, 1e-3, 0.0, -1e-3])
    g2 = torch.tensor([-1.0, -1.0, 0.0, 1.0, 2.0, 0.0])
    lr, b1, b2 = 0.1, 0.9, 0.99
    for sparse in (False, True):
        p = torch.nn.Parameter(p0.clone())
        opt = T.Lion([p], lr=lr, sparse=sparse)
        ref, m = p0.clone(), torch.zeros(6)
        for g in (g1, g2):
            p.grad = g.clone()
            opt.step()
            upd = torch.sign(b1 * m + (1 - b1) * g)
            new_m = b2 * m + (1 - b2) * g
            if sparse:
                upd = upd * (g != 0)
                new_m = torch.where(g != 0, new_m, m)
            ref = ref - lr * upd
            m = new_m
            assert torch.allclose(p.detach(), ref, atol=1e-7), sparse
    # dense Lion moves an untouched weight once it has momentum, SparseLion does not
    assert ref[2] == p0[2]


def test_char_windows_match_reference_construction():
    text = "The quick brown fox jumps over the lazy dog.\nPack my box!"
    F = 7
    data = T.CharWindows(text, F, device="cpu")
    assert len(data) == len(text) - (F + 1)
    idx = torch.tensor([0, 3, len(data) - 1])
    x, y, _ = data.batch(idx)
    for r, i in enumerate(idx.tolist()):
        feat = torch.tensor([ord(ch) for ch in text[i:i + F]], dtype=torch.float32)
        assert torch.equal(x[r], (feat - 64 + 0.5) / 64)        # single_text_dataset.py:15-16
        assert y[r, 0].item() == ord(text[i + F])
    xs, ys, ids = data.sample(5, torch.Generator().manual_seed(1))
    assert xs.shape == (5, F) and ys.shape == (5, 1) and ids.max() < len(data)
    assert len(T.synthetic_text(1000)) == 1000


def test_char_sequences_shifted_targets_and_mamba_config():
    text = "The quick brown fox jumps over the lazy dog."
    data = T.CharSequences(text, 6, device="cpu")
    x, y, _ = data.batch(torch.tensor([0, 4]))
    assert x.shape == (2, 5, 1) and y.shape == (2, 5, 1) and x.dtype == torch.long
    assert x[1, :, 0].tolist() == [ord(c) for c in text[4:9]]
    assert y[1, :, 0].tolist() == [ord(c) for c in text[5:10]]   # features shifted by one
    cfg = T.load_config(CFG, ["net=mamba", "net.n_layer=2", "data.max_features=16"])
    assert cfg.net.model_type == "mamba" and cfg.net.dt_rank == "auto" and cfg.net.d_state == 32
    assert isinstance(T.make_data(cfg, text, device="cpu"), T.CharSequences)
