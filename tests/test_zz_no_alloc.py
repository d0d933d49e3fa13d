"""The allocation-free forward entry point (pp_forward_noalloc_f32) against pp_forward_ws_f32,
through the C ABI: same bits, on two calls in a row (the second on a smaller input, so stale
scratch contents of the first would show), plus its size limit.  Collected last on purpose."""
import ctypes

import pytest
import torch

from pp_testlib import make_inputs

pytestmark = pytest.mark.gpu


def _call_noalloc(lib, PF, x, w_t, spec, method):
    B, I = x.shape
    K, O = w_t.shape[1], w_t.shape[2]
    y = torch.full((B, O), float("nan"), device=x.device)
    rc = lib.pp_forward_noalloc_f32(
        PF._ptr(x), PF._ptr(w_t), PF._ptr(y), PF._ptr(None), B, I, O, spec.n, spec.segments,
        int(spec.discontinuous), spec.length, spec.node_array(), spec.out_scale, method,
        PF._stream(x.device))
    return rc, y


def _call_ws(lib, PF, x, w_t, spec, method):
    B, I = x.shape
    O = w_t.shape[2]
    nbytes = lib.pp_forward_workspace_bytes(B, I, O, spec.n, spec.segments,
                                            int(spec.discontinuous), spec.length, method)
    ws = torch.empty(max(nbytes, 16), device=x.device, dtype=torch.uint8)
    y = torch.full((B, O), float("nan"), device=x.device)
    rc = lib.pp_forward_ws_f32(
        PF._ptr(x), PF._ptr(w_t), PF._ptr(y), PF._ptr(None), PF._ptr(ws), nbytes, B, I, O, spec.n,
        spec.segments, int(spec.discontinuous), spec.length, spec.node_array(), spec.out_scale,
        method, PF._stream(x.device))
    return rc, y


def test_noalloc_forward_matches_workspace_forward_twice():
    from language_interpolation_amd import _lib
    from language_interpolation_amd import functional as PF
    lib = _lib.load()
    torch.cuda.set_device(0)
    # first call: cfg2 hidden layer at a large batch; second: a smaller layer and batch
    for (B, I, O, n, S, disc) in ((4096 + 5, 250, 250, 3, 2, False), (500, 64, 96, 2, 2, True)):
        x, w, _ = make_inputs(B, I, O, n, S, disc, "normal", seed=60 + B % 7)
        xd = x.cuda()
        w_t = w.cuda().permute(1, 2, 0).contiguous()
        spec = PF.make_spec(n, S, disc, in_features=I)
        for method in (_lib.METHOD_AUTO, _lib.FWD_FP16X2, _lib.FWD_BF16X3, _lib.FWD_GATHER):
            assert lib.pp_auto_method(0, B, I, O, n, S, int(disc), 2.0) == _lib.FWD_FP16X2
            rc1, y1 = _call_noalloc(lib, PF, xd, w_t, spec, method)
            rc2, y2 = _call_ws(lib, PF, xd, w_t, spec, method)
            torch.cuda.synchronize()
            assert rc1 == 0 and rc2 == 0, (rc1, rc2, method)
            assert torch.equal(y1, y2), (B, method)
            assert torch.isfinite(y1).all()


def test_noalloc_forward_rejects_oversized_shapes_and_prepacked():
    from language_interpolation_amd import _lib
    from language_interpolation_amd import functional as PF
    lib = _lib.load()
    B, I, O, n, S = 512, 1024, 1024, 4, 2      # K = 7: 3*7*64*1024*32 bytes of pieces >> 8 MiB
    need = lib.pp_forward_workspace_bytes(B, I, O, n, S, 0, 2.0, _lib.FWD_FP16X2)
    assert need > _lib.NOALLOC_SCRATCH_BYTES
    x = torch.zeros(B, I, device="cuda")
    w_t = torch.zeros(I, 7, O, device="cuda")
    spec = PF.make_spec(n, S, False, in_features=I)
    rc, _ = _call_noalloc(lib, PF, x, w_t, spec, _lib.FWD_FP16X2)
    assert rc == -7
    assert b"PP_NOALLOC_SCRATCH_BYTES" in lib.pp_error_string(rc)
    rc, y = _call_noalloc(lib, PF, x, w_t, spec, _lib.FWD_GATHER)   # no scratch: always fits
    torch.cuda.synchronize()
    assert rc == 0 and float(y.abs().max()) == 0.0
    rc, _ = _call_noalloc(lib, PF, x, w_t, spec, _lib.FWD_FP16X2 | _lib.WS_PREPACKED)
    assert rc == -5


def test_noalloc_forward_on_a_random_sequence_of_shapes():
    """60 calls in a row on randomly drawn layers and batch sizes (automatic method; the shared
    scratch array keeps each call's pieces until the next overwrites them): the same bits as
    pp_forward_ws_f32 every time; where the workspace query says the automatic method needs more
    than the array holds, the bits of pp_forward_f32 (the automatic choice without scratch) and
    the size refusal for the scratch method asked for by name."""
    import random

    from language_interpolation_amd import _lib
    from language_interpolation_amd import functional as PF
    lib = _lib.load()
    torch.cuda.set_device(0)
    rng = random.Random(77)
    ran = refused = 0
    for _ in range(60):
        disc = rng.random() < 0.4
        n = rng.choice([1, 2, 3, 4, 5, 8]) if disc else rng.choice([2, 3, 4, 5, 8])
        S = rng.choice([1, 2, 2, 2, 4, 16, 128])
        I, O = rng.choice([16, 33, 64, 100, 250, 256, 400]), rng.choice([16, 40, 96, 128, 250, 300])
        B = rng.choice([1, 100, 383, 384, 1000, 2049, 4096, 8200])
        x, w, _ = make_inputs(B, I, O, n, S, disc, rng.choice(["normal", "uniform", "edges"]),
                              seed=rng.randrange(1 << 20))
        xd = x.cuda()
        w_t = w.cuda().permute(1, 2, 0).contiguous()
        spec = PF.make_spec(n, S, disc, in_features=I)
        need = lib.pp_forward_workspace_bytes(B, I, O, n, S, int(disc), 2.0, _lib.METHOD_AUTO)
        rc1, y1 = _call_noalloc(lib, PF, xd, w_t, spec, _lib.METHOD_AUTO)
        rc2, y2 = _call_ws(lib, PF, xd, w_t, spec, _lib.METHOD_AUTO)
        torch.cuda.synchronize()
        assert rc2 == 0
        if need > _lib.NOALLOC_SCRATCH_BYTES:
            # the automatic choice falls back to the method pp_forward_f32 runs (no scratch);
            # the method that needs the scratch, asked for by name, is refused
            assert rc1 == 0, (B, I, O, n, S, disc, need)
            y3 = torch.full_like(y1, float("nan"))
            rc3 = lib.pp_forward_f32(
                PF._ptr(xd), PF._ptr(w_t), PF._ptr(y3), PF._ptr(None), B, I, O, n, S, int(disc),
                spec.length, spec.node_array(), spec.out_scale, _lib.METHOD_AUTO,
                PF._stream(xd.device))
            rc4, _ = _call_noalloc(lib, PF, xd, w_t, spec, _lib.FWD_FP16X2)
            torch.cuda.synchronize()
            assert rc3 == 0 and torch.equal(y1, y3), (B, I, O, n, S, disc)
            assert rc4 == -7, (B, I, O, n, S, disc, need)
            refused += 1
        else:
            assert rc1 == 0, (B, I, O, n, S, disc, need)
            assert torch.equal(y1, y2), (B, I, O, n, S, disc)
            ran += 1
    assert ran >= 40, (ran, refused)


def test_noalloc_forward_under_stream_capture():
    """The entry point inside a HIP graph capture (the callers it exists for): after one call
    outside capture -- the first call on a device looks up the scratch address -- a captured call
    replays to the same bits as pp_forward_ws_f32, twice, the second replay on new inputs."""
    from language_interpolation_amd import _lib
    from language_interpolation_amd import functional as PF
    lib = _lib.load()
    torch.cuda.set_device(0)
    B, I, O, n, S, disc = 2048, 250, 250, 3, 2, False
    x, w, _ = make_inputs(B, I, O, n, S, disc, "normal", seed=91)
    xd = x.cuda()
    w_t = w.cuda().permute(1, 2, 0).contiguous()
    spec = PF.make_spec(n, S, disc, in_features=I)
    rc, _ = _call_noalloc(lib, PF, xd, w_t, spec, _lib.METHOD_AUTO)   # resolves the address
    assert rc == 0
    y = torch.full((B, O), float("nan"), device="cuda")
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.stream(side):
        with torch.cuda.graph(graph, stream=side):
            rc = lib.pp_forward_noalloc_f32(
                PF._ptr(xd), PF._ptr(w_t), PF._ptr(y), PF._ptr(None), B, I, O, n, S, int(disc),
                spec.length, spec.node_array(), spec.out_scale, _lib.METHOD_AUTO,
                PF._stream(xd.device))
    assert rc == 0
    torch.cuda.current_stream().wait_stream(side)
    for rep in range(2):
        if rep:
            xd.copy_(make_inputs(B, I, O, n, S, disc, "uniform", seed=92)[0])
        graph.replay()
        rc2, y2 = _call_ws(lib, PF, xd, w_t, spec, _lib.METHOD_AUTO)
        torch.cuda.synchronize()
        assert rc2 == 0 and torch.equal(y, y2), rep
