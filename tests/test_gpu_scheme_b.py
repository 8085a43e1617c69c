"""Scheme B of the shard group (L2Z_SCHEME_B=1): Wo and W2 sharded by COLUMNS, every rank's partial [dim]
vector summed by an all-reduce -- 2 collectives per layer instead of scheme A's 4 all-gathers (main.zig:392 / :419 are the
two products whose sums are split across ranks).  The sum over a row is split differently than in the unsharded pass, so
the bar is the LOGIT TOLERANCE of tests/test_gpu_parity.py against the CPU oracle and the unsharded HIP pass -- not bit
identity -- while the ranks must agree with EACH OTHER bit for bit (same partials, summed in rank order by every rank).

Here: emulated ranks (one process, the real per-rank launches, the all-reduce done by the driver) for N = 2, 4, 8; the
multi-process forms (peer-write reduce launch) are in tests/test_gpu_p2p.py."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

LOGIT_RTOL = 5e-5   # tests/test_gpu_parity.py's bound on |gpu - oracle|
LOGIT_ATOL = 5e-5

SHAPES = [
    # GQA toy: shards narrower than a 64-lane sweep (the per-wave kernels), uploaded from a host blob
    ("toy-gqa", dict(dim=128, hidden_dim=352, n_layers=2, n_heads=16, n_kv_heads=8, vocab_size=512, seq_len=16), True),
    # hidden shards of 1376 / 688 / 344 floats: 1376 is padded to 1536 (whole 256-float sweeps), the others are not
    ("pad-1376", dict(dim=1024, hidden_dim=2752, n_layers=2, n_heads=16, n_kv_heads=8, vocab_size=2048, seq_len=24), False),
    # the 7B widths, one layer: W2's column shard takes the wide-row kernel at N = 2 (5504 -> 5632 floats per row)
    ("7B-width", dict(dim=4096, hidden_dim=11008, n_layers=1, n_heads=32, n_kv_heads=32, vocab_size=4096, seq_len=16), False),
]


@pytest.fixture
def scheme_b(gpu):
    gpu.option_set("L2Z_SCHEME_B", 1)
    yield
    gpu.option_set("L2Z_SCHEME_B", 0)


@pytest.mark.parametrize("world", [2, 4, 8])
@pytest.mark.parametrize("name,kw,from_blob", SHAPES, ids=[s[0] for s in SHAPES])
def test_scheme_b_emulated_ranks(gpu, ck, orc, options, scheme_b, world, name, kw, from_blob):
    options(L2Z_FUSE_SMALL=0)
    cfg = ck.Config(**kw)
    seed, shared = 23, False
    blob = ck.synth_blob(cfg, shared, seed) if (from_blob or cfg.dim <= 1024) else None
    gpu.option_set("L2Z_SCHEME_B", 0)
    w0, s0 = gpu.Weights(cfg, blob, shared, seed=seed), gpu.RunState(cfg)
    gpu.option_set("L2Z_SCHEME_B", 1)
    comms = [gpu.Comm(r, world, None, 0, emulated=True) for r in range(world)]
    ws = [gpu.Weights(cfg, blob if from_blob else None, shared, seed=seed, comm=c) for c in comms]
    ss = [gpu.RunState(cfg, comm=c) for c in comms]
    assert all(s.form() & 8 for s in ss) and s0.form() & 8 == 0
    m = orc.Model(cfg.as_i32(), blob, shared) if blob is not None else None
    rng = np.random.default_rng(5)
    toks = [1] + rng.integers(2, cfg.vocab_size, 5).tolist()
    worst = 0.0
    for pos, t in enumerate(toks):
        s0.transformer(t, pos, w0)
        ref = s0.logits()
        gpu.emu_transformer(ss, ws, t, pos)
        got = ss[0].logits()
        for r in range(1, world):
            assert np.array_equal(ss[r].logits(), got), f"{name} x{world}: rank {r} differs from rank 0 at pos {pos}"
        np.testing.assert_allclose(got, ref, rtol=LOGIT_RTOL, atol=LOGIT_ATOL, err_msg=f"{name} x{world} pos {pos} vs unsharded")
        if m is not None:
            np.testing.assert_allclose(got, m.transformer(t, pos), rtol=LOGIT_RTOL, atol=LOGIT_ATOL,
                                       err_msg=f"{name} x{world} pos {pos} vs oracle")
        worst = max(worst, float(np.abs(got - ref).max()))
    print(f"scheme B {name} x{world}: max |logit - unsharded| {worst:.2e}")
    for o in ss + ws + [s0, w0] + ([m] if m is not None else []):
        o.close()
    for c in comms:
        c.close()


def test_scheme_b_objects_must_match(gpu, ck):
    """Weights built under one scheme cannot be driven by a RunState of the other (the column shards have another layout)."""
    cfg = ck.Config(dim=128, hidden_dim=352, n_layers=1, n_heads=16, n_kv_heads=8, vocab_size=512, seq_len=16)
    comm = gpu.Comm(0, 2, None, 0, emulated=True)
    comm1 = gpu.Comm(1, 2, None, 0, emulated=True)
    try:
        gpu.option_set("L2Z_SCHEME_B", 1)
        ws = [gpu.Weights(cfg, None, False, seed=1, comm=c) for c in (comm, comm1)]
        gpu.option_set("L2Z_SCHEME_B", 0)
        ss = [gpu.RunState(cfg, comm=c) for c in (comm, comm1)]
        with pytest.raises(gpu.L2ZError) as e:
            gpu.emu_transformer(ss, ws, 1, 0)
        assert "scheme" in str(e.value)
    finally:
        gpu.option_set("L2Z_SCHEME_B", 0)
    for o in ss + ws:
        o.close()
    comm.close(); comm1.close()


PREFILL_SHAPES = [
    # shards narrower than a tile, 45 tokens: the short-prompt GEMMs on K = 16 ... 64-column shards
    ("toy-gqa", dict(dim=128, hidden_dim=352, n_layers=2, n_heads=16, n_kv_heads=8, vocab_size=512, seq_len=64), 45),
    # hidden shards of 1376 floats padded to 1536 (N = 2); 150 tokens: the tile GEMMs, flash attention
    ("pad-1376", dict(dim=1024, hidden_dim=2752, n_layers=2, n_heads=16, n_kv_heads=8, vocab_size=2048, seq_len=192), 150),
    # the 7B widths, one layer, 70 tokens: column shards of 2048 ... 512 (Wo) and 5632 ... 1536 (W2) floats as the GEMMs' K
    ("7B-width", dict(dim=4096, hidden_dim=11008, n_layers=1, n_heads=32, n_kv_heads=32, vocab_size=4096, seq_len=96), 70),
]


@pytest.mark.parametrize("world", [2, 4, 8])
@pytest.mark.parametrize("name,kw,n_tok", PREFILL_SHAPES, ids=[s[0] for s in PREFILL_SHAPES])
def test_scheme_b_batched_prefill_emulated_ranks(gpu, ck, options, scheme_b, world, name, kw, n_tok):
    """The batched prompt pass on column-sharded Wo / W2 (prefill_host.cpp prefill_half_b; main.zig:999-1000 over :392 /
    :419): per layer two partial [tokens, dim] products per rank, summed over the ranks in rank order (here by the
    emulated-rank driver; real groups: the bulk all-reduce).  Two calls (pos0 > 0 for the second).  Every rank's logits
    equal rank 0's bit for bit; against the UNSHARDED batched pass they hold the logit tolerance (the products are split
    across ranks differently), KV rows likewise; decoding continues from the sharded state."""
    options(L2Z_FUSE_SMALL=0)
    cfg = ck.Config(**kw)
    gpu.option_set("L2Z_SCHEME_B", 0)
    w0, s0 = gpu.Weights(cfg, None, False, seed=29), gpu.RunState(cfg)
    gpu.option_set("L2Z_SCHEME_B", 1)
    comms = [gpu.Comm(r, world, None, 0, emulated=True) for r in range(world)]
    ws = [gpu.Weights(cfg, None, False, seed=29, comm=c) for c in comms]
    ss = [gpu.RunState(cfg, comm=c) for c in comms]
    assert all(s.form() & 8 for s in ss)
    rng = np.random.default_rng(8)
    toks = [1] + rng.integers(2, cfg.vocab_size, n_tok - 1).tolist()
    worst = 0.0
    for lo, hi in ((0, 9), (9, n_tok)):
        s0.prefill(toks[lo:hi], lo, w0)
        gpu.emu_prefill(ss, ws, toks[lo:hi], lo)
        ref, got = s0.logits(), ss[0].logits()
        for r in range(1, world):
            assert np.array_equal(ss[r].logits(), got), f"{name} x{world}: rank {r} differs from rank 0 after tokens {lo}..{hi}"
        np.testing.assert_allclose(got, ref, rtol=LOGIT_RTOL, atol=LOGIT_ATOL, err_msg=f"{name} x{world} tokens {lo}..{hi}")
        worst = max(worst, float(np.abs(got - ref).max()))
    kvd, S = cfg.kv_dim, cfg.seq_len
    kvl = kvd // world
    for l in range(cfg.n_layers):
        for nm in ("key_cache", "value_cache"):
            full = s0.read(nm, l * S * kvd, n_tok * kvd).reshape(n_tok, kvd)
            for r in range(world):
                mine = ss[r].read(nm, l * S * kvl, n_tok * kvl).reshape(n_tok, kvl)
                np.testing.assert_allclose(mine, full[:, r * kvl:(r + 1) * kvl], rtol=2e-5, atol=2e-5, err_msg=f"{nm} layer {l} rank {r}")
    nxt = s0.argmax()
    s0.transformer(nxt, n_tok, w0)
    gpu.emu_transformer(ss, ws, nxt, n_tok)
    np.testing.assert_allclose(ss[0].logits(), s0.logits(), rtol=LOGIT_RTOL, atol=LOGIT_ATOL)
    print(f"scheme B batched prefill {name} x{world}: max |logit - unsharded| {worst:.2e}")
    for o in ss + ws + [s0, w0]:
        o.close()
    for c in comms:
        c.close()


def test_rccl_allreduce_call_path_world1(gpu, ck, scheme_b):
    """Scheme B over RCCL cannot run with 2 ranks on one GPU (RCCL refuses); a 1-rank communicator still walks the call
    path of the N > 1 leg: dlsym of ncclAllReduce, the column-shard mat-vecs into the partial buffer, ncclAllReduce(part
    -> x) captured in the step graph, the logits' ncclAllGather.  One rank's "partial" is the whole sum, so here the
    logits must match the communicator-free pass at the tolerance and the tokens exactly."""
    cfg = ck.Config(dim=64, hidden_dim=172, n_layers=2, n_heads=4, n_kv_heads=2, vocab_size=512, seq_len=32)
    blob = ck.synth_blob(cfg, False, 55)
    gpu.option_set("L2Z_SCHEME_B", 0)
    w0, s0 = gpu.Weights(cfg, blob, False), gpu.RunState(cfg)
    gpu.option_set("L2Z_SCHEME_B", 1)
    s0.greedy_begin([3, 4])
    ref = s0.greedy_run(w0, cfg.seq_len)
    comm = gpu.Comm(0, 1, gpu.Comm.unique_id(), 0)
    w1, s1 = gpu.Weights(cfg, blob, False, comm=comm), gpu.RunState(cfg, comm=comm)
    assert s1.form() & 8
    s1.greedy_begin([3, 4])
    got = s1.greedy_run(w1, cfg.seq_len)
    assert np.array_equal(got, ref)
    s1.transformer(1, 0, w1)
    s0.transformer(1, 0, w0)
    np.testing.assert_allclose(s1.logits(), s0.logits(), rtol=LOGIT_RTOL, atol=LOGIT_ATOL)
    # the batched prompt pass on the same group: its all-reduce is ncclAllReduce over the [tokens, dim] partial
    prompt = [1] + list(range(5, 24))
    s1.prefill(prompt, 0, w1)
    s0.prefill(prompt, 0, w0)
    np.testing.assert_allclose(s1.logits(), s0.logits(), rtol=LOGIT_RTOL, atol=LOGIT_ATOL)
    for o in (s0, s1, w0, w1):
        o.close()
    comm.close()
