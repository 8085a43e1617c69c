"""The N > 1 data path with REAL processes: 2 and 4 ranks, one process each, all on the one GPU
of the test box, exchanging activations through the peer-write all-gather (IPC-mapped arenas,
direct stores + flags, csrc/p2p.hip).  Every rank's tokens and logits must equal the unsharded
run's bit for bit (a row's dot product does not depend on who owns the row)."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))

MODELS = [
    ("toy-gqa", dict(dim=64, hidden_dim=176, n_layers=3, n_heads=8, n_kv_heads=4, vocab_size=512, seq_len=96), False, 2),
    ("toy-gqa", dict(dim=64, hidden_dim=176, n_layers=3, n_heads=8, n_kv_heads=4, vocab_size=512, seq_len=96), False, 4),
    ("stories15M-shape-2layers", dict(dim=288, hidden_dim=768, n_layers=2, n_heads=6, n_kv_heads=6, vocab_size=32000, seq_len=256), True, 2),
    ("wide-rows", dict(dim=1024, hidden_dim=4096, n_layers=2, n_heads=8, n_kv_heads=8, vocab_size=4096, seq_len=320), False, 4),
    # n = 4096: the row kernel, whose writer lanes push their outputs to the peers themselves
    ("row-kernel-7B-width", dict(dim=4096, hidden_dim=8192, n_layers=2, n_heads=32, n_kv_heads=8, vocab_size=8192, seq_len=320), False, 4),
    ("row-kernel-7B-width", dict(dim=4096, hidden_dim=8192, n_layers=2, n_heads=32, n_kv_heads=8, vocab_size=8192, seq_len=320), False, 2),
    ("row-kernel-7B-width", dict(dim=4096, hidden_dim=8192, n_layers=2, n_heads=32, n_kv_heads=8, vocab_size=8192, seq_len=320), False, 8),
]


def grid_cap(world: int) -> str:
    """All ranks share ONE GPU here: a mat-vec launch that may be waiting for a peer's kernel must
    leave that kernel room to run (on real multi-GPU nodes every rank has its own chip)."""
    return str(max(32, 512 // world))


def run_ranks(tmp_path, world, spec, env_extra=None, timeout=300):
    (tmp_path / "model.json").write_text(json.dumps(spec))
    env = dict(os.environ, L2Z_P2P_TIMEOUT_S="30", L2Z_GRID_CAP=grid_cap(world))
    env.update(env_extra or {})
    procs = [subprocess.Popen([sys.executable, os.path.join(HERE, "p2p_worker.py"), str(r), str(world),
                               str(tmp_path), str(tmp_path / "model.json")], env=env,
                              stdout=subprocess.PIPE, stderr=subprocess.STDOUT) for r in range(world)]
    outs = []
    for p in procs:
        try:
            out, _ = p.communicate(timeout=timeout)
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            raise
        outs.append(out.decode(errors="replace"))
    for r, p in enumerate(procs):
        assert p.returncode == 0, f"rank {r} failed:\n{outs[r][-2000:]}"
    return outs


# consume: consumers read the LL words from their own landing slot (no gather launches; the default up to 2 ranks or for
# rows narrower than 4096, L2Z_P2P_CONSUME=1 here);
# gather: one gather launch per gathered vector, which also sends (L2Z_P2P_CONSUME=0); default: what the library picks by
# shape.  (Until round 5 also "gather-push" -- producers' epilogues send where a gather launch collects, L2Z_P2P_PUSH=2 --
# and "nopush": the form measured slower on every bed and its knob are gone.)
CASES = ([(m, "consume") for m in MODELS] + [(MODELS[0], "gather"), (MODELS[4], "gather"), (MODELS[3], "default")] +
         [(MODELS[5], "gather")])
MODE_ENV = {"consume": {"L2Z_P2P_CONSUME": "1"}, "gather": {"L2Z_P2P_CONSUME": "0"}, "default": {}}


@pytest.mark.parametrize("model,mode", CASES, ids=[f"{m[0]}-x{m[3]}-{mode}" for m, mode in CASES])
def test_multiprocess_peer_write_gather_is_bit_identical(gpu, ck, tmp_path, model, mode, options):
    name, kw, shared, world = model
    options(L2Z_FUSE_SMALL=0)  # the unsharded reference runs the launches the shards run
    cfg = ck.Config(**kw)
    steps = min(cfg.seq_len - 2, 300)
    on_device = cfg.dim >= 4096  # big shapes: seeded weights generated on the device by every rank
    spec = dict(cfg=kw, shared=shared, seed=33, prompt=[5, 9, 11], steps=steps, blob=not on_device)
    run_ranks(tmp_path, world, spec, MODE_ENV[mode])
    # unsharded reference in this process
    blob = None if on_device else ck.synth_blob(cfg, shared, 33)
    w, s = gpu.Weights(cfg, blob, shared, seed=33), gpu.RunState(cfg)
    s.greedy_begin(spec["prompt"])
    toks = s.greedy_run(w, steps)
    logits = s.logits()
    s.transformer(int(toks[-1]), len(toks) % cfg.seq_len, w)
    logits2, am = s.logits(), s.argmax()
    for r in range(world):
        o = np.load(tmp_path / f"out_{r}.npz")
        assert np.array_equal(o["toks"], toks), f"rank {r} tokens"
        assert np.array_equal(o["logits"], logits), f"rank {r} logits"
        assert np.array_equal(o["logits2"], logits2) and int(o["am"]) == am, f"rank {r} stepped call"
    s.close(); w.close()


@pytest.mark.parametrize("world,mode", [(2, "consume"), (4, "gather"), (8, "gather"), (4, "scheme_b")])
def test_argmax_tie_across_ranks_takes_the_lowest_index(gpu, ck, tmp_path, world, mode, options):
    """main.zig:715-726 over a SHARDED vocabulary: greedy steps of a shard group exchange one (max, first index) candidate
    per rank instead of gathering the logits (csrc/misc_kernels.hip argmax_kernel, ArgmaxArgs::xchg).  The classifier is
    rigged so that the largest logit is attained at two indices that live on DIFFERENT ranks, bit for bit: every step must
    produce the lower one, as the unsharded pass does (strict '>', :720), and the stepped API (full logits gather +
    l2z_argmax) must agree."""
    from p2p_worker import tie_classifier_rows
    options(L2Z_FUSE_SMALL=0, L2Z_PREFILL=0)
    kw = dict(dim=64, hidden_dim=176, n_layers=2, n_heads=8, n_kv_heads=8, vocab_size=512, seq_len=48)
    cfg = ck.Config(**kw)
    per = cfg.vocab_size // world
    # a < b and c < d, each pair on two different ranks; the higher index sits on a LOWER... and on a higher rank both ways
    rows = [per - 3, (world - 1) * per + 5, per + 7 if world > 2 else 9, (world - 1) * per + 1, 17]
    spec = dict(cfg=kw, shared=False, seed=41, prompt=[], steps=40, tie_rows=rows)
    env = {"consume": {"L2Z_P2P_CONSUME": "1"}, "gather": {"L2Z_P2P_CONSUME": "0"}, "scheme_b": {"L2Z_SCHEME_B": "1"}}[mode]
    run_ranks(tmp_path, world, spec, env)
    blob = ck.synth_blob(cfg, False, 41)
    tie_classifier_rows(ck, cfg, blob, False, rows)
    w, s = gpu.Weights(cfg, blob, False), gpu.RunState(cfg)
    s.greedy_begin([])
    toks = s.greedy_run(w, 40)
    lg = s.logits()
    tied = sum(int(t) in (rows[0], rows[2]) for t in toks)
    assert tied >= 20 and not set(toks.tolist()) & {rows[1], rows[3]}, f"the rigged rows must win most steps, by their lower index: {toks}"
    if int(toks[-1]) in (rows[0], rows[2]):
        assert np.sum(lg == lg.max()) == 2, "the rigged classifier ties the maximum at exactly two indices"
    for r in range(world):
        o = np.load(tmp_path / f"out_{r}.npz")
        assert np.array_equal(o["toks"], toks), f"rank {r}: {o['toks']} vs {toks}"
        if mode != "scheme_b":
            assert np.array_equal(o["logits"], lg), f"rank {r} logits (gathered on demand after the exchanging steps)"
        assert int(o["am"]) == int(np.argmax(o["logits2"])), f"rank {r}: l2z_argmax after a full-logits pass"
    s.close(); w.close()


SCHEME_B = [(MODELS[0], "reduce"), (MODELS[1], "reduce"), (MODELS[3], "reduce"), (MODELS[5], "reduce"), (MODELS[4], "reduce")]


@pytest.mark.parametrize("model,mode", SCHEME_B, ids=[f"{m[0]}-x{m[3]}-{mode}" for m, mode in SCHEME_B])
def test_multiprocess_scheme_b_allreduce(gpu, ck, tmp_path, model, mode, options):
    """Scheme B with real processes (L2Z_SCHEME_B=1: Wo / W2 sharded by columns, the ranks' partial [dim] vectors pushed as
    LL words into every peer's slot and summed in rank order by the reduce launch, csrc/p2p.hip): 2 collectives per layer.
    The ranks must agree with each other BIT FOR BIT; against the unsharded pass fed the same tokens the logits hold the
    parity tests' tolerance (the row sums are split differently).  The reduce launch sends the rank's partial itself."""
    name, kw, shared, world = model
    options(L2Z_FUSE_SMALL=0, L2Z_PREFILL=0)
    cfg = ck.Config(**kw)
    steps = min(cfg.seq_len - 2, 120)
    on_device = cfg.dim >= 4096
    spec = dict(cfg=kw, shared=shared, seed=33, prompt=[5, 9, 11], steps=steps, blob=not on_device)
    run_ranks(tmp_path, world, spec, {"L2Z_SCHEME_B": "1"})
    outs = [np.load(tmp_path / f"out_{r}.npz") for r in range(world)]
    for r in range(1, world):
        for k in ("toks", "logits", "logits2"):
            assert np.array_equal(outs[r][k], outs[0][k]), f"rank {r} {k} differs from rank 0"
        assert int(outs[r]["am"]) == int(outs[0]["am"])
    # the unsharded pass over the SAME token sequence (the sharded run's tokens forced as the prompt, main.zig:999-1000)
    toks = outs[0]["toks"]
    blob = None if on_device else ck.synth_blob(cfg, shared, 33)
    w, s = gpu.Weights(cfg, blob, shared, seed=33), gpu.RunState(cfg)
    s.greedy_begin(toks.tolist())
    s.greedy_run(w, len(toks))
    np.testing.assert_allclose(outs[0]["logits"], s.logits(), rtol=5e-5, atol=5e-5)
    s.transformer(int(toks[-1]), len(toks) % cfg.seq_len, w)
    np.testing.assert_allclose(outs[0]["logits2"], s.logits(), rtol=5e-5, atol=5e-5)
    print(f"scheme B {name} x{world} {mode}: max |logit - unsharded| {np.abs(outs[0]['logits2'] - s.logits()).max():.2e}")
    s.close(); w.close()


@pytest.mark.parametrize("world", [2, 4, 8])
def test_multiprocess_scheme_b_batched_prefill(gpu, ck, tmp_path, world, options):
    """Scheme B's batched prompt pass with REAL processes: per layer two bulk all-reduces of the ranks' partial [tokens,
    dim] products -- reduce-scatter through the arenas' bulk regions (every peer gets the columns of ITS slice, adds the
    blocks in rank order), then the bulk all-gather of the summed slices (csrc/p2p.hip, comm.cpp comm_bulk_allreduce).
    The greedy loop takes it for its 150-token prompt; l2z_prefill in two calls.  Ranks bit-identical to each other;
    against the unsharded pass fed the same tokens: the logit tolerance."""
    options(L2Z_FUSE_SMALL=0)
    kw = dict(dim=512, hidden_dim=1408, n_layers=2, n_heads=16, n_kv_heads=8, vocab_size=1024, seq_len=256)
    cfg = ck.Config(**kw)
    rng = np.random.default_rng(3)
    prompt = rng.integers(2, cfg.vocab_size, 150).tolist()
    pf = [1] + prompt[:149]
    spec = dict(cfg=kw, shared=False, seed=41, prompt=prompt, steps=170, blob=False, prefill=pf, prefill_split=9)
    run_ranks(tmp_path, world, spec, {"L2Z_SCHEME_B": "1"})
    outs = [np.load(tmp_path / f"out_{r}.npz") for r in range(world)]
    for r in range(1, world):
        for k in ("toks", "logits", "pf_logits"):
            assert np.array_equal(outs[r][k], outs[0][k]), f"rank {r} {k} differs from rank 0"
    toks = outs[0]["toks"]
    assert toks[:150].tolist() == prompt
    w, s = gpu.Weights(cfg, None, False, seed=41), gpu.RunState(cfg)
    options(L2Z_PREFILL=0)                 # the reference steps through: a loop that is all prompt would compute no logits
    s.greedy_begin(toks.tolist())          # the sharded run's tokens forced (main.zig:999-1000)
    s.greedy_run(w, len(toks))
    np.testing.assert_allclose(outs[0]["logits"], s.logits(), rtol=5e-5, atol=5e-5)
    s.prefill(pf[:9], 0, w); s.prefill(pf[9:], 9, w)
    np.testing.assert_allclose(outs[0]["pf_logits"], s.logits(), rtol=5e-5, atol=5e-5)
    print(f"scheme B batched prefill x{world} processes: max |logit - unsharded| {np.abs(outs[0]['pf_logits'] - s.logits()).max():.2e}")
    s.close(); w.close()


def test_landing_slots_too_small_are_refused(gpu, tmp_path):
    kw = MODELS[2][1]  # vocab 32000: half of it rounds up to 16384 words, too few
    spec = dict(cfg=kw, shared=True, seed=1, prompt=[], steps=4, expect="slot_error")
    run_ranks(tmp_path, 2, spec)
    assert (tmp_path / "ok_0").exists() and (tmp_path / "ok_1").exists()


@pytest.mark.parametrize("mode", ["consume", "gather"])
def test_dead_peer_fails_fast(gpu, tmp_path, mode):
    """One timeout, not steps x gathers timeouts: after the first wait gives up, every later wait
    sees the latched error and returns at once (2 s timeout, 64 queued steps x 13 gathers)."""
    kw = MODELS[0][1]
    spec = dict(cfg=kw, shared=False, seed=1, prompt=[], steps=80, expect="peer_dies")
    run_ranks(tmp_path, 2, spec, dict(MODE_ENV[mode], L2Z_P2P_TIMEOUT_S="2"), timeout=120)
    took = float((tmp_path / "ok_0").read_text().split()[0])
    assert took < 20.0, f"rank 0 needed {took:.1f} s to report the dead peer"


@pytest.mark.parametrize("world", [2, 4])
def test_multiprocess_sharded_prefill_through_bulk_regions(gpu, ck, tmp_path, world, options):
    """Row-sharded batched prefill with REAL processes: every rank computes its column block of each
    [tokens, n] activation matrix, pushes it into the peers' bulk regions (plain 16-byte peer stores +
    one flag per sender, csrc/p2p.hip) and unpacks what the peers pushed.  Both callers: the greedy
    loop with a 150-token prompt (prefill, then sharded decode steps on the LL transport), and
    l2z_prefill itself in two calls (pos0 > 0; the second call's 141 tokens take the tile GEMMs, the
    first one's 9 the skinny ones).  Tokens, logits and this rank's KV shard equal the unsharded run's
    bit for bit."""
    options(L2Z_FUSE_SMALL=0)
    kw = dict(dim=512, hidden_dim=1408, n_layers=2, n_heads=16, n_kv_heads=8, vocab_size=1024, seq_len=384)
    cfg = ck.Config(**kw)
    rng = np.random.default_rng(8)
    prompt = rng.integers(2, cfg.vocab_size, 150).tolist()
    pf = [1] + rng.integers(2, cfg.vocab_size, 149).tolist()
    steps = 230
    spec = dict(cfg=kw, shared=False, seed=41, prompt=prompt, steps=steps, blob=False, prefill=pf, prefill_split=9)
    run_ranks(tmp_path, world, spec)
    w, s = gpu.Weights(cfg, None, False, seed=41), gpu.RunState(cfg)
    s.greedy_begin(prompt)
    toks = s.greedy_run(w, steps)
    logits = s.logits()
    s2 = gpu.RunState(cfg)
    s2.prefill(pf[:9], 0, w)
    s2.prefill(pf[9:], 9, w)
    pf_logits = s2.logits()
    kvd = cfg.kv_dim
    key0 = s2.read("key_cache", 0, len(pf) * kvd).reshape(len(pf), kvd)
    assert len(toks) == steps and toks[:150].tolist() == prompt
    for r in range(world):
        o = np.load(tmp_path / f"out_{r}.npz")
        assert np.array_equal(o["toks"], toks), f"rank {r} tokens"
        assert np.array_equal(o["logits"], logits), f"rank {r} logits after the greedy loop"
        assert np.array_equal(o["pf_logits"], pf_logits), f"rank {r} logits after l2z_prefill"
        kvl = kvd // world
        assert np.array_equal(o["pf_key0"].reshape(len(pf), kvl), key0[:, r * kvl:(r + 1) * kvl]), f"rank {r} key cache"
    for o_ in (s, s2, w):
        o_.close()


def test_multiprocess_sharded_prefill_needs_bulk_regions(gpu, ck, tmp_path, options):
    """L2Z_P2P_BULK_MB=0: no bulk regions in the arenas.  l2z_prefill on the sharded runstate is refused
    (L2Z_ERR_INVALID, nothing launched) and the greedy loop steps through its prompt instead -- with
    tokens and logits equal to the unsharded STEPPED loop's, bit for bit."""
    options(L2Z_FUSE_SMALL=0, L2Z_PREFILL=0)
    kw = dict(dim=512, hidden_dim=1408, n_layers=2, n_heads=16, n_kv_heads=8, vocab_size=1024, seq_len=128)
    cfg = ck.Config(**kw)
    prompt = np.random.default_rng(8).integers(2, cfg.vocab_size, 40).tolist()
    spec = dict(cfg=kw, shared=False, seed=41, prompt=prompt, steps=60, blob=False, prefill=[1] + prompt, prefill_split=9,
                expect="no_bulk")
    run_ranks(tmp_path, 2, spec, {"L2Z_P2P_BULK_MB": "0"})
    w, s = gpu.Weights(cfg, None, False, seed=41), gpu.RunState(cfg)
    s.greedy_begin(prompt)
    toks = s.greedy_run(w, 60)
    for r in range(2):
        o = np.load(tmp_path / f"out_{r}.npz")
        assert np.array_equal(o["toks"], toks) and np.array_equal(o["logits"], s.logits()), f"rank {r}"
    s.close(); w.close()


def test_bench_legs_on_one_gpu(gpu, tmp_path):
    """bench.py --gpus 2 with both ranks on the one GPU of this box: every transport runs as its own leg
    (child process per rank), all of them are reported, and the headline is the fastest leg whose ranks
    agree.  RCCL refuses two ranks on one device, so its leg must FAIL here -- and the line must say so
    instead of the run dying with it; both peer-write legs must work.  The fourth leg (persistent launches) is
    refused by this narrow shape and must be reported as not run."""
    import socket
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    root = os.path.dirname(HERE)
    env = dict(os.environ, L2Z_BENCH_LEG_TIMEOUT_S="200", L2Z_P2P_TIMEOUT_S="20", L2Z_BENCH_NO_SHARDED_PREFILL="1")
    p = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                        "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.join(root, "bench.py"),
                        "--gpus", "2", "--steps", "48", "--warmup", "1", "--workload", "stories110M"],
                       env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=900)
    lines = [ln for ln in p.stdout.decode().splitlines() if ln.startswith("{")]
    assert p.returncode == 0 and len(lines) == 1, p.stdout.decode()[-3000:] + p.stderr.decode()[-3000:]
    out = json.loads(lines[0])
    legs = {l["transport"]: l for l in out["comm"]["legs"]}
    assert set(legs) == {"rccl", "p2p-gather", "p2p-consume", "rccl-allreduce", "p2p-allreduce"}
    assert legs["p2p-allreduce"]["ok"] and legs["p2p-allreduce"]["scheme"] == "B" and legs["p2p-gather"]["scheme"] == "A"
    assert legs["p2p-allreduce"]["gathers"] == 2 * 12 + 1 and legs["p2p-gather"]["gathers"] == 4 * 12 + 1
    assert out["comm"]["scheme_b"]["transport"] == "p2p-allreduce" and out["comm"]["scheme_a"]["transport"] in ("p2p-gather", "p2p-consume")
    ok = [t for t, l in legs.items() if l["ok"]]
    assert "p2p-gather" in ok and "p2p-consume" in ok, legs
    for t in ok:
        assert legs[t]["ranks_agree"] and legs[t]["steps"] == 48 and legs[t]["tokens_per_s"] > 0
    # round 6: every peer-write leg carries the cross-device diagnostics measured in its own run (here both ranks sit on
    # one GPU: the numbers are the chip's own fine-grained memory, and say so) and the predicted-vs-measured row
    for t in ("p2p-gather", "p2p-consume", "p2p-allreduce"):
        if t not in ok:
            continue
        cd, pm = legs[t]["cross_device"], legs[t]["predicted_vs_measured"]
        assert 0.0 < cd["ll_word_round_trip_us"]["1"] < 1000.0 and 0.0 < cd["peer_copy_16KB_us"]["1"] < 10000.0, cd
        assert cd["devices"] == [0, 0] and "ONE GPU" in cd["note"]
        assert pm["structure"] == t and pm["tokens_per_s_free_handovers"] > 0 and pm["measured_over_predicted"] > 0, pm
        assert legs[t]["handover_latency_floor_ms_per_token"] > 0
    # round 6: the headline is the fastest agreeing leg of EITHER scheme; the best of each scheme is reported beside it
    assert out["comm"]["transport"] == max(ok, key=lambda t: legs[t]["tokens_per_s"])
    assert out["comm"]["scheme"] == legs[out["comm"]["transport"]]["scheme"]
    assert out["comm"]["scheme_a"]["transport"] == max([t for t in ok if legs[t]["scheme"] == "A"], key=lambda t: legs[t]["tokens_per_s"])
    assert out["value"] == legs[out["comm"]["transport"]]["tokens_per_s"] and out["n_gpus"] == 2
    if not legs["rccl"]["ok"]:
        assert legs["rccl"]["why"], legs["rccl"]
        assert out["comm"]["rccl"]["initialised"] is False
    print({t: (l["ok"], l.get("tokens_per_s"), l.get("why")) for t, l in legs.items()})


@pytest.mark.parametrize("world", [2, 8])
def test_solo_rank_runs_every_structure(gpu, ck, world):
    """l2z_comm_p2p_connect_solo (measurement support, bench.py extra.scaling_model.solo_rank): ONE rank of an N-rank group
    alone, the peers' arenas a local sink, every hand-over's wait satisfied by its own zeroed landing slots.  Every leg's structure
    must run its whole pass that way -- no wait may block, nothing may time out -- and produce finite numbers (they mean
    nothing: the peers' slices read as zeros)."""
    kw = dict(dim=4096, hidden_dim=8192, n_layers=2, n_heads=32, n_kv_heads=8, vocab_size=8192, seq_len=320)
    cfg = ck.Config(**kw)
    forms = [({"L2Z_P2P_CONSUME": 1}, 0), ({"L2Z_P2P_CONSUME": 0}, 0), ({"L2Z_SCHEME_B": 1}, 8)]
    reset = {"L2Z_P2P_CONSUME": -1, "L2Z_SCHEME_B": 0}
    gpu.option_set("L2Z_P2P_TIMEOUT_S", 5)
    try:
        for opts, want in forms:
            for k, v in opts.items():
                gpu.option_set(k, v)
            try:
                comm = gpu.Comm(0, world, None, 0)
                comm.p2p_export(max(cfg.dim, cfg.hidden_dim, cfg.vocab_size, world * cfg.dim), max(cfg.dim, cfg.hidden_dim))
                comm.p2p_connect_solo()
                w = gpu.Weights(cfg, None, False, seed=3, comm=comm)
                s = gpu.RunState(cfg, comm=comm)
            finally:
                for k in opts:
                    gpu.option_set(k, reset[k])
            assert (s.form() & 12) == want, (opts, s.form())
            s.greedy_begin([5, 6])
            toks = s.greedy_run(w, 300)          # past pos 256: the split attention form as well
            assert len(toks) >= 2 and np.isfinite(s.logits()).all(), opts
            s.close(); w.close(); comm.close()
    finally:
        gpu.option_set("L2Z_P2P_TIMEOUT_S", 20)
