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


@pytest.mark.parametrize("name,kw,shared,world", MODELS, ids=[f"{m[0]}-x{m[3]}" for m in MODELS])
def test_multiprocess_peer_write_gather_is_bit_identical(gpu, ck, tmp_path, name, kw, shared, world):
    cfg = ck.Config(**kw)
    steps = min(cfg.seq_len - 2, 300)
    on_device = cfg.dim >= 4096  # big shapes: seeded weights generated on the device by every rank
    spec = dict(cfg=kw, shared=shared, seed=33, prompt=[5, 9, 11], steps=steps, blob=not on_device)
    (tmp_path / "model.json").write_text(json.dumps(spec))
    env = dict(os.environ, L2Z_P2P_TIMEOUT_S="60")
    procs = [subprocess.Popen([sys.executable, os.path.join(HERE, "p2p_worker.py"), str(r), str(world),
                               str(tmp_path), str(tmp_path / "model.json")], env=env,
                              stdout=subprocess.PIPE, stderr=subprocess.STDOUT) for r in range(world)]
    outs = []
    for p in procs:
        try:
            out, _ = p.communicate(timeout=300)
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            raise
        outs.append(out.decode(errors="replace"))
    for r, p in enumerate(procs):
        assert p.returncode == 0, f"rank {r} failed:\n{outs[r][-2000:]}"
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
