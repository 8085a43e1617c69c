"""Random shapes x world sizes with REAL processes on one GPU through the peer-write transport
(tests/p2p_worker.py): tokens and logits of every rank must equal the unsharded run bit for bit.  A third of the cases
run scheme B (L2Z_SCHEME_B=1: Wo / W2 by columns, reduce launches): the ranks must equal EACH OTHER bit for bit and the
unsharded pass over the same tokens within the parity tolerance; a sixth keep a gather launch per vector.
usage: fuzz_p2p.py [n_configs] [seed]"""
import json, os, subprocess, sys, tempfile
ROOT = os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np, __graft_entry__ as ge
pkg = ge.load_package(); B, ck = pkg.binding, pkg.checkpoint
B.option_set("L2Z_FUSE_SMALL", 0)  # the unsharded reference runs the launches the shards run
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
bad = 0
for it in range(int(sys.argv[1]) if len(sys.argv) > 1 else 8):
    world = int(rng.choice([2, 3, 4, 5, 8]))
    hs = int(rng.choice([2, 6, 16, 64, 128]))
    n_kv = world * int(rng.choice([1, 2])); n_heads = n_kv * int(rng.choice([1, 2]))
    dim = hs * n_heads
    kw = dict(dim=dim, hidden_dim=world * int(rng.integers(3, 500)), n_layers=int(rng.integers(1, 3)), n_heads=n_heads,
              n_kv_heads=n_kv, vocab_size=world * int(rng.integers(3, 400)), seq_len=int(rng.choice([48, 300])))
    cfg = ck.Config(**kw); shared = bool(rng.integers(0, 2))
    steps = cfg.seq_len - 2
    mode = str(rng.choice(["consume", "consume", "consume", "gather", "scheme-b", "scheme-b"]))
    mode_env = {"consume": {"L2Z_P2P_CONSUME": "1"}, "gather": {"L2Z_P2P_CONSUME": "0"}, "scheme-b": {"L2Z_SCHEME_B": "1"}}[mode]
    with tempfile.TemporaryDirectory() as d:
        # prompts of 4 tokens and more go through the row-sharded batched prefill (bulk regions of the arenas)
        n_prompt = int(rng.choice([2, 5, 40, min(100, cfg.seq_len - 8)]))
        prompt = rng.integers(2, cfg.vocab_size, n_prompt).tolist()
        spec = dict(cfg=kw, shared=shared, seed=it, prompt=prompt, steps=steps, blob=True)
        json.dump(spec, open(os.path.join(d, "m.json"), "w"))
        procs = [subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "p2p_worker.py"), str(r), str(world), d,
                                   os.path.join(d, "m.json")], stdout=subprocess.PIPE, stderr=subprocess.STDOUT,
                                  env=dict(os.environ, L2Z_P2P_TIMEOUT_S="60", L2Z_GRID_CAP=str(max(32, 512 // world)), **mode_env)) for r in range(world)]
        outs = [p.communicate(timeout=300)[0].decode(errors="replace") for p in procs]
        ok = all(p.returncode == 0 for p in procs)
        if ok and mode == "scheme-b":
            o0 = np.load(os.path.join(d, "out_0.npz"))
            for r in range(1, world):
                o = np.load(os.path.join(d, f"out_{r}.npz"))
                ok = ok and np.array_equal(o["toks"], o0["toks"]) and np.array_equal(o["logits"], o0["logits"])
            # the unsharded pass over the SAME tokens (forced as the prompt), stepped
            B.option_set("L2Z_PREFILL", 0)
            w, s = B.Weights(cfg, ck.synth_blob(cfg, shared, it), shared), B.RunState(cfg)
            s.greedy_begin(o0["toks"].tolist()); s.greedy_run(w, len(o0["toks"]))
            ok = ok and bool(np.allclose(o0["logits"], s.logits(), rtol=5e-5, atol=5e-5))
            s.close(); w.close()
        elif ok:
            # shards whose rows are not multiples of 4 step through their prompt (prefill_usable): the unsharded
            # reference must then step too -- the batched pass equals the stepped one only within the tolerance
            shard_prefills = all(v % 4 == 0 for v in (dim, kw["hidden_dim"], hs, dim // world, kw["hidden_dim"] // world)) and hs <= 256
            B.option_set("L2Z_PREFILL", 1 if shard_prefills else 0)
            w, s = B.Weights(cfg, ck.synth_blob(cfg, shared, it), shared), B.RunState(cfg)
            s.greedy_begin(prompt); toks = s.greedy_run(w, steps); lg = s.logits()
            for r in range(world):
                o = np.load(os.path.join(d, f"out_{r}.npz"))
                ok = ok and np.array_equal(o["toks"], toks) and np.array_equal(o["logits"], lg)
            s.close(); w.close()
        else:
            print("\n".join(x[-400:] for x in outs))
    print(("ok " if ok else "BAD"), mode, "world", world, kw, "shared", int(shared), "prompt", n_prompt); bad += not ok
print("bad:", bad)
