"""Long runs of the opt-in decode forms against the default chain (a race in the persistent launches' LDS hand-overs or in the
epochs of the self-pushed words would be rare, not systematic): greedy tokens over whole contexts, several seeds, every
form; all must equal the chain's.  usage: soak_forms.py [seeds] [full7b]"""
import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "."))
import numpy as np, __graft_entry__ as ge
pkg = ge.load_package(); B, ck = pkg.binding, pkg.checkpoint
n_seeds = int(sys.argv[1]) if len(sys.argv) > 1 else 3
shapes = [("7B-width x4 layers, ctx 2048", ck.Config(dim=4096, hidden_dim=11008, n_layers=4, n_heads=32, n_kv_heads=32, vocab_size=32000, seq_len=2048), False)]
if len(sys.argv) > 2:
    cfg7, sh7 = {n: (c, s) for n, c, s in ck.iter_configs()}["llama2-7b"]
    shapes.append(("llama2-7b", cfg7, sh7))
FORMS = {"chain": {}, "duo": {"L2Z_DUO": 1, "L2Z_OVERLAP": 0}, "two chains": {"L2Z_DUO": 1, "L2Z_OVERLAP": 1}, "engine": {"L2Z_ENGINE": 1}}
RESET = {"L2Z_DUO": 0, "L2Z_OVERLAP": 1, "L2Z_ENGINE": 0}
B.option_set("L2Z_PREFILL", 0)
bad = 0
for name, cfg, shared in shapes:
    for seed in range(n_seeds):
        w = B.Weights(cfg, None, shared, seed=900 + seed)
        ref = duo_ref = None
        for form, opts in FORMS.items():
            for k, v in opts.items(): B.option_set(k, v)
            s = B.RunState(cfg)
            for k in opts: B.option_set(k, RESET[k])
            s.greedy_begin([])
            toks = s.greedy_run(w, cfg.seq_len)
            lg = s.logits()
            s.close()
            if ref is None:
                ref = (toks, lg)
                print(f"{name} seed {seed}: chain produced {len(toks)} tokens")
                continue
            # the duo forms take other attention blocks from pos 128 on (same mathematics, other bits): tokens can part there
            n_cmp = len(ref[0]) if form == "engine" else min(128, len(ref[0]))
            same = np.array_equal(toks[:n_cmp], ref[0][:n_cmp]) and (form != "engine" or np.array_equal(lg, ref[1]))
            first = next((i for i in range(min(len(toks), len(ref[0]))) if toks[i] != ref[0][i]), None)
            print(f"  {'ok ' if same else 'BAD'} {form}: {n_cmp} tokens compared" + ("" if first is None else f", first difference at {first}"))
            bad += not same
            if form == "duo":
                duo_ref = (toks, lg)
            if form == "two chains":  # the same kernels and attention forms as the one-chain duo pass: the whole run, bit for bit
                same2 = np.array_equal(toks, duo_ref[0]) and np.array_equal(lg, duo_ref[1])
                print(f"  {'ok ' if same2 else 'BAD'} two chains vs duo: {len(toks)} tokens + final logits")
                bad += not same2
        w.close()
print("bad:", bad)
