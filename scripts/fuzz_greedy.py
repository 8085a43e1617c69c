"""Device greedy loop (batched prompt pass included) vs the stepped host loop on random shapes,
prompt lengths and call patterns.  usage: fuzz_greedy.py [n] [seed]"""
import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, __graft_entry__ as ge
pkg = ge.load_package(); B, ck = pkg.binding, pkg.checkpoint
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0); bad = 0
for it in range(int(sys.argv[1]) if len(sys.argv) > 1 else 30):
    hs = int(rng.choice([4, 8, 16, 48, 64])); n_kv = int(rng.choice([1, 2, 3])); n_heads = n_kv * int(rng.choice([1, 2, 4]))
    seq = int(rng.choice([20, 70, 140, 600]))
    cfg = ck.Config(hs * n_heads, 4 * int(rng.integers(8, 300)), int(rng.integers(1, 3)), n_heads, n_kv, int(rng.choice([50, 700])), seq)
    w = B.Weights(cfg, None, bool(rng.integers(0, 2)), seed=it); s1, s2 = B.RunState(cfg), B.RunState(cfg)
    n_prompt = int(rng.choice([0, 1, 3, 4, 5, 16, 17, 33, 64, 65, 130, seq - 1, seq])); n_prompt = min(n_prompt, seq)
    prompt = rng.integers(2, cfg.vocab_size, n_prompt).tolist()
    if n_prompt > 6 and rng.integers(0, 4) == 0: prompt[int(rng.integers(0, n_prompt))] = 1  # a BOS inside
    n_steps = int(rng.choice([seq, seq + 5, max(1, n_prompt - 1), n_prompt + 3, 7]))
    first = int(rng.choice([n_steps, 1, max(1, n_prompt), 3]))
    s2.greedy_begin(prompt)
    got = s2.greedy_run(w, min(first, n_steps)).tolist()
    if first < n_steps and (not got or got[-1] != 1): got += s2.greedy_run(w, n_steps - first).tolist()
    tok, want, ok = 1, [], True
    for pos in range(min(n_steps, seq)):
        s1.transformer(tok, pos, w)
        if pos < n_prompt: nxt = prompt[pos]
        else:
            nxt = s1.argmax()
            if len(got) > pos and got[pos] != nxt:
                lg = np.sort(s1.logits()); ok = ok and lg[-1] - lg[-2] < 1e-4; break
        want.append(nxt)
        if nxt == 1: break
        tok = nxt
    else:
        ok = ok and len(got) == len(want)
    ok = ok and got[:len(want)] == want
    print(("ok " if ok else "BAD"), cfg, "prompt", n_prompt, "steps", n_steps, "first", first, "got", len(got), "want", len(want)); bad += not ok
    for o in (s1, s2, w): o.close()
print("bad:", bad)
