import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "."))
import numpy as np, __graft_entry__ as ge
pkg = ge.load_package(); B, ck = pkg.binding, pkg.checkpoint; orc = ge.load_oracle()
rng = np.random.default_rng(5); bad = 0
for it in range(10):
    hs = int(rng.choice([4, 16, 48, 64, 128, 256])); n_kv = int(rng.choice([1, 2, 4])); n_heads = n_kv * int(rng.choice([1, 2, 8]))
    seq = int(rng.choice([1024, 2048, 4096])); dim = hs * n_heads
    cfg = ck.Config(dim, 2 * dim, 1, n_heads, n_kv, 64, seq)
    blob = ck.synth_blob(cfg, True, seed=it)
    w = B.Weights(cfg, blob, True); s = B.RunState(cfg); m = orc.Model(cfg.as_i32(), blob, True)
    toks = rng.integers(0, 64, seq).tolist(); worst = 0.0
    for pos, t in enumerate(toks):
        ref = m.transformer(t, pos); s.transformer(t, pos, w)
        if pos % 257 == 0 or pos >= seq - 2:
            worst = max(worst, float(np.abs(s.logits() - ref).max() / (1e-3 + np.abs(ref).max())))
    # prefill of the whole context vs stepped
    pf = "-"
    if hs % 4 == 0:
        s2 = B.RunState(cfg); s2.prefill(toks, 0, w)
        pf = float(np.abs(s2.logits() - s.logits()).max() / (1e-3 + np.abs(s.logits()).max())); s2.close()
    ok = worst < 2e-4 and (pf == "-" or pf < 2e-4); bad += not ok
    print(("ok " if ok else "BAD"), cfg, f"rel {worst:.1e} prefill {pf}")
    s.close(); w.close(); m.close()
print("bad:", bad)
