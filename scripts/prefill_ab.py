"""Interleaved A/B of prefill knobs on one shape: prefill_ab.py <shape> <n_tokens> <rounds> "K=V,K=V" ..."""
import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "."))
import numpy as np, __graft_entry__ as ge
pkg = ge.load_package(); B, ck = pkg.binding, pkg.checkpoint
shape, n, rounds = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
variants = sys.argv[4:] or [""]
cfg, shared = {k: (c, sh) for k, c, sh in ck.iter_configs()}[shape]
w = B.Weights(cfg, None, shared, seed=1); s = B.RunState(cfg)
toks = [1] + np.random.default_rng(1).integers(2, cfg.vocab_size, n - 1).tolist()
DEF = {"L2Z_PF_FUSE_PLANES": 1, "L2Z_PF_CHUNK": 0, "L2Z_PF_PANEL": 1, "L2Z_PF_PANEL_MAX": -1, "L2Z_PREFILL": 1, "L2Z_PF_X3": 1, "L2Z_PF_X3_STREAM_MIN": 33}   # (round 6: the other prefill knobs are gone)
res = [[] for _ in variants]
logits = [None for _ in variants]
flops = 2.0 * n * (cfg.n_layers * (2 * cfg.dim * cfg.dim + 2 * cfg.dim * cfg.kv_dim + 3 * cfg.dim * cfg.hidden_dim))
for r in range(rounds + 1):
    for i, v in enumerate(variants):
        kv = dict(x.split("=") for x in v.split(",") if x)
        for k, val in kv.items(): B.option_set(k, int(val))
        t0 = time.perf_counter(); s.prefill(toks, 0, w); dt = time.perf_counter() - t0
        for k in kv: B.option_set(k, DEF[k])
        if r > 0: res[i].append(dt)
        logits[i] = s.logits()
for i, (v, xs) in enumerate(zip(variants, res)):
    m = float(np.median(xs))
    same = "" if i == 0 else ("  logits == first variant's: %s" % np.array_equal(logits[i], logits[0]))
    print(f"{shape} prefill {n} [{v or 'defaults'}]: median {m*1e3:8.2f} ms  min {min(xs)*1e3:8.2f}  = {flops/m/1e12:6.1f} TFLOP/s ({flops/m/1e12/157.3:.3f} of the f32 MFMA peak){same}")
