import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "."))
import numpy as np, __graft_entry__ as ge
pkg = ge.load_package(); B, ck = pkg.binding, pkg.checkpoint
for name, cfg, shared in [("llama2-7b", ck.LLAMA2_7B, False), ("stories110M", ck.STORIES110M, True), ("stories15M", ck.STORIES15M, True)]:
    w = B.Weights(cfg, None, shared, seed=1); s = B.RunState(cfg)
    rng = np.random.default_rng(1)
    sizes = [int(x) for x in os.environ["PF_SIZES"].split(",")] if os.environ.get("PF_SIZES") else ((16, 64, 256, 512) if cfg.seq_len >= 600 else (16, 64, 250))
    for n in [x for x in sizes if x <= cfg.seq_len - 1]:
        toks = [1] + rng.integers(2, cfg.vocab_size, n - 1).tolist()
        s.prefill(toks, 0, w)                      # warm (allocations)
        t0 = time.perf_counter(); s.prefill(toks, 0, w); dt = time.perf_counter() - t0
        # token by token through the device loop
        if os.environ.get("PF_NO_STEPPED"):
            print(f"{name}: prompt {n:4d} tokens: prefill {dt*1e3:8.2f} ms"); continue
        s.greedy_begin(toks[1:]); s.greedy_run(w, 1); s.synchronize()
        s.greedy_begin(toks[1:]); t1 = time.perf_counter(); s.greedy_run(w, 1); s.greedy_run(w, n - 1); s.synchronize(); dt2 = time.perf_counter() - t1  # first call < prompt: stepped loop
        flops = 2.0 * n * (cfg.n_layers * (2 * cfg.dim * cfg.dim + 2 * cfg.dim * cfg.kv_dim + 3 * cfg.dim * cfg.hidden_dim))
        print(f"{name}: prompt {n:4d} tokens: prefill {dt*1e3:8.2f} ms ({n/dt:9.0f} tok/s, {flops/dt/1e12:6.1f} TFLOP/s) vs one-by-one {dt2*1e3:8.2f} ms  -> {dt2/dt:5.1f}x")
    s.close(); w.close()
