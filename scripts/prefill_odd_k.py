"""What the batched prefill costs on widths that are NOT multiples of 64 (the direct-to-LDS kernels' stage): the register-staged
fallbacks against the nearest width that takes the direct-to-LDS kernels.  stories15M (dim 288 = 4.5 x 64) and stories42M
(hidden_dim 1376 = 21.5 x 64) are such models.   python scripts/prefill_odd_k.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, __graft_entry__ as ge
pkg = ge.load_package(); B, ck = pkg.binding, pkg.checkpoint
cases = [("stories15M   dim 288 hidden 768 ", ck.Config(288, 768, 6, 6, 6, 32000, 1024)),
         ("neighbour    dim 320 hidden 768 ", ck.Config(320, 768, 6, 5, 5, 32000, 1024)),
         ("stories42M   dim 512 hidden 1376", ck.Config(512, 1376, 8, 8, 8, 32000, 1024)),
         ("neighbour    dim 512 hidden 1408", ck.Config(512, 1408, 8, 8, 8, 32000, 1024))]
for name, cfg in cases:
    w = B.Weights(cfg, None, True, seed=3); s = B.RunState(cfg)
    row = []
    for n in (16, 64, 200, 512):
        toks = [1] + np.random.default_rng(1).integers(2, cfg.vocab_size, n - 1).tolist()
        s.prefill(toks, 0, w)
        xs = []
        for _ in range(5):
            t0 = time.perf_counter(); s.prefill(toks, 0, w); xs.append(time.perf_counter() - t0)
        row.append(f"{n} tok {min(xs) * 1e3:6.3f} ms")
    print(name, " | ".join(row), flush=True)
    s.close(); w.close()
