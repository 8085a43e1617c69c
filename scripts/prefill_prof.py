"""One prefill of N tokens on a given shape (for rocprofv3):  prefill_prof.py [shape] [n]"""
import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "."))
import numpy as np, __graft_entry__ as ge
pkg = ge.load_package(); B, ck = pkg.binding, pkg.checkpoint
shape = sys.argv[1] if len(sys.argv) > 1 else "llama2-7b"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 256
cfg, shared = {k: (c, sh) for k, c, sh in ck.iter_configs()}[shape]
w = B.Weights(cfg, None, shared, seed=1); s = B.RunState(cfg)
toks = [1] + np.random.default_rng(1).integers(2, cfg.vocab_size, n - 1).tolist()
for _ in range(3):
    s.prefill(toks, 0, w)
s.synchronize(); s.close(); w.close()
