"""timeline of one block of the wq slab launch (debug build only)"""
import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "."))
import numpy as np, __graft_entry__ as ge
pkg = ge.load_package(); B, ck = pkg.binding, pkg.checkpoint
cfg, shared = {k: (c, sh) for k, c, sh in ck.iter_configs()}["llama2-7b"]
w = B.Weights(cfg, None, shared, seed=1); s = B.RunState(cfg)
n = int(sys.argv[1]); toks = [1] + list(range(2, n + 1))
for blk in [int(x) for x in sys.argv[2:]] or [0]:
    B.option_set("L2Z_PF_SLAB_DBG", 64 + 256 * blk)
    if os.environ.get("L2Z_PF_SLAB_NST"): B.option_set("L2Z_PF_SLAB_NST", int(os.environ["L2Z_PF_SLAB_NST"]))
    for _ in range(2): s.prefill(toks, 0, w)
    t = s.read("pf_sk_part", 22000000, 256).view(np.uint32)
    k = int(t[255]); t = t[:k].astype(np.int64); t = (t - t[0]) % (1 << 32)
    print(f"block {blk}: {k} stamps (100 MHz ticks -> us = /100):")
    print("  X issued / per stage [waited, barrier passed, multiplied]:")
    print("  (shader cycles)")
    body = t[1:]
    for i in range(0, len(body) - 1, 3):
        print("  ", " ".join(f"{x:7d}" for x in body[i:i + 3]))
    print("  end", t[-1])
