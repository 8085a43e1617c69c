import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "."))
import numpy as np, __graft_entry__ as ge
pkg = ge.load_package(); B, ck = pkg.binding, pkg.checkpoint
for wl in ("stories15M", "llama2-7b"):
    cfg, shared = {n: (c, sh) for n, c, sh in ck.iter_configs()}[wl]
    w = B.Weights(cfg, None, shared, seed=1)
    for ng in (0, 1):
        B.option_set("L2Z_NO_GRAPH", ng)
        s = B.RunState(cfg)
        s.greedy_begin([]); s.synchronize()
        t0 = time.perf_counter(); s.greedy_run(w, 1); s.synchronize(); t1 = time.perf_counter()
        s.greedy_run(w, 1); s.synchronize(); t2 = time.perf_counter()
        print(f"{wl} no_graph={ng}: first step {1e3*(t1-t0):.1f} ms, second {1e3*(t2-t1):.2f} ms")
        s.close()
    B.option_set("L2Z_NO_GRAPH", 0)
    w.close()
