import os, sys, ctypes as C
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "."))
os.environ["L2Z_LIB"] = os.path.join(os.environ.get("GRAFT_REPO_ROOT", "."), "scripts/xlib/libllama2_hip.so")
import numpy as np, __graft_entry__ as ge
pkg = ge.load_package(); B, ck = pkg.binding, pkg.checkpoint
L = B.lib()
for name, cfg, shared in [("7B", ck.LLAMA2_7B, False), ("15M", ck.STORIES15M, True)]:
    w = B.Weights(cfg, None, shared, seed=1); s = B.RunState(cfg)
    names = ["issue loads", "q/K arrive+pre", "scores", "barrier", "softmax", "V accumulate", "part+barrier", "final reduce"]
    for pos in (0, 63, 255):
        for p in range(max(0, pos - 2), pos + 1):
            s.transformer(1, p, w)
        s.synchronize()
        ts = (C.c_longlong * 16)(); L.l2z_dbg_ts(ts)
        d = [ts[i + 1] - ts[i] for i in range(7)]
        print(name, "pos", pos, "total cycles", ts[7] - ts[0], " | ".join(f"{n}:{v}" for n, v in zip(names[1:], d)))
    s.close(); w.close()
