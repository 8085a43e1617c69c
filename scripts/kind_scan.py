"""Per-kind kernel duration (l2z_time_kind: back to back, one event pair) under launch-time knobs.
usage: kind_scan.py <workload> "K=V,K=V" ...   ("" = defaults)"""
import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "."))
import numpy as np, __graft_entry__ as ge
pkg = ge.load_package(); B, ck = pkg.binding, pkg.checkpoint
wl = sys.argv[1]
variants = sys.argv[2:] or [""]
cfg, shared = {n: (c, sh) for n, c, sh in ck.iter_configs()}[wl]
w = B.Weights(cfg, None, shared, seed=2024)
DEF = {"L2Z_ATTN_SPLIT": -1, "L2Z_ATTN_SPLIT_POS": -1, "L2Z_FUSE_SMALL": 1}   # (round 6: the mat-vec launch knobs this scanned are constants now)
s = B.RunState(cfg)
res = {}
for rnd in range(3):
    for v in variants:
        kv = dict(x.split("=") for x in v.split(",") if x)
        for k, val in kv.items(): B.option_set(k, int(val))
        for kind in ("qkv", "wo", "ffn13", "ffn2", "cls"):
            ms, n = s.time_kind(kind, 8, w, reps=4)
            res.setdefault(v, {}).setdefault(kind, []).append(ms * 1e3)
        for k in kv: B.option_set(k, DEF[k])
for v, d in res.items():
    print(f"{wl} [{v or 'defaults'}]: " + "  ".join(f"{k} {np.median(x):6.2f} us" for k, x in d.items()))
