"""Per-kind kernel duration (l2z_time_kind: back to back, one event pair) under knobs that apply at RunState creation
(one RunState per variant).  usage: kind_ab.py <workload> <pos> "K=V,K=V" ...   ("" = defaults)"""
import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "."))
import numpy as np, __graft_entry__ as ge
pkg = ge.load_package(); B, ck = pkg.binding, pkg.checkpoint
wl, pos = sys.argv[1], int(sys.argv[2])
variants = sys.argv[3:] or [""]
cfg, shared = {n: (c, sh) for n, c, sh in ck.iter_configs()}[wl]
w = B.Weights(cfg, None, shared, seed=2024)
DEF = {"L2Z_DUO": 1, "L2Z_OVERLAP": 1, "L2Z_ROW_BLOCKS": 2}
states = []
for v in variants:
    kv = dict(x.split("=") for x in v.split(",") if x)
    for k, val in kv.items(): B.option_set(k, int(val))
    states.append(B.RunState(cfg))
    for k in kv: B.option_set(k, DEF.get(k, 0))
res = {}
for rnd in range(3):
    for v, s in zip(variants, states):
        for kind in ("qkv", "attn", "wo", "ffn13", "ffn2", "cls"):
            ms, n = s.time_kind(kind, pos, w, reps=4)
            res.setdefault(v, {}).setdefault(kind, []).append(ms * 1e3)
for v, d in res.items():
    print(f"{wl} pos {pos} [{v or 'defaults'}]: " + "  ".join(f"{k} {np.median(x):6.2f} us" for k, x in d.items()))
