"""Decode attention time per layer vs position (7B shape), HIP events in situ, for each split setting."""
import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "."))
import numpy as np, __graft_entry__ as ge
pkg = ge.load_package(); B, ck = pkg.binding, pkg.checkpoint
cfg = ck.LLAMA2_7B
w = B.Weights(cfg, None, False, seed=1)
res = {}
for split in os.environ.get("SPLITS", "-1,0,16").split(","):
    B.option_set("L2Z_ATTN_SPLIT", int(split))
    s = B.RunState(cfg)
    for pos in (0, 63, 255, 256, 511, 1023, 2047):
        tot, wo = 0.0, 0.0
        for rep in range(3):
            r = s.profile_forward(1, pos, w)
            tot += r["attn"][0] / r["attn"][1]
            wo += r["wo"][0] / r["wo"][1]
        res.setdefault(pos, {})[split] = (tot / 3 * 1e3, wo / 3 * 1e3)
    s.close()
for pos, d in res.items():
    print(f"pos {pos:5d}: " + "  ".join(f"split={k}: attn {v[0]:6.1f} us wo {v[1]:5.1f} us" for k, v in d.items()))
