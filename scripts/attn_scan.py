import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "."))
import numpy as np, __graft_entry__ as ge
pkg = ge.load_package(); B, ck = pkg.binding, pkg.checkpoint
cfg = ck.LLAMA2_7B
w = B.Weights(cfg, None, False, seed=1)
res = {}
for split in os.environ.get("SPLITS", "0,8,16").split(","):
    os.environ["L2Z_ATTN_SPLIT"] = split
    s = B.RunState(cfg)
    for pos in (0, 31, 63, 127, 255, 383, 511, 767, 1023, 1535, 2047):
        tot = 0.0
        for rep in range(3):
            r = s.profile_forward(1, pos, w)
            tot += r["attn"][0] / r["attn"][1]
        res.setdefault(pos, {})[split] = tot / 3 * 1e3
    s.close()
for pos, d in res.items():
    print(f"pos {pos:5d}: " + "  ".join(f"split={k}: {v:6.1f} us" for k, v in d.items()))
