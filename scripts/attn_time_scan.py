"""Decode attention duration per layer (l2z_time_kind: every layer's launch back to back between one event
pair -- comparable with rocprofv3) vs position, on any workload.  usage: attn_time_scan.py <workload> pos pos ..."""
import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "."))
import numpy as np, __graft_entry__ as ge
pkg = ge.load_package(); B, ck = pkg.binding, pkg.checkpoint
wl = sys.argv[1]
cfg, shared = {n: (c, sh) for n, c, sh in ck.iter_configs()}[wl]
w = B.Weights(cfg, None, shared, seed=2024)
s = B.RunState(cfg)
out = []
for pos in [int(p) for p in sys.argv[2:]]:
    if pos >= cfg.seq_len:
        continue
    xs = [s.time_kind("attn", pos, w, reps=4)[0] * 1e3 for _ in range(3)]
    kv_mb = 8 * (pos + 1) * cfg.kv_dim / 1e6
    out.append(f"pos {pos:5d}: {np.median(xs):6.2f} us  ({kv_mb:6.2f} MB of K/V rows -> {kv_mb / np.median(xs) * 1e3 / 1e3:5.2f} TB/s)")
print(f"{wl} attention per layer, back to back:\n  " + "\n  ".join(out))
