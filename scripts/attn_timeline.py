"""The split decode-attention kernel at a long context as the GPU's own clock saw it: wall-clock stamps (100 MHz) kept in
registers by every block (measurement build: scripts/timeline_build.sh, -DL2Z_TIMELINE) and stored at the block's end.
Eager launches.  Per launch, relative to the FIRST block's entry: when the blocks entered, had their K / V loads issued,
had the scores, the weights, the reduced partial, had drained it and counted their arrival, and when the last arriver of
each head had combined -- the kernel's critical path, phase by phase.
usage: L2Z_LIB=llama2.zig_amd/libllama2_hip_tl.so L2Z_NO_GRAPH=1 attn_timeline.py <workload> <pos> [passes]"""
import os, sys, ctypes as C
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "."))
import numpy as np, __graft_entry__ as ge
pkg = ge.load_package(); B, ck = pkg.binding, pkg.checkpoint
wl, pos = sys.argv[1], int(sys.argv[2])
passes = int(sys.argv[3]) if len(sys.argv) > 3 else 4
cfg, shared = {n: (c, sh) for n, c, sh in ck.iter_configs()}[wl]
w = B.Weights(cfg, None, shared, seed=2024)
s = B.RunState(cfg)
for _ in range(passes):
    s.transformer(1, pos, w)
s.synchronize()
ms, n = s.time_kind("attn", pos, w, reps=4)
L = B.lib()
n_max, nb = 512, 512
buf = (C.c_longlong * (n_max * nb * 8))()
assert L.l2z_attn_timeline_dump(buf, n_max) == 0
t = np.frombuffer(buf, dtype=np.int64).reshape(n_max, nb, 8)
n_launch = passes * cfg.n_layers
t = t[n_launch - 2 * cfg.n_layers:n_launch]          # the last two passes
names = ["entered", "K / V loads issued", "scores in LDS", "weights in LDS", "partial reduced", "drained + arrival counted", "combined (last arriver)"]
rows = []
for r in t:
    nblk = int((r[:, 7] > 0).sum())
    r = r[:nblk].astype(np.float64)
    t0 = r[:, 0].min()
    last = r[:, 7] == 2
    rel = (r[:, :7] - t0) / 100.0
    row = [nblk]
    for i in range(6):
        row += [rel[:, i].mean(), rel[:, i].max()]
    row += [rel[last, 6].mean(), rel[last, 6].max()]
    # per block durations of the phases
    d = np.diff(r[:, :6], axis=1) / 100.0
    row += list(d.mean(axis=0)) + [((r[last, 6] - r[last, 5]) / 100.0).mean()]
    rows.append(row)
a = np.array(rows)
m = a.mean(axis=0)
print(f"# split attention timeline from in-kernel stamps: {wl}, pos {pos}, {int(m[0])} blocks per launch, {len(a)} launches averaged; back-to-back kernel time (l2z_time_kind) {ms * 1e3:.2f} us")
print("\nus after the first block's entry (mean over blocks | last block):\n")
print("| stamp | mean | last block |\n|---|---:|---:|")
for i, nm in enumerate(names):
    print(f"| {nm} | {m[1 + 2 * i]:.2f} | {m[2 + 2 * i]:.2f} |")
print("\nper block, us spent between consecutive stamps (mean over blocks):\n")
print("| entry -> loads issued | -> scores | -> weights | -> partial reduced | -> drained + counted | last arriver: -> combined |\n|" + "---:|" * 6)
print("| " + " | ".join(f"{x:.2f}" for x in m[15:21]) + " |")
