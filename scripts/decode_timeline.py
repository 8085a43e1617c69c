"""The decode pass as the GPU's own clock saw it: wall-clock stamps written by the duo mat-vec kernels of the measurement
build (scripts/timeline_build.sh, -DL2Z_TIMELINE), eager launches.  Per kind of launch: when the last block was past the
hint gate, had x staged, finished its first unit, left the unit loop, had its hand-over stores acknowledged -- relative to
the moment its PRODUCER's last block left the unit loop (the earliest the input could have been complete).
usage: L2Z_LIB=llama2.zig_amd/libllama2_hip_tl.so L2Z_NO_GRAPH=1 [mode knobs] decode_timeline.py <workload> <tokens> [pos0]"""
import os, sys, ctypes as C
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "."))
import numpy as np, __graft_entry__ as ge
pkg = ge.load_package(); B, ck = pkg.binding, pkg.checkpoint
wl, toks = sys.argv[1], int(sys.argv[2])
pos0 = int(sys.argv[3]) if len(sys.argv) > 3 else 0
cfg, shared = {n: (c, sh) for n, c, sh in ck.iter_configs()}[wl]
w = B.Weights(cfg, None, shared, seed=2024)
s = B.RunState(cfg)
B.option_set("L2Z_PREFILL", 0)
s.greedy_begin(list(range(2, 2 + pos0)) if pos0 else [])
s.greedy_run(w, pos0 + toks); s.synchronize()
L = B.lib()
n_max = 16384
buf = (C.c_longlong * (n_max * 8))()
assert L.l2z_timeline_dump(buf, n_max) == 0
t = np.frombuffer(buf, dtype=np.int64).reshape(n_max, 8)
t = t[t[:, 1] != 0]
per_tok = 4 * cfg.n_layers + 1
t = t[-per_tok * min(toks, 6):]                 # the last tokens
kind = {(1, cfg.dim // 4): "qkv", (2, cfg.dim // 4): "wo", (3, cfg.dim // 4): "ffn13", (2, cfg.hidden_dim // 4): "ffn2", (4, cfg.dim // 4): "cls"}
rows = []
for r in t:
    k = kind.get((int(r[0] & 0xffffffff) >> 16, int(r[0] & 0xffff)))
    rows.append((k, bool(r[0] >> 32), r[1], r[2], r[3], r[4], r[5], r[6], (1 << 62) - r[7]))
# host enqueue order == data-flow order: the producer of a launch is the previous mat-vec (attention sits between qkv and wo)
us = lambda a, b: (a - b) / 100.0
acc = {}
for i in range(1, len(rows)):
    k, ll, entry, gate, staged, first, done, acked, first_done = rows[i]
    pk, _, pentry, _, _, _, pdone, packed, pfirst_done = rows[i - 1]
    d = acc.setdefault(k, [])
    d.append((us(entry, pdone), us(gate, pdone) if ll and gate else np.nan, us(staged, pdone), us(first, pdone), us(done, pdone),
              us(done, staged), us(acked, done) if acked else np.nan, us(done, first_done), us(packed, pdone) if packed else np.nan))
print(f"# decode timeline from in-kernel stamps: {wl}, pos0 {pos0}, {os.environ.get('L2Z_OVERLAP_EDGES', '')} overlap={os.environ.get('L2Z_OVERLAP', '1')}")
print("\nus relative to the moment the PRODUCER's last block left its unit loop (producer = the previous mat-vec; for wo that is qkv, with the attention launch in between):\n")
print("| launch | n | block 0 entered | last block past the hint | last block staged x | last block did its first unit | last block out of the loop | staged -> out of loop | own stores acknowledged after exit | spread of the blocks' exits | producer's stores acknowledged after its exit |")
print("|---|---:|---:|---:|---:|---:|---:|---:|---:|---:|---:|")
for k in ("qkv", "wo", "ffn13", "ffn2", "cls"):
    if k in acc:
        a = np.array(acc[k], dtype=float)
        m = np.nanmean(a, axis=0)
        print(f"| {k} | {len(a)} | " + " | ".join(f"{x:.2f}" for x in m) + " |")
tok_us = us(rows[-1][6], rows[-1 - per_tok][6])
print(f"\nlast token: {tok_us:.1f} us between the classifier's exits")
