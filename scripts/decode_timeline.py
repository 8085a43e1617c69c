"""The decode pass as the GPU's own clock saw it: wall-clock stamps (100 MHz) kept in registers by every block of the duo
mat-vec kernels of the measurement build (scripts/timeline_build.sh, -DL2Z_TIMELINE) and stored when the block is done;
eager launches (the host numbers them).  Per kind of launch, relative to the moment its PRODUCER's last block left the
unit loop -- the earliest its input could have been complete:
  entered / past the hint gate / x staged / first unit done / out of the unit loop   (last block each; first block for entry)
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
import time
s.greedy_run(w, pos0 + 2); s.synchronize()
t0 = time.perf_counter(); n = len(s.greedy_run(w, toks)); s.synchronize(); dt = time.perf_counter() - t0
L = B.lib()
n_max, nb = 2048, 256
buf = (C.c_longlong * (n_max * nb * 8))()
assert L.l2z_timeline_dump(buf, n_max) == 0
t = np.frombuffer(buf, dtype=np.int64).reshape(n_max, nb, 8)
per_tok = 4 * cfg.n_layers + 1
n_launch = per_tok * (pos0 + 2 + toks)
assert n_launch <= n_max, "too many launches for the stamp buffer: fewer tokens"
t = t[n_launch - per_tok * min(toks, 6):n_launch]
kind = {(1, cfg.dim // 4): "qkv", (2, cfg.dim // 4): "wo", (3, cfg.dim // 4): "ffn13", (2, cfg.hidden_dim // 4): "ffn2", (4, cfg.dim // 4): "cls"}
rows = []
for r in t:
    g = int(r[0, 7])                    # blocks of the launch
    r = r[:g]
    k = kind.get((int(r[0, 0] & 0xffffffff) >> 16, int(r[0, 0] & 0xffff)))
    ll = bool(r[0, 0] >> 32)
    rows.append(dict(k=k, ll=ll, entry0=r[:, 1].min(), entry=r[:, 1].max(), gate=r[:, 2].max() if ll else 0, staged=r[:, 3].max(),
                     first=r[:, 4].max(), done=r[:, 5].max(), done0=r[:, 5].min(), acked=r[:, 6].max(), gate0=r[:, 2].min() if ll else 0,
                     stage_span=(r[:, 3] - r[:, 2]).mean() if ll else (r[:, 3] - r[:, 1]).mean(), first_span=(r[:, 4] - r[:, 3]).mean()))
us = lambda a, b: (a - b) / 100.0
acc = {}
for i in range(1, len(rows)):
    c, p = rows[i], rows[i - 1]   # host enqueue order == data-flow order (attention sits between qkv and wo)
    acc.setdefault(c["k"], []).append((us(c["entry0"], p["done"]), us(c["gate0"], p["done"]) if c["ll"] else np.nan, us(c["gate"], p["done"]) if c["ll"] else np.nan,
                                       us(c["staged"], p["done"]), us(c["first"], p["done"]), us(c["done"], p["done"]), us(c["done"], c["staged"]),
                                       us(c["done"], c["done0"]), us(p["acked"], p["done"]) if p["acked"] else np.nan, c["stage_span"] / 100.0, c["first_span"] / 100.0))
mode = " ".join(f"{k}={v}" for k, v in os.environ.items() if k.startswith("L2Z_O") or k == "L2Z_DUO") or "defaults"
print(f"# decode timeline from in-kernel stamps: {wl}, pos0 {pos0}, [{mode}], {n / dt:.1f} tok/s ({1e3 * dt / n:.3f} ms/token) with the stamps, eager launches")
print("\nus after the PRODUCER's last block left its unit loop (producer = the previous mat-vec; for wo that is qkv, with the attention launch in between):\n")
print("| launch | n | first block entered | first block past the hint | last block past the hint | last block staged x | last block did its first unit | last block out of the loop | "
      "staged -> out of loop | spread of the blocks' exits | producer's hand-over stores acknowledged | per block: gate (or entry) -> staged | per block: staged -> first unit |")
print("|---|" + "---:|" * 12)
for k in ("qkv", "wo", "ffn13", "ffn2", "cls"):
    if k in acc:
        a = np.array(acc[k], dtype=float)
        with np.errstate(all="ignore"):
            m = np.nanmean(a, axis=0)
        print(f"| {k} | {len(a)} | " + " | ".join("" if np.isnan(x) else f"{x:.2f}" for x in m) + " |")
tok_us = us(rows[-1]["done"], rows[-1 - per_tok]["done"])
print(f"\nlast token: {tok_us:.1f} us between the classifier's exits\n")
