"""Where a persistent decode launch (engine.hip) spends its time, by the GPU's own clock: stamps kept by block-thread 0
(streaming waves) and the gatherer's lane 0 of the measurement build (scripts/timeline_build.sh), eager launches.
usage: L2Z_LIB=llama2.zig_amd/libllama2_hip_tl.so L2Z_NO_GRAPH=1 L2Z_ENGINE=1 engine_timeline.py <tokens>"""
import os, sys, time, ctypes as C
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "."))
import numpy as np, __graft_entry__ as ge
pkg = ge.load_package(); B, ck = pkg.binding, pkg.checkpoint
toks = int(sys.argv[1]) if len(sys.argv) > 1 else 4
cfg, shared = {n: (c, sh) for n, c, sh in ck.iter_configs()}["llama2-7b"]
w = B.Weights(cfg, None, shared, seed=2024); s = B.RunState(cfg)
B.option_set("L2Z_PREFILL", 0)
s.greedy_begin([]); s.greedy_run(w, 2); s.synchronize()
t0 = time.perf_counter(); n = len(s.greedy_run(w, toks)); s.synchronize(); dt = time.perf_counter() - t0
L = B.lib(); nmax, nb, ns = 1024, 256, 32
buf = (C.c_longlong * (nmax * nb * ns))()
assert L.l2z_engine_timeline_dump(buf, nmax) == 0
t = np.frombuffer(buf, dtype=np.int64).reshape(nmax, nb, ns).astype(np.float64) / 100.0   # us
per_tok = cfg.n_layers + 1
first = per_tok * 2 + per_tok * (toks - 1) + 1       # a middle-layer launch of the last token onwards
rows = t[first:first + cfg.n_layers - 2]
print(f"# engine timeline: llama2-7b, {n / dt:.1f} tok/s ({1e3 * dt / n:.3f} ms/token) with the stamps, eager; {len(rows)} launches [wo, w1|w3, w2, q|k|v], mean over launches")
names = ["wo", "w1|w3", "w2", "q|k|v"]
e0 = rows[:, :, 0]
print("\nus after the launch's first block entered (mean over launches of: first block / median block / last block):\n")
print("| mat-vec | gatherer past the gate | x staged | streaming waves began to wait | x ready (their view) | their last unit done | outputs published | waited for x (median block) | streamed (median block) |")
print("|---|---:|---:|---:|---:|---:|---:|---:|---:|")
base = e0.min(axis=1, keepdims=True)
def f(a):
    a = a - base
    return f"{a.min(axis=1).mean():.1f} / {np.median(a, axis=1).mean():.1f} / {a.max(axis=1).mean():.1f}"
for k in range(4):
    g_gate, g_staged, g_pub = rows[:, :, 4 + 6 * k], rows[:, :, 5 + 6 * k], rows[:, :, 6 + 6 * k]
    s_wait, s_ready, s_end = rows[:, :, 1 + 6 * k], rows[:, :, 2 + 6 * k], rows[:, :, 3 + 6 * k]
    print(f"| {names[k]} | {f(g_gate)} | {f(g_staged)} | {f(s_wait)} | {f(s_ready)} | {f(s_end)} | {f(g_pub)} | {np.median(s_ready - s_wait, axis=1).mean():.1f} | {np.median(s_end - s_ready, axis=1).mean():.1f} |")
end = rows[:, :, 3 + 6 * 3]
print(f"\nlaunch: first block entered -> last block's last unit {(end.max(axis=1) - e0.min(axis=1)).mean():.1f} us; entry spread of the blocks {(e0.max(axis=1) - e0.min(axis=1)).mean():.1f} us")
