"""N greedy decode steps of one workload and nothing else (for rocprofv3 --kernel-trace timelines; mode knobs from the
environment).  usage: decode_steps.py <workload> <steps> [pos0]"""
import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "."))
import __graft_entry__ as ge
pkg = ge.load_package(); B, ck = pkg.binding, pkg.checkpoint
wl, steps = sys.argv[1], int(sys.argv[2])
pos0 = int(sys.argv[3]) if len(sys.argv) > 3 else 0
cfg, shared = {n: (c, sh) for n, c, sh in ck.iter_configs()}[wl]
w = B.Weights(cfg, None, shared, seed=2024)
s = B.RunState(cfg)
B.option_set("L2Z_PREFILL", 0)
s.greedy_begin(list(range(2, 2 + pos0)) if pos0 else [])
if pos0:
    s.greedy_run(w, pos0)
s.greedy_run(w, 4); s.synchronize()
t0 = time.perf_counter()
n = len(s.greedy_run(w, steps)); s.synchronize()
dt = time.perf_counter() - t0
print(f"{wl} pos0={pos0}: {n / dt:.1f} tok/s ({1e3 * dt / n:.4f} ms/token)")
