"""Interleaved A/B of tuning knobs on one workload, in ONE process: each variant gets its own
RunState (options are applied at creation), the variants take turns for `rounds` rounds of `steps`
greedy steps, and the median tokens/s per variant is printed.  Single bench.py runs differ by +-3 %
from one process to the next on this box (clocks / power); interleaving takes that out.

usage: ab.py <workload> <steps> <rounds> [pos0] "NAME=V,NAME=V" "NAME=V" ...   ("" = defaults)"""
import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "."))
import numpy as np, __graft_entry__ as ge
pkg = ge.load_package(); B, ck = pkg.binding, pkg.checkpoint
wl, steps, rounds = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
rest = sys.argv[4:]
pos0 = 0
if rest and rest[0].isdigit():
    pos0, rest = int(rest[0]), rest[1:]
variants = rest or [""]
shapes = {n: (c, sh) for n, c, sh in ck.iter_configs()}
cfg, shared = shapes[wl]
w = B.Weights(cfg, None, shared, seed=2024)
DEFAULTS = {}
states = []
for v in variants:
    kv = dict(x.split("=") for x in v.split(",") if x)
    for k, val in kv.items():
        B.option_set(k, int(val))
    s = B.RunState(cfg)
    # graphs are captured at the first run, with the options in force THEN (launch-time knobs such as
    # the attention split are read while the launches are enqueued): capture both graph kinds now
    s.greedy_begin([]); s.greedy_run(w, 2); s.transformer(1, 0, w); s.synchronize()
    states.append(s)
    for k in kv:  # back to the default for the next variant (options apply at RunState creation)
        B.option_set(k, {"L2Z_ATTN_SPLIT": -1, "L2Z_ATTN_SPLIT_POS": -1, "L2Z_FUSE_SMALL": 1, "L2Z_NO_GRAPH": 0, "L2Z_P2P_CONSUME": -1,
                         "L2Z_ARGMAX_XCHG": 1}.get(k, 0))
# the variants' arithmetic side by side: logits of one pass at pos0 and the first greedy tokens, against variant 0
ref_logits = ref_toks = None
for v, s in zip(variants, states):
    B.option_set("L2Z_PREFILL", 0)
    s.greedy_begin(list(range(2, 2 + pos0)) if pos0 else [])
    toks = np.array(s.greedy_run(w, pos0 + 24))
    s.transformer(7, pos0, w); lg = s.logits().copy()
    if ref_logits is None:
        ref_logits, ref_toks = lg, toks
    print(f"  [{v or 'defaults'}] logits == variant 0: {bool(np.array_equal(lg, ref_logits))} (max |d| {float(np.abs(lg - ref_logits).max()):.3g}), "
          f"tokens == variant 0: {bool(np.array_equal(toks, ref_toks))}", flush=True)
res = [[] for _ in variants]
for r in range(rounds + 1):
    for i, s in enumerate(states):
        # start at pos0: the prompt positions are forced tokens, stepped one by one (no prefill)
        B.option_set("L2Z_PREFILL", 0)
        s.greedy_begin(list(range(2, 2 + pos0)) if pos0 else [])
        if pos0:
            s.greedy_run(w, pos0)
        s.synchronize()
        t0 = time.perf_counter()
        n = len(s.greedy_run(w, steps))
        s.synchronize()
        dt = time.perf_counter() - t0
        if r > 0:
            res[i].append(n / dt)
for v, xs in zip(variants, res):
    xs = np.array(xs)
    print(f"{wl} pos0={pos0} [{v or 'defaults'}]: median {np.median(xs):9.1f} tok/s  min {xs.min():9.1f}  max {xs.max():9.1f}  "
          f"({1e3 / np.median(xs):.4f} ms/token)")
