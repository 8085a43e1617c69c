"""Experiment (round 6, review item 5): what would the split attention kernel gain if `pos` came as a kernel argument
(set per replay with hipGraphExecKernelNodeSetParams) instead of a dependent read of device memory?
l2z_time_kind("attn") at the 7B shape, pos 2047 / 1023 / 300, with L2Z_ATTN_POS_ARG 0 / 1 (the same launches back
to back; only the source of pos differs)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as ge
pkg = ge.load_package(); B, ck = pkg.binding, pkg.checkpoint
cfg = ck.LLAMA2_7B
w = B.Weights(cfg, None, False, seed=1); s = B.RunState(cfg)
s.greedy_begin([]); s.greedy_run(w, 4)
for pos in (2047, 1023, 300):
    for rnd in range(2):
        for v in (0, 1):
            B.option_set("L2Z_ATTN_POS_ARG", v)
            best = min(s.time_kind("attn", pos, w, reps=8)[0] for _ in range(5))
            print(f"pos {pos} pos-by-value={v}: {best * 1e3:.2f} us per layer", flush=True)
B.option_set("L2Z_ATTN_POS_ARG", 0)
