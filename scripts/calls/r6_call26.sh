#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
{
python scripts/prefill_ab.py llama2-7b 1024 3 ""
python - <<'PY'
import os, sys, time
sys.path.insert(0, "."); sys.path.insert(0, "scripts")
import numpy as np, __graft_entry__ as ge, perf_floor
pkg = ge.load_package(); B, ck = pkg.binding, pkg.checkpoint
cfg = ck.LLAMA2_7B
for seed in (1, 2024, 7):
    w, s = B.Weights(cfg, None, False, seed=seed), B.RunState(cfg)
    print("seed %d: 1024 tokens %.2f ms, 512 tokens %.2f ms" % (seed, perf_floor.prefill_ms(B, ck, w, s, cfg, 1024), perf_floor.prefill_ms(B, ck, w, s, cfg, 512)))
    s.close(); w.close()
PY
rocm-smi --showclocks --showpower 2>/dev/null | head -20
} > gpurun_out/r6_26_state_probe2.txt 2>&1; cat gpurun_out/r6_26_state_probe2.txt
