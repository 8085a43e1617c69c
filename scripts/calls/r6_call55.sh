#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
{
for n in 48 64 100 128; do python scripts/prefill_ab.py llama2-7b $n 5 "L2Z_PF_FUSE_PLANES=0" ""; done
timeout 900 python -m pytest tests/test_gpu_x3.py -q -x 2>&1 | tail -3
timeout 1500 python -m pytest tests/test_gpu_fullsize.py -q -x -k "7b_prefill" -s 2>&1 | grep -E "7B prefill|passed|failed|Error" | tail -12
timeout 1200 python -m pytest tests -m gpu -q -x -k "prefill and not perf and not 7b" 2>&1 | tail -3
bash scripts/pf_prof.sh llama2-7b 64 | head -11
} > gpurun_out/r6_55_deferred_sums.txt 2>&1
cat gpurun_out/r6_55_deferred_sums.txt
