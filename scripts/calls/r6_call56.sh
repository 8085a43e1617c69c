#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
{
for n in 48 64 100 128; do python scripts/prefill_ab.py llama2-7b $n 5 "L2Z_PF_FUSE_PLANES=0" ""; done
bash scripts/pf_prof.sh llama2-7b 64 | head -11
timeout 900 python -m pytest tests/test_gpu_x3.py -q -x 2>&1 | tail -3
} > gpurun_out/r6_56_deferred_sums_unrolled.txt 2>&1
cat gpurun_out/r6_56_deferred_sums_unrolled.txt
