#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
{
for n in 64 128; do python scripts/prefill_ab.py llama2-7b $n 5 ""; done
timeout 2400 python -m pytest tests -m gpu -q --deselect tests/test_gpu_perf_gate.py 2>&1 | tail -8
} > gpurun_out/r6_46_full_suite.txt 2>&1
cat gpurun_out/r6_46_full_suite.txt
