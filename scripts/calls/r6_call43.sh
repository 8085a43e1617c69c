#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
{
for n in 72 96 100 128; do python scripts/prefill_ab.py llama2-7b $n 5 ""; done
timeout 1500 python -m pytest tests/test_gpu_fullsize.py -q -x -k "7b_prefill" -s 2>&1 | grep -E "7B prefill|passed|failed|Error" | tail -12
timeout 1200 python -m pytest tests/test_gpu_x3.py -q -x 2>&1 | tail -3
} > gpurun_out/r6_43_stream_cost20.txt 2>&1
cat gpurun_out/r6_43_stream_cost20.txt
