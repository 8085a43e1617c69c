#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
{
for n in 64 56 49; do python scripts/prefill_ab.py llama2-7b $n 5 ""; done
bash scripts/pf_prof.sh llama2-7b 64 | head -12
timeout 1200 python -m pytest tests -m gpu -q -x -k "prefill and not perf" 2>&1 | tail -3
} > gpurun_out/r6_41_stream_12_waves.txt 2>&1
cat gpurun_out/r6_41_stream_12_waves.txt
