#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
{
for n in 64 56 49; do python scripts/prefill_ab.py llama2-7b $n 5 ""; done
for n in 48 40 32 24; do python scripts/prefill_ab.py llama2-7b $n 5 "" "L2Z_PF_X3_STREAM_MIN=17"; done
timeout 1200 python -m pytest tests/test_gpu_x3.py -q -x 2>&1 | tail -5
timeout 1200 python -m pytest tests -m gpu -q -x -k "prefill and not perf" 2>&1 | tail -5
bash scripts/pf_prof.sh llama2-7b 64 | head -12
} > gpurun_out/r6_38_stream_16_waves.txt 2>&1
cat gpurun_out/r6_38_stream_16_waves.txt
