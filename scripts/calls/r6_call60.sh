#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
{
for n in 32 28 24 17; do python scripts/prefill_ab.py llama2-7b $n 5 "" "L2Z_PF_X3_STREAM_MIN=17"; done
} > gpurun_out/r6_60_stream_min_after_deferral.txt 2>&1
cat gpurun_out/r6_60_stream_min_after_deferral.txt
