#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
{
for n in 48 44 40 36 33; do python scripts/prefill_ab.py llama2-7b $n 5 "" "L2Z_PF_X3_STREAM_MIN=33"; done
} > gpurun_out/r6_47_stream_min.txt 2>&1
cat gpurun_out/r6_47_stream_min.txt
