#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
{
timeout 300 python scripts/x3_accuracy.py llama2-7b 128 L2Z_PF_X3_EXP=1
for n in 128 100 64 49 32; do
timeout 300 python scripts/prefill_ab.py llama2-7b $n 5 "" "L2Z_PF_X3_EXP=1,L2Z_PF_X3_STREAM_MIN=17"
done
} > gpurun_out/r6_31_stream_ks1.txt 2>&1
cat gpurun_out/r6_31_stream_ks1.txt
