#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
{
python scripts/x3_accuracy.py llama2-7b 128
python scripts/x3_accuracy.py llama2-7b 100
python scripts/x3_accuracy.py llama2-7b 64 L2Z_PF_X3_STREAM_MIN=17
python scripts/x3_accuracy.py llama2-7b 33 L2Z_PF_X3_STREAM_MIN=17
python scripts/x3_accuracy.py llama2-7b 20 L2Z_PF_X3_STREAM_MIN=17
for n in 128 100 96 64 48 32 20; do
python scripts/prefill_ab.py llama2-7b $n 5 "L2Z_PF_X3=0" "L2Z_PF_X3_STREAM_MIN=200" "L2Z_PF_X3_STREAM_MIN=17"
done
} > gpurun_out/r6_18_x3_stream.txt 2>&1
cat gpurun_out/r6_18_x3_stream.txt
