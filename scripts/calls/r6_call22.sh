#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
{
python scripts/x3_accuracy.py llama2-7b 128
for n in 128 100 64 32; do
python scripts/prefill_ab.py llama2-7b $n 5 "L2Z_PF_X3_STREAM_MIN=200" "L2Z_PF_X3_STREAM_MIN=17"
done
} > gpurun_out/r6_22_x3_stream.txt 2>&1
cat gpurun_out/r6_22_x3_stream.txt
L2Z_PF_X3_STREAM_MIN=17 bash scripts/pf_prof.sh llama2-7b 128 > gpurun_out/r6_22_prefill128_stream_kernels.md 2>&1
head -12 gpurun_out/r6_22_prefill128_stream_kernels.md
L2Z_PF_X3_STREAM_MIN=17 bash scripts/pf_prof.sh llama2-7b 64 > gpurun_out/r6_22_prefill64_stream_kernels.md 2>&1
head -12 gpurun_out/r6_22_prefill64_stream_kernels.md
