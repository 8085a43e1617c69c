#!/bin/bash
# round 6, call 11: bf16 three-term split in the tile GEMM -- accuracy against float64 and A/B timing
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
{
python scripts/x3_accuracy.py llama2-7b 128
python scripts/x3_accuracy.py llama2-7b 512
python scripts/x3_accuracy.py stories110M 300
for n in 100 128 256 512 1024; do
  python scripts/prefill_ab.py llama2-7b $n 5 "L2Z_PF_X3=0" "L2Z_PF_X3=1"
done
python scripts/prefill_ab.py stories110M 300 5 "L2Z_PF_X3=0" "L2Z_PF_X3=1"
python scripts/prefill_ab.py stories110M 1024 5 "L2Z_PF_X3=0" "L2Z_PF_X3=1"
} > gpurun_out/r6_11_x3.txt 2>&1
tail -40 gpurun_out/r6_11_x3.txt
