#!/bin/bash
# round 6, call 12: planes form of the bf16 tile GEMM -- accuracy, A/B timing, the prefill parity tests
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
{
python scripts/x3_accuracy.py llama2-7b 512
python scripts/x3_accuracy.py stories110M 300
python scripts/x3_accuracy.py stories15M 200
for n in 100 128 256 512 1024; do
  python scripts/prefill_ab.py llama2-7b $n 5 "L2Z_PF_X3=0" "L2Z_PF_X3=1"
done
python scripts/prefill_ab.py stories110M 300 5 "L2Z_PF_X3=0" "L2Z_PF_X3=1"
python scripts/prefill_ab.py stories110M 1024 5 "L2Z_PF_X3=0" "L2Z_PF_X3=1"
} > gpurun_out/r6_12_x3.txt 2>&1
tail -25 gpurun_out/r6_12_x3.txt
timeout 1500 python -m pytest tests -m gpu -x -q -k "prefill" > gpurun_out/r6_12_tests.txt 2>&1
tail -15 gpurun_out/r6_12_tests.txt
