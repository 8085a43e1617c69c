#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
{
for n in 40 48 56 65 72 80 88 96 97 112; do
python scripts/prefill_ab.py llama2-7b $n 5 "L2Z_PF_X3_STREAM_MIN=200" "L2Z_PF_X3_STREAM_MIN=17"
done
python scripts/prefill_ab.py stories110M 100 5 "L2Z_PF_X3=0" "L2Z_PF_X3_STREAM_MIN=200" "L2Z_PF_X3_STREAM_MIN=17"
python scripts/prefill_ab.py stories110M 128 5 "L2Z_PF_X3=0" "L2Z_PF_X3_STREAM_MIN=200" "L2Z_PF_X3_STREAM_MIN=17"
python scripts/prefill_ab.py stories110M 256 5 "L2Z_PF_X3=0" ""
python scripts/prefill_ab.py stories42M 200 5 "L2Z_PF_X3=0" ""
python scripts/prefill_ab.py stories15M 200 5 "L2Z_PF_X3=0" ""
} > gpurun_out/r6_23_crossover.txt 2>&1
cat gpurun_out/r6_23_crossover.txt
