#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
{
for n in 128 256 512 1024; do
  python scripts/prefill_ab.py llama2-7b $n 5 "L2Z_PF_X3=0" "L2Z_PF_X3_FORM=0" "L2Z_PF_X3_FORM=1" "L2Z_PF_X3_FORM=2"
done
} > gpurun_out/r6_13_x3_forms.txt 2>&1
tail -25 gpurun_out/r6_13_x3_forms.txt
