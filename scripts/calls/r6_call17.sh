#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
{
for n in 128 512; do
echo "default build:"; python scripts/prefill_ab.py llama2-7b $n 3 ""
for e in 1 2 29 31; do
  echo "L2Z_X3_EXP=$e:"; L2Z_LIB=$PWD/llama2.zig_amd/exp/libl2z_x3e$e.so python scripts/prefill_ab.py llama2-7b $n 3 ""
done
done
} > gpurun_out/r6_17_x3_parts.txt 2>&1
cat gpurun_out/r6_17_x3_parts.txt
