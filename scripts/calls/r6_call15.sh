#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
{
echo "default build:"; python scripts/prefill_ab.py llama2-7b 1024 3 ""
for e in 1 2 4 8 16 29 31; do
  echo "L2Z_X3_EXP=$e:"; L2Z_LIB=$PWD/llama2.zig_amd/exp/libl2z_x3e$e.so python scripts/prefill_ab.py llama2-7b 1024 3 ""
done
} > gpurun_out/r6_15_x3_parts.txt 2>&1
cat gpurun_out/r6_15_x3_parts.txt
