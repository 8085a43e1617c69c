#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
{
for n in 64 100 128; do
echo "W non-temporal (default build):"; python scripts/prefill_ab.py llama2-7b $n 5 ""
echo "W by the default policy (L2Z_X3_EXP=128):"; L2Z_LIB=$PWD/llama2.zig_amd/exp/libl2z_x3e128.so python scripts/prefill_ab.py llama2-7b $n 5 ""
done
} > gpurun_out/r6_36_stream_w_nt.txt 2>&1
cat gpurun_out/r6_36_stream_w_nt.txt
