#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
{
for n in 64 128; do
echo "default build (bare barrier in the stream loop):"; python scripts/prefill_ab.py llama2-7b $n 5 ""
for e in 32 64; do
  echo "L2Z_X3_EXP=$e:"; L2Z_LIB=$PWD/llama2.zig_amd/exp/libl2z_x3e$e.so python scripts/prefill_ab.py llama2-7b $n 3 ""
done
done
timeout 900 python -m pytest tests -m gpu -q -x -k "prefill and not perf" 2>&1 | tail -3
} > gpurun_out/r6_33_stream_loads.txt 2>&1
cat gpurun_out/r6_33_stream_loads.txt
