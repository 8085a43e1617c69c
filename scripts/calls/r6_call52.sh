#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
{
for n in 48 64 100 128; do
echo "a tile's ranges on ids 8 apart (default build):"; python scripts/prefill_ab.py llama2-7b $n 5 ""
echo "on consecutive ids (L2Z_X3_EXP=256):"; L2Z_LIB=$PWD/llama2.zig_amd/exp/libl2z_x3e256.so python scripts/prefill_ab.py llama2-7b $n 5 ""
done
L2Z_LIB=$PWD/llama2.zig_amd/exp/libl2z_x3tl.so timeout 300 python scripts/x3_timeline.py llama2-7b 64 | grep -E "^[qW]|siblings|drained|read and"
timeout 1200 python -m pytest tests -m gpu -q -x -k "prefill and not perf" 2>&1 | tail -3
} > gpurun_out/r6_52_stream_xcd_siblings.txt 2>&1
cat gpurun_out/r6_52_stream_xcd_siblings.txt
