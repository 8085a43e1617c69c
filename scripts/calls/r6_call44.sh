#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
{
for n in 64 128; do L2Z_LIB=$PWD/llama2.zig_amd/exp/libl2z_x3tl.so timeout 300 python scripts/x3_timeline.py llama2-7b $n; done
} > gpurun_out/r6_44_stream_timeline.txt 2>&1
cat gpurun_out/r6_44_stream_timeline.txt
