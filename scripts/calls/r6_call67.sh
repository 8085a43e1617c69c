#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
{
for n in 160 192 240 256; do
echo "default (128 x 64 tiles from 257 tokens):"; python scripts/prefill_ab.py llama2-7b $n 3 "L2Z_PF_CHUNK=1024"
for m in 129 193; do echo "128 x 64 tiles from $m tokens:"; L2Z_LIB=$PWD/llama2.zig_amd/exp/libl2z_t$m.so python scripts/prefill_ab.py llama2-7b $n 3 "L2Z_PF_CHUNK=1024"; done
done
} > gpurun_out/r6_67_tile128_small_chunks.txt 2>&1
cat gpurun_out/r6_67_tile128_small_chunks.txt
