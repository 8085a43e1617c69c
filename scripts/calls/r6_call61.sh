#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
{
for n in 136 144 160 176 192 208 224 240 272 288 320; do python scripts/prefill_ab.py llama2-7b $n 3 "" "L2Z_PF_CHUNK=1024" "L2Z_PF_CHUNK=128"; done
} > gpurun_out/r6_61_chunk_plan.txt 2>&1
cat gpurun_out/r6_61_chunk_plan.txt
