#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
{
for n in 176 224 232 320 336 352 368 384 416 448 640 700; do python scripts/prefill_ab.py llama2-7b $n 3 "" "L2Z_PF_CHUNK=1024"; done
} > gpurun_out/r6_62_chunk_plan2.txt 2>&1
cat gpurun_out/r6_62_chunk_plan2.txt
