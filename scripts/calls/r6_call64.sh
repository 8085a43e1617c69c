#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
{
for n in 576 704 768 800 832 900 960 1000; do python scripts/prefill_ab.py llama2-7b $n 3 "" "L2Z_PF_CHUNK=1024"; done
} > gpurun_out/r6_64_chunk_plan3.txt 2>&1
cat gpurun_out/r6_64_chunk_plan3.txt
