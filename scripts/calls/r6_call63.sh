#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
{
for n in 368 384 200; do python scripts/prefill_ab.py llama2-7b $n 3 "" "L2Z_PF_CHUNK=1024"; done
timeout 1500 python -m pytest tests -m gpu -q -x -k "prefill and not perf" 2>&1 | tail -3
( timeout 900 python scripts/fuzz_prefill.py 40 71 wide; timeout 900 python scripts/fuzz_prefill.py 60 72 ) 2>&1 | grep -E "^bad:|BAD|ERR|cases" | tail -6
} > gpurun_out/r6_63_chunk_plan_tests.txt 2>&1
cat gpurun_out/r6_63_chunk_plan_tests.txt
