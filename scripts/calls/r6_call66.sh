#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
( timeout 1500 python scripts/fuzz_prefill.py 120 91 wide; timeout 1500 python scripts/fuzz_prefill.py 120 92; timeout 900 python scripts/fuzz_shards.py 60 93; timeout 600 python scripts/fuzz_greedy.py 100 94 ) > gpurun_out/r6_66_fuzz_more.txt 2>&1
grep -E "^bad:|BAD|ERR|refused" gpurun_out/r6_66_fuzz_more.txt | sort | uniq -c | head -12
