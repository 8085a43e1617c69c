#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
AMD_LOG_LEVEL=3 python scripts/prefill_ab.py llama2-7b 100 1 "" > gpurun_out/r6_51_log.txt 2>&1
grep -n -i "invalid\|error\|fail\|exceed\|too large\|lds\|launch" gpurun_out/r6_51_log.txt | grep -v "hipSuccess" | tail -30
