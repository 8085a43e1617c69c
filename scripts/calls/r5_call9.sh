#!/bin/bash
# round 5, ninth GPU call: is 2 resident row-kernel blocks per CU still the best choice (whole token, and per kind)?
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
timeout 500 python scripts/ab.py llama2-7b 64 4 "" "L2Z_ROW_BLOCKS=3" "L2Z_ROW_BLOCKS=4" "L2Z_ROW_BLOCKS=1" 2>&1 | tail -4
timeout 300 python scripts/kind_scan.py llama2-7b "" "L2Z_ROW_BLOCKS=1" "L2Z_ROW_BLOCKS=3" "L2Z_ROW_BLOCKS=4" "L2Z_ROW_TAIL_SKIP=0" 2>&1 | tail -6
