#!/bin/bash
cd "$GRAFT_REPO_ROOT"; O=gpurun_out; mkdir -p $O
export TMPDIR=/tmp
python scripts/ab.py llama2-7b 128 3 "" "L2Z_ROW_BLOCKS=3" "L2Z_ROW_BLOCKS=4" "L2Z_ATTN_BLOCK=256" "L2Z_MAX_BLOCKS_PER_CU=4" 2>&1 | tee $O/r2j_ab7b.txt
python scripts/ab.py llama2-7b 48 3 200 "" "L2Z_ATTN_BLOCK=256" 2>&1 | tee -a $O/r2j_ab7b.txt
