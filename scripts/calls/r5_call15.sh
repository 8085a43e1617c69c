#!/bin/bash
# round 5, call 15: blocks dealt to the ranges (one panel per block and launch)
cd /tmp && export TMPDIR=/tmp
repo=$GRAFT_REPO_ROOT; O=$repo/gpurun_out
{
for n in 20 32 48 64; do python $repo/scripts/prefill_ab.py llama2-7b $n 8 "" 2>&1 | grep prefill; done
} > $O/r05q_panel_one_range_per_block.txt 2>&1
cat $O/r05q_panel_one_range_per_block.txt
cd $repo && timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py tests/test_gpu_fuzz.py -m gpu -x -q -k "panel or prefill" 2>&1 | tail -3
