#!/bin/bash
# round 5, call 21: a 65 ... 96-token tail cut in two chunks of the panel kernel's range
cd /tmp && export TMPDIR=/tmp
repo=$GRAFT_REPO_ROOT
{
for n in 129 144 160 161 257 272 288 650; do python $repo/scripts/prefill_ab.py llama2-7b $n 5 "" 2>&1 | grep prefill; done
python $repo/scripts/prefill_ab.py stories110M 80 10 "" 2>&1 | grep prefill
} > $repo/gpurun_out/r05z_prefill_tail_split.txt 2>&1
cat $repo/gpurun_out/r05z_prefill_tail_split.txt
cd $repo && timeout 1500 python -m pytest tests -m gpu -x -q -k "prefill" 2>&1 | tail -3
