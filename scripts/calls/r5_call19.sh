#!/bin/bash
# round 5, call 19: the short-prompt GEMMs with the next step's operand reads requested before this step's MFMAs
cd /tmp && export TMPDIR=/tmp
repo=$GRAFT_REPO_ROOT
{
for n in 4 8 16; do
  python $repo/scripts/prefill_ab.py llama2-7b $n 8 "" 2>&1 | grep prefill
  L2Z_LIB=$repo/llama2.zig_amd/exp/libl2z_before.so python $repo/scripts/prefill_ab.py llama2-7b $n 8 "" 2>&1 | grep prefill | sed "s/^/   [before] /"
done
python $repo/scripts/prefill_ab.py stories110M 16 20 "" 2>&1 | grep prefill
L2Z_LIB=$repo/llama2.zig_amd/exp/libl2z_before.so python $repo/scripts/prefill_ab.py stories110M 16 20 "" 2>&1 | grep prefill | sed "s/^/   [before] /"
} > $repo/gpurun_out/r05u_skinny_operand_prefetch.txt 2>&1
cat $repo/gpurun_out/r05u_skinny_operand_prefetch.txt
cd $repo && timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py -m gpu -x -q -k "prefill" 2>&1 | tail -2
