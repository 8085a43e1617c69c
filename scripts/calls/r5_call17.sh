#!/bin/bash
# round 5, call 17: where the operand reads of the next MFMA step are placed -- the library (before this step's MFMAs, the
# order pinned, hipcc's waits) against builds 32 (after the MFMAs: the first form), 64 (inline-asm reads, exact waits, MFMAs
# back to back) and 128 (inline-asm reads in the middle of the step's MFMAs)
cd /tmp && export TMPDIR=/tmp
repo=$GRAFT_REPO_ROOT
{
for n in 20 32 48 64; do
  python $repo/scripts/prefill_ab.py llama2-7b $n 8 "" 2>&1 | grep prefill
  for v in 32 64 128; do
    L2Z_LIB=$repo/llama2.zig_amd/exp/libl2z_pn$v.so python $repo/scripts/prefill_ab.py llama2-7b $n 8 "" 2>&1 | grep prefill | sed "s/^/   [build $v] /"
  done
done
} > $repo/gpurun_out/r05s_panel_operand_prefetch.txt 2>&1
cat $repo/gpurun_out/r05s_panel_operand_prefetch.txt
cd $repo && timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "panel" 2>&1 | tail -2
for v in 64 128; do L2Z_LIB=$repo/llama2.zig_amd/exp/libl2z_pn$v.so timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "panel_kernel" 2>&1 | tail -1; done
