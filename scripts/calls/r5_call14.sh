#!/bin/bash
# round 5, call 14: the reduce launches with eight loads in flight per thread (same add order)
cd /tmp && export TMPDIR=/tmp
repo=$GRAFT_REPO_ROOT; O=$repo/gpurun_out
{
for n in 20 32 48 64; do python $repo/scripts/prefill_ab.py llama2-7b $n 8 "" 2>&1 | grep prefill; done
for n in 32 64; do
  rm -rf /tmp/pe; timeout 300 rocprofv3 --kernel-trace -d /tmp/pe -o p -- python $repo/scripts/prefill_prof.py llama2-7b $n > /tmp/pe.log 2>&1 || tail -3 /tmp/pe.log
  python $repo/scripts/rocprof_summary.py $(find /tmp/pe -name "*.db" | head -1) "$n tokens" | grep "panel"
done
} > $O/r05o_reduce_batched.txt 2>&1
cat $O/r05o_reduce_batched.txt
cd $repo && timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py -m gpu -x -q -k "panel or prefill" 2>&1 | tail -3
