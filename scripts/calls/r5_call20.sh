#!/bin/bash
# round 5, call 20: where a stage's direct-to-LDS loads are issued -- at the stage's start (library), one between the MFMA
# steps at a time (build 64), all after the stage's last step (build 128)
cd /tmp && export TMPDIR=/tmp
repo=$GRAFT_REPO_ROOT
{
for n in 20 32 48 64; do
  python $repo/scripts/prefill_ab.py llama2-7b $n 8 "" 2>&1 | grep prefill
  for v in 64 128; do
    L2Z_LIB=$repo/llama2.zig_amd/exp/libl2z_pn$v.so python $repo/scripts/prefill_ab.py llama2-7b $n 8 "" 2>&1 | grep prefill | sed "s/^/   [build $v] /"
  done
done
} > $repo/gpurun_out/r05v_panel_dma_placement.txt 2>&1
cat $repo/gpurun_out/r05v_panel_dma_placement.txt
for v in 64 128; do L2Z_LIB=$repo/llama2.zig_amd/exp/libl2z_pn$v.so timeout 900 python -m pytest $repo/tests/test_gpu_parity.py -m gpu -x -q -k "panel_kernel" 2>&1 | tail -1; done
