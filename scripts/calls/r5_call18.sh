#!/bin/bash
# round 5, call 18: one / two token tiles on 8 waves with 64-k stages (build 64) against 4 waves with 128-k stages
cd /tmp && export TMPDIR=/tmp
repo=$GRAFT_REPO_ROOT
{
for n in 20 32; do
  python $repo/scripts/prefill_ab.py llama2-7b $n 8 "" 2>&1 | grep prefill
  L2Z_LIB=$repo/llama2.zig_amd/exp/libl2z_pn64.so python $repo/scripts/prefill_ab.py llama2-7b $n 8 "" 2>&1 | grep prefill | sed "s/^/   [8 waves, 64-k stages] /"
  L2Z_LIB=$repo/llama2.zig_amd/exp/libl2z_pn64.so python $repo/scripts/prefill_ab.py llama2-7b 16 8 "L2Z_PF_PANEL_MIN=1" 2>&1 | grep prefill | sed "s/^/   [8 waves, 64-k stages, 16 tokens on the panel kernel] /"
done
python $repo/scripts/prefill_ab.py llama2-7b 16 8 "" "L2Z_PF_PANEL_MIN=1" 2>&1 | grep prefill
} > $repo/gpurun_out/r05t_panel_waves_low_tiles.txt 2>&1
cat $repo/gpurun_out/r05t_panel_waves_low_tiles.txt
