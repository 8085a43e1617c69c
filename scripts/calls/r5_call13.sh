#!/bin/bash
# round 5, call 13: ring depth at three / four token tiles (4 waves x 6 buffers of 64 k against 8 waves x 3)
cd /tmp && export TMPDIR=/tmp
repo=$GRAFT_REPO_ROOT; O=$repo/gpurun_out
{
for n in 40 48 64; do  # (needs a library built with the six-buffer instantiation: commit history)
  python $repo/scripts/prefill_ab.py llama2-7b $n 8 "" 2>&1 | grep prefill
  L2Z_LIB=$repo/llama2.zig_amd/exp/libl2z_pn16.so python $repo/scripts/prefill_ab.py llama2-7b $n 8 "" 2>&1 | grep prefill | sed 's/^/   [4 waves x 6 buffers] /'
done
} > $O/r05m_panel_depth.txt 2>&1
cat $O/r05m_panel_depth.txt
