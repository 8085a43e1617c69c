#!/bin/bash
# per-rank kernel time of the row-sharded prefill (emulated ranks on one GPU): rocprofv3 kernel trace of
# scripts/sharded_prefill_emu.py for one world size; the summary's total / (4 passes x world) = one
# rank's launches of one 512-token prefill.
world=${1:-8}; tag=${2:-r02}
repo=${GRAFT_REPO_ROOT:-$PWD}; O=$repo/gpurun_out; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_sp$world
rocprofv3 --kernel-trace --stats -d /tmp/prof_sp$world -o p -- python $repo/scripts/sharded_prefill_emu.py llama2-7b 512 $world > /tmp/prof_sp$world.log 2>&1 || tail -5 /tmp/prof_sp$world.log
tail -3 /tmp/prof_sp$world.log
db=$(find /tmp/prof_sp$world -name "*.db" | head -1)
python $repo/scripts/rocprof_summary.py $db "($tag): rocprofv3 --kernel-trace --stats -- python scripts/sharded_prefill_emu.py llama2-7b 512 $world  (4 unsharded prefills, then 4 prefills of $world emulated ranks)" > $O/${tag}_sharded_prefill_w${world}_kernel_stats.md
head -30 $O/${tag}_sharded_prefill_w${world}_kernel_stats.md
