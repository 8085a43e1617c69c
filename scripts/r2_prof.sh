#!/bin/bash
# rocprofv3 kernel-trace summaries for profiles/ (tag = $1)
tag=${1:-r02}
repo=${GRAFT_REPO_ROOT:-$PWD}; O=$repo/gpurun_out; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for wl in stories15M stories110M llama2-7b; do
  rm -rf /tmp/prof_$wl
  rocprofv3 --kernel-trace --stats -d /tmp/prof_$wl -o p -- python $repo/bench.py --workload $wl --steps 64 --warmup 2 --no-cpu-baseline --no-extra > /tmp/prof_$wl.log 2>&1 || tail -5 /tmp/prof_$wl.log
  db=$(find /tmp/prof_$wl -name "*.db" | head -1)
  python $repo/scripts/rocprof_summary.py $db "round 2 ($tag): rocprofv3 --kernel-trace --stats -- python bench.py --workload $wl --steps 64 --warmup 2 --no-cpu-baseline --no-extra" > $O/${tag}_${wl}_kernel_stats.md
  head -14 $O/${tag}_${wl}_kernel_stats.md
  tail -1 /tmp/prof_$wl.log | head -c 400; echo
done
