repo=${GRAFT_REPO_ROOT:-$PWD}; cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_110; rocprofv3 --kernel-trace --stats -d /tmp/prof_110 -o p -- python $repo/scripts/prefill_prof.py stories110M ${1:-256} > /tmp/p110.log 2>&1 || tail -3 /tmp/p110.log
python $repo/scripts/rocprof_summary.py $(find /tmp/prof_110 -name "*.db" | head -1) "110M prefill ${1:-256}" | head -14
