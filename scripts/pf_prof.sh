#!/bin/bash
# kernel table of one prefill: pf_prof.sh [shape] [n_tokens]
repo=${GRAFT_REPO_ROOT:-$PWD}; cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_pf; rocprofv3 --kernel-trace --stats -d /tmp/prof_pf -o p -- python $repo/scripts/prefill_prof.py ${1:-llama2-7b} ${2:-512} > /tmp/pf.log 2>&1 || tail -3 /tmp/pf.log
python $repo/scripts/rocprof_summary.py $(find /tmp/prof_pf -name "*.db" | head -1) "rocprofv3 --kernel-trace --stats -- python scripts/prefill_prof.py ${1:-llama2-7b} ${2:-512} (3 prefills)"
