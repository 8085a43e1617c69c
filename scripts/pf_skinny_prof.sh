#!/bin/bash
# kernel table of a short-prompt prefill (skinny kernels): pf_skinny_prof.sh [n_tokens] [form...]
repo=${GRAFT_REPO_ROOT:-$PWD}; cd /tmp && export TMPDIR=/tmp
n=${1:-16}; shift
for form in ${@:-1 2}; do
rm -rf /tmp/prof_sk$form
L2Z_PF_SKINNY_FORM=$form rocprofv3 --kernel-trace --stats -d /tmp/prof_sk$form -o p -- python $repo/scripts/prefill_prof.py llama2-7b $n > /tmp/prof_sk.log 2>&1 || tail -5 /tmp/prof_sk.log
db=$(find /tmp/prof_sk$form -name "*.db" | head -1)
python $repo/scripts/rocprof_summary.py $db "skinny form $form, $n tokens" | head -14
done
