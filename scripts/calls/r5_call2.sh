#!/bin/bash
# round 5, second GPU call: the new code paths -- the panel prefill kernel and the sharded candidate exchange -- correctness
# first, then their measurements.
cd "$GRAFT_REPO_ROOT"; O=gpurun_out; mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -q -x -rA -k "panel or sharded_prefill_emulated or tie_across or multiprocess or solo_rank or 7b_prefill_equals or greedy_prompt_through_prefill or prefill_equals_token" > $O/r05b_pytest_new.log 2>&1; echo "pytest rc=$?" | tee -a $O/r05b_pytest_new.log
grep -E "passed|failed|^FAILED|^ERROR|panel kernel|Error|error" $O/r05b_pytest_new.log | tail -n 25
( for n in 8 16 24 32; do
    timeout 300 python scripts/prefill_ab.py llama2-7b $n 5 "" "L2Z_PF_PANEL=0" "L2Z_PF_PANEL_FUSE=0"
  done ) > $O/r05b_prefill_panel_ab.txt 2>&1; cat $O/r05b_prefill_panel_ab.txt
timeout 600 python scripts/solo_rank.py llama2-7b 128 > $O/r05b_solo_rank.md 2>&1; cat $O/r05b_solo_rank.md
( cd /tmp
  for n in 16 32; do
    rm -rf /tmp/prof_pf$n
    rocprofv3 --kernel-trace --stats -d /tmp/prof_pf$n -o p -- python $GRAFT_REPO_ROOT/scripts/prefill_prof.py llama2-7b $n > /tmp/prof_pf$n.log 2>&1 || tail -5 /tmp/prof_pf$n.log
    python $GRAFT_REPO_ROOT/scripts/rocprof_summary.py $(find /tmp/prof_pf$n -name "*.db" | head -1) "round 5 (r05b): rocprofv3 --kernel-trace --stats -- python scripts/prefill_prof.py llama2-7b $n (3 prefills)" > $GRAFT_REPO_ROOT/$O/r05b_prefill${n}_llama2-7b.md
    head -12 $GRAFT_REPO_ROOT/$O/r05b_prefill${n}_llama2-7b.md
  done )
