#!/bin/bash
# round 5, fifth GPU call: fixes of call 4 (scheme-B prefill with processes, solo rank N = 2), kernel tables of the
# 32 / 64 / 128-token prefills (where does a 64-token chunk's time go on the panel kernel?)
cd "$GRAFT_REPO_ROOT"; O=gpurun_out; mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -q -x -rA -k "scheme_b_batched or rccl_allreduce or solo_rank or perf_floor" > $O/r05e_pytest_new.log 2>&1; echo "pytest(new) rc=$?" | tee -a $O/r05e_pytest_new.log
grep -E "passed|failed|^FAILED|^ERROR|scheme B batched|common factor|Error" $O/r05e_pytest_new.log | tail -n 25
timeout 600 python -u scripts/solo_rank.py llama2-7b 128 > $O/r05e_solo_rank.md 2>&1; cat $O/r05e_solo_rank.md
( cd /tmp
  for n in 32 64 128; do
    rm -rf /tmp/prof_pf$n
    rocprofv3 --kernel-trace --stats -d /tmp/prof_pf$n -o p -- python $GRAFT_REPO_ROOT/scripts/prefill_prof.py llama2-7b $n > /tmp/prof_pf$n.log 2>&1 || tail -5 /tmp/prof_pf$n.log
    python $GRAFT_REPO_ROOT/scripts/rocprof_summary.py $(find /tmp/prof_pf$n -name "*.db" | head -1) "round 5 (r05e): rocprofv3 --kernel-trace --stats -- python scripts/prefill_prof.py llama2-7b $n (3 prefills)" > $GRAFT_REPO_ROOT/$O/r05e_prefill${n}_llama2-7b.md
    head -14 $GRAFT_REPO_ROOT/$O/r05e_prefill${n}_llama2-7b.md
  done )
