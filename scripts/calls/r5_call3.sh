#!/bin/bash
# round 5, third GPU call: candidate exchange (all p2p forms), the panel prefill kernel's forms, then the whole suite.
cd "$GRAFT_REPO_ROOT"; O=gpurun_out; mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -q -x -rA -k "panel or tie_across or multiprocess or solo_rank or streams-30" > $O/r05c_pytest_new.log 2>&1; echo "pytest(new) rc=$?" | tee -a $O/r05c_pytest_new.log
grep -E "passed|failed|^FAILED|^ERROR|panel kernel|Error" $O/r05c_pytest_new.log | tail -n 25
( for n in 20 32; do
    timeout 300 python scripts/prefill_ab.py llama2-7b $n 5 "" "L2Z_PF_PANEL=0" "L2Z_PF_PANEL_FORM=1"
  done
  for n in 40 48 64; do
    timeout 300 python scripts/prefill_ab.py llama2-7b $n 5 "" "L2Z_PF_PANEL_MAX=64"
  done
  timeout 300 python scripts/prefill_ab.py llama2-7b 16 5 "" "L2Z_PF_PANEL_MIN=1" "L2Z_PF_PANEL_MIN=1,L2Z_PF_PANEL_FORM=1"
) > $O/r05c_prefill_panel_ab.txt 2>&1; cat $O/r05c_prefill_panel_ab.txt
timeout 600 python scripts/solo_rank.py llama2-7b 128 > $O/r05c_solo_rank.md 2>&1; cat $O/r05c_solo_rank.md
timeout 1500 python -m pytest tests -m gpu -q -rA --durations=8 > $O/r05c_pytest_gpu.log 2>&1; echo "pytest(all) rc=$?" | tee -a $O/r05c_pytest_gpu.log
grep -E "passed|failed|^FAILED|^ERROR|us per launch" $O/r05c_pytest_gpu.log | tail -n 30
