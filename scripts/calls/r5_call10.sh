#!/bin/bash
# round 5, tenth GPU call: the 8-wave form of the panel kernel for three / four token tiles
cd "$GRAFT_REPO_ROOT"; O=gpurun_out; mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -q -x -rA -k "panel or streams-60 or streams-30 or 7b_prefill_equals" > $O/r05j_pytest_new.log 2>&1; echo "pytest(new) rc=$?" | tee -a $O/r05j_pytest_new.log
grep -E "passed|failed|^FAILED|^ERROR|panel|Error" $O/r05j_pytest_new.log | tail -n 12
( for n in 40 48 64; do timeout 300 python scripts/prefill_ab.py llama2-7b $n 5 "" "L2Z_PF_PANEL_WAVES=4" "L2Z_PF_PANEL=0"; done ) > $O/r05j_panel_waves_ab.txt 2>&1; cat $O/r05j_panel_waves_ab.txt
