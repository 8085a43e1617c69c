#!/bin/bash
# round 6, GPU call 2: panel kernel forms of round 6 (3 tiles x 512, 4 x 384, 5 / 6 tiles): parity + A/B
cd ${GRAFT_REPO_ROOT:-$PWD}; mkdir -p gpurun_out; O=gpurun_out
python -m pytest tests -m gpu -x -q -k "panel or prefill" 2>&1 | tail -15 > $O/r6_2_tests.txt
for n in 40 48 56 64; do
  python scripts/prefill_ab.py llama2-7b $n 6 "" "L2Z_PF_PANEL_FORM=9" >> $O/r6_2_panel_ab.txt 2>&1
done
for n in 72 80 88 96; do
  python scripts/prefill_ab.py llama2-7b $n 6 "" "L2Z_PF_PANEL_MAX=64" "L2Z_PF_PANEL=0" >> $O/r6_2_panel_ab.txt 2>&1
done
cat $O/r6_2_tests.txt $O/r6_2_panel_ab.txt
