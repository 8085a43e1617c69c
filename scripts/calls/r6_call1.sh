#!/bin/bash
# round 6, GPU call 1: new tests, pos-by-value experiment, panel forms A/B, prefill kernel tables at 64 / 100 / 128 tokens
cd ${GRAFT_REPO_ROOT:-$PWD}; mkdir -p gpurun_out; O=gpurun_out
python -m pytest tests -m gpu -x -q -k "refused_not_rerouted or bench_legs_on_one_gpu or stories42M" 2>&1 | tail -15 > $O/r6_1_tests.txt
python scripts/attn_pos_arg.py > $O/r6_1_attn_pos_arg.txt 2>&1
for n in 64 48 40; do
  python scripts/prefill_ab.py llama2-7b $n 6 "" "L2Z_PF_PANEL_FORM=1" "L2Z_PF_PANEL_FORM=2" "L2Z_PF_PANEL_FORM=3" >> $O/r6_1_panel_forms.txt 2>&1
done
for n in 64 100 128; do bash scripts/pf_prof.sh llama2-7b $n > $O/r6_1_prefill${n}_kernels.md 2>&1; done
python scripts/prefill_ab.py llama2-7b 100 4 "" >> $O/r6_1_panel_forms.txt 2>&1
python scripts/prefill_ab.py llama2-7b 128 4 "" >> $O/r6_1_panel_forms.txt 2>&1
cat $O/r6_1_tests.txt $O/r6_1_attn_pos_arg.txt $O/r6_1_panel_forms.txt
