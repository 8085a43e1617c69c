#!/bin/bash
# round 6, GPU call 3: W1|W3 two-range mixed launch (97-128 tokens): parity + A/B + kernel table
cd ${GRAFT_REPO_ROOT:-$PWD}; mkdir -p gpurun_out; O=gpurun_out
python -m pytest tests -m gpu -x -q -k "prefill or panel" 2>&1 | tail -15 > $O/r6_3_tests.txt
for n in 100 112 128; do
  python scripts/prefill_ab.py llama2-7b $n 6 "" "L2Z_PF_PAIR_MIX=1" "L2Z_PF_PAIR_MIX=2" "L2Z_PF_PAIR_MIX=3" >> $O/r6_3_pair_mix.txt 2>&1
done
python scripts/prefill_ab.py llama2-7b 64 4 "" >> $O/r6_3_pair_mix.txt 2>&1
bash scripts/pf_prof.sh llama2-7b 128 > $O/r6_3_prefill128_kernels.md 2>&1
cat $O/r6_3_tests.txt $O/r6_3_pair_mix.txt; head -12 $O/r6_3_prefill128_kernels.md
