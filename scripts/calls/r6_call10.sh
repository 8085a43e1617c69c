#!/bin/bash
# round 6, GPU call 10: K rounded up to whole stages in every prefill GEMM (register-staged fallbacks deleted): suite + odd-K timing + fuzz
cd ${GRAFT_REPO_ROOT:-$PWD}; mkdir -p gpurun_out; O=gpurun_out
timeout 2400 python -m pytest tests -m gpu -q 2>&1 | tail -25 > $O/r6_10_pytest_gpu.txt
python scripts/prefill_odd_k.py > $O/r6_10_prefill_odd_k.txt 2>&1
( timeout 900 python scripts/fuzz_prefill.py 60 71; timeout 900 python scripts/fuzz_prefill.py 30 72 wide ) > $O/r6_10_fuzz_full.txt 2>&1
grep -E "^bad:|BAD|ERR" $O/r6_10_fuzz_full.txt | head > $O/r6_10_fuzz.txt
python scripts/prefill_ab.py llama2-7b 16 4 "" > $O/r6_10_ab.txt 2>&1; python scripts/prefill_ab.py llama2-7b 128 4 "" >> $O/r6_10_ab.txt 2>&1; python scripts/prefill_ab.py stories110M 200 4 "" >> $O/r6_10_ab.txt 2>&1
cat $O/r6_10_pytest_gpu.txt $O/r6_10_prefill_odd_k.txt $O/r6_10_fuzz.txt $O/r6_10_ab.txt
