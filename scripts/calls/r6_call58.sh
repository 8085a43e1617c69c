#!/bin/bash
cd "$GRAFT_REPO_ROOT"; O=gpurun_out; mkdir -p $O
timeout 2400 python -m pytest tests -m gpu -q -rA --durations=8 > $O/r06_final_pytest_gpu.log 2>&1; echo "pytest rc=$?" | tee -a $O/r06_final_pytest_gpu.log
grep -E "passed|failed|^FAILED|^ERROR" $O/r06_final_pytest_gpu.log | tail -n 8 | tee $O/r06_final_pytest_gpu_tail.txt
grep -E "max \|diff\||max \|logit|identical|margin|vs oracle|vs the stepped|host replay|scheme B|common factor|panel kernel|W2 launch|us per layer|solo rank|stream form" $O/r06_final_pytest_gpu.log | head -150 > $O/r06_final_parity_numbers.txt
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
